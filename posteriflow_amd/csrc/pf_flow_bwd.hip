// pf_flow_bwd.hip -- backward of the rational-quadratic spline with linear tails (the element-wise
// transform nflows' MaskedPiecewiseRationalQuadraticAutoregressiveTransform applies, executed under
// src/ahsd/models/flows.py:615-617), hand-derived.
//
// One lane per (row, feature) pair.  Given the pair's input u, its raw parameters
// (K widths, K heights, K-1 derivatives, nflows layout) and the incoming gradients gy = dL/dy and
// gl = dL/dlogabsdet it returns dL/d(raw parameters) and the direct part of dL/du.
//
// Inside [-tb, tb], with bin b such that kx_b <= u < kx_{b+1}:
//   W = kx_{b+1} - kx_b, Hh = ky_{b+1} - ky_b, delta = Hh / W, th = (u - kx_b) / W, tt = th (1 - th)
//   A  = delta th^2 + dl tt              Dn = delta + (dl + dr - 2 delta) tt
//   E  = dr th^2 + 2 delta tt + dl (1 - th)^2
//   y  = ky_b + Hh A / Dn                lad = 2 log delta + log E - 2 log Dn
// The partials with respect to (th, delta, Hh, dl, dr, ky_b) are closed-form; th and delta are then
// chained to (u, kx_b, W, Hh), the knots to the softmax of the raw widths / heights (interior knots
// only -- the two outer knots are pinned to -tb / tb), the knot derivatives to softplus of the raw
// values (the two outer ones are the constant that makes the derivative 1).  Outside the interval
// the transform is the identity: dL/du = gy and no parameter gradient.
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <cstdint>

#include "../../include/pf_hip.h"
#include "pf_rqs_bwd.h"

namespace pf {
namespace {

struct RqsBwdArgs {
    const float* u;        // [n, D]
    const float* params;   // [n, D, 3K-1]
    const float* gy;       // [n, D]
    const float* glad;     // [n]
    float* gparams;        // [n, D, 3K-1]
    float* gu;             // [n, D]
    int64_t n;
    int D;
    RqsConsts c;
};

template <int MAXB>
__global__ __launch_bounds__(256) void rqs_backward_kernel(RqsBwdArgs a) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= a.n * a.D) return;
    const int64_t row = idx / a.D;
    const int P = 3 * a.c.K - 1;
    a.gu[idx] = rqs_backward_pair<MAXB>(a.params + idx * P, a.gparams + idx * P, a.u[idx], a.gy[idx], a.glad[row], a.c);
}
}  // namespace

int rqs_backward(const PfFlowDesc& d, float deriv_const, const float* u, const float* params, const float* gy,
                 const float* glad, int64_t n, float* gparams, float* gu, hipStream_t s) {
    RqsBwdArgs a{u, params, gy, glad, gparams, gu, n, d.features,
                 RqsConsts{d.num_bins, d.tail_bound, d.min_bin_width, d.min_bin_height, d.min_derivative, deriv_const}};
    const int64_t pairs = n * d.features;
    if (d.num_bins <= 16) rqs_backward_kernel<16><<<dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0, s>>>(a);
    else rqs_backward_kernel<32><<<dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0, s>>>(a);
    return launch_status();
}
}  // namespace pf
