// pf_flow_fwd_kernel.h -- fused forward pass of the masked-autoregressive
// rational-quadratic-spline flow for gfx950 (MI355X): ALL layers of
//   [ReversePermutation, MADE conditioner, RQS elementwise + log|det J|]
// plus the N(0,I) base log-density run in ONE kernel; nothing but x, ctx, z,
// logdet, nll touches HBM besides the (L2-resident) packed weights.
//
// Replaces (reference file:line):
//   NSFPosteriorFlow.forward            src/ahsd/models/flows.py:610-618
//   NSFPosteriorFlow.compute_psd_aware_nll (log_sigma = 0)  flows.py:727-779
//   which execute nflows CompositeTransform/ReversePermutation/MADE/
//   MaskedPiecewiseRationalQuadraticAutoregressiveTransform (built flows.py:459-529).
//
// Work decomposition (DESIGN.md "Kernels"):
//   workgroup = 16*R batch rows, NW = H/32 waves (512 threads at H = 256).
//   Everything is computed TRANSPOSED: out^T[unit, row] = W[unit, k] . act^T[k, row],
//   so the weights are the MFMA A operand (streamed from L2 exactly once per
//   workgroup, in pre-packed fragment order, straight into VGPRs) and the batch
//   rows are the 16 MFMA columns.  Wave w owns the hidden-tile pair (w, T-1-w)
//   (degree-sorted 16-unit tiles, pf_layout.h) and the spline-feature pair
//   (w, D-1-w): the block-triangular masks make the pair's useful k-range the
//   same for every wave, so all waves run one static schedule and all-zero
//   fragments are simply absent from the stream.  The residual state h lives in
//   fp32 accumulator registers for the whole layer; activations are exchanged
//   between waves through LDS in B-fragment order (one barrier per GEMM).
//   Weight streaming: each wave walks ONE linear fragment stream, fully
//   unrolled per layer, keeping a window of 16 unconditional fragment loads
//   (16 KiB) in flight in registers across phases, barriers and layers, so the
//   compiler's counted vmcnt waits are exact.  Biases are staged a layer ahead
//   through LDS (a late bias load would drain the window).  In the final masked layer wave w owns spline
//   the 3K-1 raw parameters of a feature come out of three 16-row MFMA tiles
//   (widths | heights | derivatives); they are transposed through LDS and the
//   spline (softmax, cumsum, bin search, rational quadratic, log-det) is
//   evaluated by ONE lane per (row, feature) pair, all pairs of the workgroup
//   at once -- no cross-lane traffic, 2.5x fewer wave-instructions than a
//   4-lanes-per-pair in-register evaluation (measured: the kernel is
//   instruction-issue-bound, profiles/README.md).
#pragma once
#ifndef PF_FWD_BENDS
#define PF_FWD_BENDS 1     // 1: activation fragments requested from both ends inwards (see load_b_ends)
#endif
#if PF_FWD_BENDS
#define PF_LOAD_B load_b_ends
#else
#define PF_LOAD_B load_b
#endif
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <cstdlib>
#include <type_traits>

#include "pf_flow_params.h"

namespace pf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- small device helpers ------------------------------------------------------
// FAST = bf16 mode: bare v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp; __expf / __logf compile to sequences with
// denormal-range fix-ups here).  The fp32 parity mode of the forward and D-pass kernels calls the library routines
// (the forward parity test sits at the fp32 noise floor: 1e-5 relative on the density is also what the CPU fp32
// oracle reaches).  pf_*_acc: the bare instructions corrected to ~1 ulp -- exp and log carry the rounding error of
// the base-2 scaling in a second term, division refines the reciprocal with two fused steps, log(1 + e) switches to
// its series below e = 0.01 -- used by the fp32 incremental inverse, whose spline is on its critical path.
__device__ __forceinline__ float pf_exp_acc(float v) {
    const float t = v * 1.44269502f;                                           // float(log2 e)
    const float r = __builtin_fmaf(v, 1.44269502f, -t) + v * 1.92596299e-8f;   // what t lost + log2 e - float(log2 e)
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, r * 0.693147181f, e);
}
__device__ __forceinline__ float pf_log_acc(float v) {
    const float l2 = __builtin_amdgcn_logf(v);
    return __builtin_fmaf(l2, 0.693147182f, l2 * -1.90465498e-9f);
}
__device__ __forceinline__ float pf_div_acc(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.f), r, r);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
template <bool FAST> __device__ __forceinline__ float pf_exp(float v) {
    return FAST ? __builtin_amdgcn_exp2f(v * 1.44269504f) : expf(v);
}
template <bool FAST> __device__ __forceinline__ float pf_log(float v) {
    return FAST ? __builtin_amdgcn_logf(v) * 0.693147181f : logf(v);
}
template <bool FAST> __device__ __forceinline__ float pf_div(float a, float b) {
    return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b;
}
template <bool FAST> __device__ __forceinline__ float pf_softplus(float u) {
    // torch F.softplus: beta = 1, threshold = 20
    if (FAST) return u > 20.f ? u : pf_log<true>(1.f + pf_exp<true>(u));
    return u > 20.f ? u : log1pf(expf(u));
}
__device__ __forceinline__ float pf_softplus_acc(float u) {
    const float e = pf_exp_acc(u);
    const float small = e * (1.f - e * (0.5f - e * 0.333333343f));             // log(1 + e), e < 0.01: error < e^4 / 4
    return u > 20.f ? u : (e < 0.01f ? small : pf_log_acc(1.f + e));
}
template <bool FAST> __device__ __forceinline__ float pf_sigmoid(float v) {
    return pf_div<FAST>(1.f, 1.f + pf_exp<FAST>(-v));
}

constexpr int kParStride = 52;   // floats per (row, feature) pair in the LDS transpose (48 used)

// v_max_f32 / v_max3_f32 without the NaN-quieting v_max x, x that fmaxf costs under IEEE mode (rqs_pair_fast16 only;
// a NaN operand gives the other operand, like fmaxf)
__device__ __forceinline__ float pf_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float pf_max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// rqs_pair for K = 16 in the throughput (bf16) mode: the same spline, ~40 % fewer instructions and shallower
// dependency chains.  Differences from the step-by-step form below (all at fp32 rounding level, none in what is
// computed): the exponentials are one fma + v_exp each (log2 e folded in), the softmax sums are trees, the knots are a
// fused running sum k[i+1] = k[i] + 2B min + (2B (1 - 16 min) / sum) e[i] instead of cumsum-then-affine, and the bin is
// found by a 4-level binary search over the 17 knots (select trees on knots, heights and derivatives: 57 selects
// instead of 6 per bin) -- knots increase strictly (min bin width), so it finds searchsorted's bin, including x = B
// (last knot + 1e-6: never compared, the search ends in bin 15).
// (the parameters in registers: uw / uh = the 16 raw widths / heights (overwritten), kd[1 .. 15] = the raw interior derivatives,
// kd[0] and kd[16] are set here; rqs_pair_fast16 below loads them from a transpose, the mid-batch kernel hands them over from
// its accumulators)
__device__ __forceinline__ void rqs_fast16_regs(float (&uw)[16], float (&uh)[16], float (&kd)[17], float x, const FwdParams& p,
                                                float& y, float& ld) {
    const float tb = p.tail_bound, span = 2.f * tb;
    kd[0] = p.deriv_const; kd[16] = p.deriv_const;
    constexpr float kL2E = 1.44269504f;
    auto max16 = [](const float (&v)[16]) {
        const float a = pf_max3(v[0], v[1], v[2]), b = pf_max3(v[3], v[4], v[5]), c = pf_max3(v[6], v[7], v[8]);
        const float d = pf_max3(v[9], v[10], v[11]), e = pf_max3(v[12], v[13], v[14]);
        return pf_max3(pf_max3(a, b, c), pf_max3(d, e, v[15]), v[15]);
    };
    const float mw = -max16(uw) * kL2E, mh = -max16(uh) * kL2E;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uw[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(uw[i], kL2E, mw));
        uh[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(uh[i], kL2E, mh));
    }
    auto sum16 = [](const float (&v)[16]) {
        return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) +
               (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    };
    const float aw = span * (1.f - p.min_w * 16.f) * __builtin_amdgcn_rcpf(sum16(uw)), bw = span * p.min_w;
    const float ah = span * (1.f - p.min_h * 16.f) * __builtin_amdgcn_rcpf(sum16(uh)), bh = span * p.min_h;
    float kw[17], kh[17];
    kw[0] = -tb; kh[0] = -tb;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        kw[i + 1] = __builtin_fmaf(aw, uw[i], kw[i] + bw);
        kh[i + 1] = __builtin_fmaf(ah, uh[i], kh[i] + bh);
    }
    kw[16] = tb; kh[16] = tb;
    // binary search: after level j the candidate knots are a window of 2^(4-j) + 1 consecutive knots
    float w9[9], h9[9], d9[9];
    const bool c1 = x >= kw[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) { w9[i] = c1 ? kw[8 + i] : kw[i]; h9[i] = c1 ? kh[8 + i] : kh[i]; d9[i] = c1 ? kd[8 + i] : kd[i]; }
    float w5[5], h5[5], d5[5];
    const bool c2 = x >= w9[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) { w5[i] = c2 ? w9[4 + i] : w9[i]; h5[i] = c2 ? h9[4 + i] : h9[i]; d5[i] = c2 ? d9[4 + i] : d9[i]; }
    float w3[3], h3[3], d3[3];
    const bool c3 = x >= w5[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { w3[i] = c3 ? w5[2 + i] : w5[i]; h3[i] = c3 ? h5[2 + i] : h5[i]; d3[i] = c3 ? d5[2 + i] : d5[i]; }
    const bool c4 = x >= w3[1];
    const float xl = c4 ? w3[1] : w3[0], xr = c4 ? w3[2] : w3[1];
    const float yl = c4 ? h3[1] : h3[0], yr = c4 ? h3[2] : h3[1];
    const float dl_raw = c4 ? d3[1] : d3[0], dr_raw = c4 ? d3[2] : d3[1];
    const float w = xr - xl, h = yr - yl;
    const float dl = p.min_d + pf_softplus<true>(dl_raw);
    const float dr = p.min_d + pf_softplus<true>(dr_raw);
    const float rw = __builtin_amdgcn_rcpf(w);
    const float delta = h * rw;
    const float th = (x - xl) * rw;
    const float tt = th * (1.f - th);
    const float numer = h * (delta * th * th + dl * tt);
    const float den = delta + (dl + dr - 2.f * delta) * tt;
    const float omt = 1.f - th;
    const float dnum = delta * delta * (dr * th * th + 2.f * delta * tt + dl * omt * omt);
    const bool inside = (x >= -tb) && (x <= tb);
    y = inside ? yl + numer * __builtin_amdgcn_rcpf(den) : x;
    ld = inside ? (__builtin_amdgcn_logf(dnum) - 2.f * __builtin_amdgcn_logf(den)) * 0.693147181f : 0.f;
}
__device__ __forceinline__ void rqs_pair_fast16(const float* par, float x, const FwdParams& p, float& y, float& ld) {
    float uw[16], uh[16], kd[17];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(par + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(par + 16 + 4 * q);
        const f32x4 c = *reinterpret_cast<const f32x4*>(par + 32 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { uw[4 * q + e] = a[e]; uh[4 * q + e] = b[e]; kd[1 + 4 * q + e] = c[e]; }
    }
    rqs_fast16_regs(uw, uh, kd, x, p, y, ld);
}

// Forward RQS of one (row, feature) pair by one lane.  par: 16 raw widths | 16 raw heights |
// 15 raw derivatives (rows >= K unused).  Follows nflows' rational_quadratic_spline /
// unconstrained_rational_quadratic_spline (tails = 'linear') step by step: softmax,
// min + (1 - min K) softmax, sequential cumsum, affine to [-tb, tb] with pinned ends,
// searchsorted with the last knot + 1e-6, derivative = min_d + softplus(raw), boundary
// derivative from the constant log(exp(1 - min_d) - 1).
template <bool FAST>
__device__ __forceinline__ void rqs_pair(const float* par, float x, int K, const FwdParams& p,
                                         float& y, float& ld) {
#ifndef PF_NO_FAST16
    if constexpr (FAST) {
        if (K == 16) { rqs_pair_fast16(par, x, p, y, ld); return; }
    }
#endif
    float uw[16], uh[16], ud[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(par + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(par + 16 + 4 * q);
        const f32x4 c = *reinterpret_cast<const f32x4*>(par + 32 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { uw[4 * q + e] = a[e]; uh[4 * q + e] = b[e]; ud[4 * q + e] = c[e]; }
    }
    const float tb = p.tail_bound;
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < K) { mw = fmaxf(mw, uw[i]); mh = fmaxf(mh, uh[i]); }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uw[i] = i < K ? pf_exp<FAST>(uw[i] - mw) : 0.f; sw += uw[i];
        uh[i] = i < K ? pf_exp<FAST>(uh[i] - mh) : 0.f; sh += uh[i];
    }
    const float cw = pf_div<FAST>(1.f - p.min_w * (float)K, sw);
    const float ch = pf_div<FAST>(1.f - p.min_h * (float)K, sh);
    const float span = 2.f * tb;
    float cumw = 0.f, cumh = 0.f;
    float xl = -tb, xr = tb, yl = -tb, yr = tb;
    float dl_raw = p.deriv_const, dr_raw = p.deriv_const;
    bool prev_ge = true;                     // x >= left knot of bin 0 (x is inside)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < K) {
            cumw += p.min_w + cw * uw[i];
            cumh += p.min_h + ch * uh[i];
            const bool last = i == K - 1;
            const float kr = last ? tb : span * cumw - tb;       // right knot of bin i (widths)
            const float hr = last ? tb : span * cumh - tb;       // right knot of bin i (heights)
            const float dr = last ? p.deriv_const : ud[i];       // raw derivative at that knot
            const bool ge = x >= (last ? tb + 1e-6f : kr);       // searchsorted: last knot + eps
            const bool sel = prev_ge && !ge;                     // x falls into bin i
            xr = sel ? kr : xr; yr = sel ? hr : yr; dr_raw = sel ? dr : dr_raw;
            xl = ge ? kr : xl;  yl = ge ? hr : yl;  dl_raw = ge ? dr : dl_raw;
            prev_ge = ge;
        }
    }
    const float w = xr - xl, h = yr - yl;
    const float dl = p.min_d + pf_softplus<FAST>(dl_raw);
    const float dr = p.min_d + pf_softplus<FAST>(dr_raw);
    const float delta = pf_div<FAST>(h, w);
    const float th = pf_div<FAST>(x - xl, w);
    const float tt = th * (1.f - th);
    const float numer = h * (delta * th * th + dl * tt);
    const float den = delta + (dl + dr - 2.f * delta) * tt;
    const float omt = 1.f - th;
    const float dnum = delta * delta * (dr * th * th + 2.f * delta * tt + dl * omt * omt);
    const bool inside = (x >= -tb) && (x <= tb);
    y = inside ? yl + pf_div<FAST>(numer, den) : x;
    ld = inside ? pf_log<FAST>(dnum) - 2.f * pf_log<FAST>(den) : 0.f;
}


// Inverse RQS of one (row, feature) pair by one lane: given y (the transform's output) find
// x.  nflows rational_quadratic_spline(inverse=True): bin search on the cumulative HEIGHTS,
// root of the quadratic a t^2 + b t + c via 2c / (-b - sqrt(b^2 - 4ac)); returns the
// log-det of the inverse map (= -forward log-det at the root).  `bad` is set where the
// discriminant is negative (the reference asserts there, flows.py:635-642).
template <bool FAST>
__device__ __forceinline__ void rqs_pair_inverse(const float* par, float yin, int K, const FwdParams& p,
                                                 float& x, float& ld, bool& bad) {
    float uw[16], uh[16], ud[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(par + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(par + 16 + 4 * q);
        const f32x4 c = *reinterpret_cast<const f32x4*>(par + 32 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { uw[4 * q + e] = a[e]; uh[4 * q + e] = b[e]; ud[4 * q + e] = c[e]; }
    }
    const float tb = p.tail_bound;
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < K) { mw = fmaxf(mw, uw[i]); mh = fmaxf(mh, uh[i]); }
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uw[i] = i < K ? pf_exp<FAST>(uw[i] - mw) : 0.f; sw += uw[i];
        uh[i] = i < K ? pf_exp<FAST>(uh[i] - mh) : 0.f; sh += uh[i];
    }
    const float cw = pf_div<FAST>(1.f - p.min_w * (float)K, sw);
    const float ch = pf_div<FAST>(1.f - p.min_h * (float)K, sh);
    const float span = 2.f * tb;
    float cumw = 0.f, cumh = 0.f;
    float xl = -tb, xr = tb, yl = -tb, yr = tb;
    float dl_raw = p.deriv_const, dr_raw = p.deriv_const;
    bool prev_ge = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < K) {
            cumw += p.min_w + cw * uw[i];
            cumh += p.min_h + ch * uh[i];
            const bool last = i == K - 1;
            const float kr = last ? tb : span * cumw - tb;
            const float hr = last ? tb : span * cumh - tb;
            const float dr = last ? p.deriv_const : ud[i];
            const bool ge = yin >= (last ? tb + 1e-6f : hr);     // search on the heights
            const bool sel = prev_ge && !ge;
            xr = sel ? kr : xr; yr = sel ? hr : yr; dr_raw = sel ? dr : dr_raw;
            xl = ge ? kr : xl;  yl = ge ? hr : yl;  dl_raw = ge ? dr : dl_raw;
            prev_ge = ge;
        }
    }
    const float w = xr - xl, h = yr - yl;
    const float dl = p.min_d + pf_softplus<FAST>(dl_raw);
    const float dr = p.min_d + pf_softplus<FAST>(dr_raw);
    const float delta = pf_div<FAST>(h, w);
    const float dy = yin - yl;
    const float s2 = dl + dr - 2.f * delta;
    const float a = dy * s2 + h * (delta - dl);
    const float b = h * dl - dy * s2;
    const float c = -delta * dy;
    const float disc = b * b - 4.f * a * c;
    const float root = pf_div<FAST>(2.f * c, -b - (FAST ? __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) : sqrtf(fmaxf(disc, 0.f))));
    const float tt = root * (1.f - root);
    const float den = delta + s2 * tt;
    const float omt = 1.f - root;
    const float dnum = delta * delta * (dr * root * root + 2.f * delta * tt + dl * omt * omt);
    const bool inside = (yin >= -tb) && (yin <= tb);
    bad = inside && !(disc >= 0.f);
    x = inside ? root * w + xl : yin;
    ld = inside ? -(pf_log<FAST>(dnum) - 2.f * pf_log<FAST>(den)) : 0.f;
}

// ---- the kernel ------------------------------------------------------------------
// LDS carve (bytes), all 16-B aligned:
//   ctx    : CKM * R KiB            context in B-fragment order (zero padded to CKM k-steps)
//   act0   : HK * R KiB             activations in B-fragment order
//   P      : max(HK*R KiB, D*16*52*4)  second activation buffer, aliased by the spline transpose
//   xb0/1  : 16 * 16R floats each   layer input x^T / output z^T (double buffer)
//   bias0/1: NT * 192 floats each   this / next layer's biases
//   ldb    : D * 16R floats         per-(feature,row) log-det partials
template <int B, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, N - 1>(f);
    }
}
template <int V> using ic = std::integral_constant<int, V>;

#ifdef PF_ABLATE_BUILD
#define PF_ABL(mask) (p.ablate & (mask))
#else
#define PF_ABL(mask) false
#endif

// DROP: training forward with dropout in the residual blocks (flow_train_kernel below); never set for the inverse
template <bool BF16, int NT, int R, int CKM, int DENSE, bool INV, bool DROP>
__device__ __forceinline__ void flow_body(const FwdParams& p) {
    using S = Sched<BF16, NT, CKM, DENSE>;
    constexpr bool FAST = BF16;
    constexpr int NW = NT / 2, HK = S::HK, KHS = S::KHS, KOS = S::KOS, NF = S::NF;
    constexpr int COLS = 16 * R;
#ifndef PF_FWD_W
#define PF_FWD_W 16
#endif
    constexpr int W = PF_FWD_W;                         // frags in flight per wave
    constexpr int NE = (NF + W - 1) / W * W;            // schedule length rounded to the window
    constexpr bool CTX_REGS = (R == 1) && CKM > 0 && CKM <= 9;   // context B fragments live in registers
    static_assert(W <= kWindowPad, "stream pad too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FlowPlan& L = p.plan;
    const int D = L.D, K = L.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * COLS;
    const int tA = wave, tB = NT - 1 - wave;
    // per-wave constants, forced into SGPRs
    const int kA = __builtin_amdgcn_readfirstlane(L.kH[tA]);      // W0/W1 entries [0,kA): tile A
    const int fA = __builtin_amdgcn_readfirstlane(L.featA[wave]); // spline features (-1: none)
    const int fB = __builtin_amdgcn_readfirstlane(L.featB[wave]);
    const int kOA = __builtin_amdgcn_readfirstlane(fA >= 0 ? L.kO[fA] : 0);   // out entries [0,kOA): feature A

    const size_t p_bytes = max((size_t)HK * R * kFragBytes, (size_t)D * 16 * kParStride * sizeof(float));
    char* s_ctx = smem;
    char* s_act0 = s_ctx + (size_t)CKM * R * kFragBytes;
    char* s_act1 = s_act0 + (size_t)HK * R * kFragBytes;          // region P
    float* s_par = reinterpret_cast<float*>(s_act1);
    float* s_xb0 = reinterpret_cast<float*>(s_act1 + p_bytes);
    float* s_xb1 = s_xb0 + 16 * COLS;
    float* s_xb2 = s_xb1 + 16 * COLS;                   // inverse only (forward: unused)
    float* s_bias = s_xb2 + 16 * COLS;                  // [2][NT][192]
    float* s_ldb = s_bias + 2 * NT * kBiasFloatsPerTile;

    // ---- weight stream: buffer resource over this wave's region, register window ---------
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.packed) + (int64_t)wave * L.fragsPerWave * kFragBytes, 0,
        (int)(L.fragsPerWave * kFragBytes), 0x00020000);
    u32x4 win[W];
    // forward walks the layers 0..L-1 once; the inverse walks L-1..0 with D conditioner
    // passes per layer (nflows AutoregressiveTransform.inverse), re-reading the layer's frags
    int lbase = INV ? (L.L - 1) * NF * kFragBytes : 0;  // byte offset of the current layer's frags
    int nbase = 0;                                      // ... of the frags consumed after this pass
    // entry E of the current pass was just consumed: refill its slot with entry E + W
    auto refill = [&](auto e) {
        constexpr int E = decltype(e)::value;
        constexpr int EN = (E + W) % NE;
        constexpr bool wrap = (E + W) >= NE;
        if constexpr (EN < NF) {
            const int off = PF_ABL(2) ? 0 : (wrap ? nbase : lbase) + EN * kFragBytes;
            win[EN % W] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, off, 0);
        }
    };

    // prologue of the weight stream: first window of layer 0, requested before the staging loads so that
    // its L2 latency runs under theirs
    static_for<0, W>([&](auto e) {
        constexpr int E = decltype(e)::value;
        if constexpr (E < NF) win[E] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, lbase + E * kFragBytes, 0);
    });
    // ---- stage context (B-fragment order), x^T and layer 0's biases -------------------
    for (int s = tid; s < CKM * R * 64; s += NW * 64) {
        const int ln = s & 63, r = (s >> 6) % R, ks = s / (64 * R);
        const int gg = ln >> 4, cc = ln & 15;
        int64_t row = row0 + 16 * r + cc;
        if (row >= p.batch) row = p.batch - 1;
        if (INV) row /= (p.batch / p.ctx_rows);          // sample i uses context row i / (B / ctx_rows)
        const float* src = p.ctx + row * L.C;
        if (BF16) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = 32 * ks + 8 * gg + j;
                v[j] = (__bf16)(col < L.C ? src[col] : 0.f);
            }
            *reinterpret_cast<bf16x8*>(s_ctx + (size_t)s * 16) = v;
        } else {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 16 * ks + 4 * gg + e;
                v[e] = col < L.C ? src[col] : 0.f;
            }
            *reinterpret_cast<f32x4*>(s_ctx + (size_t)s * 16) = v;
        }
    }
    for (int s = tid; s < 16 * COLS; s += NW * 64) {
        const int d = s / COLS, col = s % COLS;
        float v = 0.f;
        if (d < D) {
            int64_t row = row0 + col;
            if (row >= p.batch) row = p.batch - 1;
            if (INV) {
                v = p.x[row * D + d];                    // z, in the last layer's coordinates
            } else {
                // layer 0 sees reverse(x[:, ar_perm]): position d <- source D-1-d
                const int sd = L.additive ? d : D - 1 - d;
                const int src = p.ar_perm ? p.ar_perm[sd] : sd;
                v = p.x[row * D + src];
            }
        }
        s_xb0[s] = v;
        s_xb1[s] = 0.f;
        s_xb2[s] = 0.f;
    }
    const float* gbias = reinterpret_cast<const float*>(p.packed + L.weightBytes);
    {
        const float* b0 = gbias + (INV ? L.bias_index(L.L - 1, 0) : 0);
        for (int s = tid; s < NT * kBiasFloatsPerTile; s += NW * 64) s_bias[s] = b0[s];
    }
    for (int s = tid; s < D * COLS; s += NW * 64) s_ldb[s] = 0.f;
    __syncthreads();

    // B fragments of one LDS buffer -> registers
    auto load_b = [&](const char* src, auto n, u32x4 (&bk)[decltype(n)::value][R]) {
        constexpr int N = decltype(n)::value;
#pragma unroll
        for (int ks = 0; ks < N; ++ks)
#pragma unroll
            for (int r = 0; r < R; ++r)
                bk[ks][r] = *reinterpret_cast<const u32x4*>(src + ((size_t)(ks * R + r) * 64 + lane) * 16);
    };
    // the same, requested from both ends inwards (0, N-1, 1, N-2, ...): gemm_pair's k-step I multiplies fragment I (first tile,
    // ascending) or N-1-I (second tile, descending) depending on the wave, so in this order step I needs only the first 2 (I + 1)
    // reads instead of all of them (the first MFMA after every barrier waited for the whole buffer)
    auto load_b_ends = [&](const char* src, auto n, u32x4 (&bk)[decltype(n)::value][R]) {
        constexpr int N = decltype(n)::value;
        static_for<0, N>([&](auto i) {
            constexpr int I = decltype(i)::value;
            constexpr int ks = (I & 1) ? N - 1 - (I >> 1) : (I >> 1);
#pragma unroll
            for (int r = 0; r < R; ++r)
                bk[ks][r] = *reinterpret_cast<const u32x4*>(src + ((size_t)(ks * R + r) * 64 + lane) * 16);
        });
    };
    u32x4 cb[CTX_REGS ? CKM : 1][R];
    if constexpr (CTX_REGS) load_b(s_ctx, ic<CKM>{}, cb);

    // hoisted context projections (pf_flow_ctx.hip): 4 consecutive sorted units of one tile for
    // this lane's batch column, already relu'd / sigmoid'ed
    auto load_proj = [&](int l, int j, int tile, int r) -> f32x4 {
        int64_t row = row0 + 16 * r + c;
        if (row >= p.batch) row = p.batch - 1;
        if (INV) row /= (p.batch / p.ctx_rows);
        // fragment order of pf_flow_ctx.hip: [row/64][tile u][(row/16)%4][g*16 + row%16][4]
        const size_t u = ((size_t)l * 3 + j) * NT + tile;
        const size_t off = ((((size_t)(row >> 6) * (3 * L.L * NT) + u) * 4 + ((row >> 4) & 3)) * 64
                            + (g * 16 + (row & 15))) * 4;
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.cproj) + off);
    };
    const bool hoisted = CKM == 0 && p.cproj != nullptr;
    const bool additive = hoisted && L.additive;        // masked-context conditioner (flows.py:186-234)
    // position of feature d in the next layer's input (ReversePermutation unless masked-context)
    auto rv = [&](int d) { return L.additive ? d : D - 1 - d; };

    float ld_pair = 0.f;                                // log-det of this thread's (feature,row) pair

    // acc[r] += A(window slot of entry E) . B
    auto mma = [&](auto e, const u32x4 (&b)[R], f32x4 (&acc)[R]) {
        constexpr int E = decltype(e)::value;
        const u32x4 a = win[E % W];
        if (PF_ABL(4)) { asm volatile("" :: "v"(a)); return; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (BF16) {
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                                 __builtin_bit_cast(bf16x8, b[r]), acc[r], 0, 0, 0);
            } else {
                const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b[r]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc[r], 0, 0, 0);
            }
        }
    };
    // write 16 units x COLS activations of tile `tile` into a B-fragment buffer
    auto store_act = [&](char* dst, int tile, const f32x4 (&v)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (BF16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[r][e];
                *reinterpret_cast<bf16x4*>(dst + ((size_t)((tile >> 1) * R + r) * 64 + lane) * 16 + (tile & 1) * 8) = o;
            } else {
                *reinterpret_cast<f32x4*>(dst + ((size_t)(tile * R + r) * 64 + lane) * 16) = v[r];
            }
        }
    };
    auto zero = [&](f32x4 (&v)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto barrier = [&]() { if (!PF_ABL(8)) __syncthreads(); };

    uint32_t bad_pair = 0;
    const int n_pass = INV ? D : 1;
    // inverse buffers rotate: yb = this layer's known output, xc = current estimate of its
    // input (conditioner input), yn = the previous layer's output being assembled
    float* yb = s_xb0; float* xc = s_xb1; float* yn = s_xb2;
    for (int it = 0; it < L.L * n_pass; ++it) {
        const int step = it / n_pass, pass = it - step * n_pass;
        const int l = INV ? L.L - 1 - step : step;      // layer being evaluated
        const bool last_pass = pass == n_pass - 1;
        float* xin = INV ? xc : ((l & 1) ? s_xb1 : s_xb0);
        float* xout = (l & 1) ? s_xb0 : s_xb1;
        const int bsel = INV ? (step & 1) : (l & 1);
        const float* bias = s_bias + bsel * NT * kBiasFloatsPerTile;
        {   // where the stream continues after this pass
            const int lnext = INV ? (last_pass ? l - 1 : l) : l + 1;
            nbase = (lnext < 0 ? 0 : lnext) * NF * kFragBytes;
        }
        auto load_bias = [&](int tile, int slot) {
            return *reinterpret_cast<const f32x4*>(bias + tile * kBiasFloatsPerTile + slot * 16 + 4 * g);
        };
        // next layer's biases: issued now (oldest loads of the layer), parked in LDS at its end
        f32x4 nbA = {0.f, 0.f, 0.f, 0.f}, nbB = nbA;
        {
            const int ln = INV ? (l > 0 ? l - 1 : 0) : (l + 1 < L.L ? l + 1 : l);
            if (lane < kBiasFloatsPerTile / 4) {
                nbA = *reinterpret_cast<const f32x4*>(gbias + L.bias_index(ln, tA) + 4 * lane);
                nbB = *reinterpret_cast<const f32x4*>(gbias + L.bias_index(ln, tB) + 4 * lane);
            }
        }
        f32x4 prA[3][R], prB[3][R];
        if constexpr (CKM == 0) {
            if (hoisted) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < R; ++r) { prA[j][r] = load_proj(l, j, tA, r); prB[j][r] = load_proj(l, j, tB, r); }
            }
        }
        // masked GEMM over a tile / feature pair, entries [E0, E0+N): entry i < nA is k-step i of
        // the first accumulator, otherwise k-step N-1-i of the second (pf_layout.h); every
        // k-step index is a compile-time constant, only the accumulator choice is per wave
        auto gemm_pair = [&](auto e0, auto n, int nA, const u32x4 (&bk)[HK][R], f32x4 (&a0)[R], f32x4 (&a1)[R]) {
            constexpr int E0 = decltype(e0)::value, N = decltype(n)::value;
            static_for<0, N>([&](auto i) {
                constexpr int I = decltype(i)::value;
                constexpr int KA = I < HK ? I : HK - 1, KB = (N - 1 - I) < HK ? (N - 1 - I) : HK - 1;
                if (I < nA) mma(ic<E0 + I>{}, bk[KA], a0);
                else mma(ic<E0 + I>{}, bk[KB], a1);
                refill(ic<E0 + I>{});
            });
        };
        auto gemm_ctx = [&](auto e0, f32x4 (&a0)[R], f32x4 (&a1)[R]) {
            constexpr int E0 = decltype(e0)::value;
            static_for<0, CKM>([&](auto k) {
                constexpr int KS = decltype(k)::value;
                if constexpr (CTX_REGS) {
                    mma(ic<E0 + 2 * KS>{}, cb[KS], a0);
                    mma(ic<E0 + 2 * KS + 1>{}, cb[KS], a1);
                } else {
                    u32x4 b[1][R];
                    load_b(s_ctx + (size_t)KS * R * kFragBytes, ic<1>{}, b);
                    mma(ic<E0 + 2 * KS>{}, b[0], a0);
                    mma(ic<E0 + 2 * KS + 1>{}, b[0], a1);
                }
                refill(ic<E0 + 2 * KS>{});
                refill(ic<E0 + 2 * KS + 1>{});
            });
        };

        // ---- initial layer: h = W_in x + b_in + relu(W_c ctx + b_c) ---------------------
        f32x4 hA[R], hB[R];
        {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
                if (BF16) {
                    bf16x8 b;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = xin[((8 * g + j) & 15) * COLS + 16 * r + c];
                        const __bf16 hi = (__bf16)v;
                        b[j] = g < 2 ? hi : (__bf16)(v - (float)hi);
                    }
                    accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, win[S::E_IN % W]), b, accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, win[(S::E_IN + 1) % W]), b, accB, 0, 0, 0);
                } else {
                    const f32x4 afA = __builtin_bit_cast(f32x4, win[S::E_IN % W]);
                    const f32x4 afB = __builtin_bit_cast(f32x4, win[(S::E_IN + 1) % W]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xv = xin[(4 * g + e) * COLS + 16 * r + c];
                        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(afA[e], xv, accA, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(afB[e], xv, accB, 0, 0, 0);
                    }
                }
                hA[r] = accA; hB[r] = accB;
            }
            refill(ic<S::E_IN>{});
            refill(ic<S::E_IN + 1>{});
            const f32x4 biA = load_bias(tA, kSlotIn), biB = load_bias(tB, kSlotIn);
#pragma unroll
            for (int r = 0; r < R; ++r) { hA[r] += biA; hB[r] += biB; }
            if constexpr (CKM == 0) {
                if (hoisted) {
#pragma unroll
                    for (int r = 0; r < R; ++r) { hA[r] += prA[0][r]; hB[r] += prB[0][r]; }
                }
            }
            if constexpr (CKM > 0) {
                f32x4 cA[R], cB[R];
                zero(cA); zero(cB);
                gemm_ctx(ic<S::E_CTX>{}, cA, cB);
                const f32x4 bcA = load_bias(tA, kSlotCtx), bcB = load_bias(tB, kSlotCtx);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hA[r][e] += fmaxf(cA[r][e] + bcA[e], 0.f);
                        hB[r][e] += fmaxf(cB[r][e] + bcB[e], 0.f);
                    }
            }
        }

        // ---- residual blocks ---------------------------------------------------------------
        static_for<0, 2>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            constexpr int EB = S::E_BLK + b * S::BLK;
            f32x4 tAv[R], tBv[R];
            u32x4 bk[HK][R];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) { tAv[r][e] = fmaxf(hA[r][e], 0.f); tBv[r][e] = fmaxf(hB[r][e], 0.f); }
            store_act(s_act0, tA, tAv);
            store_act(s_act0, tB, tBv);
            barrier();
            PF_LOAD_B(s_act0, ic<HK>{}, bk);
            zero(tAv); zero(tBv);
            gemm_pair(ic<EB>{}, ic<KHS>{}, kA, bk, tAv, tBv);
            {
                const f32x4 b0A = load_bias(tA, kSlotBlk + 3 * b), b0B = load_bias(tB, kSlotBlk + 3 * b);
                if constexpr (CKM == 0) {
                    if (additive) {          // t = W0 relu(h) + b0 + ctx_layer(ctx)
#pragma unroll
                        for (int r = 0; r < R; ++r) { tAv[r] += prA[1 + b][r]; tBv[r] += prB[1 + b][r]; }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        tAv[r][e] = fmaxf(tAv[r][e] + b0A[e], 0.f);
                        tBv[r][e] = fmaxf(tBv[r][e] + b0B[e], 0.f);
                    }
                if constexpr (DROP) {
                    // nflows MaskedResidualBlock.forward: dropout after the second activation.  C fragment: element e
                    // of lane (g, c) is sorted position 16 tile + 4 g + e of row row0 + 16 r + c
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t hr = drop_row_hash(p.drop_seed, (uint32_t)(row0 + 16 * r + c));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            tAv[r][e] *= drop_factor(hr, 2 * l + b, 16 * tA + 4 * g + e, p.drop_thresh, p.drop_scale);
                            tBv[r][e] *= drop_factor(hr, 2 * l + b, 16 * tB + 4 * g + e, p.drop_thresh, p.drop_scale);
                        }
                    }
                }
            }
            store_act(s_act1, tA, tAv);
            store_act(s_act1, tB, tBv);
            barrier();
            PF_LOAD_B(s_act1, ic<HK>{}, bk);
            zero(tAv); zero(tBv);
            gemm_pair(ic<EB + KHS>{}, ic<KHS>{}, kA, bk, tAv, tBv);
            const f32x4 b1A = load_bias(tA, kSlotBlk + 3 * b + 1), b1B = load_bias(tB, kSlotBlk + 3 * b + 1);
            if constexpr (CKM > 0) {
                f32x4 gA[R], gB[R];
                zero(gA); zero(gB);
                gemm_ctx(ic<EB + 2 * KHS>{}, gA, gB);
                const f32x4 bgA = load_bias(tA, kSlotBlk + 3 * b + 2), bgB = load_bias(tB, kSlotBlk + 3 * b + 2);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hA[r][e] += (tAv[r][e] + b1A[e]) * pf_sigmoid<FAST>(gA[r][e] + bgA[e]);
                        hB[r][e] += (tBv[r][e] + b1B[e]) * pf_sigmoid<FAST>(gB[r][e] + bgB[e]);
                    }
            } else if (hoisted && !additive) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    hA[r] += (tAv[r] + b1A) * prA[1 + b][r];
                    hB[r] += (tBv[r] + b1B) * prB[1 + b][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) { hA[r] += tAv[r] + b1A; hB[r] += tBv[r] + b1B; }
            }
        });

        // ---- final masked layer: wave w produces the raw spline parameters of features (w, D-1-w) --
        store_act(s_act0, tA, hA);          // no activation in front of the final layer
        store_act(s_act0, tB, hB);
        barrier();
        f32x4 pA[3][R], pB[3][R];
        {
            u32x4 bk[HK][R];
            PF_LOAD_B(s_act0, ic<HK>{}, bk);
            static_for<0, 3>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                zero(pA[q]); zero(pB[q]);
                gemm_pair(ic<S::E_OUT + q * KOS>{}, ic<KOS>{}, kOA, bk, pA[q], pB[q]);
            });
            // schedule padding: keep the window rolling up to the next multiple of W
            static_for<NF, NE - NF>([&](auto e) { refill(e); });
        }
        // park the next layer's biases (the same wave reads them back next layer)
        if (last_pass && lane < kBiasFloatsPerTile / 4) {
            float* nb = s_bias + (bsel ^ 1) * NT * kBiasFloatsPerTile;
            *reinterpret_cast<f32x4*>(nb + tA * kBiasFloatsPerTile + 4 * lane) = nbA;
            *reinterpret_cast<f32x4*>(nb + tB * kBiasFloatsPerTile + 4 * lane) = nbB;
        }
#ifndef PF_FWD_TOUCH
#define PF_FWD_TOUCH 0          // timing experiment: fragments of the NEXT layer pulled towards L2 under the spline
#endif
        if constexpr (!INV && PF_FWD_TOUCH > 0) {
            // no fragment is consumed during the spline, so the window does not move and the memory pipe idles: touch the
            // next layer's entries W .. W + T - 1 (one dword per 16 bytes of a lane = all 8 lines of a fragment, LDS-DMA
            // into the forward's unused x buffer: no register, nothing waits on it)
#pragma unroll
            for (int t = 0; t < PF_FWD_TOUCH; ++t)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)s_xb2, 4,
                                                         lane * 16, nbase + (W + t) * kFragBytes, 0, 0);
        }
        // ---- spline: transpose the parameters through LDS, one lane per (row, feature) pair ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            auto put = [&](int feat, int tile, const f32x4 (&pp)[3][R]) {
                if (feat < 0) return;
                float* dst = s_par + (size_t)(feat * 16 + c) * kParStride + 4 * g;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *reinterpret_cast<f32x4*>(dst + 16 * q) = pp[q][r] + load_bias(tile, kSlotOut + q);
            };
            put(fA, tA, pA);
            put(fB, tB, pB);
            barrier();
            if (tid < D * 16) {
                const int feat = tid >> 4, col = tid & 15;
                if constexpr (INV) {
                    const float yv = yb[feat * COLS + 16 * r + col];
                    float xv, ld; bool bad;
                    rqs_pair_inverse<FAST>(s_par + (size_t)tid * kParStride, yv, K, p, xv, ld, bad);
                    xc[feat * COLS + 16 * r + col] = xv;         // conditioner input of the next pass
                    if (last_pass) {
                        if (r == 0) ld_pair += ld; else s_ldb[feat * COLS + 16 * r + col] += ld;
                        bad_pair |= bad ? (1u << r) : 0u;
                        // undo this layer's ReversePermutation: previous layer's output position D-1-feat
                        yn[rv(feat) * COLS + 16 * r + col] = xv;
                    }
                } else {
                    const float xv = xin[feat * COLS + 16 * r + col];
                    if (p.u_save) {          // training: the backward re-evaluates each conditioner from this
                        const int64_t row = row0 + 16 * r + col;
                        if (row < p.batch) p.u_save[((int64_t)l * p.batch + row) * D + feat] = xv;
                    }
                    float y, ld;
                    if (PF_ABL(1)) { y = xv + s_par[(size_t)tid * kParStride]; ld = 0.f; }
                    else rqs_pair<FAST>(s_par + (size_t)tid * kParStride, xv, K, p, y, ld);
                    if (r == 0) ld_pair += ld; else s_ldb[feat * COLS + 16 * r + col] += ld;
                    // the next layer starts with ReversePermutation: position D-1-feat
                    xout[rv(feat) * COLS + 16 * r + col] = y;
                }
            }
            if (r + 1 < R) barrier();      // the transpose buffer is reused by the next column group
        }
        lbase = nbase;
        barrier();
        if (INV && last_pass) {
            // next (= previous) layer: its output is what we just assembled; estimate restarts at 0
            float* t = yb; yb = yn; yn = t;
            for (int s = tid; s < D * COLS; s += NW * 64) xc[s] = 0.f;
            barrier();
        }
    }

    // ---- epilogue: sum log-dets over features, base log-density, stores -----------------
    if (tid < D * 16) s_ldb[(tid >> 4) * COLS + (tid & 15)] = ld_pair;
    __syncthreads();
    if constexpr (INV) {
        // yb holds reverse(u_0) = x[:, ar_perm]; undo the autoregressive order (flows.py:648)
        if (tid < D * 16 && p.fail_flags && bad_pair) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((bad_pair >> r) & 1) {
                    const int64_t row = row0 + 16 * r + (tid & 15);
                    if (row < p.batch) atomicOr(p.fail_flags + row, 1u);
                }
        }
        if (tid < COLS) {
            const int64_t row = row0 + tid;
            if (row < p.batch) {
                float ld = 0.f;
                for (int f = 0; f < D; ++f) ld += s_ldb[f * COLS + tid];
                for (int d = 0; d < D; ++d) {
                    const int src = p.ar_perm ? p.ar_perm[d] : d;      // ar_inv_perm
                    if (p.z) p.z[row * D + d] = yb[src * COLS + tid];
                }
                if (p.logdet) p.logdet[row] = ld;
            }
        }
    } else {
    const float* zfin = (L.L & 1) ? s_xb1 : s_xb0;     // stored reversed (see above)
    float my_nll = 0.f, my_cnt = 0.f;                  // this lane's row for the in-kernel loss reduction
    if (p.zero_pair && blockIdx.x == 0 && tid < 2 * PF_REDUCE_SLOTS) p.zero_pair[tid] = 0.f;
    if (tid < COLS) {
        const int64_t row = row0 + tid;
        if (row < p.batch) {
            float ld = 0.f, q = 0.f, sls = 0.f;
            for (int f = 0; f < D; ++f) ld += s_ldb[f * COLS + tid];
            for (int d = 0; d < D; ++d) {
                const float zv = zfin[rv(d) * COLS + tid];
                if (p.log_sigma) {           // PSDScaledNormal.log_prob, flows.py:73-83
                    const float ls = p.log_sigma[row * D + d];
                    const float zs = zv / expf(ls);
                    q += zs * zs; sls += ls;
                } else {
                    q += zv * zv;
                }
                if (p.z) p.z[row * D + d] = zv;
            }
            if (p.logdet) p.logdet[row] = ld;
            // nll = -(log N(z; 0, diag(e^ls)^2) + logdet)
            my_nll = 0.5f * (q + 2.f * sls + (float)D * 1.8378770664093453f) - ld;
            my_cnt = 1.f;
            if (p.nll) p.nll[row] = my_nll;
        }
    }
    // (sum nll, rows) += this workgroup's rows: the rows sit in the first COLS <= 32 lanes of wave 0;
    // wave shuffle reduction, then one pair of float atomics per workgroup into slot (blockIdx mod PF_REDUCE_SLOTS):
    // 256 workgroups ending together on ONE address pair serialised for 4.6 us (102.5 vs 97.9 us per launch)
    static_assert(COLS <= 64, "the loss reduction assumes the rows sit in wave 0");
    if (p.nll_sum && tid < 64) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { my_nll += __shfl_xor(my_nll, o, 64); my_cnt += __shfl_xor(my_cnt, o, 64); }
        if (tid == 0) { float* acc = p.nll_sum + 2 * (blockIdx.x % PF_REDUCE_SLOTS); atomicAdd(acc, my_nll); atomicAdd(acc + 1, my_cnt); }
    }
    }
}

template <bool BF16, int NT, int R, int CKM, int DENSE, bool INV>
__global__ __launch_bounds__(NT * 32) void flow_kernel(const FwdParams p) {
    flow_body<BF16, NT, R, CKM, DENSE, INV, false>(p);
}
// training forward of a flow with dropout > 0 (nflows applies it inside every residual block in train mode)
template <bool BF16, int NT, int R, int CKM, int DENSE>
__global__ __launch_bounds__(NT * 32) void flow_train_kernel(const FwdParams p) {
    flow_body<BF16, NT, R, CKM, DENSE, false, true>(p);
}

// ---- host side ---------------------------------------------------------------------------
inline size_t fwd_lds_bytes(const FlowPlan& L, int R) {
    const size_t pb = std::max((size_t)L.HK * R * kFragBytes, (size_t)L.D * 16 * kParStride * sizeof(float));
    return (size_t)L.CKM * R * kFragBytes + (size_t)L.HK * R * kFragBytes + pb
         + (size_t)(3 * 16 * 16 * R + 2 * L.NT * kBiasFloatsPerTile + L.D * 16 * R) * sizeof(float);
}

template <bool BF16, int NT, int R, int CKM, int DENSE, bool INV, bool DROP>
inline int launch_variant(const FwdParams& p, hipStream_t s) {
    const unsigned grid = (unsigned)((p.batch + 16 * R - 1) / (16 * R));
    const size_t lds = fwd_lds_bytes(p.plan, R);
    auto kern = [] {
        if constexpr (DROP) return flow_train_kernel<BF16, NT, R, CKM, DENSE>;
        else return flow_kernel<BF16, NT, R, CKM, DENSE, INV>;
    }();
    if (lds > 64 * 1024 &&
        !opt_in_lds(reinterpret_cast<const void*>(kern), (int)lds))
        return PF_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 32), lds, s, p);
    return launch_status();
}

// all built variants of one (precision, NT): R in {1,2} x {masked}, R = 1 x {dense}
template <bool BF16, int NT, int CKM, bool INV, bool DROP = false>
inline int launch_ckm(const FwdParams& p, int R, hipStream_t s) {
    if (p.plan.dense == 1) return launch_variant<BF16, NT, 1, CKM, 1, INV, DROP>(p, s);
    if constexpr (BF16 && NT == 16) {
        if (p.plan.dense == 2) {
            if (R >= 2) return launch_variant<BF16, NT, 2, CKM, 2, INV, DROP>(p, s);
            return launch_variant<BF16, NT, 1, CKM, 2, INV, DROP>(p, s);
        }
    }
    if (p.plan.dense != 0) return PF_ERR_UNSUPPORTED;
    if constexpr (BF16 && NT == 16 && !INV && !DROP) {   // large batches, LeanNPE-sized hidden width: 48 rows per workgroup
        if (R == 3) return launch_variant<BF16, NT, 3, CKM, 0, INV, DROP>(p, s);
    }
    if (R >= 2) return launch_variant<BF16, NT, 2, CKM, 0, INV, DROP>(p, s);
    return launch_variant<BF16, NT, 1, CKM, 0, INV, DROP>(p, s);
}

}  // namespace pf
