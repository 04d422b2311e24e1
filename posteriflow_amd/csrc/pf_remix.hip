// pf_remix.hip -- training-example assembly on the GPU (reference: RemixDataset.__getitem__,
// experiments/remix_data.py:218-299, the deterministic part).  Given the per-example decisions
// (noise row, per-signal amplitude factor and circular shift, dropped detectors) it produces
//   strain[b,d,t] = f32(noise[row_b,d,t]) + sum_k  s_bk * f32(signal[start_b+k, d, (t - ds_bk) mod T])
// with the sum over k taken in storage order in fp32 without contraction (so it is bit-identical to
// the reference's numpy arithmetic), dropped detectors replaced by the fill rows, and the network SNR
//   sqrt(sum over kept detectors and t of sig_sum^2)                       (remix_data.py:286).
// HBM-bound gather: per example 98 KB of fp16 noise + 98 KB per signal in, 196 KB of fp32 strain out.
// One workgroup = one (example, detector, 2048-sample chunk); a thread owns 8 consecutive samples:
// 16-byte fp16 loads (two aligned vectors + a uniform funnel shift for the circular offset), 2 x 16-byte
// fp32 stores.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <cstdint>

#include "../../include/pf_hip.h"

// bit-exactness with the numpy arithmetic needs separately rounded multiply and add.  The pragma
// governs expressions written in THIS file (header intrinsics such as __fmul_rn are plain `a * b`
// compiled under the default contract(fast) and do get fused into v_fma_mix_f32).
#pragma clang fp contract(off)

namespace pf {
namespace {
constexpr int kT = 16384, kChunk = 2048, kThreads = 256, kChunks = kT / kChunk, kMaxSig = 5;

struct RemixArgs {
    const __half* noise;
    const __half* signals;
    int64_t n_noise, n_signals, n_fill;
    const int64_t* noise_row;
    const int64_t* sig_start;
    const int32_t* nsig;
    const float* scale;
    const int32_t* shift;
    const int32_t* fill_row;
    const float* fill;
    float* strain;
    float* sig_sum;
    double* part;      // [batch][3][kChunks] sum of sig_sum^2 (0 for dropped detectors)
};

// 8 consecutive fp16 samples of a circularly shifted row, starting at (t0 - sh) mod T with t0 a multiple
// of 8: the two aligned 16-byte vectors that cover the window, funnel-shifted by the (workgroup-uniform)
// misalignment.  Both vectors wrap independently because T is a multiple of 8.
struct Half8 { uint32_t w[4]; };
__device__ __forceinline__ Half8 load_shifted(const __half* row, int t0, int sh) {
    const int src = (t0 - sh) & (kT - 1);
    const int a0 = src & ~7, a1 = (a0 + 8) & (kT - 1), o = src & 7;
    const uint4 lo = *reinterpret_cast<const uint4*>(row + a0);
    const uint4 hi = *reinterpret_cast<const uint4*>(row + a1);
    const uint32_t c[9] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, 0u};
    Half8 r;
    const int byte_off = (o & 1) * 2;
    switch (o >> 1) {       // uniform across the workgroup: one of four straight-line variants
    case 0:
#pragma unroll
        for (int j = 0; j < 4; ++j) r.w[j] = __builtin_amdgcn_alignbyte(c[j + 1], c[j], byte_off);
        break;
    case 1:
#pragma unroll
        for (int j = 0; j < 4; ++j) r.w[j] = __builtin_amdgcn_alignbyte(c[j + 2], c[j + 1], byte_off);
        break;
    case 2:
#pragma unroll
        for (int j = 0; j < 4; ++j) r.w[j] = __builtin_amdgcn_alignbyte(c[j + 3], c[j + 2], byte_off);
        break;
    default:
#pragma unroll
        for (int j = 0; j < 4; ++j) r.w[j] = __builtin_amdgcn_alignbyte(c[j + 4], c[j + 3], byte_off);
        break;
    }
    return r;
}
__device__ __forceinline__ float half_at(const Half8& v, int i) {
    const uint32_t w = v.w[i >> 1];
    return __half2float(__ushort_as_half(static_cast<unsigned short>((i & 1) ? (w >> 16) : (w & 0xffffu))));
}

__global__ __launch_bounds__(kThreads) void remix_kernel(RemixArgs a) {
    const int chunk = blockIdx.x % kChunks;
    const int det = (blockIdx.x / kChunks) % 3;
    const int64_t b = blockIdx.x / (kChunks * 3);
    const int tid = threadIdx.x;
    const int t0 = chunk * kChunk + tid * 8;          // this thread's 8 consecutive samples

    int ns = a.nsig[b];
    ns = ns < 0 ? 0 : (ns > kMaxSig ? kMaxSig : ns);
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    for (int k = 0; k < ns; ++k) {                    // storage order: the fp32 sum is order-dependent
        const int64_t row = a.sig_start[b] + k;
        if (row < 0 || row >= a.n_signals) continue;
        const float sc = a.scale[b * kMaxSig + k];
        const int sh = ((a.shift[b * kMaxSig + k] % kT) + kT) % kT;
        const Half8 v = load_shifted(a.signals + (row * 3 + det) * kT, t0, sh);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float scaled = half_at(v, i) * sc;  // rounded product
            s[i] = s[i] + scaled;                     // then rounded sum
        }
    }
    const int64_t nrow = a.noise_row[b];
    const int fr = a.fill_row ? a.fill_row[b * 3 + det] : -1;
    const bool dropped = fr >= 0;
    const int64_t base = (b * 3 + det) * kT + t0;
    float out[8];
    if (dropped) {
        const bool have = fr < a.n_fill;
        const float4* fp = reinterpret_cast<const float4*>(a.fill + static_cast<int64_t>(have ? fr : 0) * kT + t0);
        const float4 f0 = have ? fp[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 f1 = have ? fp[1] : make_float4(0.f, 0.f, 0.f, 0.f);
        out[0] = f0.x; out[1] = f0.y; out[2] = f0.z; out[3] = f0.w;
        out[4] = f1.x; out[5] = f1.y; out[6] = f1.z; out[7] = f1.w;
    } else if (nrow >= 0 && nrow < a.n_noise) {
        const uint4 q = *reinterpret_cast<const uint4*>(a.noise + (nrow * 3 + det) * kT + t0);
        const Half8 nv{{q.x, q.y, q.z, q.w}};
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = half_at(nv, i) + s[i];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = 0.f + s[i];
    }
    float4* dst = reinterpret_cast<float4*>(a.strain + base);
    dst[0] = make_float4(out[0], out[1], out[2], out[3]);
    dst[1] = make_float4(out[4], out[5], out[6], out[7]);
    if (a.sig_sum) {
        float4* d2 = reinterpret_cast<float4*>(a.sig_sum + base);
        d2[0] = make_float4(s[0], s[1], s[2], s[3]);
        d2[1] = make_float4(s[4], s[5], s[6], s[7]);
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += static_cast<double>(s[i]) * static_cast<double>(s[i]);
    // workgroup sum (fixed order: lanes by xor-shuffle, then the 4 waves in order)
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double wsum[kThreads / 64];
    if ((tid & 63) == 0) wsum[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) a.part[(b * 3 + det) * kChunks + chunk] = dropped ? 0.0 : ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

__global__ void remix_snr_kernel(const double* part, float* snr, int64_t batch) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double s = 0.0;
    for (int i = 0; i < 3 * kChunks; ++i) s += part[b * 3 * kChunks + i];
    snr[b] = static_cast<float>(sqrt(s));
}
}  // namespace

int64_t remix_workspace_bytes(int64_t batch) { return batch * 3 * kChunks * static_cast<int64_t>(sizeof(double)); }

int remix_forward(const void* noise, int64_t n_noise, const void* signals, int64_t n_signals,
                  const int64_t* noise_row, const int64_t* sig_start, const int32_t* nsig, const float* scale,
                  const int32_t* shift, const int32_t* fill_row, const float* fill, int64_t n_fill, int64_t batch,
                  float* strain, float* sig_sum, float* net_snr, void* ws, hipStream_t s) {
    RemixArgs a{static_cast<const __half*>(noise), static_cast<const __half*>(signals), n_noise, n_signals, n_fill,
                noise_row, sig_start, nsig, scale, shift, fill_row, fill, strain, sig_sum,
                static_cast<double*>(ws)};
    remix_kernel<<<dim3(static_cast<unsigned>(batch * 3 * kChunks)), dim3(kThreads), 0, s>>>(a);
    if (net_snr)
        remix_snr_kernel<<<dim3(static_cast<unsigned>((batch + 255) / 256)), dim3(256), 0, s>>>(
            static_cast<const double*>(ws), net_snr, batch);
    return launch_status();
}
}  // namespace pf
