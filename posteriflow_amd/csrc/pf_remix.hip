// pf_remix.hip -- training-example assembly on the GPU (reference: RemixDataset.__getitem__,
// experiments/remix_data.py:218-299, the deterministic part).  Given the per-example decisions
// (noise row, per-signal amplitude factor and circular shift, dropped detectors) it produces
//   strain[b,d,t] = f32(noise[row_b,d,t]) + sum_k  s_bk * f32(signal[start_b+k, d, (t - ds_bk) mod T])
// with the sum over k taken in storage order in fp32 without contraction (so it is bit-identical to
// the reference's numpy arithmetic), dropped detectors replaced by the fill rows, and the network SNR
//   sqrt(sum over kept detectors and t of sig_sum^2)                       (remix_data.py:286).
// HBM-bound gather: per example 98 KB of fp16 noise + 98 KB per signal in, 196 KB of fp32 strain out.
// One workgroup = one (example, detector, 2048-sample chunk); a wave instruction touches 64
// consecutive samples (128 B of fp16 in, 256 B of fp32 out).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

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

__global__ __launch_bounds__(kThreads) void remix_kernel(RemixArgs a) {
    const int chunk = blockIdx.x % kChunks;
    const int det = (blockIdx.x / kChunks) % 3;
    const int64_t b = blockIdx.x / (kChunks * 3);
    const int tid = threadIdx.x;

    int ns = a.nsig[b];
    ns = ns < 0 ? 0 : (ns > kMaxSig ? kMaxSig : ns);
    const __half* sp[kMaxSig];
    float sc[kMaxSig];
    int sh[kMaxSig];
#pragma unroll
    for (int k = 0; k < kMaxSig; ++k) {
        sp[k] = nullptr; sc[k] = 1.f; sh[k] = 0;
        if (k < ns) {
            const int64_t row = a.sig_start[b] + k;
            if (row >= 0 && row < a.n_signals) sp[k] = a.signals + (row * 3 + det) * kT;
            sc[k] = a.scale[b * kMaxSig + k];
            sh[k] = ((a.shift[b * kMaxSig + k] % kT) + kT) % kT;
        }
    }
    const int64_t nrow = a.noise_row[b];
    const __half* np = (nrow >= 0 && nrow < a.n_noise) ? a.noise + (nrow * 3 + det) * kT : nullptr;
    const int fr = a.fill_row ? a.fill_row[b * 3 + det] : -1;
    const bool dropped = fr >= 0;
    const float* fp = (dropped && fr < a.n_fill) ? a.fill + static_cast<int64_t>(fr) * kT : nullptr;

    const int64_t base = (b * 3 + det) * kT;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < kChunk / kThreads; ++j) {
        const int t = chunk * kChunk + j * kThreads + tid;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxSig; ++k)
            if (sp[k]) {
                const float scaled = __half2float(sp[k][(t - sh[k]) & (kT - 1)]) * sc[k];   // rounded product
                s = s + scaled;                                                             // then rounded sum
            }
        acc += static_cast<double>(s) * static_cast<double>(s);
        const float out = dropped ? (fp ? fp[t] : 0.f) : (np ? __half2float(np[t]) : 0.f) + s;
        a.strain[base + t] = out;
        if (a.sig_sum) a.sig_sum[base + t] = s;
    }
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
    return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}
}  // namespace pf
