// pf_embed.hip -- the convolutional stem of PosteriFlow's strain embedding on gfx950.
//
// Replaces (reference src/ahsd/models/lean_npe.py):
//   :207      nan_to_num / clamp of the raw whitened strain
//   :210-212  per-detector, per-window log mean-square ("energy branch")
//   :216-217  asinh -> Conv1d(1,32,k64,s8) GELU Conv1d(32,64,k16,s4) GELU
//             Conv1d(64,128,k8,s4) GELU Conv1d(128,192,k4,s2) GELU -> tokens [N, 61, 192]
//
// Every convolution is an implicit GEMM.  Activations are kept POSITION-MAJOR ([pos][ch], channels
// contiguous), so the im2col row of output position p -- taps p*s .. p*s+k-1, all channels -- is one
// CONTIGUOUS span of k*Cin elements starting at p*s*Cin: the GEMM's A rows are overlapping windows of
// the activation array and nothing is ever materialised.  The reduction length is 64 (conv1) or 512
// (conv2-4).  As in the flow kernels the product is formed transposed,
//   out^T[ch, pos] = W[ch, kk] . act^T[kk, pos],
// so the weights are the MFMA A operand (pre-packed fragments, streamed from L2 exactly once per
// workgroup by the wave that owns that channel tile) and 16 output positions are the MFMA columns.
// The input span of a workgroup is staged once in LDS, de-interleaved by (position mod stride) and
// by 64-byte channel blocks, which makes every B-fragment read of a wave one contiguous 1 KiB.
// conv1 reads the raw fp32 strain with coalesced 16-byte loads and fuses sanitise + asinh + the
// log-energy windows of the same samples.
//
//   bf16 mode: v_mfma_f32_16x16x32_bf16, activations stored bf16 between layers (fp32 tokens out)
//   f32  mode: v_mfma_f32_16x16x4_f32,   activations fp32 throughout (parity mode)
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "../../include/pf_hip.h"
#include "pf_math.h"

namespace pf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kStemLayers = 4;
#ifndef PF_STEM_HOIST
#define PF_STEM_HOIST 0
#endif
#ifndef PF_STEM_ABLATE
#define PF_STEM_ABLATE 0      // experiments on conv3 / conv4 (side builds only): 1 no staging loads, 2 no k-loop, 4 no stores
#endif
struct StemLayer { int cin, cout, kw, stride, lin, lout; };
// 16384 -> 2041 -> 507 -> 125 -> 61   (lean_npe.py:158-163)
__host__ __device__ constexpr StemLayer stem_layer(int i) {
    return i == 0 ? StemLayer{1, 32, 64, 8, 16384, 2041}
         : i == 1 ? StemLayer{32, 64, 16, 4, 2041, 507}
         : i == 2 ? StemLayer{64, 128, 8, 4, 507, 125}
                  : StemLayer{128, 192, 4, 2, 125, 61};
}

struct ConvParams {
    const void* in;         // FIRST: fp32 strain [N][16384]; else activations [N][lin][cin]
    const u32x4* wfrags;    // [cout/16][ksteps][64 lanes]
    const float* bias;      // [cout]
    void* out;              // [N][lout][cout]  (LAST: fp32)
    float* log_energy;      // FIRST / FUSE: [N][16]
    int64_t n_seq;
    const u32x4* wfrags0;   // FUSE (conv1 computed inside conv2's staging): conv1's fragments, bias; `in` is the strain
    const float* bias0;
    // training forward (pf_embed_train_forward): what the backward kernels read
    void* dact;             // [N][lout][cout] activation type: gelu'(pre-activation), or null
    void* sig;              // FIRST: [N][16384] activation type: asinh(sanitised strain), the im2col source of conv1's weight gradient
};

__device__ __forceinline__ float gelu_exact(float x) {         // nn.GELU() default: erf form
    return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
}

// rows per (position mod stride, channel block) of a layer's LDS input image.  (Odd counts -- consecutive row groups 16 banks
// apart for the image STORES -- were tried in round 4: SQ_LDS_BANK_CONFLICT of the fused kernel stayed at 169.08 M cycles per
// 12 288 sequences and the time did not move; the counted cycles belong to the 16-byte fragment reads, LABLOG R4.12.)
__host__ __device__ constexpr int image_rows(int span, int stride) { return (span + stride - 1) / stride + 1; }

// lean_npe.py:207: nan -> 0, +-inf -> +-100, clamp to [-100, 100] (v_med3 + one compare / select)
__device__ __forceinline__ float sanitize(float x) {
    const float c = __builtin_amdgcn_fmed3f(x, -100.f, 100.f);
    return (x != x) ? 0.f : c;
}

// slot swizzle of a 64-byte LDS row (4 slots of 16 B): (4 - (row >> 2)) & 3, see the staging code
__device__ __forceinline__ int row_swz(int q) { return (4 - ((q >> 2) & 3)) & 3; }

// One workgroup: CG*16 output positions of one sequence; wave w owns channel tiles w*TPW..
template <bool BF16, int LAYER, int CG, int NWAVES, bool FUSE = false>
__global__ __launch_bounds__(NWAVES * 64) void conv_gemm_kernel(const ConvParams p) {
    static_assert(!FUSE || LAYER == 1, "only conv1 -> conv2 is fused");
    constexpr StemLayer SL = stem_layer(LAYER);
    constexpr bool FIRST = LAYER == 0, LAST = LAYER == kStemLayers - 1;
    constexpr int CIN = SL.cin, COUT = SL.cout, KW = SL.kw, S = SL.stride, LIN = SL.lin, LOUT = SL.lout;
    constexpr int KK = KW * CIN;                       // reduction length (64 or 512)
    constexpr int KSTEP = BF16 ? 32 : 16;              // kk per fragment
    constexpr int NKS = KK / KSTEP;
    constexpr int ESZ = BF16 ? 2 : 4;                  // activation element size
    constexpr int CHB = 64 / ESZ;                      // channels per 64-byte LDS row (32 | 16)
    constexpr int TPW = (COUT / 16) / NWAVES;          // channel tiles per wave
    static_assert(TPW * NWAVES * 16 == COUT, "channel tiles must divide over the waves");
    constexpr int P = CG * 16;                         // output positions per workgroup
    constexpr int SPAN = (P - 1) * S + KW;             // input positions needed
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t n = blockIdx.y;
    const int p0 = blockIdx.x * P;                     // first output position
    const int in0 = p0 * S;                            // first input position of the span

    // weight fragments: register double buffer in chunks of CH k-steps per tile; bias (bf16: the accumulators start from it,
    // the fp32 parity mode adds it last, as the reference does)
    constexpr int CH0 = TPW > 1 ? 4 : 8;
    constexpr int CH = NKS < CH0 ? NKS : CH0;
    const float* bias = p.bias;
    const u32x4* wf[TPW];
    f32x4 b4[TPW];
    f32x4 acc[TPW][CG];
    u32x4 a0[TPW][CH], a1[TPW][CH];
    auto prime = [&]() {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = wave * TPW + t;
            wf[t] = p.wfrags + (size_t)tile * NKS * 64 + lane;
            b4[t] = *reinterpret_cast<const f32x4*>(bias + tile * 16 + 4 * g);
#pragma unroll
            for (int k = 0; k < CH; ++k) a0[t][k] = wf[t][(size_t)k * 64];
        }
    };
#if PF_STEM_HOIST
    // fused kernel: conv2's first weight chunk is requested before the strain is staged (its L2 round trip otherwise sits
    // between the conv1 phase and the first conv2 MFMA of every workgroup)
    if constexpr (FUSE) { prime(); __builtin_amdgcn_sched_barrier(0); }
#endif

    // ---- stage the input span ------------------------------------------------------------------
    if constexpr (FIRST) {
        // raw strain: coalesced float4 loads; sanitise (lean_npe.py:207), energy of the 2048 samples
        // this workgroup "owns", asinh, store as a contiguous signal in LDS
        const float* src = reinterpret_cast<const float*>(p.in) + n * LIN;
        float sq[2] = {0.f, 0.f};
        // (one chunk per iteration on purpose: this layer runs 16 small workgroups per CU, which hide the load
        // latency; requesting all chunks first costs registers = occupancy and measured 1.5x slower here)
        for (int i = tid * 4; i < SPAN; i += NWAVES * 64 * 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in0 + i + 3 < LIN) v = *reinterpret_cast<const f32x4*>(src + in0 + i);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) if (in0 + i + e < LIN) v[e] = src[in0 + i + e];
            float s4 = 0.f;                                                // a chunk never straddles a window (i % 4 == 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = sanitize(v[e]);
                s4 += x * x;
                v[e] = BF16 ? asinh_fast(x) : asinhf(x);
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) sq[w] += ((i >> 10) == w && i < P * S) ? s4 : 0.f;   // own range: 2 windows of 1024
            const bool keep = p.sig && i < P * S && in0 + i + 3 < LIN;     // this workgroup's own 2048 samples
            if (BF16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(smem + (size_t)i * 2) = o;
                if (keep) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.sig) + n * LIN + in0 + i) = o;
            } else {
                *reinterpret_cast<f32x4*>(smem + (size_t)i * 4) = v;
                if (keep) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.sig) + n * LIN + in0 + i) = v;
            }
        }
        // window energies: wave shuffle reduction, then one atomic per wave into LDS
        float* s_e = reinterpret_cast<float*>(smem + (((size_t)SPAN * ESZ + 15) & ~(size_t)15));
        if (tid < 2) s_e[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            float v = sq[w];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0) atomicAdd(&s_e[w], v);
        }
        __syncthreads();
        if (tid < 2 && p.log_energy) {
            const int w = 2 * blockIdx.x + tid;
            if (w < 16) p.log_energy[n * 16 + w] = logf(s_e[tid] * (1.f / 1024.f) + 1e-8f);
        }
    } else if constexpr (FUSE) {
        // conv1 (+ sanitise, asinh, log-energy windows, GELU) evaluated HERE for the SPAN conv1 outputs this
        // workgroup's conv2 positions need, written straight into conv2's LDS input image: the 1.6 GB conv1
        // activation never exists in HBM.  (1) strain span -> contiguous signal in LDS, (2) conv1 as MFMA over
        // 16-position tiles (both channel tiles per wave, weights in registers), (3) image rows.
        constexpr StemLayer L0 = stem_layer(0);
        constexpr int S0 = L0.stride, KW0 = L0.kw, LIN0 = L0.lin, LOUT0 = L0.lout;
        constexpr int CB = CIN / CHB;
        constexpr int Q = image_rows(SPAN, S);
        constexpr int NPT = (SPAN + 15) / 16;                       // conv1 position tiles
        constexpr int NSIG = (16 * NPT - 1) * S0 + KW0;             // strain samples those tiles read
        constexpr int NKS0 = KW0 / KSTEP;
        constexpr size_t IMG = (size_t)S * CB * Q * 64;
        char* sig = smem + IMG;
        float* s_e = reinterpret_cast<float*>(sig + (((size_t)NSIG * ESZ + 15) & ~(size_t)15));
        const float* src = reinterpret_cast<const float*>(p.in) + n * LIN0;
        const int s0 = in0 * S0;                                   // first strain sample of the span
        constexpr int OWN = P * S * S0;                            // samples this workgroup owns (energy windows)
        constexpr int NWIN = OWN / 1024;
        float sq[NWIN];
#pragma unroll
        for (int w = 0; w < NWIN; ++w) sq[w] = 0.f;
        for (int i = tid * 4; i < NSIG; i += NWAVES * 64 * 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s0 + i + 3 < LIN0) v = *reinterpret_cast<const f32x4*>(src + s0 + i);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) if (s0 + i + e < LIN0) v[e] = src[s0 + i + e];
            float s4 = 0.f;                                                // a chunk never straddles a window (i % 4 == 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = sanitize(v[e]);
                s4 += x * x;
                v[e] = BF16 ? asinh_fast(x) : asinhf(x);
            }
#pragma unroll
            for (int w = 0; w < NWIN; ++w) sq[w] += ((i >> 10) == w && i < OWN) ? s4 : 0.f;
            if (BF16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(sig + (size_t)i * 2) = o;
            } else {
                *reinterpret_cast<f32x4*>(sig + (size_t)i * 4) = v;
            }
        }
        if (tid < NWIN) s_e[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            float v = sq[w];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0) atomicAdd(&s_e[w], v);
        }
        // conv1 on the staged signal
        u32x4 a1f[2][NKS0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < NKS0; ++ks) a1f[t][ks] = p.wfrags0[(size_t)(t * NKS0 + ks) * 64 + lane];
        f32x4 b1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b1[t] = *reinterpret_cast<const f32x4*>(p.bias0 + 16 * t + 4 * g);
        for (int pt = wave; pt < NPT; pt += NWAVES) {
            // bf16: the accumulators start from the bias (one packed add per pair less in the epilogue); the fp32 parity
            // mode keeps the reference's order (bias last)
            f32x4 acc1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc1[t] = BF16 ? b1[t] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS0; ++ks) {
                const int idx = S0 * (16 * pt + c) + KSTEP * ks + (KSTEP / 4) * g;
                const u32x4 b = *reinterpret_cast<const u32x4*>(sig + (size_t)idx * ESZ);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (BF16) {
                        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a1f[t][ks]), __builtin_bit_cast(bf16x8, b), acc1[t], 0, 0, 0);
                    } else {
                        const f32x4 af = __builtin_bit_cast(f32x4, a1f[t][ks]), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4)
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q4], bf[q4], acc1[t], 0, 0, 0);
                    }
                }
            }
            const int pos = 16 * pt + c;                            // conv1 output position within the span
            const int r = pos % S, q = pos / S;
            if (q < Q) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 v;
                    if constexpr (BF16) {
                        v = gelu_erf_fast4(acc1[t]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_exact(acc1[t][e] + b1[t][e]);
                    }
                    const bool pad = in0 + pos >= LOUT0;                            // beyond conv1's output: zero padding
                    const int ch = 16 * t + 4 * g;                                  // first of this lane's 4 channels
                    const int cb = ch / CHB, byte = (ch % CHB) * ESZ;
                    char* dst = smem + ((size_t)((r * CB + cb) * Q + q) * 64) + ((((byte >> 4) ^ row_swz(q)) << 4) | (byte & 15));
                    if constexpr (BF16) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                        u32x2 ow = __builtin_bit_cast(u32x2, o);                    // (the select on the packed pair: 2 instead of 4)
                        ow[0] = pad ? 0u : ow[0]; ow[1] = pad ? 0u : ow[1];
                        *reinterpret_cast<u32x2*>(dst) = ow;
                    } else {
                        if (pad) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(dst) = v;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < NWIN && p.log_energy) {
            const int w = NWIN * blockIdx.x + tid;
            if (w < 16) p.log_energy[n * 16 + w] = logf(s_e[tid] * (1.f / 1024.f) + 1e-8f);
        }
    } else {
        // activations [pos][CIN] -> LDS rows (r = pos % S, channel block, q = pos / S), 64 B each
        constexpr int CB = CIN / CHB;                  // channel blocks per position
        constexpr int Q = image_rows(SPAN, S);      // rows per (r, block)
        const char* src = reinterpret_cast<const char*>(p.in) + ((size_t)n * LIN) * CIN * ESZ;
        constexpr int CHUNKS = CIN * ESZ / 16;         // 16-byte chunks per position
        // all of a thread's 16-byte chunks are requested before the first one is stored: a loop that loads
        // and stores one chunk per iteration pays one HBM latency per chunk (17 in a row for conv2)
        constexpr int NTHR = NWAVES * 64;
        constexpr int ITERS = (SPAN * CHUNKS + NTHR - 1) / NTHR;
        u32x4 v[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tid + it * NTHR;
            const int pos = i / CHUNKS, ch16 = i - pos * CHUNKS;
            v[it] = u32x4{0u, 0u, 0u, 0u};
            if (i < SPAN * CHUNKS && in0 + pos < LIN && !((PF_STEM_ABLATE & 1) && LAYER >= 2 && p.n_seq > 0))
                v[it] = *reinterpret_cast<const u32x4*>(src + ((size_t)(in0 + pos) * CIN * ESZ) + ch16 * 16);
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tid + it * NTHR;
            if (i < SPAN * CHUNKS) {
                const int pos = i / CHUNKS, ch16 = i - pos * CHUNKS;
                const int r = pos % S, q = pos / S;
                const int cb = ch16 / 4, sub = ch16 & 3;   // 4 chunks per 64-byte row
                // 16-byte slot XOR-swizzled by the row: ds_read_b128 serves lanes in the groups {0-3,12-15,20-27}, ...
                // (8 lanes of k-group g, 8 of g+1), which collide pairwise on plain 64-byte rows
                *reinterpret_cast<u32x4*>(smem + ((size_t)((r * CB + cb) * Q + q) * 64) + ((sub ^ row_swz(q)) << 4)) = v[it];
            }
        }
        __syncthreads();
    }

    // B fragment (ks, column group cg): 8 (bf16) / 4 (fp32) consecutive kk for output position
    // p0 + 16 cg + c at k-slot g
    auto bfrag = [&](int ks, int cg) -> u32x4 {
        if constexpr (FIRST) {
            // kk = tap: sample index S*p + KSTEP*ks + (KSTEP/4)*g .. ; contiguous signal
            const int idx = S * (16 * cg + c) + KSTEP * ks + (KSTEP / 4) * g;
            return *reinterpret_cast<const u32x4*>(smem + (size_t)idx * ESZ);
        } else {
            constexpr int CB = CIN / CHB;
            constexpr int Q = image_rows(SPAN, S);
            const int kk0 = KSTEP * ks;                // kk = tap * CIN + ch
            const int tap = kk0 / CIN, cb = (kk0 % CIN) / CHB;
            const int r = tap % S, q = 16 * cg + c + tap / S;
            return *reinterpret_cast<const u32x4*>(smem + ((size_t)((r * CB + cb) * Q + q) * 64) + ((g ^ row_swz(q)) << 4));
        }
    };

    // A wave's TPW channel tiles run TOGETHER (every B fragment read from LDS feeds TPW MFMAs, every weight fragment is read
    // from L2 once per workgroup); the dispatched geometries have one tile per wave.
#if PF_STEM_HOIST
    if constexpr (!FUSE) prime();
#else
    prime();
#endif
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) acc[t][cg] = BF16 ? b4[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    // fully unrolled: with the k-step a compile-time constant the LDS address of every B fragment is an
    // immediate offset from one per-lane base (tap / channel-block arithmetic folds away)
#pragma unroll
    for (int k0 = 0; k0 < (((PF_STEM_ABLATE & 2) && LAYER >= 2) ? 0 : NKS); k0 += 2 * CH) {
        if (k0 + CH < NKS) {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int k = 0; k < CH; ++k) a1[t][k] = wf[t][(size_t)(k0 + CH + k) * 64];
        }
        auto run = [&](const u32x4 (&a)[TPW][CH], int kb) {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                    const u32x4 b = bfrag(kb + k, cg);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        if (BF16) {
                            acc[t][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a[t][k]), __builtin_bit_cast(bf16x8, b), acc[t][cg], 0, 0, 0);
                        } else {
                            const f32x4 af = __builtin_bit_cast(f32x4, a[t][k]), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[t][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc[t][cg], 0, 0, 0);
                        }
                    }
                }
            }
        };
        run(a0, k0);
        if (k0 + CH < NKS) {
            if (k0 + 2 * CH < NKS) {
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int k = 0; k < CH; ++k) a0[t][k] = wf[t][(size_t)(k0 + 2 * CH + k) * 64];
            }
            run(a1, k0 + CH);
        }
    }
    // epilogue: bias + GELU, 4 consecutive channels of one position per lane
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave * TPW + t;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const int pos = p0 + 16 * cg + c;
            if (pos < LOUT && !((PF_STEM_ABLATE & 4) && LAYER >= 2 && p.n_seq > 0)) {
                f32x4 v;
                if constexpr (BF16) {
                    v = gelu_erf_fast4(acc[t][cg]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_exact(acc[t][cg][e] + b4[t][e]);
                }
                const size_t off = ((size_t)n * LOUT + pos) * COUT + tile * 16 + 4 * g;
                if (p.dact) {                                       // training: gelu'(pre-activation) for the backward
                    f32x4 dv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float ye, de;
                        if constexpr (BF16) gelu_fast_pair(acc[t][cg][e], ye, de);
                        else de = gelu_grad_f32(acc[t][cg][e] + b4[t][e]);
                        dv[e] = de;
                    }
                    if constexpr (BF16) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)dv[e];
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.dact) + off) = o;
                    } else {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dact) + off) = dv;
                    }
                }
                if (BF16 && !LAST) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + off) = o;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + off) = v;
                }
            }
        }
    }
}

// ---- conv3 / conv4, bf16: persistent workgroups --------------------------------------------------
// The per-sequence workgroups above run load -> LDS -> k-loop -> GELU -> store as one latency chain, and with two of them per
// CU the three phases measured additive (no loads -0.21 ms, no k-loop -0.20, no GELU + stores -0.27 of conv3 + conv4's 0.76 ms
// per 12 288 sequences, LABLOG R4.12).  Here ONE workgroup per CU walks its share of the sequences:
//   * the wave's weight tile (16 k-steps = 64 VGPRs) is read from L2 once per workgroup and stays in registers;
//   * the input span of the sequence after the next is in flight (registers, buffer loads whose range check supplies the zero
//     padding) while the next one's image sits in LDS;
//   * the k-loop of sequence i + 1 (MFMA + LDS reads) and the epilogue of sequence i (GELU on VALU, stores) are ONE
//     branch-free block over two accumulator sets -- the stores of positions >= LOUT are dropped by the buffer range check
//     instead of an exec-mask branch -- so the scheduler can put the epilogue's VALU work into the MFMA / LDS shadows;
//   * per sequence: barrier -> staged registers to LDS -> request -> barrier -> that block.  The only VMEM wait of the loop
//     (in front of the LDS stores) meets loads requested a whole block earlier.
// One workgroup = one whole sequence (conv3: 125 of 128 positions, conv4: 61 of 64), wave w owns channel tile w.
#ifndef PF_STEM_SCHED
#define PF_STEM_SCHED 1       // 1: sched_group_barrier pattern (MFMA, LDS read, a few VALU) over the fused block; 0: scheduler's choice
#endif
#ifndef PF_STEM_AHEAD
#define PF_STEM_AHEAD 3
#endif
#ifndef PF_STEM_VALU
#define PF_STEM_VALU 4
#endif
template <int LAYER, int CG, int NWAVES, bool DACT>
__global__ __launch_bounds__(NWAVES * 64) void conv_persist_kernel(const ConvParams p) {
    constexpr StemLayer SL = stem_layer(LAYER);
    constexpr bool LAST = LAYER == kStemLayers - 1;
    constexpr int CIN = SL.cin, COUT = SL.cout, KW = SL.kw, S = SL.stride, LIN = SL.lin, LOUT = SL.lout;
    constexpr int KK = KW * CIN, NKS = KK / 32, CHB = 32, CB = CIN / CHB;
    constexpr int P = CG * 16, SPAN = (P - 1) * S + KW, Q = image_rows(SPAN, S);
    constexpr int CHUNKS = CIN * 2 / 16, NTHR = NWAVES * 64, ITERS = (SPAN * CHUNKS + NTHR - 1) / NTHR;
    constexpr int OESZ = LAST ? 4 : 2;
    constexpr unsigned kOob = 0x7fffff00u;                            // beyond every range below: loads return 0, stores are dropped
    static_assert(COUT / 16 == NWAVES && P >= LOUT && LAYER >= 2, "one tile per wave, one sequence per workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t n0 = blockIdx.y, stride = gridDim.y, N = p.n_seq;

    u32x4 aw[NKS];
#pragma unroll
    for (int k = 0; k < NKS; ++k) aw[k] = p.wfrags[((size_t)wave * NKS + k) * 64 + lane];
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + wave * 16 + 4 * g);

    u32x4 v[ITERS];
    auto request = [&](int64_t n) {                                  // the sequence's [LIN][CIN] activations, 16 B per lane
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (size_t)n * LIN * CIN * 2, 0, LIN * CIN * 2, 0x00020000);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tid + it * NTHR;                           // chunk i = position i / CHUNKS, 16-byte piece i % CHUNKS:
            v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 0);   // byte i * 16; positions >= LIN read as zeros
        }
    };
    auto stash = [&]() {                                             // -> rows (pos % S, channel block, pos / S), see above
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tid + it * NTHR;
            if (i < SPAN * CHUNKS) {
                const int pos = i / CHUNKS, ch16 = i - pos * CHUNKS;
                const int r = pos % S, q = pos / S, cb = ch16 / 4, sub = ch16 & 3;
                *reinterpret_cast<u32x4*>(smem + ((size_t)((r * CB + cb) * Q + q) * 64) + ((sub ^ row_swz(q)) << 4)) = v[it];
            }
        }
    };
    auto bfrag = [&](int ks, int cg) -> u32x4 {
        const int kk0 = 32 * ks, tap = kk0 / CIN, cb = (kk0 % CIN) / CHB;
        const int r = tap % S, q = 16 * cg + c + tap / S;
        return *reinterpret_cast<const u32x4*>(smem + ((size_t)((r * CB + cb) * Q + q) * 64) + ((g ^ row_swz(q)) << 4));
    };
    auto kloop = [&](f32x4 (&acc)[CG]) {
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) acc[cg] = b4;
#pragma unroll
        for (int k = 0; k < NKS; ++k)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
                acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw[k]),
                                                                  __builtin_bit_cast(bf16x8, bfrag(k, cg)), acc[cg], 0, 0, 0);
    };
    // GELU + stores of sequence n; lane (g, c): channels 16 wave + 4 g .. + 3 of position 16 cg + c
    unsigned ooff[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const int pos = 16 * cg + c;
        ooff[cg] = pos < LOUT ? (unsigned)((pos * COUT + wave * 16 + 4 * g)) : kOob;      // in elements (kOob stays out of range)
    }
    auto epilogue = [&](int64_t n, const f32x4 (&acc)[CG]) {
        const size_t seq = (size_t)n * LOUT * COUT;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(p.out) + seq * OESZ, 0, LOUT * COUT * OESZ, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            DACT ? reinterpret_cast<char*>(p.dact) + seq * 2 : reinterpret_cast<char*>(p.out), 0, DACT ? LOUT * COUT * 2 : 0, 0x00020000);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            const f32x4 y = gelu_erf_fast4(acc[cg]);
            if constexpr (DACT) {                                    // training: gelu'(pre-activation) for the backward
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float ye, de;
                    gelu_fast_pair(acc[cg][e], ye, de);
                    o[e] = (__bf16)de;
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rd, ooff[cg] == kOob ? kOob : ooff[cg] * 2, 0, 0);
            }
            if constexpr (!LAST) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)y[e];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ro, ooff[cg] == kOob ? kOob : ooff[cg] * 2, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro, ooff[cg] == kOob ? kOob : ooff[cg] * 4, 0, 0);
            }
        }
    };

    request(n0);
    stash();
    __syncthreads();
    if (n0 + stride < N) request(n0 + stride);
    f32x4 acc[CG];
    kloop(acc);
    int64_t n = n0;
    for (; n + stride < N; n += stride) {
        __syncthreads();                                             // every wave is done with sequence n's image
        stash();                                                     // sequence n + stride
        if (n + 2 * stride < N) request(n + 2 * stride);
        __syncthreads();
        f32x4 acc2[CG];
        kloop(acc2);                                                 // sequence n + stride: MFMA pipe, LDS
        epilogue(n, acc);                                            // sequence n: VALU, stores
#if PF_STEM_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, PF_STEM_AHEAD, 0);          // B fragments requested a few MFMAs ahead
#pragma unroll
        for (int i = 0; i < NKS * CG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read
            __builtin_amdgcn_sched_group_barrier(0x002, DACT ? 2 * PF_STEM_VALU : PF_STEM_VALU, 0);   // VALU work in its shadow
        }
#endif
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) acc[cg] = acc2[cg];
    }
    epilogue(n, acc);
}

// ---- host side ---------------------------------------------------------------------------------
struct StemGeom { int cg, nwaves; };
#ifndef PF_CONV3_WAVES
#define PF_CONV3_WAVES 8
#endif
#ifndef PF_CONV4_WAVES
#define PF_CONV4_WAVES 12
#endif
// conv3 / conv4: one channel tile per wave.  Two / three tiles per wave sharing their B fragments (4 waves), or 64 positions per
// conv3 workgroup, measure the same (LABLOG R4.12: those layers are bound by their load -> k-loop -> store latency chain)
#ifndef PF_CONV3_CG
#define PF_CONV3_CG 8
#endif
constexpr int kConv3Waves = PF_CONV3_WAVES, kConv4Waves = PF_CONV4_WAVES, kConv3CG = PF_CONV3_CG;
// positions per workgroup / waves per workgroup for each layer
constexpr StemGeom stem_geom(int layer) {
    return layer == 0 ? StemGeom{16, 2} : layer == 1 ? StemGeom{16, 4} : layer == 2 ? StemGeom{kConv3CG, kConv3Waves}
                                                                                       : StemGeom{4, kConv4Waves};
}

constexpr int kFuseCG = 4;       // conv1 -> conv2 fused: 64 conv2 positions per workgroup (32: 2.23 ms, 64: 2.07, 128: 2.13)
static size_t stem_lds_fused(bool bf16) {
    const StemLayer L = stem_layer(1), L0 = stem_layer(0);
    const int esz = bf16 ? 2 : 4, P = kFuseCG * 16, span = (P - 1) * L.stride + L.kw;
    const int chb = 64 / esz, cb = L.cin / chb, q = image_rows(span, L.stride);
    const int npt = (span + 15) / 16, nsig = (16 * npt - 1) * L0.stride + L0.kw;
    return (size_t)L.stride * cb * q * 64 + (((size_t)nsig * esz + 15) & ~(size_t)15) + 64;
}
static size_t stem_lds(int layer, bool bf16) {
    const StemLayer L = stem_layer(layer);
    const StemGeom G = stem_geom(layer);
    const int esz = bf16 ? 2 : 4, P = G.cg * 16, span = (P - 1) * L.stride + L.kw;
    if (layer == 0) return (((size_t)span * esz + 15) & ~(size_t)15) + 64;
    const int chb = 64 / esz, cb = L.cin / chb, q = image_rows(span, L.stride);
    return (size_t)L.stride * cb * q * 64;
}

int64_t stem_weight_frags(int layer, bool bf16) {
    const StemLayer L = stem_layer(layer);
    return (int64_t)(L.cout / 16) * (L.kw * L.cin / (bf16 ? 32 : 16));
}

// element map of the packed stem weights: for each layer, frags [tile][ks][lane][elems]; source =
// index into the flat concatenation of (conv weight [cout][cin][kw], bias [cout]) over the 4 layers
int64_t stem_pack_map_len(bool bf16) {
    int64_t n = 0;
    for (int l = 0; l < kStemLayers; ++l) n += stem_weight_frags(l, bf16) * (bf16 ? 512 : 256) + stem_layer(l).cout;
    return n;
}
int64_t stem_raw_count() {
    int64_t n = 0;
    for (int l = 0; l < kStemLayers; ++l) { const StemLayer L = stem_layer(l); n += (int64_t)L.cout * L.cin * L.kw + L.cout; }
    return n;
}
int64_t stem_packed_bytes(bool bf16) {
    int64_t n = 0;
    for (int l = 0; l < kStemLayers; ++l) n += stem_weight_frags(l, bf16) * 1024 + stem_layer(l).cout * 4;
    return n;
}
void stem_build_pack_map(bool bf16, int32_t* map) {
    const int per = bf16 ? 8 : 4, kstep = bf16 ? 32 : 16;
    int64_t idx = 0, raw0 = 0;
    // weights of all layers first (each layer's frags), then all biases -- mirrored by stem_offsets()
    for (int l = 0; l < kStemLayers; ++l) {
        const StemLayer L = stem_layer(l);
        const int nks = L.kw * L.cin / kstep;
        for (int tile = 0; tile < L.cout / 16; ++tile)
            for (int ks = 0; ks < nks; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < per; ++e, ++idx) {
                        const int g = lane >> 4, r16 = lane & 15;
                        const int kk = bf16 ? (32 * ks + 8 * g + e) : (16 * ks + 4 * g + e);
                        const int tap = kk / L.cin, ch = kk % L.cin, co = tile * 16 + r16;
                        map[idx] = (int32_t)(raw0 + ((int64_t)co * L.cin + ch) * L.kw + tap);
                    }
        raw0 += (int64_t)L.cout * L.cin * L.kw + L.cout;
    }
    raw0 = 0;
    for (int l = 0; l < kStemLayers; ++l) {
        const StemLayer L = stem_layer(l);
        for (int co = 0; co < L.cout; ++co, ++idx) map[idx] = (int32_t)(raw0 + (int64_t)L.cout * L.cin * L.kw + co);
        raw0 += (int64_t)L.cout * L.cin * L.kw + L.cout;
    }
}
// byte offsets inside the packed buffer: weights of layer l, then (after all weights) biases
static void stem_offsets(bool bf16, int64_t (&w_off)[kStemLayers], int64_t (&b_off)[kStemLayers]) {
    int64_t o = 0;
    for (int l = 0; l < kStemLayers; ++l) { w_off[l] = o; o += stem_weight_frags(l, bf16) * 1024; }
    for (int l = 0; l < kStemLayers; ++l) { b_off[l] = o; o += stem_layer(l).cout * 4; }
}

int64_t stem_workspace_bytes(bool bf16, int64_t n_seq) {
    // two ping-pong activation buffers, sized for the largest intermediate ([2041][32] elements)
    const int64_t esz = bf16 ? 2 : 4;
    const int64_t a = (int64_t)stem_layer(0).lout * stem_layer(0).cout, b = (int64_t)stem_layer(1).lout * stem_layer(1).cout;
    return n_seq * (a + b) * esz + 256;
}

static int stem_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 0;
        return cus > 0 ? cus : 256;
    }();
    return n;
}

template <bool BF16, int LAYER>
static int launch_layer(const ConvParams& p, hipStream_t s) {
    constexpr StemGeom G = stem_geom(LAYER);
    constexpr StemLayer L = stem_layer(LAYER);
    if constexpr (BF16 && LAYER >= 2 && G.nwaves * 16 == L.cout && G.cg * 16 >= L.lout) {
        // persistent workgroups, one per CU: the LDS request is raised above half a CU's so that two never share one
        static const bool per_seq = std::getenv("PF_STEM_PER_SEQUENCE") != nullptr;      // (the per-sequence kernel, for A/B runs)
        if (!per_seq) {
            const size_t lds1 = std::max(stem_lds(LAYER, true), (size_t)82 * 1024);
            auto k = p.dact ? conv_persist_kernel<LAYER, G.cg, G.nwaves, true> : conv_persist_kernel<LAYER, G.cg, G.nwaves, false>;
            if (!opt_in_lds(reinterpret_cast<const void*>(k), (int)lds1)) return PF_ERR_HIP;
            const unsigned gy = (unsigned)std::min<int64_t>(p.n_seq, stem_cu_count());
            hipLaunchKernelGGL(k, dim3(1, gy), dim3(G.nwaves * 64), lds1, s, p);
            return launch_status();
        }
    }
    const size_t lds = stem_lds(LAYER, BF16);
    auto k = conv_gemm_kernel<BF16, LAYER, G.cg, G.nwaves>;
    if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
    const unsigned gx = (L.lout + G.cg * 16 - 1) / (G.cg * 16);
    hipLaunchKernelGGL(k, dim3(gx, (unsigned)p.n_seq), dim3(G.nwaves * 64), lds, s, p);
    return launch_status();
}

template <bool BF16>
static int launch_fused12(const ConvParams& p, hipStream_t s) {
    constexpr StemLayer L = stem_layer(1);
    const size_t lds = stem_lds_fused(BF16);
    auto k = conv_gemm_kernel<BF16, 1, kFuseCG, 4, true>;
    if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
    const unsigned gx = (L.lout + kFuseCG * 16 - 1) / (kFuseCG * 16);
    hipLaunchKernelGGL(k, dim3(gx, (unsigned)p.n_seq), dim3(4 * 64), lds, s, p);
    return launch_status();
}

template <bool BF16>
static int stem_forward_t(const char* packed, const float* strain, int64_t n_seq, float* tokens,
                          float* log_energy, char* ws, hipStream_t s) {
    int64_t w_off[kStemLayers], b_off[kStemLayers];
    stem_offsets(BF16, w_off, b_off);
    const int64_t esz = BF16 ? 2 : 4;
    char* act0 = ws;
    char* act1 = ws + (((int64_t)n_seq * stem_layer(0).lout * stem_layer(0).cout * esz + 255) & ~(int64_t)255);
    ConvParams p{};
    p.n_seq = n_seq;
    auto set = [&](int l, const void* in, void* out) {
        p.in = in; p.out = out; p.wfrags = reinterpret_cast<const u32x4*>(packed + w_off[l]);
        p.bias = reinterpret_cast<const float*>(packed + b_off[l]); p.log_energy = l == 0 ? log_energy : nullptr;
    };
    int rc;
    // bf16: conv1 inside conv2's staging (2.56 -> 2.13 ms per 12 288 sequences, tokens bit-identical to the
    // 4-launch path; $PF_STEM_UNFUSED keeps that path testable).  fp32: the fused image + signal need 84 KB of LDS =
    // one workgroup per CU and measured slower (9.4 -> 10.4 ms), so the parity mode keeps four launches.
    static const bool unfused = std::getenv("PF_STEM_UNFUSED") != nullptr;       // (read once per process)
    if (BF16 && !unfused) {
        set(1, strain, act1);
        p.wfrags0 = reinterpret_cast<const u32x4*>(packed + w_off[0]);
        p.bias0 = reinterpret_cast<const float*>(packed + b_off[0]);
        p.log_energy = log_energy;
        if ((rc = launch_fused12<BF16>(p, s)) != PF_OK) return rc;
    } else {
        set(0, strain, act0);  if ((rc = launch_layer<BF16, 0>(p, s)) != PF_OK) return rc;
        set(1, act0, act1);    if ((rc = launch_layer<BF16, 1>(p, s)) != PF_OK) return rc;
    }
    set(2, act1, act0);    if ((rc = launch_layer<BF16, 2>(p, s)) != PF_OK) return rc;
    set(3, act0, tokens);  return launch_layer<BF16, 3>(p, s);
}

// training forward: the four launches of the parity path in either precision, keeping for the backward every layer's
// output act[l] ([N][lout][cout], activation type; the last layer's = the fp32 tokens), gelu'(pre-activation) dact[l] and
// the asinh signal
template <bool BF16>
static int stem_forward_train_t(const void* const wfrags[4], const float* const bias[4], const float* strain, int64_t n_seq,
                                void* sig, void* const act[3], void* const dact[4], float* tokens, float* log_energy,
                                hipStream_t s) {
    ConvParams p{};
    p.n_seq = n_seq;
    auto set = [&](int l, const void* in, void* out) {
        p.in = in; p.out = out; p.wfrags = reinterpret_cast<const u32x4*>(wfrags[l]); p.bias = bias[l];
        p.log_energy = l == 0 ? log_energy : nullptr;
        p.dact = dact[l]; p.sig = l == 0 ? sig : nullptr;
    };
    int rc;
    set(0, strain, act[0]);  if ((rc = launch_layer<BF16, 0>(p, s)) != PF_OK) return rc;
    set(1, act[0], act[1]);  if ((rc = launch_layer<BF16, 1>(p, s)) != PF_OK) return rc;
    set(2, act[1], act[2]);  if ((rc = launch_layer<BF16, 2>(p, s)) != PF_OK) return rc;
    set(3, act[2], tokens);  return launch_layer<BF16, 3>(p, s);
}
// wfrags[l]: the layer's weights as im2col fragments [cout / 16][k-steps][64] (dense_pack mode 3 = pf_embed_stem_pack's
// order); bias[l]: fp32 [cout]
int stem_forward_train(bool bf16, const void* const wfrags[4], const float* const bias[4], const float* strain, int64_t n_seq,
                       void* sig, void* const act[3], void* const dact[4], float* tokens, float* log_energy, hipStream_t s) {
    return bf16 ? stem_forward_train_t<true>(wfrags, bias, strain, n_seq, sig, act, dact, tokens, log_energy, s)
                : stem_forward_train_t<false>(wfrags, bias, strain, n_seq, sig, act, dact, tokens, log_energy, s);
}

int stem_forward(bool bf16, const char* packed, const float* strain, int64_t n_seq, float* tokens,
                 float* log_energy, char* ws, hipStream_t s) {
    return bf16 ? stem_forward_t<true>(packed, strain, n_seq, tokens, log_energy, ws, s)
                : stem_forward_t<false>(packed, strain, n_seq, tokens, log_energy, ws, s);
}

}  // namespace pf
