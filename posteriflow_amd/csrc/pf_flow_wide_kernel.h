// pf_flow_wide_kernel.h -- large-batch forward pass of the masked-autoregressive RQS flow for gfx950 (MI355X).
// Same function as pf_flow_fwd_kernel.h (NSFPosteriorFlow.forward / compute_psd_aware_nll,
// src/ahsd/models/flows.py:610-618, 727-779, executing nflows' MADE + RQS), different work split -- see
// pf_wide_layout.h for the schedule and the packed layout:
//
//   workgroup = 4 waves = 128 batch rows; one wave per SIMD (up to 512 registers); a wave owns 32 rows = one
//   v_mfma_f32_32x32x16_bf16 column tile through ALL layers.  Out^T[unit, row] = W[unit, k] . act^T[k, row] as in the
//   16-row kernel, but every wave computes ALL hidden units of its rows: the residual state h (8 tiles x 16
//   accumulator registers), the bf16 activations (2 x 64 registers) and the bf16 context (72) live in the wave's
//   registers; an accumulator tile becomes the next GEMM's B operand by a pairwise bf16 conversion, with the weight
//   fragments packed in the matching (permuted) k order.  No activation ever goes through LDS, no barrier separates
//   dependent GEMMs.
//   The four waves consume ONE common stream of weight fragments: it is fetched from L2 once per workgroup by LDS-DMA
//   (global_load_lds_dwordx4, one 1-KiB fragment per wave-instruction, every wave loads a quarter) into a two-half
//   LDS ring, and every wave reads each fragment from there (ds_read_b128, conflict-free, 1 KiB per 32-cycle MFMA
//   per SIMD = half the LDS bandwidth).  Two barriers per ring half, neither drains the pipeline:
//     A  before the first LDS read of a half: the issuing wave waits for its own DMA pieces (vmcnt), then s_barrier;
//     B  after the MFMA that consumed the last fragment of a half: s_barrier, then the DMA of the half after next.
//   Spline: the 3K-1 raw parameters of a feature come out of the final layer as accumulator tiles (widths | heights
//   of one feature = one 32-unit tile, the derivatives of two features = one tile); two features x 32 rows = 64
//   (row, feature) pairs are transposed through a wave-private LDS buffer and evaluated one pair per lane by the same
//   rqs_pair as the 16-row kernel.
//
// Numerics = the bf16 mode of the 16-row kernel: bf16 operands (x as a hi + lo pair), fp32 accumulation, fp32
// residual / bias / spline.  The accumulation ORDER differs (bias first, k permuted inside 16-wide steps), so the two
// kernels agree to fp32 rounding of the activations, not bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "pf_flow_fwd_kernel.h"
#include "pf_wide_layout.h"

namespace pf {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int wu32x4;

// LDS accesses by 32-bit LDS address: the hot addresses go through an opaque asm once per layer (so that base +
// constant stays an instruction offset instead of a hoisted register), which would otherwise cost the pointer its
// address space (flat_load instead of ds_read)
typedef __attribute__((address_space(3))) wu32x4 lds_u32x4_t;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;
typedef __attribute__((address_space(3))) float lds_f32_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ wu32x4 lds_ld_u4(uint32_t a) { return *reinterpret_cast<const lds_u32x4_t*>(a); }
__device__ __forceinline__ f32x4 lds_ld_f4(uint32_t a) { return *reinterpret_cast<const lds_f32x4_t*>(a); }
__device__ __forceinline__ void lds_st_f4(uint32_t a, f32x4 v) { *reinterpret_cast<lds_f32x4_t*>(a) = v; }
__device__ __forceinline__ float lds_ld_f(uint32_t a) { return *reinterpret_cast<const lds_f32_t*>(a); }
__device__ __forceinline__ void lds_st_f(uint32_t a, float v) { *reinterpret_cast<lds_f32_t*>(a) = v; }

template <int D, int CKS>
__global__ __launch_bounds__(256) void flow_wide_kernel(const FwdParams p) {
    namespace W = wide;
    constexpr int NF = W::n_frags(D, CKS), NFP = W::n_frags_padded(D, CKS);
    constexpr int NB = W::n_batches(D);
    constexpr int P = 4;                                   // A fragments requested ahead of their MFMA
    constexpr int XS = W::kXStride, PS = W::kParStride;
    static_assert(NFP % W::kRing == 0 && W::kEpoch % W::kWaves == 0, "ring geometry");
    static_assert(D >= 2 && D <= 16, "2 <= D <= 16");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hf = lane >> 5;
    char* const ring = smem;
    float* const s_par = reinterpret_cast<float*>(smem + W::kRing * W::kFrag) + wave * (64 * PS);
    float* const s_x = reinterpret_cast<float*>(smem + W::kRing * W::kFrag + W::kWaves * 64 * PS * 4) + wave * (2 * 32 * XS);
    float* const s_bias = reinterpret_cast<float*>(smem + W::kRing * W::kFrag + W::kWaves * 64 * PS * 4 + W::kWaves * 2 * 32 * XS * 4);
    const char* const ring_lane = ring + lane * 16;
    const int K = p.plan.K, C = p.plan.C, NL = p.plan.L;
    const int64_t row_w = (int64_t)blockIdx.x * W::kRowsPerWG + wave * W::kRowsPerWave;   // first row of this wave
    int64_t row = row_w + n;
    const bool live = row < p.batch;
    if (!live) row = p.batch - 1;

    // ---- weight stream: LDS-DMA of one ring half = kEpoch fragments, kEpoch / 4 pieces per wave -------------------
    // source = uniform 64-bit base (SGPRs) + this lane's 32-bit offset (one VGPR): global_load_lds ... saddr form; the
    // LDS destination goes to M0.  Both are re-derived at every call from opaque scalars: left to itself the compiler
    // precomputes every piece's 64-bit address and M0 value ahead of the layer and spills them (scratch reloads in
    // front of every DMA, each with a vmcnt(0)).
    const char* gstream = p.packed + (int64_t)wave * W::kFrag;                   // + (layer * NFP + frag) * 1 KiB
    const uint32_t lane16 = lane * 16;
    const uint32_t ring_w = lds_addr(ring) + wave * W::kFrag;
    auto dma_epoch = [&](const char* src_u, int half) {       // src_u: uniform address of the epoch's fragment `wave`
        const char* src = src_u;
        uint32_t dst = ring_w;
        asm volatile("" : "+s"(src), "+s"(dst));
#pragma unroll
        for (int i = 0; i < W::kEpoch / W::kWaves; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (int64_t)i * W::kWaves * W::kFrag + lane16),
                reinterpret_cast<__attribute__((address_space(3))) void*>(dst + (half * W::kEpoch + i * W::kWaves) * W::kFrag),
                16, 0, 0);
    };
    dma_epoch(gstream, 0);
    dma_epoch(gstream + (int64_t)W::kEpoch * W::kFrag, 1);

    // ---- context as B fragments (registers, whole kernel): lane (n, hf) element j of k-step ks = ctx[row][16 ks + 8 hf + j]
    bf16x8 cx[CKS > 0 ? CKS : 1];
    if constexpr (CKS > 0) {
        const float* crow = p.ctx + row * C;
        if ((C & 3) == 0) {
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks) {
                const int c0 = 16 * ks + 8 * hf;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (c0 < C) v0 = *reinterpret_cast<const f32x4*>(crow + c0);
                if (c0 + 4 < C) v1 = *reinterpret_cast<const f32x4*>(crow + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { cx[ks][j] = (__bf16)v0[j]; cx[ks][4 + j] = (__bf16)v1[j]; }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < CKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c0 = 16 * ks + 8 * hf + j;
                    cx[ks][j] = (__bf16)(c0 < C ? crow[c0] : 0.f);
                }
        }
    }
    if constexpr (CKS > 0) {
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) asm volatile("" : "+a"(cx[ks]));
    }
    // ---- x^T of this wave's rows: position d of layer 0 <- x[row][ar_perm[D-1-d]] (ReversePermutation first) ------
    float* sx_cur = s_x;
    float* sx_nxt = s_x + 32 * XS;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 8 * hf + j;
        float v = 0.f;
        if (d < D) {
            const int sd = D - 1 - d;
            v = p.x[row * D + (p.ar_perm ? p.ar_perm[sd] : sd)];
        }
        sx_cur[n * XS + d] = v;
        sx_nxt[n * XS + d] = 0.f;
    }
    const float* gbias = reinterpret_cast<const float*>(p.packed + W::stream_frags(D, CKS, NL) * W::kFrag);
    for (int s = tid; s < W::kBiasFloats / 4; s += 256)
        reinterpret_cast<f32x4*>(s_bias)[s] = reinterpret_cast<const f32x4*>(gbias)[s];
    if (p.zero_pair && blockIdx.x == 0 && tid == 0) { p.zero_pair[0] = 0.f; p.zero_pair[1] = 0.f; }

    // ---- fragment queue -----------------------------------------------------------------------------------------
    wu32x4 aq[P];
    const char* glayer = gstream;                           // uniform address of fragment `wave` of the current layer
    // per-lane LDS bases; re-"defined" (opaque asm) at the top of every layer so that base + constant stays an
    // instruction offset instead of a hoisted, spilled register
    uint32_t rl = lds_addr(ring_lane);                     // this lane's 16 bytes of ring slot 0
    uint32_t sb = lds_addr(s_bias + 4 * hf);               // bias block, this lane half's 4 units of a group of 8
    uint32_t spw = lds_addr(s_par + n * PS + 4 * hf);      // spline transpose: (row n, first feature of the batch)
    uint32_t sxc = lds_addr(s_x + n * XS);                 // x of row n, current layer / next layer
    uint32_t sxn = lds_addr(s_x + 32 * XS + n * XS);
    auto rd = [&](auto e) {                                 // request fragment E (of this layer, or E - NFP of the next)
        constexpr int E = decltype(e)::value;
        if constexpr (E % W::kEpoch == 0)                   // barrier A: the half about to be read has landed, for everyone
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        aq[E % P] = lds_ld_u4(rl + (E % W::kRing) * W::kFrag);
    };
    auto after = [&](auto e) {                              // fragment E has been consumed
        constexpr int E = decltype(e)::value;
        if constexpr ((E + 1) % W::kEpoch == 0) {           // barrier B: everyone is done with that half: refill it
            asm volatile("s_barrier" ::: "memory");
            dma_epoch(glayer + (int64_t)(E + 1 + W::kEpoch) * W::kFrag, ((E + 1) / W::kEpoch + 1) & 1);
        }
    };
    auto use = [&](auto e, const bf16x8& b, f32x16& acc) {
        constexpr int E = decltype(e)::value;
        const wu32x4 a = aq[E % P];
        rd(ic<E + P>{});
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, acc, 0, 0, 0);
        after(e);
    };
    auto bias16 = [&](int off) {                            // accumulator initialised with the tile's 32 biases
        f32x16 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = lds_ld_f4(sb + 4 * (off + 8 * q));
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * q + e] = v[e];
        }
        return r;
    };
    // registers 8 s .. 8 s + 7 of an accumulator tile -> the B fragment of k-step s of the next GEMM
    auto to_b = [&](const f32x16& v, int s, bool relu) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = v[8 * s + j];
            o[j] = (__bf16)(relu ? fmaxf(t, 0.f) : t);
        }
        // B fragments live in AGPRs (the MFMA reads them there): with h, the activations and the context all in the
        // 256 architectural VGPRs the allocator spills ~2500 registers
        asm volatile("" : "+a"(o));
        return o;
    };

    __syncthreads();                                        // x, biases staged (also: both DMA halves landed)
    static_for<0, P>([&](auto e) { rd(e); });

    float ld_acc = 0.f;
    f32x16 h[W::kTiles];
    bf16x8 bin[W::kKSteps], bout[W::kKSteps];

    for (int l = 0; l < NL; ++l) {
        asm volatile("" : "+v"(rl), "+v"(sb), "+v"(spw), "+v"(sxc), "+v"(sxn));
        // ---- stage 1: h = W_in x + b_in + relu(W_c ctx + b_c) ----------------------------------------------------
        {
            bf16x8 xhi, xlo;
            const f32x4 v0 = lds_ld_f4(sxc + 32 * hf);
            const f32x4 v1 = lds_ld_f4(sxc + 32 * hf + 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = j < 4 ? v0[j & 3] : v1[j & 3];
                const __bf16 hi = (__bf16)v;
                xhi[j] = hi;
                xlo[j] = (__bf16)(v - (float)hi);
            }
            static_for<0, W::kTiles>([&](auto tt) {
                constexpr int T = decltype(tt)::value;
                constexpr int E0 = W::e_in(CKS, T);
                f32x16 a1 = bias16(W::kBiasIn + 32 * T);
                use(ic<E0>{}, xhi, a1);
                use(ic<E0 + 1>{}, xlo, a1);
                if constexpr (CKS > 0) {
                    f32x16 a2 = bias16(W::kBiasCtx + 32 * T);
                    static_for<0, CKS>([&](auto kk) { use(ic<E0 + 2 + decltype(kk)::value>{}, cx[decltype(kk)::value], a2); });
#pragma unroll
                    for (int i = 0; i < 16; ++i) a1[i] += fmaxf(a2[i], 0.f);
                }
                h[T] = a1;
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- residual blocks: h += (W1 relu(W0 relu(h) + b0) + b1) * sigmoid(W_g ctx + b_g) -----------------------
        static_for<0, 2>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            constexpr int EB = W::e_blk(D, CKS, b);
            constexpr int EW1 = EB + W::w0_len(D);
#pragma unroll
            for (int T = 0; T < W::kTiles; ++T) { bin[2 * T] = to_b(h[T], 0, true); bin[2 * T + 1] = to_b(h[T], 1, true); }
            static_for<0, W::kTiles>([&](auto tt) {
                constexpr int T = decltype(tt)::value;
                f32x16 acc = bias16(W::kBiasBlk + 768 * b + 32 * T);
                static_for<0, W::kH16(D, T)>([&](auto kk) {
                    use(ic<EB + W::w0_off(D, T) + decltype(kk)::value>{}, bin[decltype(kk)::value], acc);
                });
                bout[2 * T] = to_b(acc, 0, true);
                bout[2 * T + 1] = to_b(acc, 1, true);
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<0, W::kTiles>([&](auto tt) {
                constexpr int T = decltype(tt)::value;
                constexpr int E0 = EW1 + W::w1_off(D, CKS, T);
                f32x16 acc = bias16(W::kBiasBlk + 768 * b + 256 + 32 * T);
                static_for<0, W::kH16(D, T)>([&](auto kk) { use(ic<E0 + decltype(kk)::value>{}, bout[decltype(kk)::value], acc); });
                if constexpr (CKS > 0) {
                    f32x16 g = bias16(W::kBiasBlk + 768 * b + 512 + 32 * T);
                    static_for<0, CKS>([&](auto kk) {
                        use(ic<E0 + W::kH16(D, T) + decltype(kk)::value>{}, cx[decltype(kk)::value], g);
                    });
#pragma unroll
                    for (int i = 0; i < 16; ++i) h[T][i] += acc[i] * pf_sigmoid<true>(g[i]);
                } else {
                    h[T] += acc;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // ---- final masked layer + spline, two features (64 (row, feature) pairs) at a time --------------------------
#pragma unroll
        for (int T = 0; T < W::kTiles; ++T) { bin[2 * T] = to_b(h[T], 0, false); bin[2 * T + 1] = to_b(h[T], 1, false); }
#pragma nounroll
        for (int m = 0; m < NB; ++m) {
            static_for<0, NB>([&](auto mm) {
                constexpr int M = decltype(mm)::value;
                if (m == M) {
                    constexpr int E0 = W::e_out(D, CKS) + W::out_off(D, M);
                    constexpr int NA = W::kO16(D, 2 * M), NBf = W::kWHb(D, M), ND = W::kDD(D, M);
                    f32x16 acc = bias16(W::kBiasOut + 96 * M);
                    static_for<0, NA>([&](auto kk) { use(ic<E0 + decltype(kk)::value>{}, bin[decltype(kk)::value], acc); });
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        lds_st_f4(spw + 4 * (8 * q), f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
                    if constexpr (2 * M + 1 < D) {
                        acc = bias16(W::kBiasOut + 96 * M + 32);
                        static_for<0, NBf>([&](auto kk) { use(ic<E0 + NA + decltype(kk)::value>{}, bin[decltype(kk)::value], acc); });
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            lds_st_f4(spw + 4 * (32 * PS + 8 * q), f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
                    }
                    acc = bias16(W::kBiasOut + 96 * M + 64);
                    static_for<0, ND>([&](auto kk) { use(ic<E0 + NA + NBf + decltype(kk)::value>{}, bin[decltype(kk)::value], acc); });
#pragma unroll
                    for (int q = 0; q < 4; ++q)      // rows u < 16: derivatives of feature 2M; u >= 16: of feature 2M + 1
                        lds_st_f4(spw + 4 * ((q >> 1) * 32 * PS + 32 + 8 * (q & 1)),
                                  f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
                }
            });
            // one lane per pair: lane (n, hf) <-> (row n, feature 2 m + hf)
            const int f = 2 * m + hf;
            if (f < D) {
                const float xv = lds_ld_f(sxc + 4 * f);
                if (p.u_save && live) p.u_save[((int64_t)l * p.batch + row) * D + f] = xv;
                float y, ld;
                rqs_pair<true>(s_par + lane * PS, xv, K, p, y, ld);
                ld_acc += ld;
                lds_st_f(sxn + 4 * (D - 1 - f), y);            // the next layer starts with ReversePermutation
            }
        }
        // ---- the pad fragments of this layer: keep the ring turning, prime the queue for the next layer ------------
        static_for<NF, NFP - NF>([&](auto e) {
            constexpr int E = decltype(e)::value;
            rd(ic<E + P>{});
            after(e);
        });
        glayer += (int64_t)NFP * W::kFrag;
        { const uint32_t t = sxc; sxc = sxn; sxn = t; }
        if (l + 1 < NL) {                                      // next layer's biases
            __syncthreads();
            const float* gb = gbias + (int64_t)(l + 1) * W::kBiasFloats;
            for (int s = tid; s < W::kBiasFloats / 4; s += 256)
                reinterpret_cast<f32x4*>(s_bias)[s] = reinterpret_cast<const f32x4*>(gb)[s];
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring's last (unused) DMA pieces must land before the LDS is released

    // ---- epilogue: log-det of a row = sum over its features (two lanes), base density, stores ------------------------
    const float ld_row = ld_acc + __shfl_xor(ld_acc, 32, 64);
    float my_nll = 0.f, my_cnt = 0.f;
    if (hf == 0 && live) {
        float q = 0.f, sls = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float zv = lds_ld_f(sxc + 4 * (D - 1 - d));  // stored reversed
            if (p.log_sigma) {                                  // PSDScaledNormal.log_prob, flows.py:73-83
                const float ls = p.log_sigma[row * D + d];
                const float zs = zv / expf(ls);
                q += zs * zs; sls += ls;
            } else {
                q += zv * zv;
            }
            if (p.z) p.z[row * D + d] = zv;
        }
        if (p.logdet) p.logdet[row] = ld_row;
        my_nll = 0.5f * (q + 2.f * sls + (float)D * 1.8378770664093453f) - ld_row;
        my_cnt = 1.f;
        if (p.nll) p.nll[row] = my_nll;
    }
    if (p.nll_sum) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { my_nll += __shfl_xor(my_nll, o, 64); my_cnt += __shfl_xor(my_cnt, o, 64); }
        if (lane == 0) { atomicAdd(p.nll_sum, my_nll); atomicAdd(p.nll_sum + 1, my_cnt); }
    }
}

}  // namespace pf
