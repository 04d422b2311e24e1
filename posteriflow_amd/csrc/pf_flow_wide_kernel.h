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

#ifndef PF_WIDE_ABLATE
#define PF_WIDE_ABLATE 0   // timing experiments only: 1 no spline, 2 no weight DMA, 4 no MFMA, 8 no sigmoid, 16 no fragment reads, 32 no bias reads
#endif
#ifndef PF_WIDE_REGSPLINE
#define PF_WIDE_REGSPLINE 1   // 1: spline parameters from the accumulator layout to one (row, feature) pair per lane by v_permlane32_swap
#endif                        // (registers) instead of through the wave-private LDS transpose (round 4, LABLOG R4.11)
#ifndef PF_WIDE_P
#define PF_WIDE_P 3        // A fragments requested ahead of their MFMA
#endif

#ifndef PF_WIDE_BUFDMA
#define PF_WIDE_BUFDMA 1   // 1: weight DMA as buffer_load_dwordx4 ... offen lds (SGPR resource + one constant lane-offset VGPR)
#endif
#ifndef PF_WIDE_PKRELU
#define PF_WIDE_PKRELU 0   // 1: ReLU on the packed bf16 pairs (v_pk_max_i16 with 0) instead of on the fp32 values
#endif
#ifndef PF_WIDE_NOSCHED
#define PF_WIDE_NOSCHED 0  // 1: no scheduling barriers between tiles
#endif
#if PF_WIDE_NOSCHED
#define PF_WIDE_SCHED_BARRIER() ((void)0)
#else
#define PF_WIDE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef PF_WIDE_SPREAD
#define PF_WIDE_SPREAD 2   // 0: the 9 DMA pieces of a ring half in one burst behind barrier B; n > 0: one piece every n fragments
#endif
#ifndef PF_WIDE_TRACE
#define PF_WIDE_TRACE 0    // diagnostic build: wave 0 of workgroup 0 accumulates s_memtime spans per stage into p.fail_flags
#endif

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
    constexpr int P = PF_WIDE_P;                           // A fragments requested ahead of their MFMA
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
    const int C = p.plan.C, NL = p.plan.L;
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
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.packed), 0, 0x7fffffff, 0x00020000);
    // piece i (0 .. kEpoch / 4 - 1) of an epoch: this wave's fragment 4 i + wave of the ring half
    auto dma_piece = [&](const char* src_u, int half, int i) {   // src_u: uniform address of the epoch's fragment `wave`
        const char* src = src_u;
        uint32_t dst = ring_w;
        asm volatile("" : "+s"(src), "+s"(dst));
        if (PF_WIDE_ABLATE & 2) return;
        if (PF_WIDE_BUFDMA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                wrsrc, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + (half * W::kEpoch + i * W::kWaves) * W::kFrag),
                16, lane16, (int)(src - p.packed) + i * W::kWaves * W::kFrag, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (int64_t)i * W::kWaves * W::kFrag + lane16),
                reinterpret_cast<__attribute__((address_space(3))) void*>(dst + (half * W::kEpoch + i * W::kWaves) * W::kFrag),
                16, 0, 0);
        }
    };
    auto dma_epoch = [&](const char* src_u, int half) {
#pragma unroll
        for (int i = 0; i < W::kEpoch / W::kWaves; ++i) dma_piece(src_u, half, i);
    };
    // diagnostic spans (PF_WIDE_TRACE): 0 stage 1, 1 W0, 2 W1 + gate, 3 final GEMMs, 4 spline, 5 pad + bias reload,
    // 6 DMA issue, 7 barrier A, 8 barrier B, 9 whole kernel
    unsigned long long tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> unsigned long long { return PF_WIDE_TRACE ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_begin = tick();
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
    if (p.zero_pair && blockIdx.x == 0 && tid < 2 * PF_REDUCE_SLOTS) p.zero_pair[tid] = 0.f;

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
        if constexpr (E % W::kEpoch == 0) {                 // barrier A: the half about to be read has landed, for everyone
            const unsigned long long t0 = tick();
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            tr[7] += tick() - t0;
        }
        if (PF_WIDE_ABLATE & 16) { aq[E % P] = wu32x4{0x3f803f80u + E, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(aq[E % P])); }
        else aq[E % P] = lds_ld_u4(rl + (E % W::kRing) * W::kFrag);
    };
    auto after = [&](auto e) {                              // fragment E has been consumed
        constexpr int E = decltype(e)::value;
        constexpr int R = (E + 1) % W::kEpoch;               // fragments consumed of the current half
        constexpr int EB = E + 1 - R;                        // ... which began at fragment EB
        if constexpr (R == 0) {                              // barrier B: everyone is done with the previous half: refill it
            const unsigned long long t0 = tick();
            asm volatile("s_barrier" ::: "memory");
            tr[8] += tick() - t0;
        }
        // the refill = the half after next; in one burst, or one piece every PF_WIDE_SPREAD fragments
        if constexpr (PF_WIDE_SPREAD == 0) {
            if constexpr (R == 0) {
                const unsigned long long t1 = tick();
                dma_epoch(glayer + (int64_t)(EB + W::kEpoch) * W::kFrag, (EB / W::kEpoch + 1) & 1);
                tr[6] += tick() - t1;
            }
        } else if constexpr (R % PF_WIDE_SPREAD == 0 && R / PF_WIDE_SPREAD < W::kEpoch / W::kWaves) {
            const unsigned long long t1 = tick();
            dma_piece(glayer + (int64_t)(EB + W::kEpoch) * W::kFrag, (EB / W::kEpoch + 1) & 1, R / PF_WIDE_SPREAD);
            tr[6] += tick() - t1;
        }
    };
    // registers 8 s .. 8 s + 7 of an accumulator tile -> the B fragment of k-step s of the next GEMM
    auto to_b = [&](const f32x16& v, int s, bool relu) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = v[8 * s + j];
            o[j] = (__bf16)((relu && !PF_WIDE_PKRELU) ? fmaxf(t, 0.f) : t);
        }
        if (relu && PF_WIDE_PKRELU) {      // a negative bf16 is a negative int16: max with 0 per 16-bit half (-0 -> +0)
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            o = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, o), zero));
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

    // ---- software pipeline of a GEMM stage ---------------------------------------------------------------------------
    // One wave per SIMD hides nothing by itself: the MFMA chain of a tile and its epilogue (bias, activation, bf16
    // conversion, ...) run back to back unless they are interleaved in program order.  So the epilogue of tile T - 1
    // is cut into 16 one-element pieces and dealt out behind the MFMAs of tile T (independent work that issues while
    // the matrix pipe runs the chain); scheduling fences keep that order.  Chains start from C = 0 (inline constant)
    // and the bias is added by the pieces; bias values travel through a two-deep queue of 4-float LDS reads, each
    // requested one quad of elements ahead of its use.
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto e, auto first, const bf16x8& b, f32x16& acc) {
        constexpr int E = decltype(e)::value;
        const wu32x4 a = aq[E % P];
        rd(ic<E + P>{});
        if (PF_WIDE_ABLATE & 4) asm volatile("" :: "v"(a), "a"(b));
        else if constexpr (decltype(first)::value)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, zero16, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, acc, 0, 0, 0);
        after(e);
    };
    auto fence = [&]() { PF_WIDE_SCHED_BARRIER(); };
    // NS MFMA steps with NP pieces dealt out behind them (all pieces first when there is no step)
    auto interleave = [&](auto ns, auto np, auto&& do_step, auto&& do_piece) {
        constexpr int NS = decltype(ns)::value, NP = decltype(np)::value;
        if constexpr (NS == 0) {
            static_for<0, NP>(do_piece);
        } else {
            static_for<0, NS>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                do_step(kk);
                constexpr int j0 = k * NP / NS, j1 = (k + 1) * NP / NS;
                static_for<j0, j1 - j0>(do_piece);
                fence();
            });
        }
    };
    f32x4 bq0[2], bq1[2];                                   // bias queues (second one: the gate / context bias)
    auto pack8 = [&](const float (&t)[8]) {                 // 8 fp32 -> one B fragment, parked in AGPRs
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)t[j];
        asm volatile("" : "+a"(o));
        return o;
    };

    for (int l = 0; l < NL; ++l) {
        asm volatile("" : "+v"(rl), "+v"(sb), "+v"(spw), "+v"(sxc), "+v"(sxn));
        unsigned long long ts = tick();
        auto span = [&](int i) { const unsigned long long t = tick(); tr[i] += t - ts; ts = t; };
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
            f32x16 a1[2], a2[2];
            bq0[0] = lds_ld_f4(sb + 4 * (W::kBiasIn));
            bq1[0] = lds_ld_f4(sb + 4 * (W::kBiasCtx));
            auto piece = [&](auto tp, auto jj) {            // element jj of tile tp: h = a1 + b_in + relu(a2 + b_c)
                constexpr int TP = decltype(tp)::value, i = decltype(jj)::value, Q = TP * 4 + i / 4;
                if constexpr (i % 4 == 0 && Q + 1 < 4 * W::kTiles) {
                    bq0[(Q + 1) & 1] = lds_ld_f4(sb + 4 * (W::kBiasIn + 8 * (Q + 1)));
                    if constexpr (CKS > 0) bq1[(Q + 1) & 1] = lds_ld_f4(sb + 4 * (W::kBiasCtx + 8 * (Q + 1)));
                }
                float v = a1[TP & 1][i] + bq0[Q & 1][i & 3];
                if constexpr (CKS > 0) v += fmaxf(a2[TP & 1][i] + bq1[Q & 1][i & 3], 0.f);
                h[TP][i] = v;
            };
            static_for<0, W::kTiles>([&](auto tt) {
                constexpr int T = decltype(tt)::value;
                constexpr int E0 = W::e_in(CKS, T);
                interleave(ic<2 + CKS>{}, ic<(T > 0 ? 16 : 0)>{},
                           [&](auto kk) {
                               constexpr int k = decltype(kk)::value;
                               if constexpr (k == 0) step(ic<E0>{}, ic<1>{}, xhi, a1[T & 1]);
                               else if constexpr (k == 1) step(ic<E0 + 1>{}, ic<0>{}, xlo, a1[T & 1]);
                               else step(ic<E0 + k>{}, ic<(k == 2)>{}, cx[k - 2], a2[T & 1]);
                           },
                           [&](auto jj) { piece(ic<(T > 0 ? T - 1 : 0)>{}, jj); });
            });
            static_for<0, 16>([&](auto jj) { piece(ic<W::kTiles - 1>{}, jj); });
            fence();
        }
        span(0);
        // ---- residual blocks: h += (W1 relu(W0 relu(h) + b0) + b1) * sigmoid(W_g ctx + b_g) -----------------------
        static_for<0, 2>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            constexpr int EB = W::e_blk(D, CKS, b);
            constexpr int EW1 = EB + W::w0_len(D);
            constexpr int OB0 = W::kBiasBlk + 768 * b, OB1 = OB0 + 256, OBG = OB0 + 512;
#pragma unroll
            for (int T = 0; T < W::kTiles; ++T) { bin[2 * T] = to_b(h[T], 0, true); bin[2 * T + 1] = to_b(h[T], 1, true); }
            {
                f32x16 acc[2];
                float t8[8];
                bq0[0] = lds_ld_f4(sb + 4 * OB0);
                auto piece = [&](auto tp, auto jj) {        // element jj of tile tp: relu(acc + b0) -> bout
                    constexpr int TP = decltype(tp)::value, i = decltype(jj)::value, Q = TP * 4 + i / 4;
                    if constexpr (i % 4 == 0 && Q + 1 < 4 * W::kTiles) bq0[(Q + 1) & 1] = lds_ld_f4(sb + 4 * (OB0 + 8 * (Q + 1)));
                    t8[i & 7] = fmaxf(acc[TP & 1][i] + bq0[Q & 1][i & 3], 0.f);
                    if constexpr ((i & 7) == 7) bout[2 * TP + i / 8] = pack8(t8);
                };
                static_for<0, W::kTiles>([&](auto tt) {
                    constexpr int T = decltype(tt)::value;
                    interleave(ic<W::kH16(D, T)>{}, ic<(T > 0 ? 16 : 0)>{},
                               [&](auto kk) {
                                   constexpr int k = decltype(kk)::value;
                                   step(ic<EB + W::w0_off(D, T) + k>{}, ic<(k == 0)>{}, bin[k], acc[T & 1]);
                               },
                               [&](auto jj) { piece(ic<(T > 0 ? T - 1 : 0)>{}, jj); });
                });
                static_for<0, 16>([&](auto jj) { piece(ic<W::kTiles - 1>{}, jj); });
                fence();
            }
            span(1);
            {
                f32x16 accw[2], accg[2];
                bq0[0] = lds_ld_f4(sb + 4 * OB1);
                if constexpr (CKS > 0) bq1[0] = lds_ld_f4(sb + 4 * OBG);
                auto piece = [&](auto tp, auto jj) {        // element jj of tile tp: h += (acc + b1) * sigmoid(g + b_g)
                    constexpr int TP = decltype(tp)::value, i = decltype(jj)::value, Q = TP * 4 + i / 4;
                    if constexpr (i % 4 == 0 && Q + 1 < 4 * W::kTiles) {
                        bq0[(Q + 1) & 1] = lds_ld_f4(sb + 4 * (OB1 + 8 * (Q + 1)));
                        if constexpr (CKS > 0) bq1[(Q + 1) & 1] = lds_ld_f4(sb + 4 * (OBG + 8 * (Q + 1)));
                    }
                    const float t = accw[TP & 1][i] + bq0[Q & 1][i & 3];
                    if constexpr (CKS > 0) {
                        const float g = accg[TP & 1][i] + bq1[Q & 1][i & 3];
                        h[TP][i] += t * ((PF_WIDE_ABLATE & 8) ? g : pf_sigmoid<true>(g));
                    } else {
                        h[TP][i] += t;
                    }
                };
                static_for<0, W::kTiles>([&](auto tt) {
                    constexpr int T = decltype(tt)::value;
                    constexpr int E0 = EW1 + W::w1_off(D, CKS, T), NW = W::kH16(D, T);
                    interleave(ic<NW + CKS>{}, ic<(T > 0 ? 16 : 0)>{},
                               [&](auto kk) {
                                   constexpr int k = decltype(kk)::value;
                                   if constexpr (k < NW) step(ic<E0 + k>{}, ic<(k == 0)>{}, bout[k], accw[T & 1]);
                                   else step(ic<E0 + k>{}, ic<(k == NW)>{}, cx[k - NW], accg[T & 1]);
                               },
                               [&](auto jj) { piece(ic<(T > 0 ? T - 1 : 0)>{}, jj); });
                });
                static_for<0, 16>([&](auto jj) { piece(ic<W::kTiles - 1>{}, jj); });
                fence();
            }
            span(2);
        });
        // ---- final masked layer + spline, two features (64 (row, feature) pairs) at a time --------------------------
#pragma unroll
        for (int T = 0; T < W::kTiles; ++T) { bin[2 * T] = to_b(h[T], 0, false); bin[2 * T + 1] = to_b(h[T], 1, false); }
        // the next layer's biases: requested now (h is dead, registers are free), parked in LDS behind the layer's end
        constexpr int NBQ = (W::kBiasFloats / 4 + 255) / 256;
        f32x4 nbias[NBQ];
        {
            const f32x4* gb = reinterpret_cast<const f32x4*>(gbias + (int64_t)(l + 1 < NL ? l + 1 : l) * W::kBiasFloats);
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int sI = tid + 256 * i;
                nbias[i] = gb[sI < W::kBiasFloats / 4 ? sI : 0];
            }
        }
#pragma nounroll
        for (int m = 0; m < NB; ++m) {
            static_for<0, NB>([&](auto mm) {
                constexpr int M = decltype(mm)::value;
                if (m == M) {
                    constexpr int E0 = W::e_out(D, CKS) + W::out_off(D, M);
                    constexpr int NA = W::kO16(D, 2 * M), NBf = W::kWHb(D, M), ND = W::kDD(D, M);
                    constexpr bool HASB = 2 * M + 1 < D;
                    constexpr int OB = W::kBiasOut + 96 * M;
                    f32x16 accA = zero16, accB = zero16, accD = zero16;
                    float t4[4];
                    // piece: element jj of output tile `which` (0 WH a, 1 WH b, 2 DD): + bias, to the spline transpose
                    bq0[0] = lds_ld_f4(sb + 4 * OB);
                    auto piece = [&](auto which, auto jj) {
                        constexpr int WH = decltype(which)::value, i = decltype(jj)::value, Q = WH * 4 + i / 4;
                        if constexpr (i % 4 == 0 && Q + 1 < 12) bq0[(Q + 1) & 1] = lds_ld_f4(sb + 4 * (OB + 8 * (Q + 1)));
                        f32x16& acc = WH == 0 ? accA : (WH == 1 ? accB : accD);
                        if constexpr (PF_WIDE_REGSPLINE) {
                            acc[i] = acc[i] + bq0[Q & 1][i & 3];           // in place: the tile is handed over in registers below
                        } else {
                        t4[i & 3] = acc[i] + bq0[Q & 1][i & 3];
                        if constexpr ((i & 3) == 3) {
                            constexpr int q = i / 4;
                            constexpr int off = WH == 0 ? 8 * q : (WH == 1 ? 32 * PS + 8 * q : (q >> 1) * 32 * PS + 32 + 8 * (q & 1));
                            lds_st_f4(spw + 4 * off, f32x4{t4[0], t4[1], t4[2], t4[3]});
                        }
                        }
                    };
                    interleave(ic<NA>{}, ic<0>{}, [&](auto kk) { step(ic<E0 + decltype(kk)::value>{}, ic<(decltype(kk)::value == 0)>{}, bin[decltype(kk)::value], accA); },
                               [&](auto jj) {});
                    if constexpr (HASB) {
                        interleave(ic<NBf>{}, ic<16>{},
                                   [&](auto kk) { step(ic<E0 + NA + decltype(kk)::value>{}, ic<(decltype(kk)::value == 0)>{}, bin[decltype(kk)::value], accB); },
                                   [&](auto jj) { piece(ic<0>{}, jj); });
                        interleave(ic<ND>{}, ic<16>{},
                                   [&](auto kk) { step(ic<E0 + NA + NBf + decltype(kk)::value>{}, ic<(decltype(kk)::value == 0)>{}, bin[decltype(kk)::value], accD); },
                                   [&](auto jj) { piece(ic<1>{}, jj); });
                    } else {
                        interleave(ic<ND>{}, ic<16>{},
                                   [&](auto kk) { step(ic<E0 + NA + decltype(kk)::value>{}, ic<(decltype(kk)::value == 0)>{}, bin[decltype(kk)::value], accD); },
                                   [&](auto jj) { piece(ic<0>{}, jj); });
                        // the bias queue still walks over the absent tile's quads: skip them
                        bq0[0] = lds_ld_f4(sb + 4 * (OB + 64));
                    }
                    static_for<0, 16>([&](auto jj) {
                        constexpr int i = decltype(jj)::value;
                        if constexpr (HASB) piece(ic<2>{}, jj);
                        else {                                   // DD with its own queue start (bq0[0] reloaded above)
                            if constexpr (i % 4 == 0 && i / 4 + 1 < 4) bq0[(i / 4 + 1) & 1] = lds_ld_f4(sb + 4 * (OB + 64 + 8 * (i / 4 + 1)));
                            if constexpr (PF_WIDE_REGSPLINE) {
                                accD[i] = accD[i] + bq0[(i / 4) & 1][i & 3];
                            } else {
                            t4[i & 3] = accD[i] + bq0[(i / 4) & 1][i & 3];
                            if constexpr ((i & 3) == 3) {
                                constexpr int q = i / 4;
                                lds_st_f4(spw + 4 * ((q >> 1) * 32 * PS + 32 + 8 * (q & 1)), f32x4{t4[0], t4[1], t4[2], t4[3]});
                            }
                            }
                        }
                    });
                    if constexpr (PF_WIDE_REGSPLINE) {
                        float uw[16], uh[16], kd[17];            // this lane's pair: raw widths | heights | derivatives
                        // lane (n, hf) holds units 8 q + 4 hf .. + 3 of each tile for row n; the lane that evaluates feature 2 M
                        // (hf = 0) needs all of tile A and units 0 .. 15 of the derivative tile, its partner all of tile B and units
                        // 16 .. 31: v_permlane32_swap exchanges exactly those halves (pf_flow_mid_kernel.h has the same hand-over).
                        // (element copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 -- clang 19)
                        static_for<0, 16>([&](auto jj) {
                            constexpr int j = decltype(jj)::value, u0 = 8 * (j >> 2) + (j & 3);
                            const float ea = accA[j], eb = HASB ? accB[j] : 0.f;
                            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
                            const unsigned s0 = sw[0], s1 = sw[1];
                            if constexpr (u0 < 16) { uw[u0] = __builtin_bit_cast(float, s0); uw[u0 + 4] = __builtin_bit_cast(float, s1); }
                            else { uh[u0 - 16] = __builtin_bit_cast(float, s0); uh[u0 - 12] = __builtin_bit_cast(float, s1); }
                        });
                        static_for<0, 8>([&](auto jj) {
                            constexpr int j = decltype(jj)::value, u0 = 8 * (j >> 2) + (j & 3);
                            const float ea = accD[j], eb = accD[j + 8];
                            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
                            const unsigned s0 = sw[0], s1 = sw[1];
                            kd[1 + u0] = __builtin_bit_cast(float, s0);
                            kd[1 + u0 + 4] = __builtin_bit_cast(float, s1);
                        });
                        fence();
                        span(3);
                        // one lane per pair: lane (n, hf) <-> (row n, feature 2 M + hf); evaluated here, inside the batch's own code
                        // path (the parameters' 49 registers do not have to survive the merge of the eight paths)
                        const int f = 2 * M + hf;
                        if (f < D) {
                            const float xv = lds_ld_f(sxc + 4 * f);
                            if (p.u_save && live) p.u_save[((int64_t)l * p.batch + row) * D + f] = xv;
                            float y, ld;
                            if (PF_WIDE_ABLATE & 1) { y = xv + uw[0]; ld = 0.f; }
                            else rqs_fast16_regs(uw, uh, kd, xv, p, y, ld);
                            ld_acc += ld;
                            lds_st_f(sxn + 4 * (D - 1 - f), y);    // the next layer starts with ReversePermutation
                        }
                        span(4);
                    } else {
                        fence();
                    }
                }
            });
            if constexpr (!PF_WIDE_REGSPLINE) {
            span(3);
            // one lane per pair: lane (n, hf) <-> (row n, feature 2 m + hf)
            const int f = 2 * m + hf;
            if (f < D) {
                const float xv = lds_ld_f(sxc + 4 * f);
                if (p.u_save && live) p.u_save[((int64_t)l * p.batch + row) * D + f] = xv;
                float y, ld;
                if (PF_WIDE_ABLATE & 1) { y = xv + s_par[lane * PS]; ld = 0.f; }
                else rqs_pair_fast16(s_par + lane * PS, xv, p, y, ld);       // the wide plans are K = 16 only
                ld_acc += ld;
                lds_st_f(sxn + 4 * (D - 1 - f), y);            // the next layer starts with ReversePermutation
            }
            span(4);
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
        if (l + 1 < NL) {                                      // next layer's biases (loaded above)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everyone is done with this layer's
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int sI = tid + 256 * i;
                if (sI < W::kBiasFloats / 4) reinterpret_cast<f32x4*>(s_bias)[sI] = nbias[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        span(5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring's last (unused) DMA pieces must land before the LDS is released

    if (PF_WIDE_TRACE && p.fail_flags && blockIdx.x == 0 && tid == 0) {
        tr[9] = tick() - t_begin;
        for (int i = 0; i < 10; ++i) reinterpret_cast<unsigned long long*>(p.fail_flags)[i] = tr[i];
    }
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
        if (lane == 0) { float* acc = p.nll_sum + 2 * ((4 * blockIdx.x + wave) % PF_REDUCE_SLOTS); atomicAdd(acc, my_nll); atomicAdd(acc + 1, my_cnt); }
    }
}

}  // namespace pf
