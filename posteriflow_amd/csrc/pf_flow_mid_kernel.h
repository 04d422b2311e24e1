// pf_flow_mid_kernel.h -- mid-batch forward pass of the masked-autoregressive RQS flow for gfx950 (MI355X): the design
// between the 16-row kernel (pf_flow_fwd_kernel.h: every workgroup streams the whole weight set for 16-48 rows) and the
// large-batch kernel (pf_flow_wide_kernel.h: 128 rows per workgroup, one wave per SIMD, 512 registers per wave).
// Same function (NSFPosteriorFlow.forward / compute_psd_aware_nll, src/ahsd/models/flows.py:610-618, 727-779, executing
// nflows' MADE + RQS), same packed stream as the large-batch kernel (PF_FLAG_WIDE layout, pf_wide_layout.h: 32-unit x 16-k
// fragments in the accumulator-permuted k order, masked k-steps absent), different work split:
//
//   workgroup = 64 batch rows = two row blocks of 32, 8 waves, TWO waves per SIMD (<= 256 registers each), so that one
//   wave's MFMA chain runs under the other's epilogue / LDS / load issue -- with one wave per SIMD those costs add
//   (LABLOG 4.9) -- on v_mfma_f32_32x32x16_bf16.
//   Front part of a layer (stage 1, two residual blocks): wave w owns ONE hidden tile (32 units; waves 0-3: tiles 0-3, waves
//   4-7: tiles 7-4, a light and a heavy tile of the masked layers per SIMD) for BOTH row blocks: a weight fragment is
//   fetched once per workgroup, straight from L2 into a register ring (buffer loads, two rings of PF_MID_P fragments
//   alternating between consecutive chains so that a chain's first fragments are requested while the previous chain runs),
//   and multiplied with both row blocks' B operands.  The residual state h (2 x 16 accumulator registers) stays in
//   registers; the bf16 activations a dependent GEMM needs from the other tiles go through LDS in B-fragment order (a
//   32 x 32 accumulator tile converted pairwise to bf16 IS two k-steps of the next GEMM's B operand in this layout's k
//   order): one barrier per dependent GEMM, two buffers.  The bf16 context lives in LDS as B fragments (36 KiB for
//   C = 288).  B operands are read per k-step, PF_MID_BD k-steps AHEAD of their MFMAs in an order pinned by sched_barrier:
//   left to the compiler every ds_read sat right in front of its MFMA and a k-step cost an LDS round trip (216 -> 192 us).
//   Back part (final masked layer + spline): wave w owns ONE feature batch (widths | heights of two features + their
//   derivatives = three 32-unit tiles, consecutive in the stream: one chain) for both row blocks, B = h from buffer 0.
//   The spline parameters go from the accumulator layout to one (row, feature) pair per lane IN REGISTERS: lane (n, hf)
//   needs what its partner (n, 1 - hf) holds of the same tile -- 24 v_permlane32_swap per row block -- and
//   rqs_fast16_regs evaluates from there (first version: wave-private LDS transposes overlaying the activation buffers,
//   a second barrier in the back part, 104 KiB of LDS for them; 183 -> 176 us).
//   What bounds it (ablations, LABLOG R4.9): LDS reads of the B operands (2 KiB per fragment and wave = 1.9 MB per layer:
//   ~50 us at the LDS peak), the MFMAs (50 us) and the weight stream (7.7 MB per workgroup = 85 us at the CU's ingest rate)
//   overlap only partly; balancing the waves' k-step counts does not change the time (built and measured).
//
// Numerics = the large-batch kernel's: bf16 operands (x as a hi + lo pair), fp32 accumulation, fp32 residual / bias /
// spline; bias added after the chain.
#pragma once
#include <hip/hip_runtime.h>

#include "pf_flow_fwd_kernel.h"
#include "pf_wide_layout.h"

#ifndef PF_MID_ABLATE
#define PF_MID_ABLATE 0   // timing experiments only: 1 no spline, 2 every weight load on one address, 4 no MFMA, 8 no barriers, 16 no sigmoid
#endif

#ifndef PF_MID_P
#define PF_MID_P 6                       // weight fragments in flight per ring
#endif
#ifndef PF_MID_PRIO
#define PF_MID_PRIO 1                    // s_setprio around the MFMA chains: the SIMD's other wave (epilogue, spline) yields issue slots to the chain (185 -> 180 us)
#endif
#ifndef PF_MID_BD
#define PF_MID_BD 2                      // k-steps of B operands (LDS) in flight ahead of the MFMAs of the front part
#endif
#ifndef PF_MID_S1_EARLY
#define PF_MID_S1_EARLY 1
#endif
#ifndef PF_MID_TRACE
#define PF_MID_TRACE 0    // diagnostic build: wave PF_MID_TRACE_WAVE of workgroup 0 accumulates s_memtime spans into p.fail_flags
#endif
#ifndef PF_MID_TRACE_WAVE
#define PF_MID_TRACE_WAVE 0
#endif

namespace pf {
namespace mid {
constexpr int kRowsPerWG = 64, kWaves = 8, kThreads = 64 * kWaves;
constexpr int kXS = 16;                                            // floats per row of the x / z exchange
constexpr int kP = PF_MID_P;                                             // weight fragments in flight per ring (two rings)
constexpr int kActBytes = 2 * 2 * wide::kKSteps * wide::kFrag;     // [buffer][row block][k-step][1 KiB]
constexpr int kRegionA = kActBytes;
constexpr int ctx_bytes(int CKS) { return 2 * CKS * wide::kFrag; }
constexpr int kXBytes = 2 * 2 * 32 * kXS * 4;                      // [row block][current | next][32 rows][kXS]
constexpr int lds_bytes(int CKS) { return kRegionA + ctx_bytes(CKS) + kXBytes + 2 * wide::kBiasFloats * 4; }   // (two bias blocks: layers alternate)
}  // namespace mid

template <int D, int CKS>
__global__ __launch_bounds__(mid::kThreads) void flow_mid_kernel(const FwdParams p) {
    namespace W = wide;
    namespace M = mid;
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    typedef __attribute__((ext_vector_type(4))) unsigned int mu32x4;
    typedef __attribute__((address_space(3))) mu32x4 lds_mu32x4_t;
    typedef __attribute__((address_space(3))) f32x4 lds_mf32x4_t;
    typedef __attribute__((address_space(3))) float lds_mf32_t;
    constexpr int NFP = W::n_frags_padded(D, CKS);
    constexpr int NB = W::n_batches(D);
    constexpr int P = M::kP, XS = M::kXS;
    static_assert(D >= 2 && D <= 16 && CKS > 0, "2 <= D <= 16, conditional flow");
    static_assert(M::lds_bytes(CKS) <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hf = lane >> 5;
    const int ub = wave & 3, rb = wave >> 2;
    char* const s_act = smem;
    char* const s_ctx = smem + M::kRegionA;
    float* const s_x = reinterpret_cast<float*>(s_ctx + M::ctx_bytes(CKS));
    float* const s_bias = s_x + 2 * 2 * 32 * XS;
    const int C = p.plan.C, NL = p.plan.L;
    const int64_t row0 = (int64_t)blockIdx.x * M::kRowsPerWG;
    int64_t row = row0 + 32 * rb + n;
    const bool live = row < p.batch;
    if (!live) row = p.batch - 1;

    auto lds_a = [](const void* q) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)q; };
    auto ld_u4 = [](uint32_t a) { return *reinterpret_cast<const lds_mu32x4_t*>(a); };
    auto st_u4 = [](uint32_t a, mu32x4 v) { *reinterpret_cast<lds_mu32x4_t*>(a) = v; };
    auto ld_f4 = [](uint32_t a) { return *reinterpret_cast<const lds_mf32x4_t*>(a); };
    auto ld_f = [](uint32_t a) { return *reinterpret_cast<const lds_mf32_t*>(a); };
    auto st_f = [](uint32_t a, float v) { *reinterpret_cast<lds_mf32_t*>(a) = v; };

    // ---- staging: context as B fragments (lane (n, hf) element j of k-step ks = ctx[row][16 ks + 8 hf + j]), x, biases ----
    for (int f = wave; f < 2 * CKS; f += M::kWaves) {
        const int rbf = f / CKS, ks = f - rbf * CKS;
        int64_t r = row0 + 32 * rbf + n;
        if (r >= p.batch) r = p.batch - 1;
        const float* crow = p.ctx + r * C;
        bf16x8 o;
        if ((C & 3) == 0) {
            const int c0 = 16 * ks + 8 * hf;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (c0 < C) v0 = *reinterpret_cast<const f32x4*>(crow + c0);
            if (c0 + 4 < C) v1 = *reinterpret_cast<const f32x4*>(crow + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (__bf16)v0[j]; o[4 + j] = (__bf16)v1[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c0 = 16 * ks + 8 * hf + j;
                o[j] = (__bf16)(c0 < C ? crow[c0] : 0.f);
            }
        }
        *reinterpret_cast<bf16x8*>(s_ctx + (size_t)(rbf * CKS + ks) * W::kFrag + lane * 16) = o;
    }
    // x^T of the 64 rows: position d of layer 0 <- x[row][ar_perm[D-1-d]] (ReversePermutation first)
    for (int i = tid; i < 64 * 16; i += M::kThreads) {
        const int r64 = i >> 4, d = i & 15;
        int64_t r = row0 + r64;
        if (r >= p.batch) r = p.batch - 1;
        float v = 0.f;
        if (d < D) {
            const int sd = D - 1 - d;
            v = p.x[r * D + (p.ar_perm ? p.ar_perm[sd] : sd)];
        }
        float* dst = s_x + ((r64 >> 5) * 2) * 32 * XS + (r64 & 31) * XS + d;
        dst[0] = v;
        dst[32 * XS] = 0.f;
    }
    const float* gbias = reinterpret_cast<const float*>(p.packed + W::stream_frags(D, CKS, NL) * W::kFrag);
    for (int s = tid; s < W::kBiasFloats / 4; s += M::kThreads)
        reinterpret_cast<f32x4*>(s_bias)[s] = reinterpret_cast<const f32x4*>(gbias)[s];
    if (p.zero_pair && blockIdx.x == 0 && tid < 2 * PF_REDUCE_SLOTS) p.zero_pair[tid] = 0.f;
    __syncthreads();

    // ---- per-lane LDS addresses ---------------------------------------------------------------------------------------
    uint32_t sb = lds_a(s_bias + 4 * hf);                                         // this layer's bias block (layer l: block l & 1)
    const uint32_t sb_flip = lds_a(s_bias + 4 * hf) ^ lds_a(s_bias + W::kBiasFloats + 4 * hf);

    // ---- weight fragments: buffer loads, entry E of the layer at byte offset `base` -----------------------------------------
    // (num_records = the packed buffer's size: a fragment request past its end would return zeros instead of touching memory)
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.packed), 0, (int)W::packed_bytes(D, CKS, NL), 0x00020000);
    const int lane16 = lane * 16;
    auto ldA = [&](int base, int E) -> mu32x4 {
        if (PF_MID_ABLATE & 2) return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, 0, 0);
        return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + E * W::kFrag, 0);
    };
    mu32x4 ring0[P], ring1[P];
    // request the first fragments of a chain of N entries starting at E0
    auto begin = [&](auto e0, auto nn, auto& ring, int base) {
        constexpr int E0 = decltype(e0)::value, N = decltype(nn)::value;
        constexpr int RP = (int)(sizeof(ring) / sizeof(ring[0]));
        static_for<0, (N < RP ? N : RP)>([&](auto i) { ring[decltype(i)::value] = ldA(base, E0 + decltype(i)::value); });
    };
    // walk a chain: fragment k + ring depth is requested behind the MFMAs of k-step k, and the B operands (activations / context in
    // LDS) of k-step k + PF_MID_BD right behind them, in that order (sched_barrier): left to the compiler every ds_read sat one instruction ahead of the MFMA
    // that needs it, and a heavy wave paid an LDS round trip per k-step (320 cycles per step against 64 of MFMA)
    auto run_b = [&](auto e0, auto nn, auto& ring, int base, auto&& ldb, auto&& mm) {
        constexpr int E0 = decltype(e0)::value, N = decltype(nn)::value;
        constexpr int RP = (int)(sizeof(ring) / sizeof(ring[0]));
        constexpr int BD = PF_MID_BD;
        mu32x4 bq[BD][2];
        if (PF_MID_PRIO) __builtin_amdgcn_s_setprio(PF_MID_PRIO);
        static_for<0, (N < BD ? N : BD)>([&](auto i) {
            bq[decltype(i)::value][0] = ldb(i, ic<0>{});
            bq[decltype(i)::value][1] = ldb(i, ic<1>{});
        });
        static_for<0, N>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const mu32x4 a = ring[k % RP];
            if constexpr (k + RP < N) ring[k % RP] = ldA(base, E0 + k + RP);
            mm(kk, ic<0>{}, a, bq[k % BD][0]);
            mm(kk, ic<1>{}, a, bq[k % BD][1]);
            if constexpr (k + BD < N) {
                bq[k % BD][0] = ldb(ic<k + BD>{}, ic<0>{});
                bq[k % BD][1] = ldb(ic<k + BD>{}, ic<1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (PF_MID_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mma = [&](auto first, const mu32x4& a, const bf16x8& b, f32x16& acc) {
        if (PF_MID_ABLATE & 4) { asm volatile("" :: "v"(a), "v"(b)); if constexpr (decltype(first)::value) acc = zero16; return; }
        if constexpr (decltype(first)::value)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, zero16, 0, 0, 0);
        else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b, acc, 0, 0, 0);
    };
    // registers 8 s .. 8 s + 7 of an accumulator tile -> the B fragment of k-step 2 T + s of the next GEMM
    auto to_b = [&](const f32x16& v, int s, bool relu) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = v[8 * s + j];
            o[j] = (__bf16)(relu ? fmaxf(t, 0.f) : t);
        }
        return __builtin_bit_cast(mu32x4, o);
    };
    unsigned long long tbar = 0;
    auto barrier = [&]() {
        if (PF_MID_ABLATE & 8) return;
        const unsigned long long t0 = PF_MID_TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (PF_MID_TRACE) tbar += __builtin_amdgcn_s_memtime() - t0;
    };

    float ld_acc[2] = {0.f, 0.f};                           // log-det terms of this lane's (row n of row block r, feature) pairs
    int lbase = 0;                                         // byte offset of the current layer's fragments
    // diagnostic spans (PF_MID_TRACE): 0 stage 1, 1 W0 (incl. its exchange), 2 W1 + gate, 3 final exchange + GEMMs, 4 transposes +
    // splines, 5 layer end (barrier + bias reload), 6 barriers (inside the other spans), 7 whole kernel
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> unsigned long long { return PF_MID_TRACE ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long t_begin = tick();
    unsigned long long ts = t_begin;
    auto span = [&](int i) { if (PF_MID_TRACE) { const unsigned long long t = tick(); tr[i] += t - ts; ts = t; } };

    uint32_t sx_base = lds_a(s_x);
    int sxc_off = 0;                                        // 0 / 1: which half of the x exchange is the current layer's input
    // per-lane LDS addresses of BOTH row blocks (the front part of a layer -- stage 1 and the residual blocks -- is split by
    // hidden tile, not by row block: see `front`)
    uint32_t act_0 = lds_a(s_act) + lane * 16;                                      // + rb * 16 KiB + buffer * 32 KiB + ks * 1 KiB
    uint32_t ctx_0 = lds_a(s_ctx) + lane * 16;                                      // + rb * CKS KiB + ks * 1 KiB
    constexpr int kActRb = W::kKSteps * W::kFrag, kCtxRb = CKS * W::kFrag;
    f32x16 h[2];                                             // residual state of this wave's tile: row block 0 | 1

    // entries of feature batch m's final-layer chain (its three tiles are consecutive in the stream)
    auto out_len_f = [](int m) constexpr { return m < 0 ? 0 : W::kO16(D, 2 * m) + (2 * m + 1 < D ? W::kWHb(D, m) : 0) + W::kDD(D, m); };
#define out_len(m) (out_len_f(m))
    // ---- front part of a layer, for the wave that owns hidden tile T (compile time) for BOTH row blocks ------------------
    // Stage 1 and the residual blocks are split over the 8 waves by TILE (wave w < 4: tile w, wave w >= 4: tile 11 - w, so
    // that the two waves of a SIMD own a light and a heavy tile of the masked layers): a weight fragment is fetched ONCE per
    // workgroup and multiplied with both row blocks' B operands (first version: a wave owned two tiles of ONE row block, both
    // row-block waves fetched every fragment, and the CU's vector-memory path -- ~95 GB/s -- was the bound: 15 MB per
    // workgroup = 158 of 243 us).  Chains alternate between the two rings; stage 1 runs on ring1.
    auto front = [&](auto tc, auto mbc, int l) {
        constexpr int T = decltype(tc)::value, MB = decltype(mbc)::value;   // MB: this wave's feature batch in the back part (< 0: none)
        (void)l;
        // ---- stage 1: h = W_in x + b_in + relu(W_c ctx + b_c) --------------------------------------------------------
        {
            bf16x8 xhi[2], xlo[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t sx = sx_base + ((r * 2 + sxc_off) * 32 * XS + n * XS) * 4;
                const f32x4 v0 = ld_f4(sx + 32 * hf), v1 = ld_f4(sx + 32 * hf + 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = j < 4 ? v0[j & 3] : v1[j & 3];
                    const __bf16 hi = (__bf16)v;
                    xhi[r][j] = hi;
                    xlo[r][j] = (__bf16)(v - (float)hi);
                }
            }
            f32x16 a1[2], a2[2];
            begin(ic<W::e_blk(D, CKS, 0) + W::w0_off(D, T)>{}, ic<W::kH16(D, T)>{}, ring0, lbase);
            run_b(ic<W::e_in(CKS, T)>{}, ic<2 + CKS>{}, ring1, lbase,
                  [&](auto kk, auto rc) -> mu32x4 {
                      constexpr int k = decltype(kk)::value, r = decltype(rc)::value;
                      if constexpr (k == 0) return __builtin_bit_cast(mu32x4, xhi[r]);
                      else if constexpr (k == 1) return __builtin_bit_cast(mu32x4, xlo[r]);
                      else return ld_u4(ctx_0 + r * kCtxRb + (k - 2) * W::kFrag);
                  },
                  [&](auto kk, auto rc, const mu32x4& a, const mu32x4& b) {
                      constexpr int k = decltype(kk)::value, r = decltype(rc)::value;
                      if constexpr (k < 2) mma(ic<(k == 0)>{}, a, __builtin_bit_cast(bf16x8, b), a1[r]);
                      else mma(ic<(k == 2)>{}, a, __builtin_bit_cast(bf16x8, b), a2[r]);
                  });
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bi = ld_f4(sb + 4 * (W::kBiasIn + 8 * (4 * T + q)));
                const f32x4 bc = ld_f4(sb + 4 * (W::kBiasCtx + 8 * (4 * T + q)));
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        h[r][4 * q + e] = a1[r][4 * q + e] + bi[e] + fmaxf(a2[r][4 * q + e] + bc[e], 0.f);
            }
        }
        span(0);
        // ---- residual blocks: h += (W1 relu(W0 relu(h) + b0) + b1) * sigmoid(W_g ctx + b_g) ---------------------------
        static_for<0, 2>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            constexpr int EB = W::e_blk(D, CKS, b), EW1 = EB + W::w0_len(D);
            constexpr int OB0 = W::kBiasBlk + 768 * b, OB1 = OB0 + 256, OBG = OB0 + 512;
            constexpr int NW = W::kH16(D, T);
            // relu(h) of every tile -> buffer 0 -> everyone's B operand
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                st_u4(act_0 + r * kActRb + (2 * T) * W::kFrag, to_b(h[r], 0, true));
                st_u4(act_0 + r * kActRb + (2 * T + 1) * W::kFrag, to_b(h[r], 1, true));
            }
            barrier();
            // W0: t = relu(W0 . + b0) -> buffer 1.  (B fragments are read from LDS per k-step: 2 KiB per weight fragment and wave)
            {
                f32x16 acc[2];
                begin(ic<EW1 + W::w1_off(D, CKS, T)>{}, ic<NW + CKS>{}, ring1, lbase);
                run_b(ic<EB + W::w0_off(D, T)>{}, ic<NW>{}, ring0, lbase,
                      [&](auto kk, auto rc) -> mu32x4 { return ld_u4(act_0 + decltype(rc)::value * kActRb + decltype(kk)::value * W::kFrag); },
                      [&](auto kk, auto rc, const mu32x4& a, const mu32x4& b) {
                          mma(ic<(decltype(kk)::value == 0)>{}, a, __builtin_bit_cast(bf16x8, b), acc[decltype(rc)::value]);
                      });
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b0 = ld_f4(sb + 4 * (OB0 + 8 * (4 * T + q)));
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r][4 * q + e] = acc[r][4 * q + e] + b0[e];
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    st_u4(act_0 + M::kActBytes / 2 + r * kActRb + (2 * T) * W::kFrag, to_b(acc[r], 0, true));
                    st_u4(act_0 + M::kActBytes / 2 + r * kActRb + (2 * T + 1) * W::kFrag, to_b(acc[r], 1, true));
                }
            }
            span(1);
            barrier();
            // W1 + gate
            {
                f32x16 accw[2], accg[2];
                // what ring0 serves next: the next block's W0, or the first final-layer chain of this wave's unit block
                if constexpr (b == 0) begin(ic<W::e_blk(D, CKS, 1) + W::w0_off(D, T)>{}, ic<NW>{}, ring0, lbase);
                else if constexpr (MB >= 0) begin(ic<W::e_out(D, CKS) + W::out_off(D, MB)>{}, ic<out_len(MB)>{}, ring0, lbase);
                run_b(ic<EW1 + W::w1_off(D, CKS, T)>{}, ic<NW + CKS>{}, ring1, lbase,
                      [&](auto kk, auto rc) -> mu32x4 {
                          constexpr int k = decltype(kk)::value, r = decltype(rc)::value;
                          if constexpr (k < NW) return ld_u4(act_0 + M::kActBytes / 2 + r * kActRb + k * W::kFrag);
                          else return ld_u4(ctx_0 + r * kCtxRb + (k - NW) * W::kFrag);
                      },
                      [&](auto kk, auto rc, const mu32x4& a, const mu32x4& b) {
                          constexpr int k = decltype(kk)::value, r = decltype(rc)::value;
                          if constexpr (k < NW) mma(ic<(k == 0)>{}, a, __builtin_bit_cast(bf16x8, b), accw[r]);
                          else mma(ic<(k == NW)>{}, a, __builtin_bit_cast(bf16x8, b), accg[r]);
                      });
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b1 = ld_f4(sb + 4 * (OB1 + 8 * (4 * T + q)));
                    const f32x4 bg = ld_f4(sb + 4 * (OBG + 8 * (4 * T + q)));
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            h[r][4 * q + e] += (accw[r][4 * q + e] + b1[e]) *
                                               ((PF_MID_ABLATE & 16) ? accg[r][4 * q + e] + bg[e] : pf_sigmoid<true>(accg[r][4 * q + e] + bg[e]));
                }
            }
            span(2);
        });
        // h -> buffer 0: the final layer's B operand (no activation in front of it)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            st_u4(act_0 + r * kActRb + (2 * T) * W::kFrag, to_b(h[r], 0, false));
            st_u4(act_0 + r * kActRb + (2 * T + 1) * W::kFrag, to_b(h[r], 1, false));
        }
    };

    // ---- back part of a layer (final masked layer + spline), for the wave that owns feature batch MB for BOTH row blocks --------
    // A batch = widths | heights of feature 2 MB, of 2 MB + 1, the derivatives of both: three 32-unit tiles whose fragments are
    // consecutive in the stream -- ONE chain of NA + NB + ND entries on ring0, every fragment multiplied with both row blocks'
    // h (B operand read from buffer 0 like the front part's; first version: a wave owned two batches of ONE row block with h
    // preloaded into 64 registers, every final-layer fragment was fetched by two waves and fed one MFMA, and with four
    // fragments in flight per wave the back part was 40 % of the kernel).  The transposes overlay the activation buffers,
    // so nobody writes one before everybody is done with h (second barrier).
    auto back = [&](auto mbc, int l) {
        constexpr int MB = decltype(mbc)::value;
        constexpr bool HASM = MB >= 0;
        barrier();                                            // h of every tile is in buffer 0
        f32x16 acc[3][2];                                     // [widths | heights of 2 MB, of 2 MB + 1, derivatives][row block]
        if constexpr (HASM) {
            constexpr int E0 = W::e_out(D, CKS) + W::out_off(D, MB);
            constexpr bool HASB = 2 * MB + 1 < D;
            constexpr int NA = W::kO16(D, 2 * MB), NBB = HASB ? W::kWHb(D, MB) : 0, ND = W::kDD(D, MB);
            static_assert(NA + NBB + ND == out_len(MB), "chain length");
#pragma unroll
            for (int c = 0; c < 3; ++c) { acc[c][0] = zero16; acc[c][1] = zero16; }
            run_b(ic<E0>{}, ic<NA + NBB + ND>{}, ring0, lbase,
                  [&](auto kk, auto rc) -> mu32x4 {
                      constexpr int k = decltype(kk)::value, r = decltype(rc)::value;
                      constexpr int kl = k < NA ? k : k < NA + NBB ? k - NA : k - NA - NBB;
                      return ld_u4(act_0 + r * kActRb + kl * W::kFrag);
                  },
                  [&](auto kk, auto rc, const mu32x4& a, const mu32x4& b) {
                      constexpr int k = decltype(kk)::value, r = decltype(rc)::value;
                      constexpr int c = k < NA ? 0 : k < NA + NBB ? 1 : 2;
                      mma(ic<0>{}, a, __builtin_bit_cast(bf16x8, b), acc[c][r]);
                  });
        }
        span(3);
        // ---- spline: one (row, feature) pair per lane -- row n of row block r, feature 2 MB + hf -- with the parameters handed over
        // IN REGISTERS.  In the accumulator layout lane (n, hf) holds units 8 q + 4 hf .. + 3 of each tile for row n; the lane that
        // evaluates feature 2 MB (hf = 0) needs all 32 units of tile A (widths | heights) and units 0 .. 15 of tile D, its partner
        // (n, hf = 1) all of tile B and units 16 .. 31 of D: v_permlane32_swap exchanges exactly those halves (lanes 32-63 of one
        // register with lanes 0-31 of another), 24 swaps per row block.  (First version: through wave-private LDS transposes that
        // overlaid the activation buffers -- 24 b128 stores + 24 b128 loads per lane and row block, two drains, and a second
        // barrier in the back part because nobody may overwrite h before everyone has multiplied it.)
        auto spline = [&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr bool HASB = 2 * MB + 1 < D;
            constexpr int OB = W::kBiasOut + 96 * MB;
            f32x16& tA = acc[0][r];
            f32x16& tB = acc[1][r];
            f32x16& tD = acc[2][r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // biases, still in the accumulator layout
                const f32x4 ba = ld_f4(sb + 4 * (OB + 8 * q));
                const f32x4 bd = ld_f4(sb + 4 * (OB + 8 * (8 + q)));
#pragma unroll
                for (int e = 0; e < 4; ++e) { tA[4 * q + e] += ba[e]; tD[4 * q + e] += bd[e]; }
                if constexpr (HASB) {
                    const f32x4 bbv = ld_f4(sb + 4 * (OB + 8 * (4 + q)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) tB[4 * q + e] += bbv[e];
                }
            }
            float uw[16], uh[16], kd[17];
            static_for<0, 16>([&](auto jj) {
                constexpr int j = decltype(jj)::value, u0 = 8 * (j >> 2) + (j & 3);      // unit of register j in the lanes hf = 0; + 4: hf = 1
                // (copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index -- clang 19)
                const float ea = tA[j], eb = tB[j];
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
                const unsigned s0 = sw[0], s1 = sw[1];      // (copies again: see above)
                if constexpr (u0 < 16) { uw[u0] = __builtin_bit_cast(float, s0); uw[u0 + 4] = __builtin_bit_cast(float, s1); }
                else { uh[u0 - 16] = __builtin_bit_cast(float, s0); uh[u0 - 12] = __builtin_bit_cast(float, s1); }
            });
            static_for<0, 8>([&](auto jj) {
                constexpr int j = decltype(jj)::value, u0 = 8 * (j >> 2) + (j & 3);
                const float ea = tD[j], eb = tD[j + 8];
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
                const unsigned s0 = sw[0], s1 = sw[1];
                kd[1 + u0] = __builtin_bit_cast(float, s0);
                kd[1 + u0 + 4] = __builtin_bit_cast(float, s1);
            });
            const int f = 2 * MB + hf;
            if (f < D) {
                const uint32_t sxr = sx_base + ((r * 2) * 32 * XS + n * XS) * 4;
                const float xv = ld_f(sxr + sxc_off * (32 * XS * 4) + 4 * f);
                const int64_t rr = row0 + 32 * r + n;
                if (p.u_save && rr < p.batch) p.u_save[((int64_t)l * p.batch + rr) * D + f] = xv;
                float y, ld;
                if (PF_MID_ABLATE & 1) { y = xv + uw[0]; ld = 0.f; }
                else rqs_fast16_regs(uw, uh, kd, xv, p, y, ld);
                ld_acc[r] += ld;
                st_f(sxr + (sxc_off ^ 1) * (32 * XS * 4) + 4 * (D - 1 - f), y);   // the next layer starts with ReversePermutation
            }
        };
        if constexpr (HASM) {
            if constexpr (!(2 * MB + 1 < D)) { acc[1][0] = zero16; acc[1][1] = zero16; }
            spline(ic<0>{});
            spline(ic<1>{});
        }
        span(4);
    };

    // wave w owns hidden tile w (w < 4) or 11 - w (w >= 4) in the front part and, in the back part, feature batch NB - 1 - w
    // (w < 4: the long chains go to the waves with the short hidden tiles) or w - 4 (w >= 4); both parts' code is per wave
    // (tile and batch are compile-time constants there)
#define mb_of(w) ((w) < 4 ? NB - 1 - (w) : ((w) - 4 < NB - 4 ? (w) - 4 : -1))
    auto stage1_begin = [&](int base) {                     // the first fragments of this wave's stage-1 chain -> ring1
#ifndef PF_MID_TILEMAP
#define PF_MID_TILEMAP 0   // wave -> hidden tile: 0: (0 1 2 3 | 7 6 5 4) light + heavy per SIMD; 1: (7 5 3 1 | 6 4 2 0) like with like
#endif
#if PF_MID_TILEMAP == 0
#define PF_MID_T0 0
#define PF_MID_T1 1
#define PF_MID_T2 2
#define PF_MID_T3 3
#define PF_MID_T4 7
#define PF_MID_T5 6
#define PF_MID_T6 5
#define PF_MID_T7 4
#else
#define PF_MID_T0 7
#define PF_MID_T1 5
#define PF_MID_T2 3
#define PF_MID_T3 1
#define PF_MID_T4 6
#define PF_MID_T5 4
#define PF_MID_T6 2
#define PF_MID_T7 0
#endif
        if (wave == 0) begin(ic<W::e_in(CKS, PF_MID_T0)>{}, ic<2 + CKS>{}, ring1, base);
        else if (wave == 1) begin(ic<W::e_in(CKS, PF_MID_T1)>{}, ic<2 + CKS>{}, ring1, base);
        else if (wave == 2) begin(ic<W::e_in(CKS, PF_MID_T2)>{}, ic<2 + CKS>{}, ring1, base);
        else if (wave == 3) begin(ic<W::e_in(CKS, PF_MID_T3)>{}, ic<2 + CKS>{}, ring1, base);
        else if (wave == 4) begin(ic<W::e_in(CKS, PF_MID_T4)>{}, ic<2 + CKS>{}, ring1, base);
        else if (wave == 5) begin(ic<W::e_in(CKS, PF_MID_T5)>{}, ic<2 + CKS>{}, ring1, base);
        else if (wave == 6) begin(ic<W::e_in(CKS, PF_MID_T6)>{}, ic<2 + CKS>{}, ring1, base);
        else begin(ic<W::e_in(CKS, PF_MID_T7)>{}, ic<2 + CKS>{}, ring1, base);
    };
    stage1_begin(0);
    for (int l = 0; l < NL; ++l) {
        // per-lane LDS bases re-"defined" at the top of every layer: left alone, the compiler hoists every base + constant out
        // of the layer loop into its own register and spills them (37 spilled VGPRs, reloaded behind the weight loads in
        // flight); this way base + constant stays an instruction offset (all of them fit the 16-bit field)
        asm volatile("" : "+v"(act_0), "+v"(ctx_0), "+v"(sb), "+v"(sx_base));
        if (wave == 0) front(ic<PF_MID_T0>{}, ic<mb_of(0)>{}, l);
        else if (wave == 1) front(ic<PF_MID_T1>{}, ic<mb_of(1)>{}, l);
        else if (wave == 2) front(ic<PF_MID_T2>{}, ic<mb_of(2)>{}, l);
        else if (wave == 3) front(ic<PF_MID_T3>{}, ic<mb_of(3)>{}, l);
        else if (wave == 4) front(ic<PF_MID_T4>{}, ic<mb_of(4)>{}, l);
        else if (wave == 5) front(ic<PF_MID_T5>{}, ic<mb_of(5)>{}, l);
        else if (wave == 6) front(ic<PF_MID_T6>{}, ic<mb_of(6)>{}, l);
        else front(ic<PF_MID_T7>{}, ic<mb_of(7)>{}, l);
        // the next layer's biases: requested now, parked in LDS behind the layer's end (their L2 round trip runs under the back part)
        constexpr int NBQ4 = W::kBiasFloats / 4;
        static_assert(NBQ4 <= 2 * M::kThreads, "two bias quads per thread");
        f32x4 nb0, nb1;
        {
            const f32x4* gb = reinterpret_cast<const f32x4*>(gbias + (int64_t)(l + 1 < NL ? l + 1 : l) * W::kBiasFloats);
            nb0 = gb[tid];
            nb1 = gb[tid + M::kThreads < NBQ4 ? tid + M::kThreads : 0];
        }
        // next layer's stage 1 -> ring1, free from here.  Behind the last layer there is no next layer: the request is aimed at
        // layer 0 (never multiplied) -- entry 146 of a "layer L" would lie up to 146 KiB behind the stream, past the end of the
        // packed buffer when L < 7 (72 KiB of zero tail + 11 KiB of biases per layer)
        const int next_base = l + 1 < NL ? lbase + NFP * W::kFrag : 0;
        if (PF_MID_S1_EARLY) stage1_begin(next_base);
        if (wave == 0) back(ic<mb_of(0)>{}, l);
        else if (wave == 1) back(ic<mb_of(1)>{}, l);
        else if (wave == 2) back(ic<mb_of(2)>{}, l);
        else if (wave == 3) back(ic<mb_of(3)>{}, l);
        else if (wave == 4) back(ic<mb_of(4)>{}, l);
        else if (wave == 5) back(ic<mb_of(5)>{}, l);
        else if (wave == 6) back(ic<mb_of(6)>{}, l);
        else back(ic<mb_of(7)>{}, l);
        if (!PF_MID_S1_EARLY) stage1_begin(next_base);
        sxc_off ^= 1;
        lbase += NFP * W::kFrag;
        // the next layer's biases go to the OTHER bias block (nobody reads it during this layer), so the barrier that ends the layer
        // -- every spline has written its z -- publishes them too (first version: one block, a second barrier behind the stores)
        {
            f32x4* nb = reinterpret_cast<f32x4*>(s_bias + ((l + 1) & 1) * W::kBiasFloats);
            nb[tid] = nb0;
            if (tid + M::kThreads < NBQ4) nb[tid + M::kThreads] = nb1;
        }
        sb ^= sb_flip;
        barrier();
        span(5);
    }
    if (PF_MID_TRACE && p.fail_flags && blockIdx.x == 0 && wave == PF_MID_TRACE_WAVE && lane == 0) {
        tr[6] = tbar;
        tr[7] = tick() - t_begin;
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(p.fail_flags)[i] = tr[i];
    }

    // ---- epilogue: log-det of a row = sum over its features (two lanes x four unit blocks), base density, stores ------------
    float* const s_red = reinterpret_cast<float*>(smem);       // [row block][wave][32 rows] (the transposes are dead)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float ld_half = ld_acc[r] + __shfl_xor(ld_acc[r], 32, 64);
        if (hf == 0) s_red[(r * 8 + wave) * 32 + n] = ld_half;
    }
    __syncthreads();
    float my_nll = 0.f, my_cnt = 0.f;
    if (ub == 0) {
        const float* sr = s_red + rb * 8 * 32 + n;
        const float ld_row = ((sr[0] + sr[32]) + (sr[64] + sr[96])) + ((sr[128] + sr[160]) + (sr[192] + sr[224]));
        const uint32_t sxc = sx_base + ((rb * 2 + sxc_off) * 32 * XS + n * XS) * 4;     // the last layer's output
        if (hf == 0 && live) {
            float q = 0.f, sls = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float zv = ld_f(sxc + 4 * (D - 1 - d));     // stored reversed
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
            if (lane == 0) { float* acc = p.nll_sum + 2 * ((2 * blockIdx.x + rb) % PF_REDUCE_SLOTS); atomicAdd(acc, my_nll); atomicAdd(acc + 1, my_cnt); }
        }
    }
}

}  // namespace pf
