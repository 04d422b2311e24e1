// pf_flow_bwd_chain.hip -- the data-gradient chain of the flow's backward pass in ONE kernel (fp32).
//
// What it replaces: step 2 of posteriflow_amd/_flow_autograd.py::_flow_backward_batched, ~25 small launches per layer
// (one pf_flow_rqs_backward + 6 matmuls + ~16 element-wise ops, ~250-300 per backward of a 10-layer flow), i.e. the
// part of autograd's walk through nflows' MADE that the reference executes under experiments/train_lean_npe.py:363-368
// (loss.backward()).  The layer-batched conditioner re-evaluation (step 1) and the weight-gradient GEMMs (step 3) are
// plain batched library GEMMs and stay where they are.
//
// Per layer l, last layer first, for a workgroup's 16 batch rows (rows are independent, layers sequential):
//   spline:  (gy, glad) -> Gp = dL/d(raw spline parameters) [D (3K-1)],  gu = direct dL/du          (pf_rqs_bwd.h)
//   gh  = Wf^T Gp
//   for block j = 1, 0:   gt2 = gh * gate_j            Gc[1+j] = gh * t2_j * gate_j (1 - gate_j)
//                         gt1 = (W2_j^T gt2) * [t1_j > 0]     (* the forward's dropout factor, if any)
//                         gh += (W1_j^T gt1) * [h_j > 0]
//   Gh0 = gh              Gc[0] = gh * [pc > 0]         gu += W0^T gh        gy <- flip(gu)   (ReversePermutation)
// Gp, Gh0, Gt1, Gt2, Gc go to HBM for the weight-gradient GEMMs; h_j, t1_j, t2_j, gate_j, pc and the raw parameters
// come from the re-evaluation.
//
// Work split: workgroup = 16 rows = one MFMA column tile, NW waves (8 at H = 128 / 256, else 4); everything TRANSPOSED as in
// the forward kernels,
// out^T[unit, row] = W^T[unit, k] . in^T[k, row] on v_mfma_f32_16x16x4_f32 (exact fp32: gradients keep the 3e-4 agreement
// with autograd through the oracle).  Wave w owns unit tiles w, w + NW, ...; the B operand (the incoming gradient vector
// of all units) is exchanged through LDS as fp32 [row][k] rows, the A operand is read straight from the dense
// TRANSPOSED masked weight matrices the caller prepares each step ((W * mask)^T, row-major, k padded to 16): lane
// (r, g) of k-group q takes the 4 consecutive floats W^T[16 t + r][16 q + 4 g ..] as the A values of 4 MFMAs whose
// k order is (16 q + 4 g + e) -- the B rows in LDS are read the same way, so neither side needs packing.
// Masked zeros are multiplied (the dense count): at 16 rows per CU this kernel is bound by the 1.8 MB of weights a
// layer streams per workgroup, ~0.3 ms per launch whatever the batch up to 4096 rows.
//
// bf16 descs (BF = true; the throughput mode a trainer runs in): the same walk on v_mfma_f32_16x16x32_bf16.  The A operand
// is the PF_FLAG_BWD stream (pf_pack.hip: the transposed masked matrices as bf16 A-fragments, one contiguous 1-KiB load per
// wave and (tile, k-step) instead of sixteen 64-B row segments of the fp32 matrices -- those touched every 128-B line
// twice, half of it each time, with an L1 too small to keep the line in between); the gradient vectors cross LDS as bf16
// rows; accumulators, spline, gate / ReLU algebra and every output stay fp32 (the library's bf16 GEMMs are only 1.5x faster
// than its fp32 ones at these sizes, less than the casts of their operands cost, so the weight-gradient GEMMs stay fp32).
// 8x fewer MFMA instructions, half the weight bytes.
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <cstdint>

#include "../../include/pf_hip.h"
#include "pf_rqs_bwd.h"

namespace pf {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ChainArgs {
    PfFlowBwdChainArgs a;
    RqsConsts c;
    int D, H, L, M, PM;       // M = 3K-1, PM = D*M padded to 16
    int64_t B;
    int additive;             // masked-context conditioner (flows.py:186-234): t1 = W1 relu(h) + b1 + context_layer(ctx), no gate, no
                              // ReversePermutation; Gc[3 l + 1 + j] = dL/d(that projection) = gt1_j
};

// NW waves per workgroup, TPW = H / (16 NW) unit tiles per wave; BF: bf16 operands from the PF_FLAG_BWD stream.
// 8 waves (two per SIMD) where the tile count allows it (H = 128, 256): 569 -> 459 us per launch against 4 waves --
// a second wave per SIMD runs its MFMAs / LDS reads under the first one's loads (pitfall 9).
#ifndef PF_CHAIN_ABLATE
#define PF_CHAIN_ABLATE 0      // timing experiments (side builds): 1 no spline backward, 2 no MFMA, 4 no gradient stores, 8 no activation loads
#endif
template <int TPW, bool BF, int NW, int KMAX = 16>
__global__ __launch_bounds__(NW * 64) void flow_bwd_chain_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int D = p.D, H = p.H, M = p.M, PM = p.PM, L = p.L;
    const int64_t B = p.B, row0 = (int64_t)blockIdx.x * 16;
    const int PMS = PM + 4, HS = H + 4;
    float* s_gp = reinterpret_cast<float*>(smem);            // [16][PMS]: raw parameters, then their gradients
    float* s_v0 = s_gp + 16 * PMS;                            // [16][HS]: gradient vectors exchanged between GEMMs
    float* s_v1 = s_v0 + 16 * HS;
    float* s_gy = s_v1 + 16 * HS;                             // [16][16] dL/dy of the current layer
    float* s_gu = s_gy + 256;                                 // [16][16] direct dL/du
    float* s_part = s_gu + 256;                               // [NW][16][16] partial sums of the W0^T product
    // BF: the B operands as bf16 rows (k contiguous): the two exchange vectors alias s_v0 / s_v1, Gp gets its own image
    const int KSF = (D * M + 31) / 32, HK = H / 32, NT = H / 16;
    const int HSB = H + 8, GSB = 32 * KSF + 8;                // row strides in bf16 elements
    __bf16* s_b0 = reinterpret_cast<__bf16*>(s_v0);
    __bf16* s_b1 = reinterpret_cast<__bf16*>(s_v1);
    __bf16* s_gpb = reinterpret_cast<__bf16*>(s_part + NW * 256); // [16][GSB]
    const u32x4* frags = reinterpret_cast<const u32x4*>(p.a.packed);
    const int layer_frags = NT * KSF + 4 * NT * HK + HK;
    const int64_t my_row = row0 + c < B ? row0 + c : B - 1;   // clamped (stores are guarded)
    const bool live = row0 + c < B;
    const PfFlowBwdChainArgs& A = p.a;
    const bool additive = p.additive != 0;
    const bool has_ctx = A.pc != nullptr;                     // (gates / t2s only exist for the GLU conditioner)
    const bool glu = has_ctx && !additive;

    for (int s = tid; s < 256; s += NW * 64) {
        const int r = s >> 4, d = s & 15;
        const int64_t row = row0 + r < B ? row0 + r : B - 1;
        float v = 0.f;
        if (d < D) {
            if (A.g_nll) {       // loss = nll: dL/dz = g_nll z e^{-2 ls}
                const float ls = A.log_sigma ? A.log_sigma[row * D + d] : 0.f;
                v = A.g_nll[row] * A.nll_z[row * D + d] * __expf(-2.f * ls);
            } else v = A.g_z[row * D + d];
        }
        s_gy[s] = v;
    }
    __syncthreads();

    // out^T tile(s) of this wave = W^T[16 t + r][k] . s_in[col][k], k = 0 .. K-1 (K a multiple of 16)
    auto gemm = [&](const float* WT, int ldk, int K, const float* s_in, int lds_stride, f32x4 (&acc)[TPW]) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wrow[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) wrow[i] = WT + (size_t)(16 * (wave + NW * i) + c) * ldk + 4 * g;
        const float* brow = s_in + c * lds_stride + 4 * g;
        const int nq = K >> 4;
        // the A values of k-group q are requested two groups ahead (an L2 round trip is longer than the 16 MFMAs of a group)
        f32x4 a0[TPW], a1[TPW], a2[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            a0[i] = *reinterpret_cast<const f32x4*>(wrow[i]);
            a1[i] = *reinterpret_cast<const f32x4*>(wrow[i] + 16 * (nq > 1 ? 1 : 0));
        }
        for (int q = 0; q < nq; ++q) {
            const int qn = q + 2 < nq ? q + 2 : nq - 1;
#pragma unroll
            for (int i = 0; i < TPW; ++i) a2[i] = *reinterpret_cast<const f32x4*>(wrow[i] + 16 * qn);
            const f32x4 b = *reinterpret_cast<const f32x4*>(brow + 16 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)                      // tiles innermost: consecutive MFMAs are independent
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], b[e], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) { a0[i] = a1[i]; a1[i] = a2[i]; }
        }
    };
    // bf16: A = fragments [tile][ks][lane] of the packed stream, B = bf16 row c of s_in, k = 32 ks + 8 g ..
    // ---- BF: the fragment stream of a wave is static, so it is requested ACROSS phase and layer boundaries ----------
    // (ablations: ~300 of the 458 us were fifty dependent GEMM phases each starting with a cold round trip to the MALL)
    //   ring[ks][i]: the H x H phase in flight -- slot ks is refilled with the NEXT phase's k-step ks as soon as it is
    //   consumed, so every phase finds its fragments requested one whole phase earlier;
    //   FA[q][i]: the final layer's transpose (KSF k-steps, run-time count), PF k-steps ahead; its first PF k-steps and the
    //   first H x H phase are requested right behind the spline backward, under the Gp store and the bf16 image.
    constexpr int HKc = NW * TPW / 2;                         // H / 32
    constexpr int PF = 8;
    u32x4 ring[BF ? HKc : 1][TPW], FA[BF ? PF : 1][TPW];
    auto tile_ptr = [&](const u32x4* base, int nks, int i) { return base + (size_t)(wave + NW * i) * nks * 64 + lane; };
    auto drain = [&](f32x4 (&acc)[TPW]) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) asm volatile("s_nop 7" : "+a"(acc[i]));
        asm volatile("s_nop 15");
    };
    auto issue_phase = [&](const u32x4* base) {
#pragma unroll
        for (int ks = 0; ks < HKc; ++ks)
#pragma unroll
            for (int i = 0; i < TPW; ++i) ring[ks][i] = tile_ptr(base, HKc, i)[64 * ks];
    };
    auto run_phase = [&](const __bf16* s_in, const u32x4* next, f32x4 (&acc)[TPW]) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __bf16* brow = s_in + c * HSB + 8 * g;
#pragma unroll
        for (int ks = 0; ks < HKc; ++ks) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + 32 * ks);
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[ks][i]), b, acc[i], 0, 0, 0);
            if (next) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) ring[ks][i] = tile_ptr(next, HKc, i)[64 * ks];
            }
        }
        drain(acc);
    };
    auto f_prefill = [&](const u32x4* fbase) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int k = q < KSF ? q : KSF - 1;
#pragma unroll
            for (int i = 0; i < TPW; ++i) FA[q][i] = tile_ptr(fbase, KSF, i)[64 * k];
        }
    };
    auto f_run = [&](const u32x4* fbase, f32x4 (&acc)[TPW]) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __bf16* brow = s_gpb + c * GSB + 8 * g;
        // branch-free body: k-steps beyond KSF multiply a zero B operand and re-request the last fragment (an L1 hit).
        // With `if (k < KSF)` around each step hipcc moved the accumulators between register sets behind every
        // conditional block -- v_accvgpr_read / v_accvgpr_mov 5-7 wait states behind the MFMA where 8 are required
        // (scripts/audit_accvgpr.py found 17 such sites in the bf16 variants of this kernel).
        for (int k0 = 0; k0 < KSF; k0 += PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int k = k0 + q;
                const int kc = k < KSF ? k : KSF - 1;
                u32x4 bu = *reinterpret_cast<const u32x4*>(brow + 32 * kc);
                if (k >= KSF) bu = u32x4{0u, 0u, 0u, 0u};
                const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
#pragma unroll
                for (int i = 0; i < TPW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, FA[q][i]), b, acc[i], 0, 0, 0);
                const int kn = k + PF < KSF ? k + PF : KSF - 1;
#pragma unroll
                for (int i = 0; i < TPW; ++i) FA[q][i] = tile_ptr(fbase, KSF, i)[64 * kn];
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) asm volatile("s_nop 7" : "+a"(acc[i]));   // loop-carried accumulators stay put
        }
        drain(acc);
    };
    auto hh_phase = [&](const u32x4* lfp, int idx) { return lfp + ((size_t)NT * KSF + (size_t)idx * NT * HK) * 64; };

    // C-layout access of slab `idx` of a [slabs][B][H] tensor: units 16 t + 4 g .. + 3 of row my_row
    // ([L][B][H]: idx = l; [2][L][B][H]: idx = j L + l; Gc [L][3][B][H]: idx = 3 l + k)
    auto at = [&](const float* base, int idx, int t) { return base + ((size_t)idx * B + my_row) * H + 16 * t + 4 * g; };
    const bool cmp = BF && A.compact != 0;                   // compact mode: bf16 activations in, bf16 gradients out
    auto ld4 = [&](const float* base, int idx, int t) {
        if constexpr (PF_CHAIN_ABLATE & 8) return f32x4{0.5f, 0.25f, 0.125f, 1.f};
        else {
            if (cmp) {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) +
                                                                   ((size_t)idx * B + my_row) * H + 16 * t + 4 * g);
                return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            }
            return *reinterpret_cast<const f32x4*>(at(base, idx, t));
        }
    };
    auto cvt4 = [](const f32x4& v) { bf16x4 o; for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e]; return o; };
    auto st4 = [&](float* base, int idx, int t, const f32x4& v) {
        if constexpr (PF_CHAIN_ABLATE & 4) { if (v[0] == 1.2345e30f) *const_cast<float*>(at(base, idx, t)) = v[1]; return; }
        if (!live) return;
        if (cmp) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + ((size_t)idx * B + my_row) * H + 16 * t + 4 * g) = cvt4(v);
        else *reinterpret_cast<f32x4*>(const_cast<float*>(at(base, idx, t))) = v;
    };
    auto to_lds = [&](int which, int t, const f32x4& v) {               // exchange vector 0 / 1
        if constexpr (BF) *reinterpret_cast<bf16x4*>((which ? s_b1 : s_b0) + c * HSB + 16 * t + 4 * g) = cvt4(v);
        else *reinterpret_cast<f32x4*>((which ? s_v1 : s_v0) + c * HS + 16 * t + 4 * g) = v;
    };

    for (int l = L - 1; l >= 0; --l) {
        // ---- spline backward: raw parameters of the 16 rows -> LDS (coalesced), one lane per (row, feature) pair ----
        // (row = tid / TPR, columns strided by TPR: no integer division per element -- these three staging loops were
        // ~2000 instructions per thread and layer with it)
        constexpr int TPR = NW * 4;                              // threads per row
        const int sr = tid / TPR, sk = tid - sr * TPR;
        const int64_t srow = row0 + sr < B ? row0 + sr : B - 1;
        {
            const float* src = A.params + ((size_t)l * B + srow) * (D * M);
            for (int k = sk; k < PM; k += TPR) s_gp[sr * PMS + k] = k < D * M ? src[k] : 0.f;
        }
        __syncthreads();
        if (tid < 16 * D) {
            const int r = tid & 15, f = tid >> 4;
            const int64_t row = row0 + r < B ? row0 + r : B - 1;
            float* par = s_gp + r * PMS + f * M;
            if constexpr (PF_CHAIN_ABLATE & 1) s_gu[r * 16 + f] = s_gy[r * 16 + f] + par[0];
            else
            s_gu[r * 16 + f] = rqs_backward_pair<KMAX>(par, par, A.U[((size_t)l * B + row) * D + f], s_gy[r * 16 + f],
                                                 A.g_nll ? -A.g_nll[row] : A.g_lad[row], p.c);
        }
        __syncthreads();
        const u32x4* lf = frags + (size_t)l * layer_frags * 64;   // BF: this layer's fragments (64 lanes x 16 B each)
        if constexpr (BF) {       // the layer's stream starts here, under the Gp store and the bf16 image (not before the spline:
            f_prefill(lf);        // 128 fragment registers live across rqs_backward_pair spilled 47 VGPRs)
            issue_phase(hh_phase(lf, 2));                     // block 1's W2^T, consumed after the final layer's transpose
        }
        if (row0 + sr < B) {                                    // Gp -> HBM (weight gradient of the final layer)
            const int gld = A.gp_ld ? (int)A.gp_ld : D * M;
            if (cmp) {
                __bf16* dst = reinterpret_cast<__bf16*>(A.Gp) + ((size_t)l * B + row0 + sr) * gld;
                if (gld % 8 == 0) {                             // 16-byte stores (2-byte ones: 517 store instructions per row)
                    for (int k8 = 8 * sk; k8 < gld; k8 += 8 * TPR) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(k8 + e < D * M ? s_gp[sr * PMS + k8 + e] : 0.f);
                        *reinterpret_cast<bf16x8*>(dst + k8) = o;
                    }
                } else {
                    for (int k = sk; k < gld; k += TPR) dst[k] = (__bf16)(k < D * M ? s_gp[sr * PMS + k] : 0.f);
                }
            } else {
                float* dst = A.Gp + ((size_t)l * B + row0 + sr) * gld;
                for (int k = sk; k < gld; k += TPR) dst[k] = k < D * M ? s_gp[sr * PMS + k] : 0.f;
            }
        }
        if constexpr (BF) {                                     // and its bf16 image, the B operand of the next GEMM
            for (int k = sk; k < 32 * KSF; k += TPR) s_gpb[sr * GSB + k] = (__bf16)(k < D * M ? s_gp[sr * PMS + k] : 0.f);
            __syncthreads();
        }
        // ---- gh = Wf^T Gp ----------------------------------------------------------------------------------------
        f32x4 gh[TPW], acc[TPW];
        f32x4 gate_n[TPW], t2_n[TPW];                        // block operands, requested one GEMM ahead of their use
        if (glu) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) { gate_n[i] = ld4(A.gates, 1 * L + l, wave + NW * i); t2_n[i] = ld4(A.t2s, 1 * L + l, wave + NW * i); }
        }
        if constexpr (BF) f_run(lf, gh);
        else gemm(A.WfT + (size_t)l * H * PM, PM, PM, s_gp, PMS, gh);
        // ---- residual blocks, last first -------------------------------------------------------------------------
        for (int j = 1; j >= 0; --j) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wave + NW * i;
                f32x4 gt2 = gh[i];
                if (glu) {
                    const f32x4 gate = gate_n[i], t2 = t2_n[i];
                    f32x4 gc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gt2[e] = gh[i][e] * gate[e];
                        gc[e] = gh[i][e] * t2[e] * gate[e] * (1.f - gate[e]);
                    }
                    st4(A.Gc, 3 * l + 1 + j, t, gc);
                }
                st4(A.Gt2, j * L + l, t, gt2);
                to_lds(0, t, gt2);
            }
            __syncthreads();
            f32x4 t1[TPW], hj[TPW];                          // requested ahead of the GEMMs whose epilogues use them
#pragma unroll
            for (int i = 0; i < TPW; ++i) { t1[i] = ld4(A.t1s, j * L + l, wave + NW * i); hj[i] = ld4(A.hs, j * L + l, wave + NW * i); }
            if (cmp) {                                       // t1s = relu(t1) . factor: positive where the unit was live and kept
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t1[i][e] = t1[i][e] > 0.f ? A.drop_scale : 0.f;
            } else if (A.drop) {                             // training dropout: fold the forward's factor into the ReLU mask
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const f32x4 dr = ld4(A.drop, j * L + l, wave + NW * i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t1[i][e] = t1[i][e] > 0.f ? dr[e] : 0.f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t1[i][e] = t1[i][e] > 0.f ? 1.f : 0.f;
            }
            if (has_ctx) {                                   // the next block's gate / t2 (block 0 after block 1), or the context layer's pc
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if (j == 1) { if (glu) { gate_n[i] = ld4(A.gates, l, wave + NW * i); t2_n[i] = ld4(A.t2s, l, wave + NW * i); } }
                    else gate_n[i] = ld4(A.pc, l, wave + NW * i);
                }
            }
            if constexpr (BF) run_phase(s_b0, hh_phase(lf, 2 * j + 1), acc);
            else gemm(A.W2T + ((size_t)j * L + l) * H * H, H, H, s_v0, HS, acc);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wave + NW * i;
                f32x4 gt1;
#pragma unroll
                for (int e = 0; e < 4; ++e) gt1[e] = acc[i][e] * t1[i][e];        // t1 holds [t1 > 0] (. dropout factor)
                st4(A.Gt1, j * L + l, t, gt1);
                if (additive && has_ctx) st4(A.Gc, 3 * l + 1 + j, t, gt1);       // t1 = ... + context_layer(ctx): same gradient
                to_lds(1, t, gt1);
            }
            __syncthreads();
            if constexpr (BF) {
                if (j == 1) run_phase(s_b1, hh_phase(lf, 0), acc);
                else run_phase(s_b1, nullptr, acc);
            } else gemm(A.W1T + ((size_t)j * L + l) * H * H, H, H, s_v1, HS, acc);
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) gh[i][e] += hj[i][e] > 0.f ? acc[i][e] : 0.f;
        }
        // ---- initial layer: Gh0, Gc[0], gu += W0^T gh -------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            st4(A.Gh0, l, t, gh[i]);
            if (has_ctx) {
                const f32x4 pc = gate_n[i];
                f32x4 gc;
#pragma unroll
                for (int e = 0; e < 4; ++e) gc[e] = pc[e] > 0.f ? gh[i][e] : 0.f;
                st4(A.Gc, 3 * l, t, gc);
            }
            to_lds(0, t, gh[i]);
        }
        __syncthreads();
        {   // one output tile (the D <= 16 features), the k range split over the four waves, reduced through LDS
            f32x4 part = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                const u32x4* fr = lf + ((size_t)NT * KSF + (size_t)4 * NT * HK) * 64 + lane;
                for (int ks = wave; ks < HK; ks += NW) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(s_b0 + c * HSB + 32 * ks + 8 * g);
                    part = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[64 * ks]), b, part, 0, 0, 0);
                    asm volatile("s_nop 7" : "+a"(part));       // loop-carried accumulator: see f_run
                }
            } else {
            const float* wrow = A.W0T + ((size_t)l * 16 + c) * H + 4 * g;
            const int kq = H / (16 * NW);                    // k-groups of 16 per wave
            for (int q = wave * kq; q < (wave + 1) * kq; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + 16 * q);
                const f32x4 b = *reinterpret_cast<const f32x4*>(s_v0 + c * HS + 16 * q + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) part = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], part, 0, 0, 0);
            }
            }
            // lane (g, c): features 4 g .. 4 g + 3 of row c
            *reinterpret_cast<f32x4*>(s_part + (wave * 16 + c) * 16 + 4 * g) = part;
        }
        __syncthreads();
        {
            const int r = tid >> 4, d = tid & 15;            // 256 threads = 16 rows x 16 feature slots
            float v = 0.f;
            if (d < D && tid < 256) {
                v = s_gu[r * 16 + d];
#pragma unroll
                for (int w = 0; w < NW; ++w) v += s_part[(w * 16 + r) * 16 + d];
                s_gy[r * 16 + (additive ? d : D - 1 - d)] = v;  // through this layer's ReversePermutation (none: masked-context)
            }
        }
        __syncthreads();
    }
    if (tid < 256) {
        const int r = tid >> 4, d = tid & 15;
        if (d < D && row0 + r < B) A.g_x[(row0 + r) * D + d] = s_gy[r * 16 + d];
    }
}

}  // namespace

int flow_backward_chain(const PfFlowDesc& d, float deriv_const, const PfFlowBwdChainArgs& a, hipStream_t s) {
    ChainArgs p{};
    p.a = a;
    p.c = RqsConsts{d.num_bins, d.tail_bound, d.min_bin_width, d.min_bin_height, d.min_derivative, deriv_const};
    p.D = d.features; p.H = d.hidden_features; p.L = d.num_layers; p.M = 3 * d.num_bins - 1;
    p.PM = (p.D * p.M + 15) / 16 * 16;
    p.B = a.batch;
    p.additive = (d.reserved & PF_FLAG_MASKED_CONTEXT) ? 1 : 0;
    const bool bf = d.precision == PF_PREC_BF16;
    if (bf && p.additive) return PF_ERR_UNSUPPORTED;       // (the PF_FLAG_BWD stream is the GLU conditioner's)
    const int nw = p.H % 128 == 0 ? 8 : 4;
    const bool wide_k = d.num_bins > 16;           // fp32 only: spline backward with a 32-bin capacity
    const size_t lds = ((size_t)16 * (p.PM + 4) + 2 * 16 * (p.H + 4) + 256 + 256 + nw * 256) * sizeof(float)
                     + (bf ? (size_t)16 * (32 * ((p.D * p.M + 31) / 32) + 8) * 2 : 0);
    const unsigned grid = (unsigned)((a.batch + 15) / 16);
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024 &&
            !opt_in_lds(reinterpret_cast<const void*>(kern), (int)lds))
            return PF_ERR_HIP;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, s, p);
        return launch_status();
    };
    if (lds > 160 * 1024) return PF_ERR_UNSUPPORTED;
    if (bf) {
        if (wide_k) return PF_ERR_UNSUPPORTED;
        switch (p.H) {
        case 64: return launch(flow_bwd_chain_kernel<1, true, 4>);
        case 128: return launch(flow_bwd_chain_kernel<1, true, 8>);
        case 192: return launch(flow_bwd_chain_kernel<3, true, 4>);
        case 256: return launch(flow_bwd_chain_kernel<2, true, 8>);
        }
        return PF_ERR_UNSUPPORTED;
    }
    // fp32: also the widths and bin counts beyond the scheduled kernels' set (the generic forward's shapes with D <= 16)
#define PF_CHAIN_F32(TPW_, NW_) (wide_k ? launch(flow_bwd_chain_kernel<TPW_, false, NW_, 32>) : launch(flow_bwd_chain_kernel<TPW_, false, NW_, 16>))
    switch (p.H) {
    case 64: return PF_CHAIN_F32(1, 4);
    case 128: return PF_CHAIN_F32(1, 8);
    case 192: return PF_CHAIN_F32(3, 4);
    case 256: return PF_CHAIN_F32(2, 8);
    case 384: return PF_CHAIN_F32(3, 8);
    case 512: return PF_CHAIN_F32(4, 8);
    }
#undef PF_CHAIN_F32
    return PF_ERR_UNSUPPORTED;
}

}  // namespace pf
