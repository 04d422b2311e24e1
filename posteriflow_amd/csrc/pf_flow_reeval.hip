// pf_flow_reeval.hip -- re-evaluation of every layer's conditioner for the backward pass, in ONE launch (bf16 mode).
//
// What it replaces: step 1 of posteriflow_amd/_flow_autograd.py::_flow_backward_batched -- one addmm for the context
// projections, five layer-batched baddbmm and ~20 element-wise launches that recompute, from the layer inputs the training
// forward kept (pf_flow_forward_train), the activations the backward needs: the part of nflows' autograd graph the reference
// keeps alive under src/ahsd/models/flows.py:615-617 (MADE.forward of every layer) and walks in loss.backward()
// (experiments/train_lean_npe.py:363-368).
//
// The layers are independent given their inputs, so the grid is (16-row blocks) x (layers): 1280 workgroups for 2048 rows
// of a 10-layer flow, each streaming ONE layer's forward matrices (1.2 MB of bf16 A-fragments, the forward region of the
// PF_FLAG_BWD stream, pf_pack.hip) -- unlike the chain, which has to walk the layers in sequence on 128 workgroups.
// Same transposed MFMA form as everywhere (out^T[unit, row] = W . act^T on v_mfma_f32_16x16x32_bf16, NW = 8 waves where the
// tile count allows it (else 4), wave w owns unit tiles w, w + NW, ...; activations cross LDS as bf16 rows), same arithmetic as the bf16 forward kernel: bf16 operands,
// x as a hi + lo pair, fp32 accumulate / bias / residual / gate -- the re-evaluated activations are those of the forward
// that produced z, not those of an fp32 model evaluated on the bf16 trajectory.
//
//   pc = Wc ctx + bc                     h0 = W_in x + b_in + relu(pc)
//   gate_j = sigmoid(Wg_j ctx + bg_j)    t1_j = W1_j relu(h_j) + b1_j      t2_j = W2_j relu(t1_j) + b2_j
//   h_{j+1} = h_j + t2_j gate_j          params = Wf h_2 + bf
// written as fp32 [.., B, H] tensors in nflows unit order: exactly what the chain kernel and the weight-gradient GEMMs read.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pf_hip.h"
#include "pf_layout.h"
#include "pf_status.h"

namespace pf {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ReevalArgs {
    PfFlowReevalArgs a;
    int D, C, H, L, M;
    int64_t B;
    int64_t fwd_region;        // first forward fragment of the stream
    int layer_frags, layer_bias;
    int64_t bias_offset;       // bytes from the start of the stream to the fp32 biases
};

template <int TPW, int NW>   // NW waves, TPW = H / (16 NW) unit tiles per wave (8 waves where the tile count allows it)
__global__ __launch_bounds__(NW * 64) void flow_reeval_kernel(const ReevalArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int D = p.D, C = p.C, H = p.H, M = p.M, L = p.L, l = blockIdx.y;
    const int64_t B = p.B, row0 = (int64_t)blockIdx.x * 16;
    const int NT = H / 16, HK = H / 32, CKB = (C + 31) / 32, NTF = (D * M + 15) / 16;
    const int HSB = H + 8, CSB = 32 * CKB + 8;                 // LDS row strides in bf16 elements
    __bf16* s_ctx = reinterpret_cast<__bf16*>(smem);           // [16][CSB]
    __bf16* s_x = s_ctx + 16 * CSB;                            // [16][40]: hi | lo of the layer input
    __bf16* s_a = s_x + 16 * 40;                               // [16][HSB] activation vector handed to the next GEMM
    const PfFlowReevalArgs& A = p.a;
    const u32x4* lf = reinterpret_cast<const u32x4*>(A.packed) + (p.fwd_region + (int64_t)l * p.layer_frags) * 64;
    const float* lb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(A.packed) + p.bias_offset) + (int64_t)l * p.layer_bias;
    const int64_t my_row = row0 + c < B ? row0 + c : B - 1;
    const bool live = row0 + c < B;

    // ---- stage the operands of the first GEMMs: context rows (bf16) and the layer input as hi | lo ----
    for (int s = tid; s < 16 * 32 * CKB; s += NW * 64) {
        const int r = s / (32 * CKB), k = s - r * (32 * CKB);
        const int64_t row = row0 + r < B ? row0 + r : B - 1;
        s_ctx[r * CSB + k] = (__bf16)(k < C ? A.ctx[row * C + k] : 0.f);
    }
    for (int s = tid; s < 16 * 16; s += NW * 64) {
        const int r = s >> 4, d = s & 15;
        const int64_t row = row0 + r < B ? row0 + r : B - 1;
        const float v = d < D ? A.U[((int64_t)l * B + row) * D + d] : 0.f;
        const __bf16 hi = (__bf16)v;
        s_x[r * 40 + d] = hi;
        s_x[r * 40 + 16 + d] = (__bf16)(v - (float)hi);
    }
    __syncthreads();

    // acc[i] = W[tile t0 + wave + NW i][k-steps] . s_in  (fragments [tile][ks][lane]; tiles beyond tmax re-read tile tmax and
    // are not stored).  The fragments are requested a chunk of 4 k-steps ahead (two register chunks): a k-step of TPW MFMAs
    // is 64 cycles, an L2 round trip ~800.
    auto gemm = [&](const u32x4* fr, int nks, int t0, int tmax, const __bf16* s_in, int stride, f32x4 (&acc)[TPW]) {
        constexpr int P = 4;
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const u32x4* fa[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = t0 + wave + NW * i;
            fa[i] = fr + (size_t)(t < tmax ? t : tmax) * nks * 64 + lane;
        }
        const __bf16* brow = s_in + c * stride + 8 * g;
        u32x4 A0[P][TPW], A1[P][TPW];
        auto fetch = [&](u32x4 (&A)[P][TPW], int k0) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int k = k0 + q < nks ? k0 + q : nks - 1;           // clamped: loads stay unconditional
#pragma unroll
                for (int i = 0; i < TPW; ++i) A[q][i] = fa[i][64 * k];
            }
        };
        auto compute = [&](const u32x4 (&A)[P][TPW], int k0) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                if (k0 + q < nks) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + 32 * (k0 + q));
#pragma unroll
                    for (int i = 0; i < TPW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[q][i]), b, acc[i], 0, 0, 0);
                }
            }
        };
        fetch(A0, 0);
        for (int k0 = 0; k0 < nks; k0 += 2 * P) {
            fetch(A1, k0 + P);
            compute(A0, k0);
            fetch(A0, k0 + 2 * P);
            compute(A1, k0 + P);
        }
        // let the matrix pipe drain before the epilogue reads the accumulators: behind a loop exit hipcc (ROCm 7.2) placed
        // v_accvgpr_read right after the last MFMA with too few wait states and three of four values came back stale
        // (pf_flow_reeval.hip's final layer, found on the hardware; LABLOG.md pitfall 15)
#pragma unroll
        for (int i = 0; i < TPW; ++i) asm volatile("s_nop 7" : "+a"(acc[i]));
        asm volatile("s_nop 15");
    };
    auto bias4 = [&](const float* b, int t) { return *reinterpret_cast<const f32x4*>(b + 16 * t + 4 * g); };
    // slab `idx` of a [slabs][B][H] fp32 tensor, units 16 t + 4 g .. + 3 of this lane's row
    const bool cmp = A.compact != 0;                          // compact mode: bf16 outputs in the form the backward uses them
    auto st4 = [&](float* base, int idx, int t, const f32x4& v) {
        if (!live) return;
        const size_t off = ((size_t)idx * B + my_row) * H + 16 * t + 4 * g;
        if (cmp) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + off) = o;
        } else *reinterpret_cast<f32x4*>(base + off) = v;
    };
    auto relu4 = [](const f32x4& v) { f32x4 o; for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], 0.f); return o; };
    auto to_lds_relu = [&](int t, const f32x4& v) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(v[e], 0.f);
        *reinterpret_cast<bf16x4*>(s_a + c * HSB + 16 * t + 4 * g) = o;
    };

    const u32x4* f_in = lf;
    const u32x4* f_ctx = f_in + (size_t)NT * 64;
    const u32x4* f_blk = f_ctx + (C > 0 ? (size_t)3 * NT * CKB * 64 : 0);
    const u32x4* f_out = f_blk + (size_t)4 * NT * HK * 64;
    const float* b_in = lb;
    const float* b_ctx = b_in + H;
    const float* b_blk = b_ctx + (C > 0 ? 3 * H : 0);
    const float* b_out = b_blk + 4 * H;

    // ---- initial layer + context projections ----
    f32x4 h[TPW], gate[2][TPW], acc[TPW];
    gemm(f_in, 1, 0, NT - 1, s_x, 40, h);
#pragma unroll
    for (int i = 0; i < TPW; ++i) h[i] += bias4(b_in, wave + NW * i);
    if (C > 0) {
        gemm(f_ctx, CKB, 0, NT - 1, s_ctx, CSB, acc);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            const f32x4 pc = acc[i] + bias4(b_ctx, t);
            st4(A.pc, l, t, cmp ? relu4(pc) : pc);
#pragma unroll
            for (int e = 0; e < 4; ++e) h[i][e] += fmaxf(pc[e], 0.f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gemm(f_ctx + (size_t)(1 + j) * NT * CKB * 64, CKB, 0, NT - 1, s_ctx, CSB, acc);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int t = wave + NW * i;
                const f32x4 z = acc[i] + bias4(b_ctx + (1 + j) * H, t);
#pragma unroll
                for (int e = 0; e < 4; ++e) gate[j][i][e] = 1.f / (1.f + __expf(-z[e]));
                st4(A.gates, j * L + l, t, gate[j][i]);
            }
        }
    }
    // ---- residual blocks ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) { st4(A.hs, j * L + l, wave + NW * i, cmp ? relu4(h[i]) : h[i]); to_lds_relu(wave + NW * i, h[i]); }
        __syncthreads();
        gemm(f_blk + (size_t)(2 * j) * NT * HK * 64, HK, 0, NT - 1, s_a, HSB, acc);
        __syncthreads();                                       // every wave has read relu(h) before it is overwritten
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            const f32x4 t1 = acc[i] + bias4(b_blk + (2 * j) * H, t);
            f32x4 a1 = relu4(t1);
            if (A.drop) {                                      // training dropout, as flow_train_kernel applied it
                const f32x4 dr = *reinterpret_cast<const f32x4*>(A.drop + ((size_t)(j * L + l) * B + my_row) * H + 16 * t + 4 * g);
                a1 *= dr;
            }
            st4(A.t1s, j * L + l, t, cmp ? a1 : t1);
            to_lds_relu(t, a1);
        }
        __syncthreads();
        gemm(f_blk + (size_t)(2 * j + 1) * NT * HK * 64, HK, 0, NT - 1, s_a, HSB, acc);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = wave + NW * i;
            const f32x4 t2 = acc[i] + bias4(b_blk + (2 * j + 1) * H, t);
            if (C > 0) {
                st4(A.t2s, j * L + l, t, t2);
                h[i] += t2 * gate[j][i];
            } else {
                h[i] += t2;
            }
        }
    }
    // ---- final layer: the raw spline parameters ----
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave + NW * i;
        st4(A.h2, l, t, h[i]);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)h[i][e];       // the final layer takes h itself, not relu(h)
        *reinterpret_cast<bf16x4*>(s_a + c * HSB + 16 * t + 4 * g) = o;
    }
    __syncthreads();
    const int DM = D * M;
    for (int t0 = 0; t0 < NTF; t0 += NW * TPW) {               // NW TPW output tiles per pass, TPW per wave
        gemm(f_out, HK, t0, NTF - 1, s_a, HSB, acc);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int t = t0 + wave + NW * i;
            if (t < NTF && live) {
                const f32x4 o = acc[i] + bias4(b_out, t);
                float* dst = A.params + ((size_t)l * B + my_row) * DM + 16 * t + 4 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (16 * t + 4 * g + e < DM) dst[e] = o[e];
            }
        }
    }
}

}  // namespace

int flow_reevaluate(const FlowPlan& P, const PfFlowReevalArgs& a, hipStream_t s) {
    ReevalArgs p{};
    p.a = a;
    p.D = P.D; p.C = P.C; p.H = P.H; p.L = P.L; p.M = P.M; p.B = a.batch;
    p.fwd_region = P.fwd_region();
    p.layer_frags = P.fwd_layer_frags();
    p.layer_bias = P.fwd_layer_bias();
    p.bias_offset = P.weightBytes;
    const int CKB = P.bwd_ckb();
    const size_t lds = ((size_t)16 * (32 * CKB + 8) + 16 * 40 + 16 * (P.H + 8)) * 2;
    const dim3 grid((unsigned)((a.batch + 15) / 16), (unsigned)P.L);
    const int nw = P.H % 128 == 0 ? 8 : 4;
    if (lds > 160 * 1024) return PF_ERR_UNSUPPORTED;        // context wider than the LDS image holds (C > ~2400)
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(kern), (int)lds)) return (int)PF_ERR_HIP;
        hipLaunchKernelGGL(kern, grid, dim3(nw * 64), lds, s, p);
        return launch_status();
    };
    switch (P.H) {
    case 64: return launch(flow_reeval_kernel<1, 4>);
    case 128: return launch(flow_reeval_kernel<1, 8>);
    case 192: return launch(flow_reeval_kernel<3, 4>);
    case 256: return launch(flow_reeval_kernel<2, 8>);
    }
    return PF_ERR_UNSUPPORTED;
}

}  // namespace pf
