// pf_flow_ctx.hip -- hoisted context projections of ALL layers of the flow in one GEMM:
//   P[r][l][0][u] = relu   (Wc_l  ctx_r + bc_l )[u]     (nflows MADE.context_layer)
//   P[r][l][1+b][u] = sigmoid(Wg_lb ctx_r + bg_lb)[u]   (GLU gate of residual block b)
// for every context row r, u in degree-sorted position order.  [rows, C] x [C, 3 L H].
//
// In the reference these 3 L small GEMMs are re-executed by nflows inside every MADE call
// (flows.py:615-617), and in the inverse once per autoregressive pass for 4096 copies of ONE
// context row (flows.py:637, pipeline.py:171-173).  They do not depend on x, so they are
// hoisted; the layer chain then only reads 3 H values per row and layer.
//
// Same transposed MFMA formulation as the chain kernel: weights are the A operand (packed
// fragments, pf_layout.h "hoisted-context plans"), 64 context rows per workgroup are the
// columns of 4 MFMA column groups, the context tile is staged once in LDS in B-fragment order.
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include "pf_flow_params.h"

namespace pf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct CtxParams {
    const u32x4* frags;     // [tiles][CK][64]
    const float* bias;      // [tiles][16]
    const float* ctx;       // [rows, C]
    void* out;              // [rows/64][tiles][4][64][4] fp32 (MFMA C-fragment order)
    int64_t rows;
    int C, CK, NT, tiles, tiles_per_wave;
    int additive;           // 1: block projections are plain affine terms (masked-context conditioner)
    int rowmajor;           // 1: out = raw affine values, fp32 row-major [rows][16 tiles] (the incremental inverse's operand)
};

constexpr int kCtxRowGroups = 4;   // 64 context rows per workgroup

template <bool BF16>
__global__ __launch_bounds__(256) void ctx_project_kernel(const CtxParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 16 * kCtxRowGroups;

    // stage the 64-row context tile: coalesced 16-B global loads of the row-major tile, scattered
    // into B-fragment order on the LDS side (zero padding beyond C and beyond the last row)
    const int kw = BF16 ? 32 : 16, cpad = p.CK * kw;
    if ((p.C & 3) == 0 && (reinterpret_cast<uintptr_t>(p.ctx) & 15) == 0) {
        const int c4 = cpad / 4;
        for (int i = tid; i < 16 * kCtxRowGroups * c4; i += 256) {
            const int r = i / c4, col = (i - r * c4) * 4;
            int64_t row = row0 + r;
            if (row >= p.rows) row = p.rows - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (col < p.C) v = *reinterpret_cast<const f32x4*>(p.ctx + row * p.C + col);
            const int rg = r >> 4, cc = r & 15, ks = col / kw, rem = col - ks * kw;
            if (BF16) {
                const int gg = rem >> 3, j = rem & 7;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(smem + ((size_t)(ks * kCtxRowGroups + rg) * 64 + gg * 16 + cc) * 16 + j * 2) = o;
            } else {
                const int gg = rem >> 2;
                *reinterpret_cast<f32x4*>(smem + ((size_t)(ks * kCtxRowGroups + rg) * 64 + gg * 16 + cc) * 16) = v;
            }
        }
    } else {
        for (int s = tid; s < p.CK * kCtxRowGroups * 64; s += 256) {
            const int ln = s & 63, rg = (s >> 6) & (kCtxRowGroups - 1), ks = s >> 8;
            int64_t row = row0 + 16 * rg + (ln & 15);
            if (row >= p.rows) row = p.rows - 1;
            const float* src = p.ctx + row * p.C;
            const int gg = ln >> 4;
            if (BF16) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = 32 * ks + 8 * gg + j;
                    v[j] = (__bf16)(col < p.C ? src[col] : 0.f);
                }
                *reinterpret_cast<bf16x8*>(smem + (size_t)s * 16) = v;
            } else {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = 16 * ks + 4 * gg + e;
                    v[e] = col < p.C ? src[col] : 0.f;
                }
                *reinterpret_cast<f32x4*>(smem + (size_t)s * 16) = v;
            }
        }
    }
    __syncthreads();

    // Each wave walks its tiles with the NEXT tile's fragments already in flight (register double
    // buffer): the loop is otherwise one dependent L2 round trip per handful of MFMAs.
    constexpr int CKC = 9;                              // fragments held per buffer (k-chunk)
    const int first = (blockIdx.y * 4 + wave) * p.tiles_per_wave;
    const int n_mine = min(p.tiles_per_wave, p.tiles - first);
    const int n_chunks = (p.CK + CKC - 1) / CKC;
    const int n_steps = n_mine > 0 ? n_mine * n_chunks : 0;     // (tile, k-chunk) steps
    auto fetch = [&](int step, u32x4 (&a)[CKC]) {
        const int st = step < n_steps ? step : n_steps - 1;      // clamped: loads stay unconditional
        const int u = first + st / n_chunks, k0 = (st % n_chunks) * CKC;
        const u32x4* wf = p.frags + (size_t)u * p.CK * 64 + lane;
#pragma unroll
        for (int k = 0; k < CKC; ++k) a[k] = wf[(size_t)min(k0 + k, p.CK - 1) * 64];
    };
    f32x4 acc[kCtxRowGroups];
    auto compute = [&](int step, const u32x4 (&a)[CKC]) {
        const int u = first + step / n_chunks, kc = step % n_chunks, k0 = kc * CKC;
        if (kc == 0) {
#pragma unroll
            for (int rg = 0; rg < kCtxRowGroups; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < CKC; ++k) {
            if (k0 + k < p.CK) {
#pragma unroll
                for (int rg = 0; rg < kCtxRowGroups; ++rg) {
                    const char* bp = smem + ((size_t)((k0 + k) * kCtxRowGroups + rg) * 64 + lane) * 16;
                    if (BF16) {
                        acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a[k]), *reinterpret_cast<const bf16x8*>(bp), acc[rg], 0, 0, 0);
                    } else {
                        const f32x4 af = __builtin_bit_cast(f32x4, a[k]), bf = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc[rg], 0, 0, 0);
                    }
                }
            }
        }
        if (kc != n_chunks - 1) return;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + (size_t)u * 16 + 4 * g);
        if (p.rowmajor) {          // lane (g, column cc) holds units 16 u + 4 g .. + 3 of context row 16 rg + cc: one 16-byte store
#pragma unroll
            for (int rg = 0; rg < kCtxRowGroups; ++rg) {
                const int64_t row = row0 + 16 * rg + (lane & 15);
                if (row < p.rows)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + row * ((int64_t)p.tiles * 16) + (size_t)u * 16 + 4 * g) = acc[rg] + b;
            }
            return;
        }
        const bool is_first = ((u / p.NT) % 3) == 0;
        const bool is_gate = !is_first && !p.additive;
#pragma unroll
        for (int rg = 0; rg < kCtxRowGroups; ++rg) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = acc[rg][e] + b[e];
                if (is_gate) v[e] = BF16 ? __builtin_amdgcn_rcpf(1.f + __expf(-t)) : 1.f / (1.f + expf(-t));
                else v[e] = is_first ? fmaxf(t, 0.f) : t;
            }
            // fragment order [row block of 64][tile][row group][lane][4]: one 1-KiB contiguous
            // store per wave-instruction; rows past the end are padding
            const size_t off = ((((size_t)blockIdx.x * p.tiles + u) * kCtxRowGroups + rg) * 64 + lane) * 4;
            // always fp32: the hoisted path then computes the same function as the in-layer one
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + off) = v;
        }
    };
    if (n_steps > 0) {
        u32x4 a0[CKC], a1[CKC];
        fetch(0, a0);
        for (int step = 0; step < n_steps; step += 2) {
            fetch(step + 1, a1);
            compute(step, a0);
            fetch(step + 2, a0);
            if (step + 1 < n_steps) compute(step + 1, a1);
        }
    }
}

int64_t ctx_project_bytes(const FlowPlan& L, int64_t ctx_rows) {
    if (!L.hoist) return 0;
    const int64_t blocks = (ctx_rows + 63) / 64;          // stored in fragment order, 64-row blocks
    return blocks * 64 * 3 * L.L * L.H * 4;
}

static int launch_ctx(const CtxParams& p_in, bool bf16, hipStream_t s);

int launch_ctx_project(const FlowPlan& L, const char* packed, const float* ctx, int64_t ctx_rows,
                       void* out, hipStream_t s) {
    CtxParams p{};
    p.frags = reinterpret_cast<const u32x4*>(packed + L.ctx_frag_offset());
    p.bias = reinterpret_cast<const float*>(packed + L.ctx_bias_offset());
    p.ctx = ctx; p.out = out; p.rows = ctx_rows;
    p.C = L.C; p.CK = L.CK; p.NT = L.NT; p.tiles = 3 * L.L * L.NT; p.additive = L.additive;
    return launch_ctx(p, L.bf16 != 0, s);
}

// out[r][n] = sum_c ctx[r][c] W[n][c] + bias[n], fp32 row-major: the same kernel over any matrix packed as plain MFMA
// A-fragments [n_units / 16 tiles][k-steps][64 lanes] (pf_dense_pack_matrix) -- the per-context-row projections the
// incremental inverse reads (pf_flow_inverse_inc: [ctx_rows][L][3][H] raw affine values)
int ctx_project_rows(bool bf16, const void* wfrags, const float* bias, const float* ctx, int64_t rows, int C, int n_units,
                     float* out, hipStream_t s) {
    if (rows == 0) return PF_OK;
    const int kw = bf16 ? 32 : 16;
    CtxParams p{};
    p.frags = reinterpret_cast<const u32x4*>(wfrags);
    p.bias = bias; p.ctx = ctx; p.out = out; p.rows = rows;
    p.C = C; p.CK = (C + kw - 1) / kw; p.NT = 1; p.tiles = n_units / 16; p.additive = 1; p.rowmajor = 1;
    if ((size_t)p.CK * kCtxRowGroups * kFragBytes > 160 * 1024) return PF_ERR_UNSUPPORTED;
    return launch_ctx(p, bf16, s);
}

static int launch_ctx(const CtxParams& p_in, bool bf16, hipStream_t s) {
    CtxParams p = p_in;
    const int64_t ctx_rows = p.rows;
    const unsigned row_blocks = (unsigned)((ctx_rows + 63) / 64);
    // enough workgroups to fill the chip when there are few row blocks
    int tpw = 12;
    while (tpw > 2 && (int64_t)row_blocks * ((p.tiles + 4 * tpw - 1) / (4 * tpw)) < 1024) tpw >>= 1;
    p.tiles_per_wave = tpw;
    const unsigned chunks = (unsigned)((p.tiles + 4 * tpw - 1) / (4 * tpw));
    const size_t lds = (size_t)p.CK * kCtxRowGroups * kFragBytes;
    if (bf16) {
        auto k = ctx_project_kernel<true>;
        if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
        hipLaunchKernelGGL(k, dim3(row_blocks, chunks), dim3(256), lds, s, p);
    } else {
        auto k = ctx_project_kernel<false>;
        if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
        hipLaunchKernelGGL(k, dim3(row_blocks, chunks), dim3(256), lds, s, p);
    }
    return launch_status();
}

}  // namespace pf
