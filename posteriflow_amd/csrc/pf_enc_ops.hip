// pf_enc_ops.hip -- LayerNorm, self-attention, attention pool (forward + backward) and the token assembly of the strain
// embedding's training path: see pf_enc_ops.h.  GEMM-shaped work inside the attention runs on MFMA:
//   bf16 mode  v_mfma_f32_16x16x32_bf16 for the head-dimension products (K = 32 = one instruction per 16 x 16 tile) and
//              v_mfma_f32_16x16x16_bf16 for the products that reduce over tokens: its operand layout (k = 4 g + j) IS the
//              accumulator layout (row = 4 g + r), so P / dS feed the second product straight from registers;
//   f32 mode   v_mfma_f32_16x16x4_f32 in both places (the same trick: MFMA r of a group of four takes accumulator
//              register r as its B operand).
// Both orientations of the score tile are computed where a product needs the reduction on the other index (S^T for dQ,
// S for dK / dV): the head dimension is 32, a score tile costs one MFMA, a transpose through LDS costs more.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "pf_dense.h"
#include "pf_enc_ops.h"
#include "pf_status.h"

namespace pf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <bool BF16> __device__ __forceinline__ f32x4 load_act4(const void* base, int64_t off) {
    if constexpr (BF16) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + off);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
    }
}
template <bool BF16> __device__ __forceinline__ void store_act4(void* base, int64_t off, const f32x4& v) {
    if constexpr (BF16) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base) + off) = o;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
    }
}
__device__ __forceinline__ s16x4 cvt_bf16x4(const f32x4& v) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    return __builtin_bit_cast(s16x4, o);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over 192 features: one wave per row, lanes 0..47 hold 4 consecutive features
// ---------------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = lane < 48;
    const int col = 4 * (on ? lane : 0);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + col), bt = *reinterpret_cast<const f32x4*>(a.beta + col);
    for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < a.M; m += (int64_t)gridDim.x * 4) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (on) x = *reinterpret_cast<const f32x4*>(a.x + m * kEncD + col);
        const float mean = wave_sum(x[0] + x[1] + x[2] + x[3]) * (1.f / kEncD);
        f32x4 d = x - mean;
        if (!on) d = f32x4{0.f, 0.f, 0.f, 0.f};
        const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / kEncD);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        if (on) store_act4<BF16>(a.y, m * kEncD + col, d * rstd * gm + bt);
        if (lane == 0) { a.mean[m] = mean; a.rstd[m] = rstd; }
    }
}

// rows in flight per wave x workgroups: measured at 187 K rows (us): 4 x 1024 156, 8 x 1024 157, 8 x 512 150, 4 x 2048 142,
// 2 x 2048 131.5 (4.4 TB/s), 3 x 2048 131.4, 1 x 2048 135, 2 x 4096 145 -- all eight waves per SIMD resident, two rows each
#ifndef PF_LN_BWD_U
#define PF_LN_BWD_U 2
#endif
#ifndef PF_LN_BWD_GRID
#define PF_LN_BWD_GRID 2048
#endif
template <bool BF16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnArgs a) {
    __shared__ float s_red[2][4][kEncD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool on = lane < 48;
    const int col = 4 * (on ? lane : 0);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + col);
    f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    const uint32_t thr = enc_drop_threshold(a.drop_p);
    const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    // U rows per iteration, every load requested before the first row's arithmetic: with one row in flight and 1024 workgroups a
    // wave's ~46 rows were 46 global round trips in a row (228 us per 187 K rows = 2.9 TB/s)
    constexpr int U = PF_LN_BWD_U;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t m0 = (int64_t)blockIdx.x * 4 + wave; m0 < a.M; m0 += stride * U) {
        f32x4 x[U], dy[U], dr[U];
        float mean[U], rstd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + u * stride;
            const bool live = on && m < a.M;
            const int64_t mc = m < a.M ? m : a.M - 1;
            x[u] = dy[u] = dr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live) {
                x[u] = *reinterpret_cast<const f32x4*>(a.x + m * kEncD + col);
                dy[u] = load_act4<BF16>(a.dy, m * kEncD + col);
                if (a.dres) dr[u] = *reinterpret_cast<const f32x4*>(a.dres + m * kEncD + col);
            }
            mean[u] = a.mean[mc];
            rstd[u] = a.rstd[mc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + u * stride;
            if (m >= a.M) break;                               // (wave-uniform)
            f32x4 xh = (x[u] - mean[u]) * rstd[u];
            if (!on) xh = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 gy = dy[u] * gm;
            const float s1 = wave_sum(gy[0] + gy[1] + gy[2] + gy[3]) * (1.f / kEncD);
            const float s2 = wave_sum(gy[0] * xh[0] + gy[1] * xh[1] + gy[2] * xh[2] + gy[3] * xh[3]) * (1.f / kEncD);
            dg += dy[u] * xh;
            db += dy[u];
            if (on) {
                f32x4 dx = (gy - s1 - xh * s2) * rstd[u] + dr[u];
                *reinterpret_cast<f32x4*>(a.dx + m * kEncD + col) = dx;
                if (a.gout) {
                    if (a.drop_p > 0.f) {
                        f32x4 fac;
                        enc_drop4(a.seed, a.site, (uint32_t)(m * kEncD + col), thr, dscale, fac);
                        dx = dx * fac;
                    }
                    store_act4<BF16>(a.gout, m * kEncD + col, dx);
                }
            }
        }
    }
    if (on) {
        *reinterpret_cast<f32x4*>(&s_red[0][wave][col]) = dg;
        *reinterpret_cast<f32x4*>(&s_red[1][wave][col]) = db;
    }
    __syncthreads();
    if (threadIdx.x < kEncD) {
        const int cc = threadIdx.x;
        atomicAdd(a.dgamma + cc, s_red[0][0][cc] + s_red[0][1][cc] + s_red[0][2][cc] + s_red[0][3][cc]);
        atomicAdd(a.dbeta + cc, s_red[1][0][cc] + s_red[1][1][cc] + s_red[1][2][cc] + s_red[1][3][cc]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// self-attention, one workgroup per (head, event); T <= 192 tokens, head dimension 32
// ---------------------------------------------------------------------------------------------------------------------
// Every operand a loop re-reads lives in LDS, in the two forms the MFMAs take it:
//   row images  [192 tokens][32]: bf16 64-byte rows with the 16-byte slot XOR-ed by (row >> 2) & 3 (ds_read_b128, conflict-free),
//               fp32 rows padded to 33 floats (ds_read_b32);  = operands of the head-dimension products (k = head dim);
//   transposed  [32][kTS tokens]: operands of the token-dimension products (k = token), read 4 tokens at a time.
// (First version: the inner-loop operands came from global memory -- 47 KB of Q, K, V, dO per workgroup, five workgroups
// per CU against a 32 KB L1: every tile pair waited out an L2 round trip; backward 612 us per layer at 1024 events.)
constexpr int kTS = 196;                        // row stride (elements) of the transposed [32][T] images
constexpr int kRF = 33;                         // row stride (floats) of the fp32 row images
constexpr float kScale = 0.17677669529663687f;  // 1 / sqrt(32)
template <bool BF16> constexpr size_t row_img_bytes() { return BF16 ? (size_t)kEncMaxTokens * 64 : (size_t)kEncMaxTokens * kRF * 4; }
template <bool BF16> constexpr size_t tr_img_bytes() { return (size_t)32 * kTS * (BF16 ? 2 : 4); }

// operand of the head-dimension products: 8 bf16 (one 16-byte read) or 8 fp32 (k = 4 s + g) of one token row
template <bool BF16> struct HeadFrag { u32x4 v; float f[8]; };
template <bool BF16>
__device__ __forceinline__ void global_head_frag(HeadFrag<BF16>& fr, const void* base, int64_t row_off, bool valid, int g) {
    if constexpr (BF16) {
        fr.v = u32x4{0u, 0u, 0u, 0u};
        if (valid) fr.v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(base) + row_off + 8 * g);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) fr.f[s] = valid ? reinterpret_cast<const float*>(base)[row_off + 4 * s + g] : 0.f;
    }
}
template <bool BF16>
__device__ __forceinline__ void lds_head_frag(HeadFrag<BF16>& fr, const char* img, int row, int g) {
    if constexpr (BF16) {
        fr.v = *reinterpret_cast<const u32x4*>(img + row * 64 + ((g ^ ((row >> 2) & 3)) << 4));
    } else {
        const float* r = reinterpret_cast<const float*>(img) + row * kRF + g;
#pragma unroll
        for (int s = 0; s < 8; ++s) fr.f[s] = r[4 * s];
    }
}
template <bool BF16>
__device__ __forceinline__ f32x4 head_mma(const HeadFrag<BF16>& a, const HeadFrag<BF16>& b) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[s], b.f[s], acc, 0, 0, 0);
    }
    return acc;
}
// second product: acc += Img[16 dt + c][16 t + 4 g .. + 3] (A operand, from a transposed LDS image) x tile (B operand =
// an accumulator tile whose ROW index is the reduction index)
template <bool BF16>
__device__ __forceinline__ f32x4 token_mma(const char* img, int dt, int t, int c, int g, const f32x4& tile, f32x4 acc) {
    if constexpr (BF16) {
        const s16x4 av = *reinterpret_cast<const s16x4*>(img + ((size_t)(16 * dt + c) * kTS + 16 * t + 4 * g) * 2);
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, cvt_bf16x4(tile), acc, 0, 0, 0);
    } else {
        const f32x4 av = *reinterpret_cast<const f32x4*>(img + ((size_t)(16 * dt + c) * kTS + 16 * t + 4 * g) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], tile[r], acc, 0, 0, 0);
        return acc;
    }
}
// one [T][32] slice (row stride ld elements) -> row image and / or transposed image; rows / columns T .. 191 zeroed
template <bool BF16>
__device__ __forceinline__ void stage_slice(char* rows, char* tr, const void* src, int64_t row0, int ld, int col0, int T, int tid) {
    constexpr int ESZ = BF16 ? 2 : 4, EPC = 16 / ESZ, CPR = 32 / EPC;
    for (int i = tid; i < kEncMaxTokens * CPR; i += 256) {
        const int t = i / CPR, ch = i - t * CPR;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t < T) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + ((row0 + t) * ld + col0 + ch * EPC) * ESZ);
        if constexpr (BF16) {
            if (rows) *reinterpret_cast<u32x4*>(rows + t * 64 + ((ch ^ ((t >> 2) & 3)) << 4)) = v;
            if (tr) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<__bf16*>(tr + ((size_t)(ch * 8 + j) * kTS + t) * 2) = b[j];
            }
        } else {
            const f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rows) reinterpret_cast<float*>(rows)[t * kRF + ch * 4 + j] = f[j];
                if (tr) *reinterpret_cast<float*>(tr + ((size_t)(ch * 4 + j) * kTS + t) * 4) = f[j];
            }
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256, BF16 ? 3 : 1) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* KR = smem;
    char* VT = smem + row_img_bytes<BF16>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.x, T = a.T;
    const int64_t e = blockIdx.y, r0 = e * T;
    stage_slice<BF16>(KR, nullptr, a.qkv, r0, 3 * kEncD, kEncD + kEncHd * h, T, tid);
    stage_slice<BF16>(nullptr, VT, a.qkv, r0, 3 * kEncD, 2 * kEncD + kEncHd * h, T, tid);
    __syncthreads();
    const uint32_t thr = enc_drop_threshold(a.drop_p);
    const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const int ntile = (T + 15) >> 4;
    // the next tile's global operands are requested before this tile's arithmetic (unconditionally, on a clamped tile: a
    // conditional load would make the compiler drain the queue): requested at the top of their own tile they cost one exposed
    // round trip per tile, three per wave
    HeadFrag<BF16> qf_next;
    {
        const int q0 = 16 * (wave < ntile ? wave : ntile - 1) + c;
        global_head_frag<BF16>(qf_next, a.qkv, (r0 + (q0 < T ? q0 : T - 1)) * (3 * kEncD) + kEncHd * h, true, g);
    }
#pragma unroll 1
    for (int qt = wave; qt < ntile; qt += 4) {
        const int q = 16 * qt + c;
        const bool qv = q < T;
        HeadFrag<BF16> qf = qf_next;
        {
            const int qn = 16 * (qt + 4 < ntile ? qt + 4 : qt) + c;
            global_head_frag<BF16>(qf_next, a.qkv, (r0 + (qn < T ? qn : T - 1)) * (3 * kEncD) + kEncHd * h, true, g);
        }
        f32x4 s[12];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
            HeadFrag<BF16> kf;
            lds_head_frag<BF16>(kf, KR, 16 * kt + c, g);
            s[kt] = head_mma<BF16>(kf, qf);                    // S^T[key = 16 kt + 4 g + r][q = 16 qt + c]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (16 * kt + 4 * g + r) < T ? s[kt][r] * kScale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
            // (keeps hipcc from hoisting all twelve K fragments -- 48 registers -- in front of the first MFMA)
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 12; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = BF16 ? __expf(s[kt][r] - mx) : expf(s[kt][r] - mx);
                s[kt][r] = pv;
                l += pv;
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        if (g == 0 && qv) a.lse[(e * kEncHeads + h) * T + q] = mx + logf(l);
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kt = 0; kt < 12; ++kt) {
            f32x4 pd = s[kt] * inv;
            if (a.drop_p > 0.f) {     // factor index: [event][head][query][192 key slots]: 4-aligned, two hashes per four keys
                f32x4 fac;
                enc_drop4(a.seed, a.site, (uint32_t)(((e * kEncHeads + h) * T + q) * kEncMaxTokens + 16 * kt + 4 * g), thr, dscale, fac);
                pd = pd * fac;
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o[dt] = token_mma<BF16>(VT, dt, kt, c, g, pd, o[dt]);
        }
        if (qv) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store_act4<BF16>(a.out, (r0 + q) * kEncD + kEncHd * h + 16 * dt + 4 * g, o[dt]);
        }
    }
}

// backward, part 1: dQ.  Per query tile, key on the accumulator rows: S^T and dP^T tiles, dS^T feeds dQ^T = K^T dS^T.
template <bool BF16>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* KR = smem;
    char* VR = smem + row_img_bytes<BF16>();
    char* KT = smem + 2 * row_img_bytes<BF16>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.x, T = a.T;
    const int64_t e = blockIdx.y, r0 = e * T;
    stage_slice<BF16>(KR, KT, a.qkv, r0, 3 * kEncD, kEncD + kEncHd * h, T, tid);
    stage_slice<BF16>(VR, nullptr, a.qkv, r0, 3 * kEncD, 2 * kEncD + kEncHd * h, T, tid);
    __syncthreads();
    const uint32_t thr = enc_drop_threshold(a.drop_p);
    const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const int ntile = (T + 15) >> 4;
    const uint32_t idx_base = (uint32_t)((e * kEncHeads + h) * T) * (uint32_t)kEncMaxTokens;
    HeadFrag<BF16> qf_next, dof_next, of_next;
    float lse_next;
    auto request = [&](int qt_) {                     // (clamped: always a valid row; rows beyond T are masked by qv below)
        const int qq = 16 * qt_ + c;
        const int64_t row = r0 + (qq < T ? qq : T - 1);
        global_head_frag<BF16>(qf_next, a.qkv, row * (3 * kEncD) + kEncHd * h, true, g);
        global_head_frag<BF16>(dof_next, a.dout, row * kEncD + kEncHd * h, true, g);
        global_head_frag<BF16>(of_next, a.out, row * kEncD + kEncHd * h, true, g);
        lse_next = a.lse[(e * kEncHeads + h) * T + (qq < T ? qq : T - 1)];
    };
    request(wave < ntile ? wave : ntile - 1);
#pragma unroll 1
    for (int qt = wave; qt < ntile; qt += 4) {
        const int q = 16 * qt + c;
        const bool qv = q < T;
        HeadFrag<BF16> qf = qf_next, dof = dof_next, of = of_next;
        const float lse_cur = lse_next;
        request(qt + 4 < ntile ? qt + 4 : qt);
        // delta_q = dO[q] . O[q] over the head's 32 features: this lane's 8, then across the four lane groups
        float del_q = 0.f;
        if constexpr (BF16) {
            const bf16x8 u = __builtin_bit_cast(bf16x8, dof.v), w = __builtin_bit_cast(bf16x8, of.v);
#pragma unroll
            for (int j = 0; j < 8; ++j) del_q += (float)u[j] * (float)w[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) del_q += dof.f[j] * of.f[j];
        }
        del_q += __shfl_xor(del_q, 16);
        del_q += __shfl_xor(del_q, 32);
        const float lse_q = qv ? lse_cur : 0.f;
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
        for (int kt = 0; kt < ntile; ++kt) {
            HeadFrag<BF16> kf, vf;
            lds_head_frag<BF16>(kf, KR, 16 * kt + c, g);
            lds_head_frag<BF16>(vf, VR, 16 * kt + c, g);
            const f32x4 st = head_mma<BF16>(kf, qf), dp = head_mma<BF16>(vf, dof);
            f32x4 ds, fac = {1.f, 1.f, 1.f, 1.f};
            if (a.drop_p > 0.f) enc_drop4(a.seed, a.site, idx_base + (uint32_t)(q * kEncMaxTokens + 16 * kt + 4 * g), thr, dscale, fac);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * kt + 4 * g + r;
                const float pr = (kk < T && qv) ? (BF16 ? __expf(st[r] * kScale - lse_q) : expf(st[r] * kScale - lse_q)) : 0.f;
                ds[r] = pr * (dp[r] * fac[r] - del_q) * kScale;
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dq[dt] = token_mma<BF16>(KT, dt, kt, c, g, ds, dq[dt]);
        }
        if (qv) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store_act4<BF16>(a.dqkv, (r0 + q) * (3 * kEncD) + kEncHd * h + 16 * dt + 4 * g, dq[dt]);
        }
    }
}

// backward, part 2: dK, dV.  Per key tile, query on the accumulator rows: S and dP tiles; P . D feeds dV^T = dO^T (P . D),
// dS feeds dK^T = Q^T dS.
template <bool BF16>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* QR = smem;
    char* DOR = smem + row_img_bytes<BF16>();
    char* QT = smem + 2 * row_img_bytes<BF16>();
    char* DOT = QT + tr_img_bytes<BF16>();
    float* s_delta = reinterpret_cast<float*>(DOT + tr_img_bytes<BF16>());
    float* s_lse = s_delta + kEncMaxTokens;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int h = blockIdx.x, T = a.T;
    const int64_t e = blockIdx.y, r0 = e * T;
    stage_slice<BF16>(QR, QT, a.qkv, r0, 3 * kEncD, kEncHd * h, T, tid);
    stage_slice<BF16>(DOR, DOT, a.dout, r0, kEncD, kEncHd * h, T, tid);
    if (tid < kEncMaxTokens) {
        float dl = 0.f, ls = 0.f;
        if (tid < T) {
#pragma unroll
            for (int d4 = 0; d4 < 8; ++d4) {
                const f32x4 u = load_act4<BF16>(a.dout, (r0 + tid) * kEncD + kEncHd * h + 4 * d4);
                const f32x4 w = load_act4<BF16>(a.out, (r0 + tid) * kEncD + kEncHd * h + 4 * d4);
                dl += u[0] * w[0] + u[1] * w[1] + u[2] * w[2] + u[3] * w[3];
            }
            ls = a.lse[(e * kEncHeads + h) * T + tid];
        }
        s_delta[tid] = dl;
        s_lse[tid] = ls;
    }
    __syncthreads();
    const uint32_t thr = enc_drop_threshold(a.drop_p);
    const float dscale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const int ntile = (T + 15) >> 4;
    const uint32_t idx_base = (uint32_t)((e * kEncHeads + h) * T) * (uint32_t)kEncMaxTokens;
    HeadFrag<BF16> kf_next, vf_next;
    auto request = [&](int kt_) {
        const int kk = 16 * kt_ + c;
        const int64_t row = r0 + (kk < T ? kk : T - 1);
        global_head_frag<BF16>(kf_next, a.qkv, row * (3 * kEncD) + kEncD + kEncHd * h, true, g);
        global_head_frag<BF16>(vf_next, a.qkv, row * (3 * kEncD) + 2 * kEncD + kEncHd * h, true, g);
    };
    request(wave < ntile ? wave : ntile - 1);
#pragma unroll 1
    for (int kt = wave; kt < ntile; kt += 4) {
        const int key = 16 * kt + c;
        const bool kv = key < T;
        HeadFrag<BF16> kf = kf_next, vf = vf_next;
        request(kt + 4 < ntile ? kt + 4 : kt);
        f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
        for (int qt = 0; qt < ntile; ++qt) {
            HeadFrag<BF16> qf, dof;
            lds_head_frag<BF16>(qf, QR, 16 * qt + c, g);
            lds_head_frag<BF16>(dof, DOR, 16 * qt + c, g);
            const f32x4 sc = head_mma<BF16>(qf, kf), dp = head_mma<BF16>(dof, vf);   // [q = 16 qt + 4 g + r][key = 16 kt + c]
            const f32x4 ls4 = *reinterpret_cast<const f32x4*>(s_lse + 16 * qt + 4 * g);
            const f32x4 dl4 = *reinterpret_cast<const f32x4*>(s_delta + 16 * qt + 4 * g);
            f32x4 pd, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = 16 * qt + 4 * g + r;
                const float pr = (qq < T && kv) ? (BF16 ? __expf(sc[r] * kScale - ls4[r]) : expf(sc[r] * kScale - ls4[r])) : 0.f;
                float fac = 1.f;
                if (a.drop_p > 0.f) fac = enc_drop_hash(a.seed, a.site, idx_base + (uint32_t)(qq * kEncMaxTokens + key)) >= thr ? dscale : 0.f;
                pd[r] = pr * fac;
                ds[r] = pr * (dp[r] * fac - dl4[r]) * kScale;
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = token_mma<BF16>(DOT, dt, qt, c, g, pd, dv[dt]);
                dk[dt] = token_mma<BF16>(QT, dt, qt, c, g, ds, dk[dt]);
            }
        }
        if (kv) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                store_act4<BF16>(a.dqkv, (r0 + key) * (3 * kEncD) + kEncD + kEncHd * h + 16 * dt + 4 * g, dk[dt]);
                store_act4<BF16>(a.dqkv, (r0 + key) * (3 * kEncD) + 2 * kEncD + kEncHd * h + 16 * dt + 4 * g, dv[dt]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// attention pool: 8 fixed queries per head over the T tokens of an event (vector ALU: 8 x T scores per head)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPS = 33;        // padded row of the fp32 K / V head images in LDS
template <bool BF16, bool BWD>
__global__ __launch_bounds__(256) void pool_kernel(const PoolArgs a) {
    __shared__ float sK[kEncMaxTokens * kPS], sV[kEncMaxTokens * kPS];
    __shared__ float sQ[kEncPoolQ * kEncHd], sP[kEncPoolQ * kEncMaxTokens], sDO[kEncPoolQ * kEncHd], sDS[kEncPoolQ * kEncMaxTokens];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = a.T;
    const int64_t e = blockIdx.x, r0 = e * T;
#pragma unroll 1
    for (int h = 0; h < kEncHeads; ++h) {
        __syncthreads();
        for (int i = tid; i < T * 8; i += 256) {                // 8 groups of 4 features per token and operand
            const int t = i >> 3, d4 = i & 7;
            const f32x4 kx = load_act4<BF16>(a.kv, (r0 + t) * (2 * kEncD) + kEncHd * h + 4 * d4);
            const f32x4 vx = load_act4<BF16>(a.kv, (r0 + t) * (2 * kEncD) + kEncD + kEncHd * h + 4 * d4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { sK[t * kPS + 4 * d4 + j] = kx[j]; sV[t * kPS + 4 * d4 + j] = vx[j]; }
        }
        {
            const int qi = tid >> 5, d = tid & 31;
            sQ[tid] = a.q[qi * kEncD + kEncHd * h + d];
            if (BWD) sDO[tid] = a.dpooled[(e * kEncPoolQ + qi) * kEncD + kEncHd * h + d];
        }
        __syncthreads();
        if (tid < T) {
            float sc[kEncPoolQ];
#pragma unroll
            for (int qi = 0; qi < kEncPoolQ; ++qi) sc[qi] = 0.f;
            for (int d = 0; d < kEncHd; ++d) {
                const float kx = sK[tid * kPS + d];
#pragma unroll
                for (int qi = 0; qi < kEncPoolQ; ++qi) sc[qi] += sQ[qi * kEncHd + d] * kx;
            }
#pragma unroll
            for (int qi = 0; qi < kEncPoolQ; ++qi) sP[qi * kEncMaxTokens + tid] = sc[qi];
        }
        __syncthreads();
        // softmax over the keys: wave w handles queries 2 w, 2 w + 1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int qi = 2 * wave + u;
            float v[3], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = lane + 64 * j;
                v[j] = t < T ? sP[qi * kEncMaxTokens + t] : -INFINITY;
                mx = fmaxf(mx, v[j]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[j] = expf(v[j] - mx); l += v[j]; }
            l = wave_sum(l);
            const float inv = 1.f / l;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = lane + 64 * j;
                if (t < kEncMaxTokens) sP[qi * kEncMaxTokens + t] = t < T ? v[j] * inv : 0.f;
            }
        }
        __syncthreads();
        if constexpr (!BWD) {
            const int qi = tid >> 5, d = tid & 31;
            float o = 0.f;
            for (int t = 0; t < T; ++t) o += sP[qi * kEncMaxTokens + t] * sV[t * kPS + d];
            a.pooled[(e * kEncPoolQ + qi) * kEncD + kEncHd * h + d] = o;
        } else {
            // dp[qi][t] = dO[qi] . v[t];  ds = p (dp - sum_t p dp)
            float dp[kEncPoolQ];
            if (tid < T) {
#pragma unroll
                for (int qi = 0; qi < kEncPoolQ; ++qi) dp[qi] = 0.f;
                for (int d = 0; d < kEncHd; ++d) {
                    const float vx = sV[tid * kPS + d];
#pragma unroll
                    for (int qi = 0; qi < kEncPoolQ; ++qi) dp[qi] += sDO[qi * kEncHd + d] * vx;
                }
#pragma unroll
                for (int qi = 0; qi < kEncPoolQ; ++qi) sDS[qi * kEncMaxTokens + tid] = dp[qi];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qi = 2 * wave + u;
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = lane + 64 * j;
                    if (t < T) dot += sP[qi * kEncMaxTokens + t] * sDS[qi * kEncMaxTokens + t];
                }
                dot = wave_sum(dot);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = lane + 64 * j;
                    if (t < T) sDS[qi * kEncMaxTokens + t] = sP[qi * kEncMaxTokens + t] * (sDS[qi * kEncMaxTokens + t] - dot);
                }
            }
            __syncthreads();
            if (tid < T) {                                       // dK[t] = sum_q ds[q][t] q[q],  dV[t] = sum_q p[q][t] dO[q]
                float pq[kEncPoolQ], dsq[kEncPoolQ];
#pragma unroll
                for (int qi = 0; qi < kEncPoolQ; ++qi) { pq[qi] = sP[qi * kEncMaxTokens + tid]; dsq[qi] = sDS[qi * kEncMaxTokens + tid]; }
#pragma unroll
                for (int d4 = 0; d4 < 8; ++d4) {
                    f32x4 gk = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int qi = 0; qi < kEncPoolQ; ++qi)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            gk[j] += dsq[qi] * sQ[qi * kEncHd + 4 * d4 + j];
                            gv[j] += pq[qi] * sDO[qi * kEncHd + 4 * d4 + j];
                        }
                    store_act4<BF16>(a.dkv, (r0 + tid) * (2 * kEncD) + kEncHd * h + 4 * d4, gk);
                    store_act4<BF16>(a.dkv, (r0 + tid) * (2 * kEncD) + kEncD + kEncHd * h + 4 * d4, gv);
                }
            }
            {                                                    // dq[qi][d] += sum_t ds[qi][t] k[t][d]
                const int qi = tid >> 5, d = tid & 31;
                float acc = 0.f;
                for (int t = 0; t < T; ++t) acc += sDS[qi * kEncMaxTokens + t] * sK[t * kPS + d];
                atomicAdd(a.dq + qi * kEncD + kEncHd * h + d, acc);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// token assembly and small element-wise kernels
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(192) void tok_assemble_kernel(const float* __restrict__ stem, const float* __restrict__ extra,
                                                           const float* __restrict__ bias, int64_t B, int n_extra, int n_stem,
                                                           float* __restrict__ x0) {
    // 4 token rows per workgroup (48 lanes x 16 bytes each), blockIdx.y strides over the events: the row / token / event
    // indices are block-level integers, not three 64-bit divisions per element
    const int T = n_extra + n_stem;
    const int t = 4 * blockIdx.x + threadIdx.x / 48, c4 = 4 * (threadIdx.x % 48);
    if (t >= T) return;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias && t >= n_extra) bv = *reinterpret_cast<const f32x4*>(bias + (int64_t)t * kEncD + c4);
    for (int64_t e = blockIdx.y; e < B; e += gridDim.y) {
        f32x4 v;
        if (t < n_extra) v = *reinterpret_cast<const f32x4*>(extra + (e * n_extra + t) * kEncD + c4);
        else v = *reinterpret_cast<const f32x4*>(stem + (e * n_stem + (t - n_extra)) * kEncD + c4) + bv;
        *reinterpret_cast<f32x4*>(x0 + (e * T + t) * kEncD + c4) = v;
    }
}

template <bool BF16>
__global__ __launch_bounds__(192) void tok_backward_kernel(const float* __restrict__ dx0, const void* __restrict__ dact, int64_t B,
                                                           int n_extra, int n_det, void* __restrict__ gpad, int64_t gseq,
                                                           int64_t goff, float* __restrict__ dextra, float* __restrict__ dbias) {
    // 4 tokens per workgroup, 48 lanes x 4 channels per token row (16-byte loads, 8-byte bf16 stores); blockIdx.y = a range of
    // events; 4 events per iteration with the loads first.  (One channel per thread and one event at a time: 2-byte stores and 64
    // dependent round trips per thread, 143 us for 288 MB.)
    const int T = n_extra + 61 * n_det;
    const int t = 4 * blockIdx.x + threadIdx.x / 48, c4 = 4 * (threadIdx.x % 48);
    if (t >= T) return;
    const int64_t per = (B + gridDim.y - 1) / gridDim.y;
    const int64_t e0 = (int64_t)blockIdx.y * per, e1 = e0 + per < B ? e0 + per : B;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int j = t - n_extra, d = j >= 0 ? j / 61 : 0, pp = j >= 0 ? j - 61 * d : 0;
    constexpr int U = 4;
    for (int64_t eb = e0; eb < e1; eb += U) {
        f32x4 v[U], da[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = eb + u < e1 ? eb + u : e1 - 1;
            v[u] = *reinterpret_cast<const f32x4*>(dx0 + (e * T + t) * kEncD + c4);
            da[u] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (t >= n_extra) da[u] = load_act4<BF16>(dact, ((e * n_det + d) * 61 + pp) * kEncD + c4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t e = eb + u;
            if (e >= e1) break;
            sum += v[u];
            if (t < n_extra) {
                if (dextra) *reinterpret_cast<f32x4*>(dextra + (e * n_extra + t) * kEncD + c4) = v[u];
            } else {
                store_act4<BF16>(gpad, (e * n_det + d) * gseq + goff + (int64_t)pp * kEncD + c4, v[u] * da[u]);
            }
        }
    }
    if (dbias && e0 < e1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(dbias + (int64_t)t * kEncD + c4 + k, sum[k]);
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4*>(dst + 4 * i) = o;
    }
}

unsigned grid_for(int64_t items, int per_block, unsigned cap = 2048) {
    const int64_t b = (items + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : b > cap ? cap : b);
}

}  // namespace

int ln_forward(bool bf16, const LnArgs& a, hipStream_t s) {
    if (a.M <= 0) return PF_OK;
    const unsigned grid = grid_for(a.M, 4 * 4);
    if (bf16) hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    return launch_status();
}
int ln_backward(bool bf16, const LnArgs& a, hipStream_t s) {
    if (a.M <= 0) return PF_OK;
    const unsigned grid = grid_for(a.M, 4 * 16, PF_LN_BWD_GRID);
    if (bf16) hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    return launch_status();
}

template <bool BF16>
static int attn_launch(const AttnArgs& a, bool backward, hipStream_t s) {
    const dim3 grid(kEncHeads, (unsigned)a.B);
    auto go = [&](auto kern, size_t lds) {
        if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(kern), (int)lds)) return (int)PF_ERR_HIP;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return launch_status();
    };
    if (!backward) return go(attn_fwd_kernel<BF16>, row_img_bytes<BF16>() + tr_img_bytes<BF16>());
    const int rc = go(attn_bwd_dq_kernel<BF16>, 2 * row_img_bytes<BF16>() + tr_img_bytes<BF16>());
    if (rc != PF_OK) return rc;
    return go(attn_bwd_dkv_kernel<BF16>, 2 * row_img_bytes<BF16>() + 2 * tr_img_bytes<BF16>() + 2 * kEncMaxTokens * sizeof(float));
}
int attn_forward(bool bf16, const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0) return PF_OK;
    if (a.T < 1 || a.T > kEncMaxTokens) return PF_ERR_UNSUPPORTED;
    return bf16 ? attn_launch<true>(a, false, s) : attn_launch<false>(a, false, s);
}
int attn_backward(bool bf16, const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0) return PF_OK;
    if (a.T < 1 || a.T > kEncMaxTokens) return PF_ERR_UNSUPPORTED;
    return bf16 ? attn_launch<true>(a, true, s) : attn_launch<false>(a, true, s);
}

int pool_forward(bool bf16, const PoolArgs& a, hipStream_t s) {
    if (a.B <= 0) return PF_OK;
    if (a.T < 1 || a.T > kEncMaxTokens) return PF_ERR_UNSUPPORTED;
    if (bf16) hipLaunchKernelGGL((pool_kernel<true, false>), dim3((unsigned)a.B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pool_kernel<false, false>), dim3((unsigned)a.B), dim3(256), 0, s, a);
    return launch_status();
}
int pool_backward(bool bf16, const PoolArgs& a, hipStream_t s) {
    if (a.B <= 0) return PF_OK;
    if (a.T < 1 || a.T > kEncMaxTokens) return PF_ERR_UNSUPPORTED;
    if (bf16) hipLaunchKernelGGL((pool_kernel<true, true>), dim3((unsigned)a.B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pool_kernel<false, true>), dim3((unsigned)a.B), dim3(256), 0, s, a);
    return launch_status();
}

int tok_assemble(const float* stem_tokens, const float* extra, const float* token_bias, int64_t B, int n_extra, int n_stem,
                 float* x0, hipStream_t s) {
    if (B <= 0) return PF_OK;
    const int T = n_extra + n_stem;
    hipLaunchKernelGGL(tok_assemble_kernel, dim3((unsigned)((T + 3) / 4), (unsigned)(B < 256 ? B : 256)), dim3(192), 0, s, stem_tokens, extra,
                       token_bias, B, n_extra, n_stem, x0);
    return launch_status();
}
int tok_backward(bool bf16, const float* dx0, const void* dact, int64_t B, int n_extra, int n_det, void* gpad, int64_t gpad_seq_stride,
                 int64_t gpad_offset, float* dextra, float* dbias, hipStream_t s) {
    if (B <= 0) return PF_OK;
    const int T = n_extra + 61 * n_det;
    const unsigned ny = (unsigned)(B < 64 ? B : 64);
    const dim3 grid((unsigned)((T + 3) / 4), ny);
    if (bf16) hipLaunchKernelGGL(tok_backward_kernel<true>, grid, dim3(192), 0, s, dx0, dact, B, n_extra, n_det, gpad,
                                 gpad_seq_stride, gpad_offset, dextra, dbias);
    else hipLaunchKernelGGL(tok_backward_kernel<false>, grid, dim3(192), 0, s, dx0, dact, B, n_extra, n_det, gpad,
                            gpad_seq_stride, gpad_offset, dextra, dbias);
    return launch_status();
}
int cast_rows(bool bf16, const float* src, void* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return PF_OK;
    if (!bf16) {
        if (hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return PF_ERR_HIP;
        return PF_OK;
    }
    if (n % 4) return PF_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256 * 4)), dim3(256), 0, s, src, reinterpret_cast<__bf16*>(dst), n / 4);
    return launch_status();
}

}  // namespace pf
