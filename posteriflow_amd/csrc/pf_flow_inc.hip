// pf_flow_inc.hip -- incremental inverse (sampling direction) of the masked-autoregressive spline flow.
//
// nflows inverts an autoregressive layer with D full conditioner passes (AutoregressiveTransform.inverse,
// executed under src/ahsd/models/flows.py:637), and so does pf_flow_inverse: every pass streams the whole
// layer (~0.45 MB of fragments) although feature i's spline parameters only need the hidden units of
// degree <= i, and of those only the units of degree exactly i are NEW in pass i (their inputs -- x_0..x_{i-1}
// and the units of lower degree -- were final one pass earlier).  Here a pass computes just those units:
// with the hidden units sorted by degree they are a contiguous range of 2-3 sixteen-unit tiles per hidden
// layer (H / (D - 1) units), whose reduction runs over the units of degree <= i only, and the activations
// of all five hidden stages of the layer stay in LDS (bf16, 5 x 16 rows x 256) between passes.  Summed over the
// D passes the work is ONE masked conditioner evaluation per layer ("exact incremental
// algorithm") instead of D dense ones.  The result is the same x: no approximation is involved.
//
// One workgroup = 16 draws (one sixteen-row MFMA column tile), 4 waves, 46 KB of LDS, <= 168 VGPRs: three of them share
// a CU.  A pass is a chain of seven barrier-separated stages (six dependent GEMV-like stages + the spline) whose
// cost is latency, so independent workgroups on the same CU are what fills it; 32 / 48-draw workgroups sharing one
// weight fetch measured 25-30 % slower per draw.  Transposed MFMA form as everywhere:
// out^T[unit, row] = W[unit, k] . act^T[k, row]; a wave owns a new unit tile, its fp32 residual values h live in its
// registers through the five stages of a pass.  bf16 operands / fp32 accumulation, the input features enter the first
// masked layer as a bf16 hi + lo pair, context enters as per-context-row projections computed once by the caller
// (C = 0: none).  Spline inversion: 16 lanes per draw (lane = bin), DPP row reductions.
//
// Latency rules the code follows (each measured with the stage timestamps of $PF_INC_TRACE, scripts/trace_inc.py):
//   - every global load is issued one stage before its value is used, in straight-line code with a fixed number of
//     loads per stage (fragments beyond the units of degree <= i are fetched through a buffer resource that ends
//     before them: zeros, no traffic); a conditional load between another load and its use makes the compiler drain
//     the whole queue at that use;
//   - the head of pass i + 1 (biases, context projections, W0, first hidden matrix) is requested under the spline of
//     pass i, the two fragment buffers swap roles every pass (passes are unrolled by two);
//   - MFMA operands live in registers as dword vectors: a <8 x bfloat> that crosses a branch is split and re-packed
//     right behind its load;
//   - LDS rows have a fixed stride, so operand addresses are one VGPR + an instruction immediate.
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <mutex>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "pf_flow_fwd_kernel.h"

// side-build switches (scripts/side_obj.sh NAME pf_flow_inc -DPF_INC_ROT=1, A/B with scripts/ab_inc.sh; LABLOG R4.13):
// PF_INC_SKIP (default on, +5 %): a whole group of four masked-zero k-steps is jumped over;
// PF_INC_ROT (tile ownership rotated with the workgroup) and PF_INC_PRIO (s_setprio around the chains): measured, no gain
#ifndef PF_INC_ROT
#define PF_INC_ROT 0
#endif
#ifndef PF_INC_SKIP
#define PF_INC_SKIP 1
#endif
#ifndef PF_INC_PRIO
#define PF_INC_PRIO 0
#endif

namespace pf {
namespace {
constexpr int kParS = kParStride;          // floats per row in the spline-parameter transpose (52)
// Bytes per activation row in LDS: 256 units whatever H is (LDS offsets are then instruction immediates).  Inside a row
// the 16-byte chunk q sits at chunk (q & ~15) | ((q ^ row) & 15): a ds_read_b128 serves lanes {0-3, 12-15, 20-27}, ... in
// one pass -- every row 0..15 once, with g = lane >> 4 in {0, 1} or {2, 3} -- and this XOR sends those 16 lanes to
// the 16 different 16-byte slots of the 256-byte bank window (rows padded by 16 bytes collide on one slot per pass:
// 45 % of the LDS cycles of the padded version were bank conflicts, profiles/r01_inc_pmc.txt).
constexpr int kActStride = 256 * 2;
constexpr int kActStrideF32 = 256 * 4;

struct IncParams {
    FwdParams sp;              // spline scalars only: tail_bound, min_w, min_h, min_d, deriv_const
    const char* w;             // L layer blocks (see pf_hip.h, pf_flow_inverse_inc)
    const float* proj;         // [ctx_rows][L][3][H] fp32 in sorted-unit order, or null when C = 0
    const float* z;            // [batch][D]
    float* x;                  // [batch][D]
    float* logdet;             // [batch] or null
    uint32_t* fail;            // [batch] or null
    const int32_t* inv_perm;   // [D] or null
    int64_t batch, ctx_rows, layer_bytes;
    int64_t off_w1[2], off_w2[2], off_wf, off_bias;      // byte offsets inside a layer block (W0 at 0)
    int D, H, K, L;
    int u1[17];                // number of hidden units (sorted order) with degree <= i
    unsigned long long* trace; // $PF_INC_TRACE: stage timestamps of workgroup 0 / wave 0 (debug)
};

// MFMA operands travel as four dwords: a <8 x bfloat> that lives across a branch is split into halves and re-packed
// (v_bfi) right behind its load, which makes the load synchronous
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 bf16_of(f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    return o;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    return v;
}
// (math helpers: pf_flow_fwd_kernel.h -- bare instructions in bf16 mode, corrected to ~1 ulp in the fp32 parity mode)
template <bool FAST> __device__ __forceinline__ float exp_m(float v) { return FAST ? pf_exp<true>(v) : pf_exp_acc(v); }
template <bool FAST> __device__ __forceinline__ float log_m(float v) { return FAST ? pf_log<true>(v) : pf_log_acc(v); }
template <bool FAST> __device__ __forceinline__ float div_m(float a, float b) { return FAST ? pf_div<true>(a, b) : pf_div_acc(a, b); }
template <bool FAST> __device__ __forceinline__ float sqrt_m(float v) { return __builtin_amdgcn_sqrtf(v); }
template <bool FAST> __device__ __forceinline__ float softplus_m(float u) { return FAST ? pf_softplus<true>(u) : pf_softplus_acc(u); }
template <bool FAST> __device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        v[e] = FAST ? __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v[e] * -1.44269504f)) : pf_div_acc(1.f, 1.f + pf_exp_acc(-v[e]));
    return v;
}

// ---- lane-parallel spline inverse: 16 lanes (one DPP row) per draw, lane j = bin j ------------------------
// With one lane per draw the inversion is ~600 dependent instructions on a single wave while the other seven
// wait (28 % of the kernel); with a lane per bin the softmax sums, the knot prefix sums and the bin selection
// are 4-step DPP row operations and the chain is ~5x shorter.  Same formulas as rqs_pair_inverse; the prefix
// sums associate as a scan instead of left to right (fp32 rounding only).
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum(float v) {
    v += row_dpp<0xB1>(v); v += row_dpp<0x4E>(v); v += row_dpp<0x141>(v); v += row_dpp<0x140>(v);
    return v;
}
// max over the 16 lanes of a row for two values at once: v_max_f32_dpp directly (through fmaxf every step costs a DPP
// move plus two NaN-quieting v_max); a VGPR written by a VALU instruction needs two wait states before a DPP read
__device__ __forceinline__ void row_max2(float& a, float& b) {
    asm volatile(
        "s_nop 1\n"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(a), "+v"(b));
}
// raw transcendental instructions (1 ulp, no denormal-range fix-ups: every argument here is a normal number)
__device__ __forceinline__ float row_scan(float v) {            // inclusive prefix sum over the 16 lanes of a row
    v += row_dpp<0x111>(v); v += row_dpp<0x112>(v); v += row_dpp<0x114>(v); v += row_dpp<0x118>(v);
    return v;
}
template <bool FAST>
__device__ __forceinline__ void rqs_row16_inverse(const float* par, float yin, int K, const FwdParams& p, int j,
                                                  float& x, float& ld, bool& bad) {
    const bool live = j < K;
    const float uw = live ? par[j] : -INFINITY, uh = live ? par[16 + j] : -INFINITY;
    const float ud = par[32 + j];                                   // raw derivative at the RIGHT knot of bin j (j < K - 1)
    const float tb = p.tail_bound, span = 2.f * tb;
    // (every DPP reduction is evaluated OUTSIDE conditionals: inside `live ? ... : ...` the compiler may run it with
    // the non-live lanes masked off, and a DPP read of a disabled lane is not the lane's value)
    float mw = uw, mh = uh;
    row_max2(mw, mh);
    const float ew = live ? exp_m<FAST>(uw - mw) : 0.f, eh = live ? exp_m<FAST>(uh - mh) : 0.f;
    const float cw = div_m<FAST>(1.f - p.min_w * (float)K, row_sum(ew));
    const float ch = div_m<FAST>(1.f - p.min_h * (float)K, row_sum(eh));
    const float wj = live ? p.min_w + cw * ew : 0.f, hj = live ? p.min_h + ch * eh : 0.f;
    const float cumw = row_scan(wj), cumh = row_scan(hj);
    const bool last = j == K - 1;
    const float kr = last ? tb : span * cumw - tb, hr = last ? tb : span * cumh - tb;       // right knots of bin j
    const float dr_raw = last ? p.deriv_const : ud;
    // left knots / derivative = the right ones of bin j - 1 (row_shr:1), the interval's ends for bin 0
    float kl = row_dpp<0x111>(kr), hl = row_dpp<0x111>(hr), dl_raw = row_dpp<0x111>(dr_raw);
    if (j == 0) { kl = -tb; hl = -tb; dl_raw = p.deriv_const; }
    // searchsorted on the heights: bin j holds y if y >= its left knot and not >= its right knot (last knot + 1e-6)
    const bool ge_r = yin >= (last ? tb + 1e-6f : hr);
    const bool ge_l = j == 0 ? true : yin >= hl;
    const bool sel = live && ge_l && !ge_r;
    const float w = kr - kl, hh = hr - hl;
    const float dl = p.min_d + softplus_m<FAST>(dl_raw), dr = p.min_d + softplus_m<FAST>(dr_raw);
    const float delta = div_m<FAST>(hh, w);
    const float dy = yin - hl, s2 = dl + dr - 2.f * delta;
    const float a = dy * s2 + hh * (delta - dl), b = hh * dl - dy * s2, c0 = -delta * dy;
    const float disc = b * b - 4.f * a * c0;
    const float root = div_m<FAST>(2.f * c0, -b - sqrt_m<FAST>(fmaxf(disc, 0.f)));
    const float tt = root * (1.f - root), den = delta + s2 * tt, omt = 1.f - root;
    const float dnum = delta * delta * (dr * root * root + 2.f * delta * tt + dl * omt * omt);
    const bool inside = (yin >= -tb) && (yin <= tb);
    // exactly one bin is selected inside the interval: its values reach every lane of the row through a sum
    const float xsel = sel ? root * w + kl : 0.f;
    const float lsel = sel ? -(log_m<FAST>(dnum) - 2.f * log_m<FAST>(den)) : 0.f;
    const float bsel = (sel && !(disc >= 0.f)) ? 1.f : 0.f;
    const float xs = row_sum(xsel), ls = row_sum(lsel), bs = row_sum(bsel);
    x = inside ? xs : yin;
    ld = inside ? ls : 0.f;
    bad = inside && bs > 0.f;
}

#define PF_TR(k) do { if (p.trace && blockIdx.x == 0 && tid == 0) p.trace[((p.L - 1 - l) * 16 + i) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// kF32: the fp32 parity mode.  fp32 weights and activations, v_mfma_f32_16x16x4_f32: a fragment is 16 units x 16 k
// (lane (r, kq) holds W[r][16 q + 4 kq .. + 3], one dwordx4; B = 4 consecutive floats of the activation row), 16
// fragments per tile row at H = 256, activation rows of 1024 bytes (80 KB of LDS state: one workgroup per CU).
template <int kCols, int kThreads, bool kCtx, bool kF32>
__global__ __launch_bounds__(kThreads)
__attribute__((amdgpu_waves_per_eu(kThreads == 256 && !kF32 ? 3 : 1, kThreads == 256 && !kF32 ? 3 : 2)))
void flow_inverse_inc_kernel(const IncParams p) {
    constexpr int kRows = 16 * kCols;
    constexpr int NF = kF32 ? 16 : 8;                            // fragments (k-steps of 16 / 32) per tile row, at most
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = p.D, H = p.H, HK = kF32 ? H / 16 : H / 32;
    constexpr int AS = kF32 ? kActStrideF32 : kActStride;
    char* const act = smem;                                      // 5 stages x [kRows][H] bf16
    char* const xb = act + (size_t)5 * kRows * AS;               // [kRows][32] bf16: x hi (0..15) | lo (16..31)
    float* const xs = reinterpret_cast<float*>(xb + kRows * 64); // [kRows][16] current layer's input, fp32
    float* const ys = xs + kRows * 16;                           // [kRows][16] current layer's output
    float* const par = ys + kRows * 16;                          // [kRows][52] spline parameters of one feature
    float* const ldacc = par + kRows * kParS;                    // [kRows]
    uint32_t* const badf = reinterpret_cast<uint32_t*>(ldacc + kRows);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    // tile ownership rotates with the workgroup (PF_INC_ROT, experiment): the waves of the workgroups that share a CU own
    // their tiles on different SIMDs (the fourth wave has no tile in stages a-e, the third often none)
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = PF_INC_ROT == 0 ? wave_hw
                   : (wave_hw + (PF_INC_ROT == 1 ? (int)(blockIdx.x >> 8) : PF_INC_ROT == 2 ? (int)blockIdx.x : (int)(blockIdx.x >> 3))) & (kThreads / 64 - 1);
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    auto act_of = [&](int s) { return act + (size_t)s * kRows * AS; };
    const int lane16 = lane * 16;
    // B-operand read of k-step ks (chunk 4 ks + g of row c): swz[ks & 3] + (ks >> 2) * 256 -- see kActStride
    int swz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) swz[j] = c * AS + (((c ^ g) << 4) ^ (j << 6));

    // ---- initial state: y = z (coordinates of the last layer's output), log-det 0 --------------------
    for (int s = tid; s < kRows * 16; s += kThreads) {
        const int r = s >> 4, d = s & 15;
        int64_t row = row0 + r;
        if (row >= p.batch) row = p.batch - 1;
        ys[s] = d < D ? p.z[row * D + d] : 0.f;
    }
    if (tid < kRows) { ldacc[tid] = 0.f; badf[tid] = 0u; }
    // activations of units not computed yet are read through zero (masked) weights: they must be finite
    for (int s = tid * 16; s < 5 * kRows * AS; s += kThreads * 16) *reinterpret_cast<f32x4*>(act + s) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // this lane's context rows (one per column tile) are fixed for the whole kernel: byte offsets of their projection
    // blocks relative to the workgroup's first context row (32-bit, through a buffer resource: one VGPR per column)
    const int64_t per = p.proj ? p.batch / p.ctx_rows : 1;
    const int64_t crow_first = p.proj ? (row0 < p.batch ? row0 : p.batch - 1) / per : 0;
    const int proj_row_bytes = p.L * 3 * H * 4;
    int poff[kCols];
#pragma unroll
    for (int cc = 0; cc < kCols; ++cc) {
        int64_t row = row0 + 16 * cc + c;
        if (row >= p.batch) row = p.batch - 1;
        poff[cc] = (int)(row / per - crow_first) * proj_row_bytes;
    }
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.proj ? p.proj + crow_first * (int64_t)(p.L * 3 * H) : nullptr), 0, kCtx ? kRows * proj_row_bytes : 0, 0x00020000);
    auto proj_load = [&](int cc, int l, int which, int u) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, poff[cc] + u * 4, ((l * 3 + which) * H) * 4, 0));
    };
    for (int l = p.L - 1; l >= 0; --l) {
        const char* lw = p.w + (size_t)l * p.layer_bytes;
        const float* bias = reinterpret_cast<const float*>(lw + p.off_bias);   // b0 | b1_0 | b2_0 | b1_1 | b2_1 | bf[D][48]
        // x estimate of this layer starts at zero (only solved features are ever read through non-zero weights)
        for (int s = tid; s < kRows * 32; s += kThreads) reinterpret_cast<__bf16*>(xb)[s] = (__bf16)0.f;
        for (int s = tid; s < kRows * 16; s += kThreads) xs[s] = 0.f;
        __syncthreads();

        // Every global load of a pass is issued one stage (0.4-0.5 us, about an L2 round trip) before its value is
        // used, in straight-line code with a FIXED number of loads per stage, so that the compiler can wait for exactly
        // the fragments a stage needs while the next stage's are in flight (with a load count that depends on a branch
        // it drains everything before each MFMA chain; a k-step under a branch puts an LDS round trip between
        // consecutive MFMAs).  The reduction always runs over NF k-steps; those beyond the units of degree <= i have
        // masked-zero weights: the buffer resource of a fetch ends after kmax fragments, the loads beyond it return
        // zeros without touching memory (the load count stays fixed, the L1 / L2 traffic triangular); offsets are
        // lane * 16 (+ 4096 j) + an instruction immediate.  Two fragment buffers X / Y alternate; the head of pass
        // i + 1 (biases, context projections, W0 and the first hidden matrix) is requested while pass i computes its
        // spline parameters, so the buffers swap roles every pass.
        auto fetch = [&](u32x4 (&buf)[NF], int64_t off, int tile, int kmax) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(lw) + off + (int64_t)tile * HK * 1024, 0, kmax * 1024, 0x00020000);
#pragma unroll
            for (int ks = 0; ks < NF; ++ks)
                buf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16 + 4096 * (ks >> 2) + (ks & 3) * 1024, 0, 0);
        };
        auto gemm = [&](const u32x4 (&a)[NF], const char* src, int kmax, f32x4 (&v)[kCols]) {
#if PF_INC_PRIO
            __builtin_amdgcn_s_setprio(PF_INC_PRIO);          // (experiment) the chain's wave ahead of its SIMD's other workgroups
#endif
#pragma unroll
            for (int cc = 0; cc < kCols; ++cc) {
                // (k-steps >= kmax multiply zero weights -- see fetch -- with whatever finite activations the row holds:
                // zeros beyond H)
                const char* brow = src + 16 * cc * AS;
                f32x4 v0{0.f, 0.f, 0.f, 0.f}, v1{0.f, 0.f, 0.f, 0.f};
                if constexpr (kF32) {
                    // four fragments (64 k) per group; the MFMAs of groups beyond the units of degree <= i are skipped
                    // (the loads are not: their count must stay fixed)
#pragma unroll
                    for (int q0 = 0; q0 < NF; q0 += 4) {
                        if (q0 < kmax) {
                            f32x4 b[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(brow + swz[q] + (q0 >> 2) * 256);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 af = __builtin_bit_cast(f32x4, a[q0 + q]);
                                v0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], b[q][0], v0, 0, 0, 0);
                                v1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], b[q][1], v1, 0, 0, 0);
                                v0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], b[q][2], v0, 0, 0, 0);
                                v1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], b[q][3], v1, 0, 0, 0);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int k0 = 0; k0 < NF; k0 += 4) {      // four operand reads in flight (register budget of the 4-wave variant)
#if PF_INC_SKIP
                        // a whole group of masked-zero fragments (units of degree > i): no reads, no MFMAs.  One uniform
                        // branch per chain, not one per k-step (that puts an LDS round trip between consecutive MFMAs);
                        // the fragment LOADS stay unconditional: their count must not depend on a branch
                        if (k0 > 0 && k0 >= kmax) continue;
#endif
                        u32x4 b[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<const u32x4*>(brow + swz[ks] + (k0 >> 2) * 256);
                        v0 = mfma_bf16(a[k0], b[0], v0);
                        v1 = mfma_bf16(a[k0 + 1], b[1], v1);
                        v0 = mfma_bf16(a[k0 + 2], b[2], v0);
                        v1 = mfma_bf16(a[k0 + 3], b[3], v1);
                        if (kThreads == 256) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                v[cc] = v0 + v1;
            }
#if PF_INC_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        // activation tile of this lane (4 consecutive units of row 16 cc + c) into stage s
        auto put = [&](int s, int cc, int u, f32x4 val) {
            char* row = act_of(s) + (16 * cc + c) * AS;
            if constexpr (kF32) {
                const int q = u >> 2;                                               // 16-byte chunk of units u .. u + 3
                *reinterpret_cast<f32x4*>(row + (((q ^ c) & 15) << 4) + (q >> 4) * 256) = val;
            } else {
                const int q = u >> 3;                                               // 8 units per chunk
                *reinterpret_cast<bf16x4*>(row + (((q ^ c) & 15) << 4) + (q >> 4) * 256 + (u & 4) * 2) = bf16_of(val);
            }
        };
        auto kfrag = [&](int i) { return kF32 ? (p.u1[i] + 15) / 16 : (p.u1[i] + 31) / 32; };   // fragments holding the units of degree <= i
        // head registers of the coming pass
        // (a stage's bias travels with its weights: bX / bY belong to the fragment buffers X / Y)
        f32x4 b0, bA, bB, pr0[kCols], pg0[kCols], pg1[kCols];
        u32x4 a0;
        const int wf = wave < 3 ? wave : 2;

        // stage f of pass i (spline parameters of feature i from the last hidden state) + the spline inversion;
        // X holds the stage's weights, Y receives the first hidden matrix of pass i + 1
        auto tail = [&](int i, u32x4 (&X)[NF], f32x4& bX, u32x4 (&Y)[NF], f32x4& bY) {
            if (wave < 3) {
                f32x4 v[kCols];
                gemm(X, act_of(4), kfrag(i), v);
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc)
                    *reinterpret_cast<f32x4*>(par + (16 * cc + c) * kParS + 16 * wf + 4 * g) = v[cc] + bX;
            }
            // head of pass i + 1, requested after the stage's own operands have been consumed: a conditional load
            // between another load and its use makes the compiler drain the whole queue at that use
            if (i + 1 < D) {
                const int t_lo = p.u1[i] / 16, ntile = (p.u1[i + 1] + 15) / 16 - t_lo;
                if (wave < ntile) {
                    const int t = t_lo + wave, u = 16 * t + 4 * g;
                    b0 = *reinterpret_cast<const f32x4*>(bias + u);
                    bY = *reinterpret_cast<const f32x4*>(bias + H + u);
                    if constexpr (kCtx) {
#pragma unroll
                        for (int cc = 0; cc < kCols; ++cc) {
                            pr0[cc] = proj_load(cc, l, 0, u);
                            pg0[cc] = proj_load(cc, l, 1, u);
                        }
                    }
                    a0 = *reinterpret_cast<const u32x4*>(lw + ((size_t)t * 64 + lane) * 16);
                    fetch(Y, p.off_w1[0], t, kfrag(i + 1));
                }
            }
            __syncthreads();
            PF_TR(6);
            // ---- spline inversion: 16 lanes per draw (lane = bin), kThreads / 16 draws per sweep ----
            for (int r0 = 0; r0 < kRows; r0 += kThreads / 16) {
                const int r = r0 + (tid >> 4), j = tid & 15;
                if (r < kRows) {                                   // (uniform per 16-lane row)
                    float xv, ld;
                    bool bad;
                    rqs_row16_inverse<!kF32>(par + r * kParS, ys[r * 16 + i], p.K, p.sp, j, xv, ld, bad);
                    if (j == 0) {
                        xs[r * 16 + i] = xv;
                        const __bf16 hi = (__bf16)xv;
                        reinterpret_cast<__bf16*>(xb)[r * 32 + i] = hi;
                        reinterpret_cast<__bf16*>(xb)[r * 32 + 16 + i] = (__bf16)(xv - (float)hi);
                        ldacc[r] += ld;
                        if (bad) badf[r] = 1u;
                    }
                }
            }
            __syncthreads();
            PF_TR(7);
        };
        // pass i >= 1: the new hidden units (degree == i) through the five hidden stages, then the tail.
        // X holds the first hidden matrix on entry and stage f's weights on exit.
        auto pass = [&](int i, u32x4 (&X)[NF], f32x4& bX, u32x4 (&Y)[NF], f32x4& bY) {
            PF_TR(0);
            const int t_lo = p.u1[i - 1] / 16, ntile = (p.u1[i] + 15) / 16 - t_lo;
            const int kmax = kfrag(i);
            const bool mine = wave < ntile;                  // a wave owns a new tile across all column tiles
            const int t = t_lo + wave, u = 16 * t + 4 * g;
            f32x4 h[kCols], v[kCols];
            if (mine) {                                                             // ---- stage a: h0 = W0 . (x hi|lo) + b0 + relu(pc)
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc) {
                    f32x4 w0;
                    if constexpr (kF32) {
                        const f32x4 bx = *reinterpret_cast<const f32x4*>(xs + (16 * cc + c) * 16 + 4 * g);
                        const f32x4 af = __builtin_bit_cast(f32x4, a0);
                        f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bx[e], acc, 0, 0, 0);
                        w0 = acc + b0;
                    } else {
                        const u32x4 bx0 = *reinterpret_cast<const u32x4*>(xb + (16 * cc + c) * 64 + g * 16);
                        w0 = mfma_bf16(a0, bx0, f32x4{0.f, 0.f, 0.f, 0.f}) + b0;
                    }
                    if constexpr (kCtx) w0 = w0 + relu4(pr0[cc]);
                    h[cc] = w0;
                    put(0, cc, u, relu4(w0));
                }
            }
            __syncthreads();
            PF_TR(1);
            if (mine) {                                                             // ---- stage b: act1 = relu(W1 . act0 + b1)
                fetch(Y, p.off_w2[0], t, kmax);
                bY = *reinterpret_cast<const f32x4*>(bias + 2 * H + u);
                gemm(X, act_of(0), kmax, v);
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc)
                    put(1, cc, u, relu4(v[cc] + bX));
            }
            __syncthreads();
            PF_TR(2);
            if (mine) {                                                             // ---- stage c: h1 = h0 + (W2 . act1 + b2) sigmoid(pg0)
                fetch(X, p.off_w1[1], t, kmax);
                bX = *reinterpret_cast<const f32x4*>(bias + 3 * H + u);
                if constexpr (kCtx) {
#pragma unroll
                    for (int cc = 0; cc < kCols; ++cc)
                        pg1[cc] = proj_load(cc, l, 2, u);
                }
                gemm(Y, act_of(1), kmax, v);
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc) {
                    f32x4 d = v[cc] + bY;
                    if constexpr (kCtx) d = d * sigmoid4<!kF32>(pg0[cc]);
                    h[cc] = h[cc] + d;
                    put(2, cc, u, relu4(h[cc]));
                }
            }
            __syncthreads();
            PF_TR(3);
            if (mine) {                                                             // ---- stage d
                fetch(Y, p.off_w2[1], t, kmax);
                bY = *reinterpret_cast<const f32x4*>(bias + 4 * H + u);
                gemm(X, act_of(2), kmax, v);
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc)
                    put(3, cc, u, relu4(v[cc] + bX));
            }
            __syncthreads();
            PF_TR(4);
            // stage f's weights, one stage ahead; unconditional (out of range for waves >= 3): it sits between the
            // fetch of Y and its use
            fetch(X, p.off_wf, 3 * i + wf, wave < 3 ? kmax : 0);
            bX = *reinterpret_cast<const f32x4*>(bias + 5 * H + 48 * i + 16 * wf + 4 * g);
            if (mine) {                                                             // ---- stage e: h2 = h1 + (W2 . act3 + b2) sigmoid(pg1)
                gemm(Y, act_of(3), kmax, v);
#pragma unroll
                for (int cc = 0; cc < kCols; ++cc) {
                    f32x4 d = v[cc] + bY;
                    if constexpr (kCtx) d = d * sigmoid4<!kF32>(pg1[cc]);
                    h[cc] = h[cc] + d;
                    put(4, cc, u, h[cc]);
                }
            }
            __syncthreads();
            PF_TR(5);
            tail(i, X, bX, Y, bY);
        };
        u32x4 bufA[NF], bufB[NF];
        // feature 0 depends on no hidden unit that is new: its parameters are the bias (+ masked zeros)
        fetch(bufA, p.off_wf, wf, 0);
        bA = *reinterpret_cast<const f32x4*>(bias + 5 * H + 16 * wf + 4 * g);
        { const int i = 0; PF_TR(0); }
        tail(0, bufA, bA, bufB, bB);
        for (int i = 1; i < D; i += 2) {
            pass(i, bufB, bB, bufA, bA);
            if (i + 1 < D) pass(i + 1, bufA, bA, bufB, bB);
        }
        // this layer's input is the previous layer's output, reversed (ReversePermutation precedes every layer)
        for (int s = tid; s < kRows * 16; s += kThreads) {
            const int r = s >> 4, d = s & 15;
            ys[s] = d < D ? xs[r * 16 + (D - 1 - d)] : 0.f;
        }
        __syncthreads();
    }
    // ---- ys now holds x[:, ar_perm]; undo the autoregressive order and store ------------------------------
    if (tid < kRows) {
        const int64_t row = row0 + tid;
        if (row < p.batch) {
            for (int d = 0; d < D; ++d) {
                const int src = p.inv_perm ? p.inv_perm[d] : d;
                p.x[row * D + d] = ys[tid * 16 + src];
            }
            if (p.logdet) p.logdet[row] = ldacc[tid];
            if (p.fail && badf[tid]) atomicOr(p.fail + row, 1u);
        }
    }
}

// raw fp32 [N][K] row-major -> bf16 A fragments: fragment (tile, ks), lane (g, r): W[16 tile + r][32 ks + 8 g ..+7]
__global__ void inc_pack_frags_kernel(const float* wsrc, int n_rows, int k, __bf16* out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<int64_t>(n_rows) * k) return;
    const int ksteps = k / 32;
    const int e = i & 7, lane = (i >> 3) & 63;
    const int64_t f = i >> 9;
    const int tile = static_cast<int>(f / ksteps), ks = static_cast<int>(f % ksteps);
    out[i] = (__bf16)wsrc[static_cast<int64_t>(16 * tile + (lane & 15)) * k + 32 * ks + 8 * (lane >> 4) + e];
}
}  // namespace

int inc_pack_frags(const float* src, int n_rows, int k, void* out, hipStream_t s) {
    const int64_t tot = static_cast<int64_t>(n_rows) * k;
    inc_pack_frags_kernel<<<dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s>>>(
        src, n_rows, k, static_cast<__bf16*>(out));
    return launch_status();
}

int64_t inc_layer_bytes(int D, int H, bool f32) {
    const int64_t NT = H / 16, HK = f32 ? H / 16 : H / 32;          // 1 KB fragments per tile row
    return (NT + 4 * NT * HK + 3 * (int64_t)D * HK) * 1024 + (5 * (int64_t)H + 48 * (int64_t)D) * 4;
}

int flow_inverse_inc(const PfFlowDesc& d, float deriv_const, const int32_t* u1, const void* packed, const float* proj,
                     int64_t ctx_rows, const float* z, const int32_t* inv_perm, int64_t batch, float* x, float* logdet,
                     uint32_t* fail, hipStream_t s) {
    IncParams p{};
    p.sp.tail_bound = d.tail_bound; p.sp.min_w = d.min_bin_width; p.sp.min_h = d.min_bin_height;
    p.sp.min_d = d.min_derivative; p.sp.deriv_const = deriv_const;
    p.w = static_cast<const char*>(packed); p.proj = proj; p.z = z; p.x = x; p.logdet = logdet; p.fail = fail;
    p.inv_perm = inv_perm; p.batch = batch; p.ctx_rows = ctx_rows;
    p.D = d.features; p.H = d.hidden_features; p.K = d.num_bins; p.L = d.num_layers;
    const bool f32 = d.precision == PF_PREC_F32;
    const int64_t NT = p.H / 16, HK = f32 ? p.H / 16 : p.H / 32;
    p.off_w1[0] = NT * 1024;
    p.off_w2[0] = p.off_w1[0] + NT * HK * 1024;
    p.off_w1[1] = p.off_w2[0] + NT * HK * 1024;
    p.off_w2[1] = p.off_w1[1] + NT * HK * 1024;
    p.off_wf = p.off_w2[1] + NT * HK * 1024;
    p.off_bias = p.off_wf + 3 * (int64_t)p.D * HK * 1024;
    p.layer_bytes = inc_layer_bytes(p.D, p.H, f32);
    for (int i = 0; i <= p.D; ++i) p.u1[i] = u1[i];
    // One workgroup = 16 draws.  A pass is a chain of seven barrier-separated stages whose cost is latency, not
    // bandwidth, so the CU is filled with three independent 4-wave workgroups (46 KB of LDS, <= 168 VGPRs each) whose
    // stalls overlap; wider workgroups (32 / 48 draws sharing one weight fetch) measured 25-30 % slower per draw.
    // A wave owns one new tile per pass: at most 4 (8 with the 8-wave variant) sixteen-unit tiles may hold the
    // units of one degree.
    int max_tiles = 0;
    for (int i = 1; i < p.D; ++i) max_tiles = std::max(max_tiles, (u1[i] + 15) / 16 - u1[i - 1] / 16);
    if (max_tiles > 8) return PF_ERR_UNSUPPORTED;
    int threads = max_tiles <= 4 ? 256 : 512;
    static const bool force512 = [] { const char* ft = std::getenv("PF_INC_THREADS"); return ft && std::atoi(ft) == 512; }();
    if (force512) threads = 512;                     // (tuning knob, read once per process)
    const size_t lds = 5 * 16 * (size_t)(f32 ? kActStrideF32 : kActStride) + 16 * 64 + 16 * (16 + 16 + kParS + 2) * 4;
    using Kern = void (*)(const IncParams);
    static const Kern kerns[8] = {
        flow_inverse_inc_kernel<1, 256, false, false>, flow_inverse_inc_kernel<1, 256, true, false>,
        flow_inverse_inc_kernel<1, 512, false, false>, flow_inverse_inc_kernel<1, 512, true, false>,
        flow_inverse_inc_kernel<1, 256, false, true>,  flow_inverse_inc_kernel<1, 256, true, true>,
        flow_inverse_inc_kernel<1, 512, false, true>,  flow_inverse_inc_kernel<1, 512, true, true>};
    // the opt-in to > 64 KB of dynamic LDS is a per-DEVICE function attribute: remember it per device, under a lock
    {
        static std::mutex mu;
        static bool configured[64] = {};
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return PF_ERR_HIP;
        std::lock_guard<std::mutex> lock(mu);
        if (!configured[devid]) {
            for (const Kern k : kerns)
                if (!opt_in_lds(reinterpret_cast<const void*>(k), 96 * 1024))
                    return PF_ERR_HIP;
            configured[devid] = true;
        }
    }
    const int variant = (f32 ? 4 : 0) + (threads == 256 ? 0 : 2) + (proj ? 1 : 0);
    const bool tracing = std::getenv("PF_INC_TRACE") != nullptr;
    if (tracing && hipMalloc(&p.trace, p.L * 16 * 8 * sizeof(unsigned long long)) != hipSuccess) return PF_ERR_HIP;
    const dim3 grid(static_cast<unsigned>((batch + 15) / 16));
    kerns[variant]<<<grid, dim3(threads), lds, s>>>(p);
    if (tracing) {                   // debug: 100 MHz timestamps -> 10 ns units per stage, layers 0 and 1 as launched
        std::vector<unsigned long long> t(p.L * 16 * 8);
        if (hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(t.data(), p.trace, t.size() * sizeof(t[0]), hipMemcpyDeviceToHost) != hipSuccess) return PF_ERR_HIP;
        (void)hipFree(p.trace);
        for (int l = 0; l < 2 && l < p.L; ++l)
            for (int i = 0; i < p.D; ++i) {
                const unsigned long long* r = &t[(l * 16 + i) * 8];
                std::fprintf(stderr, "inc trace layer %d pass %2d:", l, i);
                for (int k = 1; k < 8; ++k) std::fprintf(stderr, " %5lld", (long long)(r[k] - r[k - 1]));
                std::fprintf(stderr, "  | pass %lld\n", (long long)(r[7] - r[0]));
            }
    }
    return launch_status();
}
}  // namespace pf
