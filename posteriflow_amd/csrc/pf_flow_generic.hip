// pf_flow_generic.hip -- the flow for shapes none of the scheduled kernels is built for.
//
// The scheduled kernels (pf_flow_fwd_kernel.h, pf_flow_wide_kernel.h, pf_flow_inc.hip) are specialised on H in {64 .. 256},
// K <= 16 and D <= H / 16: static per-wave fragment streams, three parameter tiles per feature, splines unrolled over 16
// bins.  The reference also builds other sizes -- FlowHead(12, 384, 24) of experiments/frozen_context_heads.py:159-163 --
// and a drop-in must evaluate them: this kernel takes any H (multiple of 16, <= 512), D <= 32, K <= 32, plain conditioner,
// forward (NSFPosteriorFlow.forward / compute_psd_aware_nll, flows.py:610-618, 727-779) and nflows' D-pass inverse
// (AutoregressiveTransform.inverse under flows.py:637), in both precisions.  Same arithmetic conventions as the scheduled
// kernels: transposed MFMA products (weights = A operand from packed fragments, the 16 rows of a workgroup = the MFMA
// columns), bf16 mode = bf16 operands (x as hi + lo) with fp32 accumulation / bias / residual / spline, fp32 mode =
// v_mfma_f32_16x16x4_f32.  What it does NOT have is their specialisation: masks are multiplied as zeros (dense count),
// the residual state lives in LDS, loops run over runtime tile and k-step counts.  Layout: pf_layout.h "generic".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "pf_flow_fwd_kernel.h"
#include "pf_status.h"

namespace pf {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int gu32x4;

constexpr int kGenWaves = 8;

// Generic-K spline of one (row, feature) pair by one lane, parameters in LDS: par[0 .. K) raw widths, [K .. 2K) raw
// heights, [2K .. 3K - 1) raw derivatives.  nflows' steps (rqs_pair in pf_flow_fwd_kernel.h follows the same ones for
// K <= 16 out of registers): softmax, min + (1 - min K) softmax, running cumsum, knots pinned to +-tail_bound, searchsorted
// with the last knot + 1e-6, derivative = min_d + softplus(raw), boundary derivative from the constant.  INV: search on
// the heights and solve the quadratic.
template <bool FAST, bool INV>
__device__ __forceinline__ void rqs_generic(const float* par, float v, int K, const FwdParams& p, float& out, float& ld, bool& bad) {
    const float tb = p.tail_bound;
    float mw = -INFINITY, mh = -INFINITY;
    for (int i = 0; i < K; ++i) { mw = fmaxf(mw, par[i]); mh = fmaxf(mh, par[K + i]); }
    float sw = 0.f, sh = 0.f;
    for (int i = 0; i < K; ++i) { sw += pf_exp<FAST>(par[i] - mw); sh += pf_exp<FAST>(par[K + i] - mh); }
    const float cw = pf_div<FAST>(1.f - p.min_w * (float)K, sw);
    const float ch = pf_div<FAST>(1.f - p.min_h * (float)K, sh);
    const float span = 2.f * tb;
    float cumw = 0.f, cumh = 0.f;
    float xl = -tb, xr = tb, yl = -tb, yr = tb;
    float dl_raw = p.deriv_const, dr_raw = p.deriv_const;
    bool prev_ge = true;
    for (int i = 0; i < K; ++i) {
        cumw += p.min_w + cw * pf_exp<FAST>(par[i] - mw);
        cumh += p.min_h + ch * pf_exp<FAST>(par[K + i] - mh);
        const bool last = i == K - 1;
        const float kr = last ? tb : span * cumw - tb;
        const float hr = last ? tb : span * cumh - tb;
        const float dr = last ? p.deriv_const : par[2 * K + i];
        const float knot = INV ? hr : kr;
        const bool ge = v >= (last ? tb + 1e-6f : knot);
        const bool sel = prev_ge && !ge;
        xr = sel ? kr : xr; yr = sel ? hr : yr; dr_raw = sel ? dr : dr_raw;
        xl = ge ? kr : xl;  yl = ge ? hr : yl;  dl_raw = ge ? dr : dl_raw;
        prev_ge = ge;
    }
    const float w = xr - xl, h = yr - yl;
    const float dl = p.min_d + pf_softplus<FAST>(dl_raw);
    const float dr = p.min_d + pf_softplus<FAST>(dr_raw);
    const float delta = pf_div<FAST>(h, w);
    const bool inside = (v >= -tb) && (v <= tb);
    bad = false;
    if constexpr (!INV) {
        const float th = pf_div<FAST>(v - xl, w);
        const float tt = th * (1.f - th);
        const float numer = h * (delta * th * th + dl * tt);
        const float den = delta + (dl + dr - 2.f * delta) * tt;
        const float omt = 1.f - th;
        const float dnum = delta * delta * (dr * th * th + 2.f * delta * tt + dl * omt * omt);
        out = inside ? yl + pf_div<FAST>(numer, den) : v;
        ld = inside ? pf_log<FAST>(dnum) - 2.f * pf_log<FAST>(den) : 0.f;
    } else {
        const float dy = v - yl;
        const float s2 = dl + dr - 2.f * delta;
        const float a = dy * s2 + h * (delta - dl);
        const float b = h * dl - dy * s2;
        const float c = -delta * dy;
        const float disc = b * b - 4.f * a * c;
        const float root = pf_div<FAST>(2.f * c, -b - (FAST ? __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) : sqrtf(fmaxf(disc, 0.f))));
        const float tt = root * (1.f - root);
        const float den = delta + s2 * tt;
        const float omt = 1.f - root;
        const float dnum = delta * delta * (dr * root * root + 2.f * delta * tt + dl * omt * omt);
        bad = inside && !(disc >= 0.f);
        out = inside ? root * w + xl : v;
        ld = inside ? -(pf_log<FAST>(dnum) - 2.f * pf_log<FAST>(den)) : 0.f;
    }
}

// conditioner re-evaluation (pf_flow_reevaluate, fp32 descs): where the conditioner's intermediate values go, fp32 [.., B, H]
struct ReevalSink {
    float* hs; float* t1s; float* t2s; float* gates; float* pc; float* h2; float* params;
    const float* drop;         // [2][L][B][H] or null
};

// MODE 0: forward, 1: D-pass inverse, 2: one layer's conditioner per workgroup (grid.y = layer), everything written out
template <bool BF16, int MODE>
__global__ __launch_bounds__(kGenWaves * 64) void flow_generic_kernel(const FwdParams p, const ReevalSink sink) {
    constexpr bool INV = MODE == 1, REEVAL = MODE == 2;
    constexpr bool FAST = BF16;
    constexpr int KSTEP = BF16 ? 32 : 16, ESZ = BF16 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FlowPlan& L = p.plan;
    const int D = L.D, H = L.H, K = L.K, M = L.M, C = L.C, NT = L.NT;
    const bool additive = L.additive != 0;       // masked-context conditioner: additive context inside the blocks, no gates,
                                                 // no ReversePermutation between the layers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;

    // ---- LDS carve: operand rows (act type, 16-byte pad per row), residual, spline parameters, small fp32 vectors --------
    const int sx = L.gKx * KSTEP * ESZ + 16, sc = L.gKc * KSTEP * ESZ + 16, sh = L.gKh * KSTEP * ESZ + 16;
    char* s_x = smem;
    char* s_ctx = s_x + 16 * sx;
    char* s_a = s_ctx + 16 * sc;
    char* s_b = s_a + 16 * sh;
    // rows of the fp32 images are padded by 16 bytes: with H (and 16 gTf) a multiple of 64 floats the 16 rows of a tile sat on
    // the same banks (PMC: 76 % of the LDS-active cycles were bank conflicts)
    const int HS = H + 4, PS = 16 * L.gTf + 4;
    float* s_h = reinterpret_cast<float*>(s_b + 16 * sh);          // [16][H + 4]
    float* s_par = s_h + 16 * HS;                                   // [16][16 gTf + 4]
    float* s_u = s_par + 16 * PS;                                   // [16][32] input of the current layer's conditioner
    float* s_y = s_u + 16 * 32;                                     // [16][32] the layer's other side (forward: output)
    float* s_ld = s_y + 16 * 32;                                    // [16][32] per-(row, feature) log-dets of a layer
    float* s_acc = s_ld + 16 * 32;                                  // [16] accumulated log-det, [16..32) bad flags

    auto store_act = [&](char* base, int stride, int r, int k, float v) {
        if constexpr (BF16) reinterpret_cast<__bf16*>(base + r * stride)[k] = (__bf16)v;
        else reinterpret_cast<float*>(base + r * stride)[k] = v;
    };
    // out^T[16 t + 4 g + e][row c] = sum_k W[16 t + r][k] act[row][k]: fragments [tile][k-step][lane], B rows from LDS
    // nks_row: k-steps per tile row in the fragment array; nks <= nks_row: the k-steps evaluated (sorted plans skip the rest)
    auto mm = [&](const gu32x4* fr, int tile, int nks_row, int nks, const char* act, int stride) -> f32x4 {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const gu32x4* a = fr + (size_t)tile * nks_row * 64 + lane;
        const char* brow = act + c * stride + g * 16;
        int ks = 0;
        for (; ks + 4 <= nks; ks += 4) {                      // four fragment loads in flight
            gu32x4 av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = a[(size_t)(ks + j) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const gu32x4 b = *reinterpret_cast<const gu32x4*>(brow + (size_t)(ks + j) * KSTEP * ESZ);
                if constexpr (BF16) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[j]), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
                } else {
                    const f32x4 af = __builtin_bit_cast(f32x4, av[j]), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc, 0, 0, 0);
                }
            }
        }
        for (; ks < nks; ++ks) {
            const gu32x4 av = a[(size_t)ks * 64];
            const gu32x4 b = *reinterpret_cast<const gu32x4*>(brow + (size_t)ks * KSTEP * ESZ);
            if constexpr (BF16) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
            } else {
                const f32x4 af = __builtin_bit_cast(f32x4, av), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc, 0, 0, 0);
            }
        }
        return acc;
    };
    // sorted plans (hidden units in degree order, csrc/pf_pack.hip): unit p has degree deg_at(p); the units of degree <= d are the
    // first cnt_le(d) positions.  A hidden -> hidden tile needs the columns of degree <= its largest row degree (nflows mask
    // deg_out >= deg_in), an output tile those of degree <= its last feature index (deg_out = f + 1 > deg_in)
    auto cnt_le = [&](int d) {
        if (d >= D - 1) return H;
        if (d <= 0) return 0;
        const int full = H / (D - 1), rem = H % (D - 1);
        return full * d + (rem < d ? rem : d);
    };
    auto ksteps_of = [&](int cols) { const int n = (cols + KSTEP - 1) / KSTEP; return n < L.gKh ? n : L.gKh; };
    auto kmax_hidden = [&](int t) {
        if (!L.gsorted) return L.gKh;
        int d = 1;
        while (d < D - 1 && cnt_le(d) <= 16 * t + 15) ++d;
        return ksteps_of(cnt_le(d));
    };
    auto kmax_out = [&](int t) {
        if (!L.gsorted) return L.gKh;
        int f = (16 * t + 15) / M;
        if (f > D - 1) f = D - 1;
        return ksteps_of(cnt_le(f));
    };

    // ---- context rows (once), zero padding of every operand image ---------------------------------------------------------
    for (int i = tid; i < 16 * (sx + sc + 2 * sh) / 4; i += kGenWaves * 64) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    __syncthreads();
    if (C > 0) {
        const int64_t rpc = p.ctx_rows > 0 ? p.batch / p.ctx_rows : 1;             // rows per context row (inverse: grouped)
        for (int i = tid; i < 16 * C; i += kGenWaves * 64) {
            const int r = i / C, k = i - r * C;
            const int64_t row = row0 + r < p.batch ? row0 + r : p.batch - 1;
            store_act(s_ctx, sc, r, k, p.ctx[(row / rpc) * C + k]);
        }
    }
    // layer input / running value: forward x[:, ar_perm]; inverse z
    if (tid < 16 * 32) {
        const int r = tid >> 5, d = tid & 31;
        const int64_t row = row0 + r < p.batch ? row0 + r : p.batch - 1;
        float v = 0.f;
        if (d < D) {
            if constexpr (!INV) v = p.x[row * D + (p.ar_perm ? p.ar_perm[d] : d)];
            else v = p.x[row * D + d];
        }
        s_y[tid] = v;
        if (tid < 32) s_acc[tid] = 0.f;
    }
    __syncthreads();

    const int layer_frags = L.gen_layer_frags(), layer_bias = L.gen_layer_bias();
    const int xh = L.gen_xh();
    // conditioner of one layer: s_u (fp32 [16][32], the layer's input order) -> s_par
    auto conditioner = [&](int l) {
        const gu32x4* fr = reinterpret_cast<const gu32x4*>(p.packed) + (size_t)l * layer_frags * 64;
        const float* bias = reinterpret_cast<const float*>(p.packed + L.weightBytes) + (size_t)l * layer_bias;
        const gu32x4* f_in = fr;
        const gu32x4* f_c = f_in + (size_t)NT * L.gKx * 64;
        const gu32x4* f_g0 = f_c + (size_t)(C > 0 ? NT * L.gKc : 0) * 64;
        const gu32x4* f_g1 = f_g0 + (size_t)(C > 0 ? NT * L.gKc : 0) * 64;
        const gu32x4* f_blk = f_g1 + (size_t)(C > 0 ? NT * L.gKc : 0) * 64;      // W1_0, W2_0, W1_1, W2_1
        const gu32x4* f_out = f_blk + (size_t)4 * NT * L.gKh * 64;
        const float* b_in = bias;
        const float* b_c = b_in + H;
        const float* b_g0 = b_c + (C > 0 ? H : 0);
        const float* b_g1 = b_g0 + (C > 0 ? H : 0);
        const float* b_blk = b_g1 + (C > 0 ? H : 0);                              // b1_0, b2_0, b1_1, b2_1
        const float* b_out = b_blk + 4 * H;
        // x operand: bf16 hi | lo halves, fp32 as is
        if (tid < 16 * 32) {
            const int r = tid >> 5, d = tid & 31;
            if (d < D) {
                const float v = s_u[r * 32 + d];
                if constexpr (BF16) {
                    const __bf16 hi = (__bf16)v;
                    reinterpret_cast<__bf16*>(s_x + r * sx)[d] = hi;
                    reinterpret_cast<__bf16*>(s_x + r * sx)[xh + d] = (__bf16)(v - (float)hi);
                } else {
                    reinterpret_cast<float*>(s_x + r * sx)[d] = v;
                }
            }
        }
        __syncthreads();
        // h = W_in x + b (+ relu(Wc ctx + bc)); s_a = relu(h)
        for (int t = wave; t < NT; t += kGenWaves) {
            f32x4 v = mm(f_in, t, L.gKx, L.gKx, s_x, sx) + *reinterpret_cast<const f32x4*>(b_in + 16 * t + 4 * g);
            f32x4 pcv = {0.f, 0.f, 0.f, 0.f};
            if (C > 0) {
                pcv = mm(f_c, t, L.gKc, L.gKc, s_ctx, sc) + *reinterpret_cast<const f32x4*>(b_c + 16 * t + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += fmaxf(pcv[e], 0.f);
            }
            if constexpr (REEVAL) {
                if (row0 + c < p.batch) {
                    const size_t o = ((size_t)l * p.batch + row0 + c) * H + 16 * t + 4 * g;
                    *reinterpret_cast<f32x4*>(sink.hs + o) = v;                                   // hs[0]: state before block 0
                    if (C > 0) *reinterpret_cast<f32x4*>(sink.pc + o) = pcv;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s_h[c * HS + 16 * t + 4 * g + e] = v[e];
                store_act(s_a, sh, c, 16 * t + 4 * g + e, fmaxf(v[e], 0.f));
            }
        }
        __syncthreads();
        for (int b = 0; b < 2; ++b) {
            const gu32x4* f1 = f_blk + (size_t)(2 * b) * NT * L.gKh * 64;
            const gu32x4* f2 = f1 + (size_t)NT * L.gKh * 64;
            for (int t = wave; t < NT; t += kGenWaves) {                           // t1 = W1 relu(h) + b1; s_b = relu(t1)
                const int kh = kmax_hidden(t);
                f32x4 v = mm(f1, t, L.gKh, kh, s_a, sh) + *reinterpret_cast<const f32x4*>(b_blk + (2 * b) * H + 16 * t + 4 * g);
                if (additive)          // t1 = W1 relu(h) + b1 + context_layer(ctx)   (flows.py:225-234)
                    v = v + mm(b == 0 ? f_g0 : f_g1, t, L.gKc, L.gKc, s_ctx, sc) + *reinterpret_cast<const f32x4*>((b == 0 ? b_g0 : b_g1) + 16 * t + 4 * g);
                f32x4 df = {1.f, 1.f, 1.f, 1.f};
                if constexpr (REEVAL) {
                    if (row0 + c < p.batch) {
                        const size_t o = (((size_t)b * L.L + l) * p.batch + row0 + c) * H + 16 * t + 4 * g;
                        *reinterpret_cast<f32x4*>(sink.t1s + o) = v;
                        if (sink.drop) df = *reinterpret_cast<const f32x4*>(sink.drop + o);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) store_act(s_b, sh, c, 16 * t + 4 * g + e, fmaxf(v[e], 0.f) * df[e]);
            }
            __syncthreads();
            for (int t = wave; t < NT; t += kGenWaves) {                           // h += (W2 . + b2) . sigmoid(Wg ctx + bg)
                f32x4 v = mm(f2, t, L.gKh, kmax_hidden(t), s_b, sh) + *reinterpret_cast<const f32x4*>(b_blk + (2 * b + 1) * H + 16 * t + 4 * g);
                const bool live = REEVAL && row0 + c < p.batch;
                const size_t o = (((size_t)b * L.L + l) * p.batch + row0 + c) * H + 16 * t + 4 * g;
                if (additive) {
                    if (live) *reinterpret_cast<f32x4*>(sink.t2s + o) = v;
                } else if (C > 0) {
                    f32x4 gt = mm(b == 0 ? f_g0 : f_g1, t, L.gKc, L.gKc, s_ctx, sc) +
                               *reinterpret_cast<const f32x4*>((b == 0 ? b_g0 : b_g1) + 16 * t + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gt[e] = pf_sigmoid<FAST>(gt[e]);
                    if (live) { *reinterpret_cast<f32x4*>(sink.t2s + o) = v; *reinterpret_cast<f32x4*>(sink.gates + o) = gt; }
                    v = v * gt;
                }
                f32x4 hv4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hv = s_h[c * HS + 16 * t + 4 * g + e] + v[e];
                    hv4[e] = hv;
                    s_h[c * HS + 16 * t + 4 * g + e] = hv;
                    store_act(s_a, sh, c, 16 * t + 4 * g + e, b == 0 ? fmaxf(hv, 0.f) : hv);   // the final layer takes h itself
                }
                if (live) {
                    if (b == 0) *reinterpret_cast<f32x4*>(sink.hs + (((size_t)1 * L.L + l) * p.batch + row0 + c) * H + 16 * t + 4 * g) = hv4;
                    else *reinterpret_cast<f32x4*>(sink.h2 + ((size_t)l * p.batch + row0 + c) * H + 16 * t + 4 * g) = hv4;
                }
            }
            __syncthreads();
        }
        for (int t = wave; t < L.gTf; t += kGenWaves) {
            const f32x4 v = mm(f_out, t, L.gKh, kmax_out(t), s_a, sh) + *reinterpret_cast<const f32x4*>(b_out + 16 * t + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) s_par[c * PS + 16 * t + 4 * g + e] = v[e];
        }
        __syncthreads();
    };

    const int n_pairs = 16 * D;
    if constexpr (REEVAL) {
        const int l = blockIdx.y;
        if (tid < 16 * 32) {
            const int r = tid >> 5, d = tid & 31;
            const int64_t row = row0 + r < p.batch ? row0 + r : p.batch - 1;
            s_u[tid] = d < D ? p.x[((size_t)l * p.batch + row) * D + d] : 0.f;          // p.x = U[L][B][D]
        }
        __syncthreads();
        conditioner(l);
        const int DM = D * M;
        for (int i = tid; i < 16 * DM; i += kGenWaves * 64) {
            const int r = i / DM, k = i - r * DM;
            if (row0 + r < p.batch) sink.params[((size_t)l * p.batch + row0 + r) * DM + k] = s_par[r * PS + k];
        }
    } else if constexpr (!INV) {
        for (int l = 0; l < L.L; ++l) {
            // ReversePermutation in front of every autoregressive layer (flows.py:459-529)
            if (tid < 16 * 32) { const int r = tid >> 5, d = tid & 31; s_u[tid] = d < D ? s_y[r * 32 + (additive ? d : D - 1 - d)] : 0.f; }
            __syncthreads();
            if (p.u_save && tid < n_pairs) {
                const int r = tid / D, d = tid - r * D;
                if (row0 + r < p.batch) p.u_save[((size_t)l * p.batch + row0 + r) * D + d] = s_u[r * 32 + d];
            }
            conditioner(l);
            if (tid < n_pairs) {
                const int r = tid & 15, f = tid >> 4;
                float y, ld; bool bad;
                rqs_generic<FAST, false>(s_par + r * PS + f * M, s_u[r * 32 + f], K, p, y, ld, bad);
                s_y[r * 32 + f] = y;
                s_ld[r * 32 + f] = ld;
            }
            __syncthreads();
            if (tid < 16) { float a = s_acc[tid]; for (int f = 0; f < D; ++f) a += s_ld[tid * 32 + f]; s_acc[tid] = a; }
            __syncthreads();
        }
        float my_nll = 0.f, my_cnt = 0.f;
        if (p.zero_pair && blockIdx.x == 0 && tid < 2 * PF_REDUCE_SLOTS) p.zero_pair[tid] = 0.f;
        if (tid < 16 && row0 + tid < p.batch) {
            const int64_t row = row0 + tid;
            float q = 0.f, sls = 0.f;
            for (int d = 0; d < D; ++d) {
                const float zv = s_y[tid * 32 + d];
                if (p.log_sigma) { const float ls = p.log_sigma[row * D + d]; const float zs = zv / expf(ls); q += zs * zs; sls += ls; }
                else q += zv * zv;
                if (p.z) p.z[row * D + d] = zv;
            }
            const float ld = s_acc[tid];
            if (p.logdet) p.logdet[row] = ld;
            my_nll = 0.5f * (q + 2.f * sls + (float)D * 1.8378770664093453f) - ld;
            my_cnt = 1.f;
            if (p.nll) p.nll[row] = my_nll;
        }
        if (p.nll_sum && tid < 64) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { my_nll += __shfl_xor(my_nll, o, 64); my_cnt += __shfl_xor(my_cnt, o, 64); }
            if (tid == 0) { float* acc = p.nll_sum + 2 * (blockIdx.x % PF_REDUCE_SLOTS); atomicAdd(acc, my_nll); atomicAdd(acc + 1, my_cnt); }
        }
    } else {
        // inverse: s_y holds the value to invert (the layer's OUTPUT side); D passes over the running estimate in s_u
        for (int l = L.L - 1; l >= 0; --l) {
            if (tid < 16 * 32) s_u[tid] = 0.f;
            __syncthreads();
            for (int pass = 0; pass < D; ++pass) {
                conditioner(l);
                if (tid < n_pairs) {
                    const int r = tid & 15, f = tid >> 4;
                    float x, ld; bool bad;
                    rqs_generic<FAST, true>(s_par + r * PS + f * M, s_y[r * 32 + f], K, p, x, ld, bad);
                    s_u[r * 32 + f] = x;
                    s_ld[r * 32 + f] = ld;
                    if (bad && pass == D - 1) s_acc[16 + r] = 1.f;
                }
                __syncthreads();
            }
            if (tid < 16) { float a = s_acc[tid]; for (int f = 0; f < D; ++f) a += s_ld[tid * 32 + f]; s_acc[tid] = a; }
            // undo the ReversePermutation that preceded this layer
            if (tid < 16 * 32) { const int r = tid >> 5, d = tid & 31; s_y[tid] = d < D ? s_u[r * 32 + (additive ? d : D - 1 - d)] : 0.f; }
            __syncthreads();
        }
        if (tid < 16 && row0 + tid < p.batch) {
            const int64_t row = row0 + tid;
            for (int d = 0; d < D; ++d) {
                const int src = p.ar_perm ? p.ar_perm[d] : d;                        // ar_inv_perm
                if (p.z) p.z[row * D + d] = s_y[tid * 32 + src];
            }
            if (p.logdet) p.logdet[row] = s_acc[tid];
            if (p.fail_flags && s_acc[16 + tid] != 0.f) atomicOr(p.fail_flags + row, 1u);
        }
    }
}

template <bool BF16, int MODE>
int launch_generic_t(const FwdParams& p, const ReevalSink& sink, hipStream_t s) {
    const size_t lds = (size_t)p.plan.gen_lds_bytes();
    auto k = flow_generic_kernel<BF16, MODE>;
    if (lds > 160 * 1024) return PF_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
    const unsigned grid = (unsigned)((p.batch + 15) / 16);
    if (grid == 0) return PF_OK;
    hipLaunchKernelGGL(k, dim3(grid, MODE == 2 ? (unsigned)p.plan.L : 1u), dim3(kGenWaves * 64), lds, s, p, sink);
    return launch_status();
}

}  // namespace

int launch_flow_generic(const FwdParams& p, bool inverse, hipStream_t s) {
    const ReevalSink none{};
    if (p.plan.bf16) return inverse ? launch_generic_t<true, 1>(p, none, s) : launch_generic_t<true, 0>(p, none, s);
    return inverse ? launch_generic_t<false, 1>(p, none, s) : launch_generic_t<false, 0>(p, none, s);
}

// fp32 conditioner re-evaluation of every layer (pf_flow_reevaluate with an fp32 desc): L = the generic plan of the desc
int flow_reevaluate_generic(const FlowPlan& L, const PfFlowReevalArgs& a, const PfFlowDesc& d, hipStream_t s) {
    if (L.bf16 || !L.generic) return PF_ERR_UNSUPPORTED;
    FwdParams p{};
    p.packed = static_cast<const char*>(a.packed);
    p.x = a.U; p.ctx = a.ctx; p.batch = a.batch; p.ctx_rows = a.batch; p.plan = L;
    p.tail_bound = d.tail_bound; p.min_w = d.min_bin_width; p.min_h = d.min_bin_height; p.min_d = d.min_derivative;
    ReevalSink sink{a.hs, a.t1s, a.t2s, a.gates, a.pc, a.h2, a.params, a.drop};
    return launch_generic_t<false, 2>(p, sink, s);
}

}  // namespace pf
