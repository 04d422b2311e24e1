// pf_enc_train.hip -- the strain embedding's TRAINING path: forward that keeps what the backward needs, and the backward,
// each ONE C call that enqueues every kernel on the caller's stream (no Python between launches, no autograd graph).
//
// Replaces, for a differentiable / train()-mode call of LeanStrainEncoder._compute_feats (src/ahsd/models/lean_npe.py:
// 199-233) and what autograd records under experiments/train_lean_npe.py:363-364:
//   stem (asinh, 4 strided Conv1d + GELU)            pf_embed.hip conv kernels (+ gelu' and the asinh signal kept)
//   + positional / detector embedding, extra tokens  tok_assemble
//   3 x pre-norm TransformerEncoderLayer (dropout)   ln_* / dense_nt / attn_* kernels below, dropout by counter hash
//   pool attention (K / V side)                      dense_nt + pool_*
// and their gradients w.r.t. every parameter, the extra (geometry) tokens, the token bias and the projected pool queries.
// The three small MLPs, the pool's query / output projections and the embeddings themselves stay host-side tensor ops
// (M = batch rows; plain library GEMMs).
//
// Raw parameter layout (fp32, one flat buffer; gradients come back in the same layout):
//   stem.{0,2,4,6}.{weight [cout][cin][kw], bias}
//   per layer l = 0..2: norm1.{weight,bias}, self_attn.in_proj_{weight [576][192], bias}, self_attn.out_proj.{weight [192][192], bias},
//                       norm2.{weight,bias}, linear1.{weight [768][192], bias}, linear2.{weight [192][768], bias}
//   pool_attn.in_proj_weight [576][192] (whole; rows 0..191 = the query projection are not read, their gradient is zero),
//   pool_attn.in_proj_bias [576]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/pf_hip.h"
#include "pf_dense.h"
#include "pf_enc_ops.h"
#include "pf_status.h"

namespace pf {

int stem_forward_train(bool bf16, const void* const wfrags[4], const float* const bias[4], const float* strain, int64_t n_seq,
                       void* sig, void* const act[3], void* const dact[4], float* tokens, float* log_energy, hipStream_t s);

namespace {

struct Conv { int cin, cout, kw, s, lin, lout; };
constexpr Conv kConv[4] = {{1, 32, 64, 8, 16384, 2041}, {32, 64, 16, 4, 2041, 507}, {64, 128, 8, 4, 507, 125}, {128, 192, 4, 2, 125, 61}};
constexpr int kTok = 61;

// ---- raw parameter offsets ------------------------------------------------------------------------------------------
struct RawLayout {
    int64_t conv_w[4], conv_b[4];
    struct Layer { int64_t n1w, n1b, inw, inb, ow, ob, n2w, n2b, w1, b1, w2, b2; } L[kEncLayers];
    int64_t pool_w, pool_b, total;
};
RawLayout raw_layout() {
    RawLayout r{};
    int64_t o = 0;
    for (int l = 0; l < 4; ++l) {
        r.conv_w[l] = o; o += (int64_t)kConv[l].cout * kConv[l].cin * kConv[l].kw;
        r.conv_b[l] = o; o += kConv[l].cout;
    }
    for (int l = 0; l < kEncLayers; ++l) {
        auto& L = r.L[l];
        L.n1w = o; o += kEncD; L.n1b = o; o += kEncD;
        L.inw = o; o += 3 * kEncD * kEncD; L.inb = o; o += 3 * kEncD;
        L.ow = o; o += kEncD * kEncD; L.ob = o; o += kEncD;
        L.n2w = o; o += kEncD; L.n2b = o; o += kEncD;
        L.w1 = o; o += kEncFF * kEncD; L.b1 = o; o += kEncFF;
        L.w2 = o; o += kEncD * kEncFF; L.b2 = o; o += kEncD;
    }
    r.pool_w = o; o += 3 * kEncD * kEncD;
    r.pool_b = o; o += 3 * kEncD;
    r.total = o;
    return r;
}

// ---- packed fragments -------------------------------------------------------------------------------------------------
struct PackLayout {
    int64_t conv_fwd[4], conv_dx[4];                       // conv_dx[0] unused
    struct Layer { int64_t in_f, in_t, o_f, o_t, w1_f, w1_t, w2_f, w2_t; } L[kEncLayers];
    int64_t kv_f, kv_t, total;                             // in units of 16 bytes
    DensePackTable tab;
};
PackLayout pack_layout(bool bf16) {
    PackLayout p{};
    const RawLayout r = raw_layout();
    int64_t o = 0;
    int n = 0;
    auto add = [&](int mode, int64_t src, int ld, int N, int K, const Conv* c) -> int64_t {
        DensePackEntry& e = p.tab.e[n++];
        e.src_off = src; e.dst_off = o; e.mode = mode; e.ld = ld; e.N = N; e.K = K;
        e.cin = c ? c->cin : 0; e.cout = c ? c->cout : 0; e.kw = c ? c->kw : 0; e.s = c ? c->s : 0;
        const int64_t at = o;
        o += dense_frag_count(bf16, N, K);
        return at;
    };
    for (int l = 0; l < 4; ++l) {
        const Conv& c = kConv[l];
        p.conv_fwd[l] = add(3, r.conv_w[l], 0, c.cout, c.kw * c.cin, &c);
        if (l > 0) p.conv_dx[l] = add(2, r.conv_w[l], 0, c.s * c.cin, (c.kw / c.s) * c.cout, &c);
    }
    for (int l = 0; l < kEncLayers; ++l) {
        auto& L = p.L[l];
        const auto& R = r.L[l];
        L.in_f = add(0, R.inw, kEncD, 3 * kEncD, kEncD, nullptr);  L.in_t = add(1, R.inw, kEncD, kEncD, 3 * kEncD, nullptr);
        L.o_f = add(0, R.ow, kEncD, kEncD, kEncD, nullptr);        L.o_t = add(1, R.ow, kEncD, kEncD, kEncD, nullptr);
        L.w1_f = add(0, R.w1, kEncD, kEncFF, kEncD, nullptr);      L.w1_t = add(1, R.w1, kEncD, kEncD, kEncFF, nullptr);
        L.w2_f = add(0, R.w2, kEncFF, kEncD, kEncFF, nullptr);     L.w2_t = add(1, R.w2, kEncFF, kEncFF, kEncD, nullptr);
    }
    p.kv_f = add(0, r.pool_w + (int64_t)kEncD * kEncD, kEncD, 2 * kEncD, kEncD, nullptr);
    p.kv_t = add(1, r.pool_w + (int64_t)kEncD * kEncD, kEncD, kEncD, 2 * kEncD, nullptr);
    p.tab.n = n;
    p.total = o;
    return p;
}

// ---- workspace ----------------------------------------------------------------------------------------------------------
struct Ws {
    int64_t sig, act[3], dact[4], stem_tok;                     // stem
    struct Layer { int64_t x, mean1, rstd1, y1, qkv, lse, o, xmid, mean2, rstd2, y2, hd, gd; } L[kEncLayers];
    int64_t x3, x3a, kv;
    // backward temporaries
    int64_t dxa, dxb, g192a[kEncLayers], g192b[kEncLayers], g768[kEncLayers], dqkv[kEncLayers], dkv, dy, gpad[4], g1, conv_dw, total;
};
struct Dims { int64_t B, N, R; int T, D, n_extra; bool bf16; int esz; bool fwd_only; };
constexpr int kGpadRows[4] = {0, 514, 128, 64};                // rows per sequence of the padded gradient images of conv2..4
constexpr int kGpadOff[4] = {0, 3, 1, 1};                      // kw / s - 1 leading zero rows
constexpr int kDxRows[4] = {0, 511, 127, 63};                  // ceil(lin / s): rows of the transposed-convolution GEMM

Ws ws_layout(const Dims& d) {
    Ws w{};
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += (bytes + 255) & ~(int64_t)255; return at; };
    const int64_t e = d.esz;
    w.sig = take(d.N * 16384 * e);
    for (int l = 0; l < 3; ++l) w.act[l] = take(d.N * kConv[l].lout * kConv[l].cout * e);
    for (int l = 0; l < 4; ++l) w.dact[l] = take(d.N * kConv[l].lout * kConv[l].cout * e);
    w.stem_tok = take(d.N * kTok * kEncD * 4);
    for (int l = 0; l < kEncLayers; ++l) {
        auto& L = w.L[l];
        // forward-only: the layers share one set of buffers.  Safe in the forward's order: a layer's input x is last read
        // by the out-projection's residual epilogue (-> xmid), FFN2 then writes the next layer's input over it
        if (d.fwd_only && l > 0) { L = w.L[0]; continue; }
        L.x = take(d.R * kEncD * 4); L.mean1 = take(d.R * 4); L.rstd1 = take(d.R * 4);
        L.y1 = take(d.R * kEncD * e); L.qkv = take(d.R * 3 * kEncD * e); L.lse = take(d.B * kEncHeads * d.T * 4);
        L.o = take(d.R * kEncD * e); L.xmid = take(d.R * kEncD * 4); L.mean2 = take(d.R * 4); L.rstd2 = take(d.R * 4);
        L.y2 = take(d.R * kEncD * e); L.hd = take(d.R * kEncFF * e); L.gd = take(d.R * kEncFF * e);
    }
    w.x3 = take(d.R * kEncD * 4);
    w.x3a = d.bf16 ? take(d.R * kEncD * e) : w.x3;
    w.kv = take(d.R * 2 * kEncD * e);
    if (d.fwd_only) { w.total = o; return w; }             // none of the backward's temporaries
    w.dxa = take(d.R * kEncD * 4); w.dxb = take(d.R * kEncD * 4);
    for (int l = 0; l < kEncLayers; ++l) {       // per layer: the weight-gradient GEMMs read them on a second stream while
        w.g192a[l] = take(d.R * kEncD * e); w.g192b[l] = take(d.R * kEncD * e);      // the chain moves on to the layer below
        w.g768[l] = take(d.R * kEncFF * e); w.dqkv[l] = take(d.R * 3 * kEncD * e);
    }
    w.dkv = take(d.R * 2 * kEncD * e);
    w.dy = take(d.R * kEncD * e);
    for (int l = 1; l < 4; ++l) w.gpad[l] = take(d.N * kGpadRows[l] * kConv[l].cout * e);
    w.g1 = take(d.N * kConv[0].lout * kConv[0].cout * e);
    w.conv_dw = take((int64_t)(32 * 64 + 64 * 512 + 128 * 512 + 192 * 512) * 4);   // conv weight gradients in im2col order
    w.total = o;
    return w;
}

int dims_of(const PfEmbedTrainDesc* desc, int64_t n_events, Dims& d) {
    if (!desc || n_events < 0) return PF_ERR_BAD_ARG;
    if (desc->precision != PF_PREC_F32 && desc->precision != PF_PREC_BF16) return PF_ERR_BAD_ARG;
    if (desc->n_detectors < 1 || desc->n_detectors > 3 || desc->n_extra_tokens < 0) return PF_ERR_BAD_ARG;
    if (!(desc->dropout_p >= 0.f && desc->dropout_p < 1.f)) return PF_ERR_BAD_ARG;
    d.B = n_events; d.D = desc->n_detectors; d.n_extra = desc->n_extra_tokens;
    d.T = d.n_extra + kTok * d.D;
    if (d.T > kEncMaxTokens) return PF_ERR_UNSUPPORTED;
    d.N = d.B * d.D; d.R = d.B * d.T;
    d.bf16 = desc->precision == PF_PREC_BF16; d.esz = d.bf16 ? 2 : 4;
    d.fwd_only = desc->forward_only != 0;
    return PF_OK;
}

#define PF_TRY(x) do { const int rc_ = (x); if (rc_ != PF_OK) return rc_; } while (0)

// The weight-gradient GEMMs are off the backward's critical path (nothing downstream reads dW) and individually latency-bound
// (1 - 2 workgroups per CU): they run on a second stream, forked from and joined back into the caller's stream with events,
// under the data-gradient chain.  One side stream and a small ring of events per host thread and device, created on first use.
struct SideStream {
    static constexpr int kEvents = 32;
    hipStream_t stream = nullptr;
    hipEvent_t ev[kEvents];
    int device = -1, next = 0;
    int ensure() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hip_failed(hipGetLastError());
        if (stream && dev == device) return PF_OK;
        if (stream) return PF_ERR_UNSUPPORTED;            // one device per host thread (one process per GPU)
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return hip_failed(hipGetLastError());
        for (int i = 0; i < kEvents; ++i)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return hip_failed(hipGetLastError());
        device = dev;
        return PF_OK;
    }
    // everything enqueued on `from` so far happens before what is enqueued on `to` from now on
    int order(hipStream_t from, hipStream_t to) {
        hipEvent_t e = ev[next];
        next = (next + 1) % kEvents;
        if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) return hip_failed(hipGetLastError());
        return PF_OK;
    }
};
thread_local SideStream g_side;

// a plain [R][K] x W^T GEMM
int linear(const Dims& d, int epi, const void* A, int K, int N, const void* frags, const float* bias, void* out, hipStream_t s,
           void* dact = nullptr, const float* resid = nullptr, const void* mul = nullptr, float drop_p = 0.f, uint32_t seed = 0,
           uint32_t site = 0, bool out_f32 = false) {
    DenseArgs a{};
    a.A = A; a.M = d.R; a.rows_per_seq = d.R > 0 ? d.R : 1; a.a_seq_stride = 0; a.lda = K;
    a.K = K; a.N = N; a.KC = K <= 256 ? K : 192;
    a.wfrags = frags; a.bias = bias;
    a.out = out; a.o_seq_stride = 0; a.ldo = N; a.o_valid_per_seq = 0; a.x_seq_stride = 0;
    a.dact = dact; a.resid = resid; a.mul = mul; a.drop_p = drop_p; a.seed = seed; a.site = site; a.out_f32 = out_f32 ? 1 : 0;
    return dense_nt(d.bf16, epi, a, s);
}
int weight_grad(const Dims& d, const void* G, int N1, const void* A, int N2, float* dW, float* db, hipStream_t s) {
    DenseTnArgs a{};
    a.G = G; a.g_seq_stride = 0; a.ldg = N1; a.A = A; a.a_seq_stride = 0; a.lda = N2;
    a.M = d.R; a.rows_per_seq = d.R > 0 ? d.R : 1; a.N1 = N1; a.N2 = N2; a.dW = dW; a.ldw = N2; a.db = db; a.splits = 0;
    return dense_tn(d.bf16, a, s);
}

__global__ __launch_bounds__(256) void drop_cast_kernel(const float* __restrict__ src, void* __restrict__ dst, int64_t n4, int bf16,
                                                        float p, uint32_t seed, uint32_t site) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const uint32_t thr = enc_drop_threshold(p);
    const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
    // 4 pieces per thread and iteration, loads first
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < n4) v[u] = *reinterpret_cast<const f32x4*>(src + 4 * i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n4) break;
            if (p > 0.f) {
                f32x4 fac;
                enc_drop4(seed, site, (uint32_t)(4 * i), thr, sc, fac);
                v[u] = v[u] * fac;
            }
            if (bf16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[u][e];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dst) + 4 * i) = o;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dst) + 4 * i) = v[u];
            }
        }
    }
}

// Conv1d weight gradient from the im2col order the GEMM produces ([cout][tap * cin + ch], contiguous float atomics) to the
// parameter's own [cout][cin][kw]: atomics scattered straight into that layout (16 lanes on 16 different 64-byte segments)
// ran at a twentieth of the contiguous rate -- 0.8 ms per convolution
__global__ __launch_bounds__(256) void conv_dw_permute_kernel(const float* __restrict__ tmp, float* __restrict__ out, int cout, int cin, int kw) {
    const int n = cout * cin * kw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int tap = i % kw, ci = (i / kw) % cin, co = i / (kw * cin);
        out[i] = tmp[co * (kw * cin) + tap * cin + ci];
    }
}

}  // namespace

int64_t enc_train_raw_count() { return raw_layout().total; }
int64_t enc_train_packed_bytes(bool bf16) { return pack_layout(bf16).total * 16; }
int enc_train_pack(bool bf16, const float* raw, void* packed, hipStream_t s) {
    const PackLayout p = pack_layout(bf16);
    return dense_pack(bf16, raw, p.tab, packed, s);
}
int64_t enc_train_workspace_bytes(const PfEmbedTrainDesc* desc, int64_t n_events) {
    Dims d;
    if (dims_of(desc, n_events, d) != PF_OK) return -1;
    return ws_layout(d).total + 256;
}

int enc_train_forward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* strain,
                      const float* extra_tokens, const float* token_bias, const float* pool_q, int64_t n_events, float* pooled,
                      float* log_energy, void* workspace, hipStream_t s) {
    Dims d;
    PF_TRY(dims_of(desc, n_events, d));
    if (d.B == 0) return PF_OK;
    const RawLayout r = raw_layout();
    const PackLayout pk = pack_layout(d.bf16);
    const Ws w = ws_layout(d);
    char* ws = reinterpret_cast<char*>(workspace);
    const dn_u32x4* fr = reinterpret_cast<const dn_u32x4*>(packed);
    const float p = desc->training ? desc->dropout_p : 0.f;
    const uint32_t seed = (uint32_t)(desc->dropout_seed ^ (desc->dropout_seed >> 32));

    // ---- stem ---------------------------------------------------------------------------------------------------------
    const void* cw[4]; const float* cb[4]; void* act[3]; void* dact[4];
    for (int l = 0; l < 4; ++l) { cw[l] = fr + pk.conv_fwd[l]; cb[l] = raw + r.conv_b[l]; dact[l] = ws + w.dact[l]; }
    for (int l = 0; l < 3; ++l) act[l] = ws + w.act[l];
    float* stem_tok = reinterpret_cast<float*>(ws + w.stem_tok);
    PF_TRY(stem_forward_train(d.bf16, cw, cb, strain, d.N, ws + w.sig, act, dact, stem_tok, log_energy, s));
    PF_TRY(tok_assemble(stem_tok, extra_tokens, token_bias, d.B, d.n_extra, kTok * d.D, reinterpret_cast<float*>(ws + w.L[0].x), s));

    // ---- token mixer ----------------------------------------------------------------------------------------------------
    for (int l = 0; l < kEncLayers; ++l) {
        const auto& L = w.L[l];
        const auto& R = r.L[l];
        const auto& P = pk.L[l];
        float* x = reinterpret_cast<float*>(ws + L.x);
        float* xmid = reinterpret_cast<float*>(ws + L.xmid);
        float* xnext = reinterpret_cast<float*>(ws + (l + 1 < kEncLayers ? w.L[l + 1].x : w.x3));
        LnArgs ln{};
        ln.x = x; ln.gamma = raw + R.n1w; ln.beta = raw + R.n1b; ln.M = d.R; ln.y = ws + L.y1;
        ln.mean = reinterpret_cast<float*>(ws + L.mean1); ln.rstd = reinterpret_cast<float*>(ws + L.rstd1);
        PF_TRY(ln_forward(d.bf16, ln, s));
        PF_TRY(linear(d, kEpiPlain, ws + L.y1, kEncD, 3 * kEncD, fr + P.in_f, raw + R.inb, ws + L.qkv, s));
        AttnArgs at{};
        at.qkv = ws + L.qkv; at.B = d.B; at.T = d.T; at.out = ws + L.o; at.lse = reinterpret_cast<float*>(ws + L.lse);
        at.drop_p = p; at.seed = seed; at.site = 4 * l + 0;
        PF_TRY(attn_forward(d.bf16, at, s));
        PF_TRY(linear(d, kEpiResid, ws + L.o, kEncD, kEncD, fr + P.o_f, raw + R.ob, xmid, s, nullptr, x, nullptr, p, seed, 4 * l + 1));
        ln.x = xmid; ln.gamma = raw + R.n2w; ln.beta = raw + R.n2b; ln.y = ws + L.y2;
        ln.mean = reinterpret_cast<float*>(ws + L.mean2); ln.rstd = reinterpret_cast<float*>(ws + L.rstd2);
        PF_TRY(ln_forward(d.bf16, ln, s));
        PF_TRY(linear(d, kEpiGelu, ws + L.y2, kEncD, kEncFF, fr + P.w1_f, raw + R.b1, ws + L.hd, s, ws + L.gd, nullptr, nullptr, p, seed,
                      4 * l + 2));
        PF_TRY(linear(d, kEpiResid, ws + L.hd, kEncFF, kEncD, fr + P.w2_f, raw + R.b2, xnext, s, nullptr, xmid, nullptr, p, seed,
                      4 * l + 3));
    }
    // ---- pool --------------------------------------------------------------------------------------------------------------
    if (d.bf16) PF_TRY(cast_rows(true, reinterpret_cast<const float*>(ws + w.x3), ws + w.x3a, d.R * kEncD, s));
    PF_TRY(linear(d, kEpiPlain, ws + w.x3a, kEncD, 2 * kEncD, fr + pk.kv_f, raw + r.pool_b + kEncD, ws + w.kv, s));
    PoolArgs po{};
    po.kv = ws + w.kv; po.q = pool_q; po.B = d.B; po.T = d.T; po.pooled = pooled;
    return pool_forward(d.bf16, po, s);
}

int enc_train_backward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* pool_q,
                       const float* grad_pooled, int64_t n_events, void* workspace, float* grad_raw, float* grad_extra,
                       float* grad_token_bias, float* grad_pool_q, hipStream_t s) {
    Dims d;
    PF_TRY(dims_of(desc, n_events, d));
    if (d.fwd_only) return PF_ERR_BAD_ARG;                  // the forward kept one layer's activations only
    const RawLayout r = raw_layout();
    if (hipMemsetAsync(grad_raw, 0, (size_t)r.total * 4, s) != hipSuccess) return hip_failed(hipGetLastError());
    if (grad_pool_q && hipMemsetAsync(grad_pool_q, 0, kEncPoolQ * kEncD * 4, s) != hipSuccess) return hip_failed(hipGetLastError());
    if (grad_token_bias && hipMemsetAsync(grad_token_bias, 0, (size_t)d.T * kEncD * 4, s) != hipSuccess) return hip_failed(hipGetLastError());
    if (d.B == 0) return PF_OK;
    const PackLayout pk = pack_layout(d.bf16);
    const Ws w = ws_layout(d);
    char* ws = reinterpret_cast<char*>(workspace);
    const dn_u32x4* fr = reinterpret_cast<const dn_u32x4*>(packed);
    const float p = desc->training ? desc->dropout_p : 0.f;
    const uint32_t seed = (uint32_t)(desc->dropout_seed ^ (desc->dropout_seed >> 32));
    float* dxa = reinterpret_cast<float*>(ws + w.dxa);
    float* dxb = reinterpret_cast<float*>(ws + w.dxb);

    // weight-gradient GEMMs go to the side stream (PF_ENC_ONE_STREAM=1 keeps everything on the caller's stream: A/B timing)
    static const bool one_stream = std::getenv("PF_ENC_ONE_STREAM") != nullptr;
    hipStream_t side = s;
    if (!one_stream) {
        PF_TRY(g_side.ensure());
        side = g_side.stream;
        PF_TRY(g_side.order(s, side));                  // the zeroing of grad_raw above, and whatever produced our inputs
    }
    // run `tn` on the side stream once everything enqueued on the main stream so far (its operands) is done
    auto on_side = [&](auto&& tn) -> int {
        if (side != s) PF_TRY(g_side.order(s, side));
        return tn(side);
    };

    float* conv_tmp[4];
    // Everything between the fork above and the join below runs as one unit whose every exit is followed by the join: a
    // failure half-way must not return while side-stream kernels still read the workspace and write grad_raw (the caller
    // would release both to an allocator that orders reuse on ITS stream only)
    auto chain = [&]() -> int {
    // ---- pool ----------------------------------------------------------------------------------------------------------
    PoolArgs po{};
    po.kv = ws + w.kv; po.q = pool_q; po.B = d.B; po.T = d.T; po.dpooled = grad_pooled; po.dkv = ws + w.dkv;
    po.dq = grad_pool_q ? grad_pool_q : dxb;          // (dxb is scratch here when the caller does not want dq; zeroed below)
    if (!grad_pool_q && hipMemsetAsync(dxb, 0, kEncPoolQ * kEncD * 4, s) != hipSuccess) return hip_failed(hipGetLastError());
    PF_TRY(pool_backward(d.bf16, po, s));
    PF_TRY(on_side([&](hipStream_t q) {
        return weight_grad(d, ws + w.dkv, 2 * kEncD, ws + w.x3a, kEncD, grad_raw + r.pool_w + (int64_t)kEncD * kEncD,
                           grad_raw + r.pool_b + kEncD, q);
    }));
    PF_TRY(linear(d, kEpiPlain, ws + w.dkv, 2 * kEncD, kEncD, fr + pk.kv_t, nullptr, dxa, s, nullptr, nullptr, nullptr, 0.f, 0, 0, true));

    // ---- token mixer, last layer first ---------------------------------------------------------------------------------
    // entering layer l: dxa = dL/dx_{l+1} (fp32), g192a[l] = act(dxa . dropout factor of the layer's second residual branch)
    {
        const int64_t n4 = d.R * kEncD / 4;
        const unsigned grid = (unsigned)((n4 + 1023) / 1024 > 2048 ? 2048 : (n4 + 1023) / 1024);
        hipLaunchKernelGGL(drop_cast_kernel, dim3(grid), dim3(256), 0, s, dxa, ws + w.g192a[kEncLayers - 1], n4, d.bf16 ? 1 : 0, p, seed,
                           (uint32_t)(4 * (kEncLayers - 1) + 3));
        PF_TRY(launch_status());
    }
    for (int l = kEncLayers - 1; l >= 0; --l) {
        const auto& L = w.L[l];
        const auto& R = r.L[l];
        const auto& P = pk.L[l];
        char* g192a = ws + w.g192a[l];
        char* g192b = ws + w.g192b[l];
        char* g768 = ws + w.g768[l];
        char* dqkv = ws + w.dqkv[l];
        // FFN: x_{l+1} = xmid + D3 . (W2 hd + b2),  hd = D2 . gelu(W1 y2 + b1)
        PF_TRY(on_side([&](hipStream_t q) { return weight_grad(d, g192a, kEncD, ws + L.hd, kEncFF, grad_raw + R.w2, grad_raw + R.b2, q); }));
        PF_TRY(linear(d, kEpiMul, g192a, kEncD, kEncFF, fr + P.w2_t, nullptr, g768, s, nullptr, nullptr, ws + L.gd));
        PF_TRY(on_side([&](hipStream_t q) { return weight_grad(d, g768, kEncFF, ws + L.y2, kEncD, grad_raw + R.w1, grad_raw + R.b1, q); }));
        PF_TRY(linear(d, kEpiPlain, g768, kEncFF, kEncD, fr + P.w1_t, nullptr, ws + w.dy, s));
        LnArgs ln{};
        ln.x = reinterpret_cast<const float*>(ws + L.xmid); ln.gamma = raw + R.n2w; ln.M = d.R;
        ln.mean = reinterpret_cast<float*>(ws + L.mean2); ln.rstd = reinterpret_cast<float*>(ws + L.rstd2);
        ln.dy = ws + w.dy; ln.dres = dxa; ln.dx = dxb; ln.gout = g192b;
        ln.dgamma = grad_raw + R.n2w; ln.dbeta = grad_raw + R.n2b; ln.drop_p = p; ln.seed = seed; ln.site = 4 * l + 1;
        PF_TRY(ln_backward(d.bf16, ln, s));
        // attention: xmid = x + D1 . (Wo O + bo)
        PF_TRY(on_side([&](hipStream_t q) { return weight_grad(d, g192b, kEncD, ws + L.o, kEncD, grad_raw + R.ow, grad_raw + R.ob, q); }));
        PF_TRY(linear(d, kEpiPlain, g192b, kEncD, kEncD, fr + P.o_t, nullptr, ws + w.dy, s));
        AttnArgs at{};
        at.qkv = ws + L.qkv; at.B = d.B; at.T = d.T; at.out = ws + L.o; at.lse = reinterpret_cast<float*>(ws + L.lse);
        at.drop_p = p; at.seed = seed; at.site = 4 * l + 0; at.dout = ws + w.dy; at.dqkv = dqkv;
        PF_TRY(attn_backward(d.bf16, at, s));
        PF_TRY(on_side([&](hipStream_t q) { return weight_grad(d, dqkv, 3 * kEncD, ws + L.y1, kEncD, grad_raw + R.inw, grad_raw + R.inb, q); }));
        PF_TRY(linear(d, kEpiPlain, dqkv, 3 * kEncD, kEncD, fr + P.in_t, nullptr, ws + w.dy, s));
        ln.x = reinterpret_cast<const float*>(ws + L.x); ln.gamma = raw + R.n1w;
        ln.mean = reinterpret_cast<float*>(ws + L.mean1); ln.rstd = reinterpret_cast<float*>(ws + L.rstd1);
        ln.dy = ws + w.dy; ln.dres = dxb; ln.dx = dxa; ln.gout = l > 0 ? ws + w.g192a[l - 1] : nullptr;
        ln.dgamma = grad_raw + R.n1w; ln.dbeta = grad_raw + R.n1b; ln.site = l > 0 ? 4 * (l - 1) + 3 : 0;
        PF_TRY(ln_backward(d.bf16, ln, s));
    }
    // ---- token assembly + stem -----------------------------------------------------------------------------------------------
    for (int l = 1; l < 4; ++l)
        if (hipMemsetAsync(ws + w.gpad[l], 0, (size_t)d.N * kGpadRows[l] * kConv[l].cout * d.esz, s) != hipSuccess)
            return hip_failed(hipGetLastError());
    PF_TRY(tok_backward(d.bf16, dxa, ws + w.dact[3], d.B, d.n_extra, d.D, ws + w.gpad[3], (int64_t)kGpadRows[3] * kEncD,
                        (int64_t)kGpadOff[3] * kEncD, grad_extra, grad_token_bias, s));
    {
        float* t0 = reinterpret_cast<float*>(ws + w.conv_dw);
        int64_t off = 0;
        for (int l = 0; l < 4; ++l) { conv_tmp[l] = t0 + off; off += (int64_t)kConv[l].cout * kConv[l].cin * kConv[l].kw; }
        if (hipMemsetAsync(t0, 0, (size_t)off * 4, s) != hipSuccess) return hip_failed(hipGetLastError());
    }
    for (int l = 3; l >= 0; --l) {
        const Conv& c = kConv[l];
        // weight gradient: G_l^T . im2col(input of layer l)
        DenseTnArgs t{};
        if (l > 0) {
            t.G = ws + w.gpad[l] + (size_t)kGpadOff[l] * c.cout * d.esz; t.g_seq_stride = (int64_t)kGpadRows[l] * c.cout;
            t.A = ws + w.act[l - 1]; t.a_seq_stride = (int64_t)c.lin * c.cin;
        } else {
            t.G = ws + w.g1; t.g_seq_stride = (int64_t)c.lout * c.cout;
            t.A = ws + w.sig; t.a_seq_stride = c.lin;
        }
        t.ldg = c.cout; t.lda = c.s * c.cin; t.M = d.N * c.lout; t.rows_per_seq = c.lout; t.N1 = c.cout; t.N2 = c.kw * c.cin;
        t.dW = conv_tmp[l]; t.ldw = c.cin * c.kw; t.conv_cin = 0; t.conv_kw = 0; t.db = grad_raw + r.conv_b[l];
        t.splits = 0;
        if (l == 0) {
            // the last weight gradient (conv1: 6.3 M positions x 32 x 64) stays on the caller's stream: on the side stream it
            // queued behind conv2's weight gradient (0.7 ms, started one kernel earlier) and the step ended with 0.25 ms of
            // one kernel on an otherwise idle GPU; here it runs beside that kernel's tail
            PF_TRY(dense_tn(d.bf16, t, s));
            break;
        }
        PF_TRY(on_side([&](hipStream_t q) { return dense_tn(d.bf16, t, q); }));
        // data gradient: transposed convolution as a GEMM over windows of the padded gradient image, times gelu' of the
        // previous layer, written into that layer's padded image (or the plain G1 of the first layer)
        const Conv& cp = kConv[l - 1];
        DenseArgs a{};
        a.A = ws + w.gpad[l]; a.M = d.N * kDxRows[l]; a.rows_per_seq = kDxRows[l]; a.a_seq_stride = (int64_t)kGpadRows[l] * c.cout;
        a.lda = c.cout; a.K = (c.kw / c.s) * c.cout; a.N = c.s * c.cin; a.KC = a.K <= 256 ? a.K : 192;
        a.wfrags = fr + pk.conv_dx[l]; a.bias = nullptr;
        if (l > 1) {
            a.out = ws + w.gpad[l - 1] + (size_t)kGpadOff[l - 1] * cp.cout * d.esz; a.o_seq_stride = (int64_t)kGpadRows[l - 1] * cp.cout;
        } else {
            a.out = ws + w.g1; a.o_seq_stride = (int64_t)cp.lout * cp.cout;
        }
        a.ldo = c.s * c.cin; a.o_valid_per_seq = (int64_t)c.lin * c.cin; a.x_seq_stride = (int64_t)c.lin * c.cin;
        a.mul = ws + w.dact[l - 1];
        PF_TRY(dense_nt(d.bf16, kEpiMul, a, s));
    }
    return PF_OK;
    };
    int rc = chain();
    if (side != s) {                                    // join: the caller's stream continues after the last weight gradient
        const int jrc = g_side.order(side, s);
        if (rc == PF_OK) rc = jrc;
    }
    if (rc != PF_OK) return rc;
    for (int l = 0; l < 4; ++l) {
        const Conv& c = kConv[l];
        hipLaunchKernelGGL(conv_dw_permute_kernel, dim3((unsigned)((c.cout * c.cin * c.kw + 1023) / 1024)), dim3(256), 0, s, conv_tmp[l],
                           grad_raw + r.conv_w[l], c.cout, c.cin, c.kw);
        PF_TRY(launch_status());
    }
    return PF_OK;
}

}  // namespace pf
