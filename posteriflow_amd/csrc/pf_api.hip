// pf_api.hip -- the extern "C" boundary declared in include/pf_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "pf_flow_params.h"
#include "pf_status.h"

namespace pf {
int64_t stem_pack_map_len(bool bf16);
int64_t stem_raw_count();
int64_t stem_packed_bytes(bool bf16);
void stem_build_pack_map(bool bf16, int32_t* map);
int64_t stem_workspace_bytes(bool bf16, int64_t n_seq);
int stem_forward(bool bf16, const char* packed, const float* strain, int64_t n_seq, float* tokens,
                 float* log_energy, char* ws, hipStream_t s);
int64_t fusion_raw_count();
int64_t fusion_packed_bytes();
int fusion_pack(const float* raw, char* packed, hipStream_t s);
int fusion_forward(const char* packed, float* tokens, int n_tokens, const float* tok_bias, const float* pool_q,
                   int64_t n_events, float* pooled, hipStream_t s);
int inc_pack_frags(const float* src, int n_rows, int k, void* out, hipStream_t s);
int64_t inc_layer_bytes(int D, int H, bool f32);
int flow_inverse_inc(const PfFlowDesc& d, float deriv_const, const int32_t* u1, const void* packed, const float* proj,
                     int64_t ctx_rows, const float* z, const int32_t* inv_perm, int64_t batch, float* x, float* logdet,
                     uint32_t* fail, hipStream_t s);
int64_t remix_workspace_bytes(int64_t batch);
int remix_forward(const void* noise, int64_t n_noise, const void* signals, int64_t n_signals,
                  const int64_t* noise_row, const int64_t* sig_start, const int32_t* nsig, const float* scale,
                  const int32_t* shift, const int32_t* fill_row, const float* fill, int64_t n_fill, int64_t batch,
                  float* strain, float* sig_sum, float* net_snr, void* ws, hipStream_t s);
int rqs_backward(const PfFlowDesc& d, float deriv_const, const float* u, const float* params, const float* gy,
                 const float* glad, int64_t n, float* gparams, float* gu, hipStream_t s);
int launch_gather(bool bf16, const float* raw, const int32_t* map, void* out, int64_t n, hipStream_t s);
int flow_backward_chain(const PfFlowDesc& d, float deriv_const, const PfFlowBwdChainArgs& a, hipStream_t s);
int flow_reevaluate_generic(const FlowPlan& L, const PfFlowReevalArgs& a, const PfFlowDesc& d, hipStream_t s);
int flow_reevaluate(const FlowPlan& P, const PfFlowReevalArgs& a, hipStream_t s);
int64_t enc_train_raw_count();
int64_t enc_train_packed_bytes(bool bf16);
int enc_train_pack(bool bf16, const float* raw, void* packed, hipStream_t s);
int ctx_project_rows(bool bf16, const void* wfrags, const float* bias, const float* ctx, int64_t rows, int C, int n_units,
                     float* out, hipStream_t s);
int diag_stream_ingest(const void* buf, int64_t bytes, int workgroups, int in_flight, unsigned* sink, hipStream_t s);
int64_t enc_train_workspace_bytes(const PfEmbedTrainDesc* desc, int64_t n_events);
int enc_train_forward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* strain,
                      const float* extra_tokens, const float* token_bias, const float* pool_q, int64_t n_events, float* pooled,
                      float* log_energy, void* workspace, hipStream_t s);
int enc_train_backward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* pool_q,
                       const float* grad_pooled, int64_t n_events, void* workspace, float* grad_raw, float* grad_extra,
                       float* grad_token_bias, float* grad_pool_q, hipStream_t s);
}  // namespace pf
#include "pf_dense.h"
#include "pf_enc_ops.h"

namespace pf {
thread_local int g_hip_error = 0;
void geom_twiddles(float* table);
int geom_features(const PfGeomArgs& a, hipStream_t st);
}

namespace {
thread_local char g_err[256] = "";
int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int layout_of(const PfFlowDesc* d, pf::FlowPlan& L) {
    if (!d) return fail(PF_ERR_BAD_ARG, "desc is null");
    const int rc = pf::make_plan(*d, L);
    if (rc != PF_OK)
        return fail(rc, "unsupported flow shape (scheduled kernels: D<=H/16, H in {64,128,192,256}, K<=16, num_blocks=2; in-layer "
                    "context: C<=288, or <=576 in bf16 at H=256 -- wider contexts need PF_FLAG_HOIST_CTX, C<=1024 / 2048 in bf16; "
                    "generic kernel: plain conditioner, H%16==0, H<=512, D<=32, K<=32, LDS image <= 160 KB)");
    if (!(d->tail_bound > 0.f)) return fail(PF_ERR_BAD_ARG, "tail_bound must be positive");
    if (d->min_bin_width * d->num_bins > 1.f || d->min_bin_height * d->num_bins > 1.f)
        return fail(PF_ERR_BAD_ARG, "minimal bin size too large for the number of bins");
    return PF_OK;
}
// entry points that compute with the packed weights: a PF_FLAG_BWD desc only describes a packing
int compute_layout_of(const PfFlowDesc* d, pf::FlowPlan& L) {
    const int rc = layout_of(d, L);
    if (rc != PF_OK) return rc;
    return L.bwd ? fail(PF_ERR_UNSUPPORTED, "PF_FLAG_BWD describes the backward chain's weight stream only") : PF_OK;
}
bool misaligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) != 0; }
}  // namespace

extern "C" {

const char* pf_last_error(void) { return g_err; }
const char* pf_version(void) { return "posteriflow_amd 0.1 (gfx950)"; }

int64_t pf_flow_raw_param_count(const PfFlowDesc* desc) {
    pf::FlowPlan L;
    if (layout_of(desc, L) != PF_OK) return -1;
    return pf::raw_param_count(L);
}
int64_t pf_flow_packed_bytes(const PfFlowDesc* desc) {
    pf::FlowPlan L;
    if (layout_of(desc, L) != PF_OK) return -1;
    return L.packed_bytes();
}
int64_t pf_flow_pack_map_len(const PfFlowDesc* desc) {
    pf::FlowPlan L;
    if (layout_of(desc, L) != PF_OK) return -1;
    return pf::pack_map_len(L);
}
int pf_flow_build_pack_map(const PfFlowDesc* desc, int32_t* map_host) {
    pf::FlowPlan L;
    int rc = layout_of(desc, L);
    if (rc != PF_OK) return rc;
    if (!map_host) return fail(PF_ERR_BAD_ARG, "map_host is null");
    return pf::build_pack_map(L, map_host);
}
int pf_flow_pack(const PfFlowDesc* desc, const float* raw, const int32_t* map, void* packed, void* stream) {
    pf::FlowPlan L;
    int rc = layout_of(desc, L);
    if (rc != PF_OK) return rc;
    if (!raw || !map || !packed) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (misaligned(map, 16) || misaligned(packed, 16)) return fail(PF_ERR_BAD_ARG, "map/packed must be 16-byte aligned");
    rc = pf::launch_pack(L, raw, map, packed, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}

int64_t pf_flow_workspace_bytes(const PfFlowDesc* desc, int64_t ctx_rows) {
    pf::FlowPlan L;
    if (compute_layout_of(desc, L) != PF_OK || ctx_rows < 0) return -1;
    return pf::ctx_project_bytes(L, ctx_rows);
}

// keep <=> (hash >> 8) >= threshold: P(drop) = round(p 2^24) / 2^24
static uint32_t drop_threshold(float p) { return static_cast<uint32_t>(std::lround((double)p * 16777216.0)); }

// hoisted plans: run the context projection into the workspace, hand it to the chain kernel
static int project_context(const pf::FlowPlan& L, pf::FwdParams& p, void* workspace, int64_t workspace_bytes,
                           hipStream_t s) {
    p.cproj = nullptr;
    if (!L.hoist) return PF_OK;
    const int64_t need = pf::ctx_project_bytes(L, p.ctx_rows);
    if (!workspace || workspace_bytes < need) return fail(PF_ERR_BAD_ARG, "workspace too small (pf_flow_workspace_bytes)");
    if (misaligned(workspace, 16)) return fail(PF_ERR_BAD_ARG, "workspace must be 16-byte aligned");
    const int rc = pf::launch_ctx_project(L, p.packed, p.ctx, p.ctx_rows, workspace, s);
    if (rc != PF_OK) return fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
    p.cproj = workspace;
    return PF_OK;
}

static int flow_forward_impl(const PfFlowDesc* desc, const void* packed, const float* x, const float* ctx,
                             const int32_t* ar_perm, const float* log_sigma, int64_t batch, float* z,
                             float* logdet, float* nll, float* layer_inputs, float* nll_sum, float* zero_pair,
                             void* workspace, int64_t workspace_bytes, void* stream, float dropout_p = 0.f,
                             uint64_t dropout_seed = 0) {
    pf::FlowPlan L;
    int rc = compute_layout_of(desc, L);
    if (rc != PF_OK) return rc;
    if (batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
    if (batch == 0) return PF_OK;
    if (!packed || !x) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (L.C > 0 && !ctx) return fail(PF_ERR_BAD_ARG, "ctx is null but context_features > 0");
    if (misaligned(packed, 16)) return fail(PF_ERR_BAD_ARG, "packed must be 16-byte aligned");
    pf::FwdParams p{};
    p.packed = static_cast<const char*>(packed);
    p.x = x; p.ctx = ctx; p.ar_perm = ar_perm; p.log_sigma = log_sigma; p.z = z; p.logdet = logdet; p.nll = nll;
    p.batch = batch; p.ctx_rows = batch; p.fail_flags = nullptr; p.plan = L; p.u_save = layer_inputs; p.nll_sum = nll_sum; p.zero_pair = zero_pair;
    p.tail_bound = desc->tail_bound; p.min_w = desc->min_bin_width; p.min_h = desc->min_bin_height;
    p.min_d = desc->min_derivative;
    p.deriv_const = (float)std::log(std::exp(1.0 - (double)desc->min_derivative) - 1.0);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(PF_ERR_BAD_ARG, "dropout_p must be in [0, 1)");
    p.drop_thresh = drop_threshold(dropout_p);
    p.drop_seed = static_cast<uint32_t>(dropout_seed ^ (dropout_seed >> 32));
    p.drop_scale = 1.f / (1.f - dropout_p);
    if (p.drop_thresh && L.wide) return fail(PF_ERR_UNSUPPORTED, "PF_FLAG_WIDE is an evaluation layout: no dropout");
    rc = project_context(L, p, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    if (rc != PF_OK) return rc;
    rc = pf::launch_flow_forward(p, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, rc == PF_ERR_HIP ? hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)) : "unsupported launch shape");
}

int pf_flow_forward_train(const PfFlowDesc* desc, const void* packed, const float* x, const float* ctx,
                          const int32_t* ar_perm, const float* log_sigma, int64_t batch, float* z,
                          float* logdet, float* nll, float* layer_inputs, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    return flow_forward_impl(desc, packed, x, ctx, ar_perm, log_sigma, batch, z, logdet, nll, layer_inputs, nullptr,
                             nullptr, workspace, workspace_bytes, stream);
}
int pf_flow_forward_train_dropout(const PfFlowDesc* desc, const void* packed, const float* x, const float* ctx,
                                  const int32_t* ar_perm, const float* log_sigma, int64_t batch, float* z,
                                  float* logdet, float* nll, float* layer_inputs, float dropout_p,
                                  uint64_t dropout_seed, void* workspace, int64_t workspace_bytes, void* stream) {
    return flow_forward_impl(desc, packed, x, ctx, ar_perm, log_sigma, batch, z, logdet, nll, layer_inputs, nullptr,
                             nullptr, workspace, workspace_bytes, stream, dropout_p, dropout_seed);
}
int pf_flow_dropout_mask(const PfFlowDesc* desc, float dropout_p, uint64_t dropout_seed, int64_t batch, float* mask,
                         void* stream) {
    pf::FlowPlan L;
    int rc = layout_of(desc, L);
    if (rc != PF_OK) return rc;
    if (L.generic) return fail(PF_ERR_UNSUPPORTED, "the generic-shape kernel is an evaluation kernel: no dropout");
    if (batch < 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return fail(PF_ERR_BAD_ARG, "need batch >= 0 and dropout_p in [0, 1)");
    if (batch == 0) return PF_OK;
    if (!mask) return fail(PF_ERR_BAD_ARG, "mask is null");
    rc = pf::launch_dropout_mask(L, drop_threshold(dropout_p), static_cast<uint32_t>(dropout_seed ^ (dropout_seed >> 32)),
                                 1.f / (1.f - dropout_p), batch, mask, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, rc == PF_ERR_HIP ? hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)) : "hidden_features > 256");
}
int pf_flow_forward_reduce(const PfFlowDesc* desc, const void* packed, const float* x, const float* ctx,
                           const int32_t* ar_perm, const float* log_sigma, int64_t batch, float* nll,
                           float* nll_sum_count, float* zero_pair, void* workspace, int64_t workspace_bytes,
                           void* stream) {
    if (!nll_sum_count) return fail(PF_ERR_BAD_ARG, "nll_sum_count is null");
    if (zero_pair == nll_sum_count) return fail(PF_ERR_BAD_ARG, "zero_pair must not be the accumulator of this launch");
    return flow_forward_impl(desc, packed, x, ctx, ar_perm, log_sigma, batch, nullptr, nullptr, nll, nullptr,
                             nll_sum_count, zero_pair, workspace, workspace_bytes, stream);
}
int pf_flow_forward(const PfFlowDesc* desc, const void* packed, const float* x, const float* ctx,
                    const int32_t* ar_perm, const float* log_sigma, int64_t batch, float* z,
                    float* logdet, float* nll, void* workspace, int64_t workspace_bytes, void* stream) {
    return pf_flow_forward_train(desc, packed, x, ctx, ar_perm, log_sigma, batch, z, logdet, nll, nullptr,
                                 workspace, workspace_bytes, stream);
}

int pf_flow_rqs_backward(const PfFlowDesc* desc, const float* u, const float* params, const float* grad_y,
                         const float* grad_logabsdet, int64_t rows, float* grad_params, float* grad_u,
                         void* stream) {
    if (!desc) return fail(PF_ERR_BAD_ARG, "desc is null");
    if (desc->features < 1 || desc->num_bins < 2 || desc->num_bins > 32)
        return fail(PF_ERR_UNSUPPORTED, "need features >= 1 and 2 <= num_bins <= 32");
    if (!(desc->tail_bound > 0.f)) return fail(PF_ERR_BAD_ARG, "tail_bound must be positive");
    if (rows < 0) return fail(PF_ERR_BAD_ARG, "negative rows");
    if (rows == 0) return PF_OK;
    if (!u || !params || !grad_y || !grad_logabsdet || !grad_params || !grad_u)
        return fail(PF_ERR_BAD_ARG, "null pointer");
    const float dc = (float)std::log(std::exp(1.0 - (double)desc->min_derivative) - 1.0);
    const int rc = pf::rqs_backward(*desc, dc, u, params, grad_y, grad_logabsdet, rows, grad_params, grad_u,
                                    static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}

int pf_flow_backward_chain(const PfFlowDesc* desc, const PfFlowBwdChainArgs* a, void* stream) {
    if (!desc || !a) return fail(PF_ERR_BAD_ARG, "null pointer");
    const int D = desc->features, H = desc->hidden_features, K = desc->num_bins;
    const bool f32d = desc->precision == PF_PREC_F32;
    const bool h_ok = H == 64 || H == 128 || H == 192 || H == 256 || (f32d && (H == 384 || H == 512));
    if (!h_ok || D < 1 || D > 16 || K < 2 || K > (f32d ? 32 : 16) || desc->num_blocks != 2 || desc->num_layers < 1 ||
        ((desc->reserved & PF_FLAG_MASKED_CONTEXT) && !f32d))
        return fail(PF_ERR_UNSUPPORTED, "backward chain: 2 blocks, D <= 16; bf16: plain conditioner, H in {64,128,192,256}, K <= 16; "
                    "fp32: also H = 384, 512, K <= 32 and the masked-context conditioner");
    if (!(desc->tail_bound > 0.f)) return fail(PF_ERR_BAD_ARG, "tail_bound must be positive");
    if (a->batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
    if (a->batch == 0) return PF_OK;
    const bool bf = desc->precision == PF_PREC_BF16;
    if (desc->precision != PF_PREC_F32 && !bf) return fail(PF_ERR_BAD_ARG, "bad precision");
    if (desc->reserved & ~PF_FLAG_MASKED_CONTEXT & ~PF_FLAG_HOIST_CTX & ~PF_FLAG_WIDE)
        return fail(PF_ERR_BAD_ARG, "pass the flow's own desc (PF_FLAG_BWD belongs to the packing calls)");
    if (bf ? !a->packed : (!a->WfT || !a->W2T || !a->W1T || !a->W0T))
        return fail(PF_ERR_BAD_ARG, bf ? "bf16 desc: args.packed (the PF_FLAG_BWD stream) is null" : "null weight pointer");
    if (bf && H % 32) return fail(PF_ERR_UNSUPPORTED, "bf16 backward chain needs H % 32 == 0");
    if (!a->U || !a->params || !a->hs || !a->t1s || !a->Gp || !a->Gh0 || !a->Gt1 || !a->Gt2 || !a->g_x)
        return fail(PF_ERR_BAD_ARG, "null pointer");
    if (a->gp_ld && (int)a->gp_ld < D * (3 * K - 1)) return fail(PF_ERR_BAD_ARG, "gp_ld is shorter than a row of Gp");
    if (a->g_nll ? (!a->nll_z || a->g_z || a->g_lad) : (!a->g_z || !a->g_lad))
        return fail(PF_ERR_BAD_ARG, "pass either (g_z, g_lad) or (g_nll, nll_z[, log_sigma])");
    const bool ctx = a->pc != nullptr, glu = ctx && !(desc->reserved & PF_FLAG_MASKED_CONTEXT);
    if (glu != (a->t2s != nullptr) || glu != (a->gates != nullptr) || ctx != (a->Gc != nullptr))
        return fail(PF_ERR_BAD_ARG, "pc and Gc (and t2s, gates for the plain conditioner) go together (all NULL for a context-free flow)");
    const void* al[] = {bf ? nullptr : a->WfT, bf ? nullptr : a->W2T, bf ? nullptr : a->W1T, bf ? nullptr : a->W0T, bf ? a->packed : nullptr,
                        a->hs, a->t1s, a->t2s, a->gates, a->pc, a->Gh0, a->Gt1, a->Gt2, a->Gc, a->drop};
    for (const void* q : al)
        if (misaligned(q, 16)) return fail(PF_ERR_BAD_ARG, "weight / activation tensors must be 16-byte aligned");
    const float dc = (float)std::log(std::exp(1.0 - (double)desc->min_derivative) - 1.0);
    const int rc = pf::flow_backward_chain(*desc, dc, *a, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, rc == PF_ERR_HIP ? hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)) : "unsupported shape");
}

int pf_flow_reevaluate(const PfFlowDesc* desc, const PfFlowReevalArgs* a, void* stream) {
    if (!desc || !a) return fail(PF_ERR_BAD_ARG, "null pointer");
    if ((desc->reserved & PF_FLAG_BWD) || ((desc->reserved & PF_FLAG_MASKED_CONTEXT) && desc->precision != PF_PREC_F32))
        return fail(PF_ERR_UNSUPPORTED, "re-evaluation kernel: the flow's own desc; the masked-context conditioner in fp32 only");
    if (desc->precision == PF_PREC_F32) {           // parity mode: the generic kernel's conditioner, exact-fp32 MFMA
        PfFlowDesc g = *desc;
        g.reserved = PF_FLAG_GENERIC | (desc->reserved & PF_FLAG_MASKED_CONTEXT);
        pf::FlowPlan G;
        const int rg = layout_of(&g, G);
        if (rg != PF_OK) return rg;
        if (a->compact) return fail(PF_ERR_BAD_ARG, "fp32 re-evaluation: compact must be 0");
        if (a->batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
        if (a->batch == 0) return PF_OK;
        if (!a->packed || !a->U || !a->hs || !a->t1s || !a->h2 || !a->params) return fail(PF_ERR_BAD_ARG, "null pointer");
        const bool cx = G.C > 0, glu = cx && !G.additive;
        if (cx != (a->ctx != nullptr) || cx != (a->t2s != nullptr) || glu != (a->gates != nullptr) || cx != (a->pc != nullptr))
            return fail(PF_ERR_BAD_ARG, "ctx, t2s, pc (and gates, for the plain conditioner) go together with context_features > 0");
        const void* al[] = {a->packed, a->hs, a->t1s, a->t2s, a->gates, a->pc, a->h2, a->drop};
        for (const void* q : al)
            if (misaligned(q, 16)) return fail(PF_ERR_BAD_ARG, "stream / activation tensors must be 16-byte aligned");
        if (G.H % 4) return fail(PF_ERR_UNSUPPORTED, "hidden_features % 4");
        const int rc = pf::flow_reevaluate_generic(G, *a, *desc, static_cast<hipStream_t>(stream));
        return rc == PF_OK ? rc : fail(rc, rc == PF_ERR_HIP ? hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)) : "unsupported shape");
    }
    if (desc->precision != PF_PREC_BF16)
        return fail(PF_ERR_BAD_ARG, "bad precision");
    PfFlowDesc d = *desc;
    d.reserved = PF_FLAG_BWD;                       // the layout of the stream it reads
    pf::FlowPlan P;
    const int rc0 = layout_of(&d, P);
    if (rc0 != PF_OK) return rc0;
    if (a->batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
    if (a->batch == 0) return PF_OK;
    if (!a->packed || !a->U || !a->hs || !a->t1s || !a->h2 || !a->params) return fail(PF_ERR_BAD_ARG, "null pointer");
    const bool ctx = P.C > 0;
    if (ctx != (a->ctx != nullptr) || ctx != (a->t2s != nullptr) || ctx != (a->gates != nullptr) || ctx != (a->pc != nullptr))
        return fail(PF_ERR_BAD_ARG, "ctx, t2s, gates and pc go together with context_features > 0");
    const void* al[] = {a->packed, a->hs, a->t1s, a->t2s, a->gates, a->pc, a->h2, a->drop};
    for (const void* q : al)
        if (misaligned(q, 16)) return fail(PF_ERR_BAD_ARG, "stream / activation tensors must be 16-byte aligned");
    const int rc = pf::flow_reevaluate(P, *a, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, rc == PF_ERR_HIP ? hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)) : "unsupported shape");
}

int pf_flow_inverse(const PfFlowDesc* desc, const void* packed, const float* z, const float* ctx,
                    int64_t ctx_rows, const int32_t* ar_inv_perm, int64_t batch, float* x,
                    float* logdet, uint32_t* fail_flags, void* workspace, int64_t workspace_bytes,
                    void* stream) {
    pf::FlowPlan L;
    int rc = compute_layout_of(desc, L);
    if (rc != PF_OK) return rc;
    if (batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
    if (batch == 0) return PF_OK;
    if (!packed || !z) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (L.C > 0) {
        if (!ctx) return fail(PF_ERR_BAD_ARG, "ctx is null but context_features > 0");
        if (ctx_rows < 1 || batch % ctx_rows != 0)
            return fail(PF_ERR_BAD_ARG, "ctx_rows must be >= 1 and divide batch");
    } else {
        ctx_rows = batch;
    }
    if (misaligned(packed, 16)) return fail(PF_ERR_BAD_ARG, "packed must be 16-byte aligned");
    pf::FwdParams p{};
    p.packed = static_cast<const char*>(packed);
    p.x = z; p.ctx = ctx; p.ar_perm = ar_inv_perm; p.log_sigma = nullptr; p.z = x; p.logdet = logdet;
    p.nll = nullptr; p.batch = batch; p.ctx_rows = ctx_rows; p.fail_flags = fail_flags; p.plan = L;
    p.tail_bound = desc->tail_bound; p.min_w = desc->min_bin_width; p.min_h = desc->min_bin_height;
    p.min_d = desc->min_derivative;
    p.deriv_const = (float)std::log(std::exp(1.0 - (double)desc->min_derivative) - 1.0);
    rc = project_context(L, p, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    if (rc != PF_OK) return rc;
    rc = pf::launch_flow_inverse(p, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, rc == PF_ERR_HIP ? hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)) : "unsupported launch shape");
}

// ---- strain-embedding stem -----------------------------------------------------------------
static int prec_ok(int32_t precision) {
    return (precision == PF_PREC_F32 || precision == PF_PREC_BF16) ? PF_OK : fail(PF_ERR_BAD_ARG, "bad precision");
}
int64_t pf_embed_stem_raw_param_count(void) { return pf::stem_raw_count(); }
int64_t pf_embed_stem_packed_bytes(int32_t precision) {
    return prec_ok(precision) == PF_OK ? pf::stem_packed_bytes(precision == PF_PREC_BF16) : -1;
}
int64_t pf_embed_stem_pack_map_len(int32_t precision) {
    return prec_ok(precision) == PF_OK ? pf::stem_pack_map_len(precision == PF_PREC_BF16) : -1;
}
int pf_embed_stem_build_pack_map(int32_t precision, int32_t* map_host) {
    if (prec_ok(precision) != PF_OK) return PF_ERR_BAD_ARG;
    if (!map_host) return fail(PF_ERR_BAD_ARG, "map_host is null");
    pf::stem_build_pack_map(precision == PF_PREC_BF16, map_host);
    return PF_OK;
}
int pf_embed_stem_pack(int32_t precision, const float* raw, const int32_t* map, void* packed, void* stream) {
    if (prec_ok(precision) != PF_OK) return PF_ERR_BAD_ARG;
    if (!raw || !map || !packed) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (misaligned(map, 16) || misaligned(packed, 16)) return fail(PF_ERR_BAD_ARG, "map/packed must be 16-byte aligned");
    const bool bf = precision == PF_PREC_BF16;
    // weights (operand type) then biases (fp32), the order of pf::stem_build_pack_map
    int64_t nbias = 0;
    const int couts[4] = {32, 64, 128, 192};
    for (int c : couts) nbias += c;
    const int64_t nw = pf::stem_pack_map_len(bf) - nbias;
    const int64_t wbytes = pf::stem_packed_bytes(bf) - nbias * 4;
    int rc = pf::launch_gather(bf, raw, map, packed, nw, static_cast<hipStream_t>(stream));
    if (rc == PF_OK)
        rc = pf::launch_gather(false, raw, map + nw, static_cast<char*>(packed) + wbytes, nbias, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}
int64_t pf_embed_stem_workspace_bytes(int32_t precision, int64_t n_sequences) {
    if (prec_ok(precision) != PF_OK || n_sequences < 0) return -1;
    return pf::stem_workspace_bytes(precision == PF_PREC_BF16, n_sequences);
}
int pf_embed_stem_forward(int32_t precision, const void* packed, const float* strain, int64_t n_sequences,
                          float* tokens, float* log_energy, void* workspace, int64_t workspace_bytes,
                          void* stream) {
    if (prec_ok(precision) != PF_OK) return PF_ERR_BAD_ARG;
    if (n_sequences < 0) return fail(PF_ERR_BAD_ARG, "negative n_sequences");
    if (n_sequences == 0) return PF_OK;
    if (n_sequences > 65535) return fail(PF_ERR_UNSUPPORTED, "at most 65535 sequences per call");
    if (!packed || !strain || !tokens) return fail(PF_ERR_BAD_ARG, "null pointer");
    const bool bf = precision == PF_PREC_BF16;
    if (!workspace || workspace_bytes < pf::stem_workspace_bytes(bf, n_sequences))
        return fail(PF_ERR_BAD_ARG, "workspace too small (pf_embed_stem_workspace_bytes)");
    if (misaligned(packed, 16) || misaligned(strain, 16) || misaligned(tokens, 16) || misaligned(workspace, 256))
        return fail(PF_ERR_BAD_ARG, "packed/strain/tokens must be 16-byte, workspace 256-byte aligned");
    const int rc = pf::stem_forward(bf, static_cast<const char*>(packed), strain, n_sequences, tokens, log_energy,
                                    static_cast<char*>(workspace), static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}

// ---- incremental inverse ------------------------------------------------------------------------
int pf_pack_bf16_frags(const float* src, int32_t n_rows, int32_t k, void* out, void* stream) {
    if (!src || !out) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (n_rows <= 0 || k <= 0 || n_rows % 16 || k % 32) return fail(PF_ERR_BAD_ARG, "rows must be a multiple of 16, k of 32");
    if (misaligned(out, 16)) return fail(PF_ERR_BAD_ARG, "out must be 16-byte aligned");
    const int rc = pf::inc_pack_frags(src, n_rows, k, out, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}
int64_t pf_flow_inc_layer_bytes(const PfFlowDesc* desc) {
    if (!desc || desc->hidden_features % 32 || desc->hidden_features < 64 || desc->hidden_features > 256) return -1;
    return pf::inc_layer_bytes(desc->features, desc->hidden_features, desc->precision == PF_PREC_F32);
}
int pf_flow_inverse_inc(const PfFlowDesc* desc, const int32_t* units_upto_degree, const void* packed,
                        const float* ctx_proj, int64_t ctx_rows, const float* z, const int32_t* ar_inv_perm,
                        int64_t batch, float* x, float* logdet, uint32_t* fail_flags, void* stream) {
    if (!desc) return fail(PF_ERR_BAD_ARG, "desc is null");
    const int D = desc->features, H = desc->hidden_features;
    if (H % 32 || H < 64 || H > 256 || D < 1 || D > 16 || D > H / 16 || desc->num_bins < 2 || desc->num_bins > 16 ||
        desc->num_blocks != 2)
        return fail(PF_ERR_UNSUPPORTED, "incremental inverse: need H in 64..256 (multiple of 32), D <= min(16, H/16), K <= 16");
    if (!(desc->tail_bound > 0.f)) return fail(PF_ERR_BAD_ARG, "tail_bound must be positive");
    if (batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
    if (batch == 0) return PF_OK;
    if (!units_upto_degree || !packed || !z || !x) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (ctx_proj && (ctx_rows < 1 || batch % ctx_rows != 0)) return fail(PF_ERR_BAD_ARG, "ctx_rows must be >= 1 and divide batch");
    if (misaligned(packed, 16) || misaligned(ctx_proj, 16)) return fail(PF_ERR_BAD_ARG, "packed / ctx_proj must be 16-byte aligned");
    if (units_upto_degree[0] != 0 || units_upto_degree[D] > H) return fail(PF_ERR_BAD_ARG, "units_upto_degree[0] must be 0 and [D] <= H");
    for (int i = 0; i < D; ++i)
        if (units_upto_degree[i] > units_upto_degree[i + 1]) return fail(PF_ERR_BAD_ARG, "units_upto_degree must be non-decreasing");
    const float dc = (float)std::log(std::exp(1.0 - (double)desc->min_derivative) - 1.0);
    const int rc = pf::flow_inverse_inc(*desc, dc, units_upto_degree, packed, ctx_proj, ctx_proj ? ctx_rows : batch, z,
                                        ar_inv_perm, batch, x, logdet, fail_flags, static_cast<hipStream_t>(stream));
    if (rc == PF_ERR_UNSUPPORTED) return fail(rc, "incremental inverse: more than 8 sixteen-unit tiles hold the units of one degree");
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}

// ---- strain-embedding token mixer (fusion transformer + attention pool) ------------------------
int64_t pf_embed_fusion_raw_param_count(void) { return pf::fusion_raw_count(); }
int64_t pf_embed_fusion_packed_bytes(void) { return pf::fusion_packed_bytes(); }
int pf_embed_fusion_pack(const float* raw, void* packed, void* stream) {
    if (!raw || !packed) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (misaligned(packed, 16) || misaligned(raw, 4)) return fail(PF_ERR_BAD_ARG, "packed must be 16-byte aligned");
    const int rc = pf::fusion_pack(raw, static_cast<char*>(packed), static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}
int pf_embed_fusion_forward(const void* packed, float* tokens, int32_t n_tokens, const float* token_bias,
                            const float* pool_queries, int64_t n_events, float* pooled, void* stream) {
    if (n_events < 0) return fail(PF_ERR_BAD_ARG, "negative n_events");
    if (n_events == 0) return PF_OK;
    if (n_tokens < 1 || n_tokens > 192) return fail(PF_ERR_UNSUPPORTED, "1 <= n_tokens <= 192 tokens per event");
    if (n_events > 0x7fffffff) return fail(PF_ERR_UNSUPPORTED, "too many events per call");
    if (!packed || !tokens || !pool_queries || !pooled) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (misaligned(packed, 16) || misaligned(tokens, 16) || misaligned(pooled, 16) || misaligned(pool_queries, 4) ||
        misaligned(token_bias, 16))
        return fail(PF_ERR_BAD_ARG, "packed/tokens/token_bias/pooled must be 16-byte aligned");
    const int rc = pf::fusion_forward(static_cast<const char*>(packed), tokens, n_tokens, token_bias, pool_queries, n_events, pooled,
                                      static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}

// ---- training-example remix --------------------------------------------------------------------
int64_t pf_remix_workspace_bytes(int64_t batch) { return batch < 0 ? -1 : pf::remix_workspace_bytes(batch); }
int pf_remix_forward(const void* noise_pool, int64_t n_noise, const void* signal_pool, int64_t n_signals,
                     const int64_t* noise_row, const int64_t* sig_start, const int32_t* nsig,
                     const float* scale, const int32_t* shift, const int32_t* fill_row, const float* fill,
                     int64_t n_fill, int64_t batch, float* strain, float* sig_sum, float* net_snr,
                     void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch < 0 || n_noise < 0 || n_signals < 0 || n_fill < 0) return fail(PF_ERR_BAD_ARG, "negative count");
    if (batch == 0) return PF_OK;
    if (batch > (int64_t(1) << 26)) return fail(PF_ERR_UNSUPPORTED, "at most 2^26 examples per call");
    if (!noise_row || !sig_start || !nsig || !scale || !shift || !strain)
        return fail(PF_ERR_BAD_ARG, "null pointer");
    if ((n_noise > 0 && !noise_pool) || (n_signals > 0 && !signal_pool) || (n_fill > 0 && !fill))
        return fail(PF_ERR_BAD_ARG, "pool pointer is null but its row count is not 0");
    if (!workspace || workspace_bytes < pf::remix_workspace_bytes(batch))
        return fail(PF_ERR_BAD_ARG, "workspace too small (pf_remix_workspace_bytes)");
    if (misaligned(workspace, 8) || misaligned(strain, 16) || misaligned(noise_pool, 16) || misaligned(signal_pool, 16) ||
        misaligned(sig_sum, 16) || misaligned(fill, 16))
        return fail(PF_ERR_BAD_ARG, "pools, fill, strain and sig_sum must be 16-byte aligned (workspace 8)");
    const int rc = pf::remix_forward(noise_pool, n_noise, signal_pool, n_signals, noise_row, sig_start, nsig, scale,
                                     shift, fill_row, fill, n_fill, batch, strain, sig_sum, net_snr, workspace,
                                     static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}

const char* pf_flow_forward_kernel_name(const PfFlowDesc* desc, int64_t batch) {
    static thread_local char name[160];
    pf::FlowPlan L;
    if (compute_layout_of(desc, L) != PF_OK) return nullptr;
    pf::forward_kernel_name(L, batch, name, sizeof(name));
    return name;
}

// ---- strain embedding: training path ------------------------------------------------------------------------------------
namespace {
const char* hip_msg() { return hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)); }
int finish(int rc, const char* unsupported) {
    if (rc == PF_OK) return rc;
    return fail(rc, rc == PF_ERR_HIP ? hip_msg() : rc == PF_ERR_UNSUPPORTED ? unsupported : "bad argument");
}
bool prec2(int32_t p) { return p == PF_PREC_F32 || p == PF_PREC_BF16; }
}  // namespace

int64_t pf_embed_train_raw_param_count(void) { return pf::enc_train_raw_count(); }
int64_t pf_embed_train_packed_bytes(int32_t precision) { return prec2(precision) ? pf::enc_train_packed_bytes(precision == PF_PREC_BF16) : -1; }
int pf_embed_train_pack(int32_t precision, const float* raw, void* packed, void* stream) {
    if (!prec2(precision) || !raw || !packed) return fail(PF_ERR_BAD_ARG, "pf_embed_train_pack: null pointer or bad precision");
    if (misaligned(packed, 16) || misaligned(raw, 4)) return fail(PF_ERR_BAD_ARG, "pf_embed_train_pack: misaligned pointer");
    return finish(pf::enc_train_pack(precision == PF_PREC_BF16, raw, packed, static_cast<hipStream_t>(stream)), "");
}
int64_t pf_embed_train_workspace_bytes(const PfEmbedTrainDesc* desc, int64_t n_events) {
    return pf::enc_train_workspace_bytes(desc, n_events);
}
int pf_embed_train_forward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* strain,
                           const float* extra_tokens, const float* token_bias, const float* pool_q, int64_t n_events,
                           float* pooled, float* log_energy, void* workspace, int64_t workspace_bytes, void* stream) {
    const int64_t need = pf::enc_train_workspace_bytes(desc, n_events);
    if (need < 0) return fail(PF_ERR_BAD_ARG, "pf_embed_train_forward: bad desc (precision, 1..3 detectors, n_extra + 61 n_det <= 192, 0 <= p < 1)");
    if (n_events == 0) return PF_OK;
    if (!packed || !raw || !strain || !pool_q || !pooled || !log_energy || !workspace)
        return fail(PF_ERR_BAD_ARG, "pf_embed_train_forward: null pointer");
    if (desc->n_extra_tokens > 0 && !extra_tokens) return fail(PF_ERR_BAD_ARG, "pf_embed_train_forward: extra_tokens is null");
    if (workspace_bytes < need) return fail(PF_ERR_BAD_ARG, "pf_embed_train_forward: workspace too small (pf_embed_train_workspace_bytes)");
    if (misaligned(workspace, 256) || misaligned(packed, 16) || misaligned(strain, 16) || misaligned(raw, 16) ||
        misaligned(pooled, 16) || misaligned(extra_tokens, 16) || misaligned(token_bias, 16))
        return fail(PF_ERR_BAD_ARG, "pf_embed_train_forward: misaligned pointer (workspace 256 B, tensors 16 B)");
    return finish(pf::enc_train_forward(desc, packed, raw, strain, extra_tokens, token_bias, pool_q, n_events, pooled, log_energy,
                                        workspace, static_cast<hipStream_t>(stream)), "unsupported token count");
}
int pf_embed_train_backward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* pool_q,
                            const float* grad_pooled, int64_t n_events, void* workspace, int64_t workspace_bytes,
                            float* grad_raw, float* grad_extra_tokens, float* grad_token_bias, float* grad_pool_q,
                            void* stream) {
    const int64_t need = pf::enc_train_workspace_bytes(desc, n_events);
    if (need < 0) return fail(PF_ERR_BAD_ARG, "pf_embed_train_backward: bad desc");
    if (!grad_raw) return fail(PF_ERR_BAD_ARG, "pf_embed_train_backward: grad_raw is null");
    if (n_events > 0 && (!packed || !raw || !pool_q || !grad_pooled || !workspace))
        return fail(PF_ERR_BAD_ARG, "pf_embed_train_backward: null pointer");
    if (n_events > 0 && workspace_bytes < need) return fail(PF_ERR_BAD_ARG, "pf_embed_train_backward: workspace too small");
    if (misaligned(workspace, 256) || misaligned(packed, 16) || misaligned(grad_raw, 16) || misaligned(grad_pooled, 16) ||
        misaligned(grad_extra_tokens, 16) || misaligned(grad_token_bias, 16))
        return fail(PF_ERR_BAD_ARG, "pf_embed_train_backward: misaligned pointer");
    return finish(pf::enc_train_backward(desc, packed, raw, pool_q, grad_pooled, n_events, workspace, grad_raw, grad_extra_tokens,
                                         grad_token_bias, grad_pool_q, static_cast<hipStream_t>(stream)), "unsupported token count");
}

// ---- building blocks -------------------------------------------------------------------------------------------------------
int64_t pf_dense_frag_bytes(int32_t precision, int32_t N, int32_t K) {
    const int ks = precision == PF_PREC_BF16 ? 32 : 16;
    if (!prec2(precision) || N <= 0 || K <= 0 || N % 16 || K % ks) return -1;
    return pf::dense_frag_count(precision == PF_PREC_BF16, N, K) * 16;
}
int pf_dense_pack_matrix(int32_t precision, const float* src, int32_t mode, int32_t ld, int32_t N, int32_t K, void* out, void* stream) {
    if (pf_dense_frag_bytes(precision, N, K) < 0 || !src || !out || (mode != 0 && mode != 1) || misaligned(out, 16))
        return fail(PF_ERR_BAD_ARG, "pf_dense_pack_matrix: N % 16, K % 32 (bf16) / 16 (fp32), mode 0 | 1, 16-byte aligned output");
    pf::DensePackTable tab{};
    tab.n = 1;
    tab.e[0].src_off = 0; tab.e[0].dst_off = 0; tab.e[0].mode = mode; tab.e[0].ld = ld; tab.e[0].N = N; tab.e[0].K = K;
    return finish(pf::dense_pack(precision == PF_PREC_BF16, src, tab, out, static_cast<hipStream_t>(stream)), "");
}
int pf_dense_pack_linear(int32_t precision, const float* weight, int32_t ld, int32_t n, int32_t k, int32_t n_padded, int32_t k_padded,
                         int32_t with_transposed, void* out, void* stream) {
    if (pf_dense_frag_bytes(precision, n_padded, k_padded) < 0 || (with_transposed && pf_dense_frag_bytes(precision, k_padded, n_padded) < 0))
        return fail(PF_ERR_BAD_ARG, "pf_dense_pack_linear: padded sizes must be multiples of 32 (bf16) / 16 (fp32)");
    if (!weight || !out || misaligned(out, 16) || n <= 0 || k <= 0 || n > n_padded || k > k_padded || ld < k)
        return fail(PF_ERR_BAD_ARG, "pf_dense_pack_linear: null / misaligned pointer, or n > n_padded, k > k_padded, ld < k");
    pf::DensePackTable tab{};
    tab.n = with_transposed ? 2 : 1;
    tab.e[0].src_off = 0; tab.e[0].dst_off = 0; tab.e[0].mode = 0; tab.e[0].ld = ld; tab.e[0].N = n_padded; tab.e[0].K = k_padded;
    tab.e[0].n_valid = n; tab.e[0].k_valid = k;
    if (with_transposed) {          // W^T [k_padded][n_padded] behind the forward form: the data gradient's operand
        tab.e[1] = tab.e[0];
        tab.e[1].dst_off = pf::dense_frag_count(precision == PF_PREC_BF16, n_padded, k_padded);
        tab.e[1].mode = 1; tab.e[1].N = k_padded; tab.e[1].K = n_padded; tab.e[1].n_valid = k; tab.e[1].k_valid = n;
    }
    return finish(pf::dense_pack(precision == PF_PREC_BF16, weight, tab, out, static_cast<hipStream_t>(stream)), "");
}
int pf_dense_nt(int32_t precision, int32_t epilogue, const PfDenseArgs* a, void* stream) {
    if (!prec2(precision) || !a) return fail(PF_ERR_BAD_ARG, "pf_dense_nt: null args or bad precision");
    if (a->M > 0 && (!a->A || !a->wfrags || !a->out || a->rows_per_seq <= 0)) return fail(PF_ERR_BAD_ARG, "pf_dense_nt: null pointer");
    if ((epilogue == PF_EPI_RESID && !a->resid) || (epilogue == PF_EPI_MUL && !a->mul))
        return fail(PF_ERR_BAD_ARG, "pf_dense_nt: the epilogue's operand is null");
    if (epilogue < PF_EPI_PLAIN || epilogue > PF_EPI_MUL) return fail(PF_ERR_BAD_ARG, "pf_dense_nt: unknown epilogue");
    // every global access of the kernels is a 16-byte piece (8 bytes for the accumulator-layout loads of an operand): check
    // each pointer and stride against the element size of the array it addresses -- the output, resid and dact arrays are
    // fp32 for the residual / multiply epilogues, in fp32 mode and with out_f32
    const int esz = precision == PF_PREC_BF16 ? 2 : 4;
    const int oesz = (epilogue == PF_EPI_RESID || epilogue == PF_EPI_MUL || precision != PF_PREC_BF16 || a->out_f32) ? 4 : 2;
    const int64_t xss = a->x_seq_stride ? a->x_seq_stride : a->o_seq_stride;
    if (misaligned(a->A, 16) || misaligned(a->wfrags, 16) || (a->lda * (int64_t)esz) % 16 || (a->a_seq_stride * esz) % 16 ||
        (a->a_chunk_stride * esz) % 16 || a->N % 16)
        return fail(PF_ERR_BAD_ARG, "pf_dense_nt: rows of A must start on 16-byte boundaries, N % 16 == 0");
    if (misaligned(a->out, 16) || (a->ldo * (int64_t)oesz) % 16 || (a->o_seq_stride * oesz) % 16)
        return fail(PF_ERR_BAD_ARG, "pf_dense_nt: rows of out must start on 16-byte boundaries (in its own element size)");
    if (misaligned(a->bias, 4) || (epilogue == PF_EPI_RESID && (misaligned(a->resid, 16) || (xss * 4) % 16)) ||
        (epilogue == PF_EPI_MUL && (misaligned(a->mul, 16) || (xss * esz) % 16)) ||
        (epilogue == PF_EPI_GELU && a->dact && (misaligned(a->dact, 16) || (xss * oesz) % 16)))
        return fail(PF_ERR_BAD_ARG, "pf_dense_nt: the epilogue's operand (resid / mul / dact) must be 16-byte aligned with 16-byte row starts");
    return finish(pf::dense_nt(precision == PF_PREC_BF16, epilogue, *a, static_cast<hipStream_t>(stream)),
                  "pf_dense_nt: KC too large for LDS, or a chunked reduction with N > 256");
}
int pf_dense_tn(int32_t precision, const PfDenseTnArgs* a, void* stream) {
    if (!prec2(precision) || !a) return fail(PF_ERR_BAD_ARG, "pf_dense_tn: null args or bad precision");
    if (a->M > 0 && (!a->G || !a->A || !a->dW || a->rows_per_seq <= 0)) return fail(PF_ERR_BAD_ARG, "pf_dense_tn: null pointer");
    const int esz = precision == PF_PREC_BF16 ? 2 : 4;
    if (misaligned(a->G, 16) || misaligned(a->A, 16) || (a->ldg * esz) % 16 || (a->lda * esz) % 16 || (a->g_seq_stride * esz) % 16 ||
        (a->a_seq_stride * esz) % 16)
        return fail(PF_ERR_BAD_ARG, "pf_dense_tn: rows of G and A must start on 16-byte boundaries");
    return finish(pf::dense_tn(precision == PF_PREC_BF16, *a, static_cast<hipStream_t>(stream)), "");
}
float pf_dropout_factor(float p, uint32_t seed, uint32_t site, uint32_t index) {
    if (!(p > 0.f)) return 1.f;
    return pf::enc_drop_hash(seed, site, index) >= pf::enc_drop_threshold(p) ? 1.f / (1.f - p) : 0.f;
}
#define PF_ENC_ENTRY(name, fn, T)                                                                              \
    int name(int32_t precision, const T* a, void* stream) {                                                    \
        if (!prec2(precision) || !a) return fail(PF_ERR_BAD_ARG, #name ": null args or bad precision");      \
        return finish(pf::fn(precision == PF_PREC_BF16, *a, static_cast<hipStream_t>(stream)), #name ": 1 <= T <= 192"); \
    }
PF_ENC_ENTRY(pf_enc_ln_forward, ln_forward, PfLnArgs)
PF_ENC_ENTRY(pf_enc_ln_backward, ln_backward, PfLnArgs)
PF_ENC_ENTRY(pf_enc_attn_forward, attn_forward, PfAttnArgs)
PF_ENC_ENTRY(pf_enc_attn_backward, attn_backward, PfAttnArgs)
PF_ENC_ENTRY(pf_enc_pool_forward, pool_forward, PfPoolArgs)
PF_ENC_ENTRY(pf_enc_pool_backward, pool_backward, PfPoolArgs)
#undef PF_ENC_ENTRY

// transposed context weights of ALL layers as ONE packed matrix P[c][(layer, j, hidden unit)] = W_{layer, j}[unit][c]
// (j = 0: MADE context layer, 1 / 2: the gates of the residual blocks): the weight operand of the context gradient
// g_ctx[row][c] = sum_{layer, j, unit} Gc[layer][j][row][unit] W_{layer, j}[unit][c]  (pf_dense_nt over 3 L slabs of Gc)
namespace {
int ctx_t_plan(const PfFlowDesc* desc, pf::FlowPlan& L) {
    const int rc = layout_of(desc, L);
    if (rc != PF_OK) return rc;
    if (L.C <= 0 || L.C % 16 || L.NB != 2 || 3 * L.L > pf::kMaxPackEntries)
        return fail(PF_ERR_UNSUPPORTED, "context gradient GEMM: plain conditioner, C % 16 == 0, at most 16 layers");
    return PF_OK;
}
}  // namespace
int64_t pf_flow_ctx_transposed_bytes(const PfFlowDesc* desc) {
    pf::FlowPlan L;
    if (ctx_t_plan(desc, L) != PF_OK) return -1;
    return pf::dense_frag_count(L.bf16 != 0, L.C, 3 * L.L * L.H) * 16;
}
int pf_flow_pack_ctx_transposed(const PfFlowDesc* desc, const float* raw, void* out, void* stream) {
    pf::FlowPlan L;
    const int rc = ctx_t_plan(desc, L);
    if (rc != PF_OK) return rc;
    if (!raw || !out || misaligned(out, 16)) return fail(PF_ERR_BAD_ARG, "pf_flow_pack_ctx_transposed: null or misaligned pointer");
    const int kstep = L.bf16 ? 32 : 16;
    const int64_t ctxp = (int64_t)L.H * L.C + L.H, hh = 2 * ((int64_t)L.H * L.H + L.H);
    pf::DensePackTable tab{};
    for (int l = 0; l < L.L; ++l)
        for (int j = 0; j < 3; ++j) {
            pf::DensePackEntry& e = tab.e[tab.n++];
            e.src_off = (int64_t)l * L.rawPerLayer + (int64_t)L.H * L.D + L.H + (j == 0 ? 0 : ctxp + (j - 1) * (ctxp + hh));
            e.dst_off = 0; e.mode = 1; e.ld = L.C; e.N = L.C; e.K = L.H;
            e.nks_total = 3 * L.L * (L.H / kstep); e.ks_off = (3 * l + j) * (L.H / kstep);
        }
    return finish(pf::dense_pack(L.bf16 != 0, raw, tab, out, static_cast<hipStream_t>(stream)), "");
}

int pf_flow_ctx_project_rows(int32_t precision, const void* wfrags, const float* bias, const float* ctx, int64_t rows,
                             int32_t context_features, int32_t n_units, float* out, void* stream) {
    if (!prec2(precision)) return fail(PF_ERR_BAD_ARG, "pf_flow_ctx_project_rows: bad precision");
    if (rows < 0 || context_features <= 0 || n_units <= 0 || n_units % 16)
        return fail(PF_ERR_BAD_ARG, "pf_flow_ctx_project_rows: rows >= 0, context_features > 0, n_units a positive multiple of 16");
    if (rows == 0) return PF_OK;
    if (!wfrags || !bias || !ctx || !out) return fail(PF_ERR_BAD_ARG, "pf_flow_ctx_project_rows: null pointer");
    if (misaligned(wfrags, 16) || misaligned(bias, 16) || misaligned(out, 16) || misaligned(ctx, 4))
        return fail(PF_ERR_BAD_ARG, "pf_flow_ctx_project_rows: wfrags / bias / out must be 16-byte aligned");
    return finish(pf::ctx_project_rows(precision == PF_PREC_BF16, wfrags, bias, ctx, rows, context_features, n_units, out,
                                       static_cast<hipStream_t>(stream)), "context width beyond the LDS tile (<= 1280 bf16 / 640 fp32)");
}

int64_t pf_flow_issued_flop_per_row(const PfFlowDesc* desc) {
    pf::FlowPlan L;
    if (compute_layout_of(desc, L) != PF_OK) return -1;
    // every 1-KiB fragment of the stream is multiplied with each row once: bf16 16 units x 32 k (wide: 32 x 16) = 512 MAC,
    // fp32 16 x 16 = 256 MAC; the zero padding behind each wave's stream (kWindowPad) is never multiplied
    const int64_t frags = (L.wide || L.generic) ? L.fragsTotal : (int64_t)L.L * L.NF * L.NW;
    return frags * (L.bf16 ? 1024 : 512);
}

int32_t pf_flow_rows_per_workgroup(const PfFlowDesc* desc, int64_t batch) {
    pf::FlowPlan L;
    if (compute_layout_of(desc, L) != PF_OK) return -1;
    return pf::rows_per_workgroup(L, batch);
}

int pf_diag_stream_ingest(const void* buf, int64_t bytes, int32_t workgroups, int32_t in_flight, uint32_t* sink, void* stream) {
    if (!buf || !sink || misaligned(buf, 16) || bytes < 128 * 1024 || workgroups < 1 || workgroups > 65535)
        return fail(PF_ERR_BAD_ARG, "pf_diag_stream_ingest: 16-byte aligned buffer of >= 128 KiB, 1..65535 workgroups");
    if (in_flight != 2 && in_flight != 4 && in_flight != 8 && in_flight != 16)
        return fail(PF_ERR_BAD_ARG, "pf_diag_stream_ingest: in_flight must be 2, 4, 8 or 16");
    return finish(pf::diag_stream_ingest(buf, bytes, workgroups, in_flight, sink, static_cast<hipStream_t>(stream)), "");
}

int pf_geom_twiddles(float* host_table) {
    if (!host_table) return fail(PF_ERR_BAD_ARG, "null pointer");
    pf::geom_twiddles(host_table);
    return PF_OK;
}

int pf_geom_features(const PfGeomArgs* a, void* stream) {
    if (!a || !a->clean || !a->twiddle || !a->spec || !a->etot || !a->rel) return fail(PF_ERR_BAD_ARG, "null pointer");
    if (a->batch < 0) return fail(PF_ERR_BAD_ARG, "negative batch");
    if (misaligned(a->clean, 16) || misaligned(a->twiddle, 8) || misaligned(a->spec, 8))
        return fail(PF_ERR_BAD_ARG, "clean must be 16-byte, twiddle / spec 8-byte aligned");
    if (a->n_det < 1 || a->n_det > 8 || a->n_bands < 1 || a->n_bands > 16 || a->band_lo < 1 || a->nf < 1 ||
        a->band_lo + a->nf > 4096 || a->maxlag < 1 || a->maxlag > 127)
        return fail(PF_ERR_UNSUPPORTED, "geometry features: n_det <= 8, n_bands <= 16, 1 <= band_lo, band_lo + nf <= 4096, maxlag <= 127");
    if (a->band_edge[0] < 0 || a->band_edge[a->n_bands] > a->nf) return fail(PF_ERR_BAD_ARG, "band edges outside the kept bins");
    for (int b = 0; b < a->n_bands; ++b)
        if (a->band_edge[b] > a->band_edge[b + 1]) return fail(PF_ERR_BAD_ARG, "band edges must be non-decreasing");
    if (a->batch == 0) return PF_OK;
    if (a->batch * (int64_t)a->n_det * (a->n_det > 1 ? (a->n_det - 1) : 1) > 0x7fffffffLL) return fail(PF_ERR_UNSUPPORTED, "batch too large for one launch");
    const int rc = pf::geom_features(*a, static_cast<hipStream_t>(stream));
    return rc == PF_OK ? rc : fail(rc, hipGetErrorString(static_cast<hipError_t>(pf::g_hip_error)));
}
}  // extern "C"
