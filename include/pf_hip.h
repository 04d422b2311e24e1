/* pf_hip.h -- C ABI of libpfhip.so: the MI355X (gfx950) hot path of PosteriFlow's
 * conditional neural-spline flow and strain embedding.
 *
 * The reference (bibinthomas123/PosteriFlow) has no FFI: its seam is the Python
 * nn.Module API.  Each entry point below names the reference method whose
 * arithmetic it replaces (paths relative to the reference repo):
 *
 *   pf_flow_forward   NSFPosteriorFlow.forward               src/ahsd/models/flows.py:610-618
 *                     + compute_psd_aware_nll (N(0,I) base)  src/ahsd/models/flows.py:727-779
 *                     (executes nflows CompositeTransform / MADE / RQS, flows.py:459-529)
 *   pf_flow_inverse   NSFPosteriorFlow.inverse (transform part) src/ahsd/models/flows.py:620-655
 *   pf_flow_pack      the per-call `weight * mask` of nflows MaskedLinear, done once
 *   pf_embed_fusion_forward  LeanStrainEncoder fusion transformer + pool attention  lean_npe.py:226-229
 *   pf_remix_forward  RemixDataset.__getitem__ (algebra)      experiments/remix_data.py:218-299
 *   pf_geom_features  CoherentEncoder._geometry_rel           src/ahsd/models/coherent_encoder.py:79-116
 *   pf_embed_train_forward / _backward  LeanStrainEncoder._compute_feats under autograd (training)  lean_npe.py:199-233,
 *                     experiments/train_lean_npe.py:363-364
 *   pf_embed_stem_forward  LeanStrainEncoder stem + energy windows  src/ahsd/models/lean_npe.py:207-217
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer unless named *_host;
 *     the library never allocates or frees device memory;
 *   - tensors are contiguous row-major fp32: x[B,D], ctx[B,C], z[B,D], logdet[B];
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*), no
 *     internal synchronisation, no global mutable state;
 *   - return value: 0 ok, PF_ERR_* (<0) otherwise; no C++ exception crosses
 *     the ABI; pf_last_error() returns a thread-local message.
 */
#ifndef PF_HIP_H
#define PF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK 0
#define PF_ERR_BAD_ARG (-1)      /* null pointer, negative size, misaligned pointer */
#define PF_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels are built for */
#define PF_ERR_HIP (-3)          /* HIP runtime error at launch */

#define PF_PREC_F32 0   /* f32-input MFMA (v_mfma_f32_16x16x4_f32): exact fp32, parity mode */
#define PF_PREC_BF16 1  /* bf16 MFMA operands, fp32 accumulate: throughput mode */

/* PfFlowDesc.reserved flag bits */
#define PF_FLAG_HOIST_CTX 1  /* evaluate the context projections of all layers once, up front,
                              * into the caller's workspace (pf_flow_workspace_bytes) */
#define PF_FLAG_MASKED_CONTEXT 2 /* the reference's masked-context conditioner (flows.py:112-360,
                              * full_context=True): context added between the two masked linears of
                              * a block instead of the GLU gate, no ReversePermutation between
                              * layers; implies PF_FLAG_HOIST_CTX.  Raw layout unchanged (the block's
                              * context_layer weight/bias take the place of the gate's). */

#define PF_FLAG_WIDE 4        /* forward only: the layout of the two kernels for many rows -- the large-batch kernel
                              * (128 rows per workgroup, every wave owns 32 rows through all layers, weights fetched
                              * once per workgroup through an LDS ring) and, since round 4, the mid-batch kernel (64 rows
                              * per workgroup, two waves per SIMD, a wave owns one hidden tile for both 32-row blocks);
                              * pf_flow_forward picks per call by rounds x measured round time (pf_flow_forward_kernel_name
                              * / pf_flow_rows_per_workgroup tell which; $PF_FLOW_MID = 0 / 1 forces).  bf16,
                              * H = 256, K = 16, plain conditioner, (D, C) in {(15, 288), (11, 288)}; other shapes:
                              * PF_ERR_UNSUPPORTED from every entry point.  The packed buffer of a PF_FLAG_WIDE desc
                              * has its own layout (pf_flow_packed_bytes / pack_map / pack with the same desc). */

#define PF_FLAG_BWD 8         /* packing only: the weight stream of the bf16 backward, bf16 MFMA A-fragments of 1 KiB per
                              * (16-unit tile, 32-wide k-step), nflows unit order, masks folded in:
                              *  transposed region (pf_flow_backward_chain), per layer: WfT [H/16][ceil(D(3K-1)/32)] |
                              *    for block j: W2T_j, W1T_j [H/16][H/32] | W0T [H/32];
                              *  forward region (pf_flow_reevaluate), per layer: Win [H/16][1] (x as hi | lo) | Wc, Wg0,
                              *    Wg1 [H/16][ceil(C/32)] (if C > 0) | W1_0, W2_0, W1_1, W2_1 [H/16][H/32] |
                              *    Wf [ceil(D(3K-1)/16)][H/32];
                              *  then fp32 biases per layer: b_in | bc, bg0, bg1 | b1_0, b2_0, b1_1, b2_1 | bf (padded to 16).
                              * bf16 precision, plain conditioner.  pf_flow_packed_bytes / pack_map_len / build_pack_map /
                              * pack take the flag; the forward, inverse and workspace entry points refuse it. */
#define PF_FLAG_GENERIC 16    /* request the generic kernel's layout (plain [tile][k-step] fragment arrays of the dense masked
                              * matrices, nflows unit order) for a shape the scheduled kernels would take: the fp32-mode
                              * conditioner re-evaluation (pf_flow_reevaluate with an fp32 desc) reads this layout; forward and
                              * inverse calls with such a desc run the generic kernel.  Shapes outside the scheduled set get the
                              * generic kernel without the flag, and then with the hidden units stored in autoregressive-degree
                              * order (block lower-triangular masks: the kernel skips the zero k-steps); a packed buffer belongs to
                              * the desc it was packed for (same flags). */

/* Plain-old-data description of one NSFPosteriorFlow (flows.py:379-548).
 * conditioner: nflows MADE, num_blocks residual blocks with GLU context gate,
 * ReversePermutation in front of every layer, tails='linear'. */
typedef struct PfFlowDesc {
    int32_t features;          /* D  (1..H/16)                                 */
    int32_t context_features;  /* C  (0 = unconditional)                       */
    int32_t hidden_features;   /* H  (64, 128, 192 or 256 in this build)       */
    int32_t num_bins;          /* K  (2..16)                                   */
    int32_t num_layers;        /* L                                            */
    int32_t num_blocks;        /* residual blocks per MADE (2)                 */
    float tail_bound;          /* spline domain [-tail_bound, tail_bound]      */
    float min_bin_width;       /* nflows DEFAULT_MIN_BIN_WIDTH  = 1e-3         */
    float min_bin_height;      /* nflows DEFAULT_MIN_BIN_HEIGHT = 1e-3         */
    float min_derivative;      /* nflows DEFAULT_MIN_DERIVATIVE = 1e-3         */
    int32_t precision;         /* PF_PREC_*                                    */
    int32_t reserved;          /* flags: PF_FLAG_*                             */
} PfFlowDesc;

/* ---- supported shapes ------------------------------------------------------
 * Scheduled kernels (static weight streams, the numbers of bench.py): hidden_features in {64, 128, 192, 256},
 * num_bins <= 16, features <= hidden_features / 16, num_blocks = 2.  Any other plain-conditioner shape with
 * hidden_features % 16 == 0, <= 512, features <= 32, num_bins <= 32 whose workgroup image fits 160 KB of LDS (the reference
 * also builds 12 x 384 x 24 heads, experiments/frozen_context_heads.py:159-163) is served by one generic kernel behind the
 * SAME entry points: pf_flow_raw_param_count / packed_bytes / pack_map_len / build_pack_map / pack, pf_flow_forward,
 * pf_flow_forward_reduce, pf_flow_forward_train (layer inputs; no dropout), pf_flow_inverse, and -- fp32 descs, features <= 16,
 * hidden_features in {64, 128, 192, 256, 384, 512}, num_bins <= 32 -- pf_flow_reevaluate (PF_FLAG_GENERIC layout),
 * pf_flow_backward_chain and pf_flow_rqs_backward.  The incremental-inverse, large-batch and bf16 backward entry points return
 * PF_ERR_UNSUPPORTED for such a shape.
 * Masked-context conditioner (PF_FLAG_MASKED_CONTEXT): forward / inverse on the scheduled kernels (context projections
 * hoisted) or the generic kernel; backward through the SAME fp32 entry points in their additive form -- pf_flow_reevaluate
 * (fp32 desc, PF_FLAG_GENERIC | PF_FLAG_MASKED_CONTEXT layout; gates = NULL) and pf_flow_backward_chain (fp32 desc with the
 * flag; t2s = gates = NULL, Gc[l][1 + j] = dL/d(block j's context projection) = Gt1[j][l]). */

/* ---- raw parameter layout -------------------------------------------------
 * One flat fp32 buffer, layer after layer, each layer in nflows state_dict
 * order (SURVEY.md 8a "state_dict layout"):
 *   initial_layer.weight[H,D] .bias[H]   context_layer.weight[H,C] .bias[H]
 *   for b in blocks: context_layer.weight[H,C] .bias[H]
 *                    linear_layers.0.weight[H,H] .bias[H]  linear_layers.1.weight[H,H] .bias[H]
 *   final_layer.weight[D*(3K-1),H] .bias[D*(3K-1)]
 * (context tensors absent when C == 0).  Masks are NOT applied by the caller. */
int64_t pf_flow_raw_param_count(const PfFlowDesc* desc);

/* ---- packed weights ---------------------------------------------------------
 * The kernels read weights pre-masked, cast to the MFMA operand type and laid
 * out as per-wave streams of MFMA A-fragments.  Packing is a device gather
 *     packed[i] = map[i] < 0 ? 0 : cast(raw[map[i]])
 * driven by an index map the library builds on the HOST once per desc. */
int64_t pf_flow_packed_bytes(const PfFlowDesc* desc);      /* size of the packed buffer        */
int64_t pf_flow_pack_map_len(const PfFlowDesc* desc);      /* number of int32 entries in map   */
int pf_flow_build_pack_map(const PfFlowDesc* desc, int32_t* map_host);
int pf_flow_pack(const PfFlowDesc* desc, const float* raw, const int32_t* map,
                 void* packed, void* stream);

/* ---- workspace ----------------------------------------------------------------
 * Bytes of caller-owned device scratch pf_flow_forward / pf_flow_inverse need for a
 * call with `ctx_rows` context rows (0 unless PF_FLAG_HOIST_CTX is set). */
int64_t pf_flow_workspace_bytes(const PfFlowDesc* desc, int64_t ctx_rows);

/* ---- forward / density ------------------------------------------------------
 * z, logdet = transform(x[:, ar_perm], ctx)
 * nll = -(log N(z; 0, diag(exp(log_sigma))^2) + logdet)   (PSDScaledNormal, flows.py:56-85)
 * ar_perm: int32[D] device pointer or NULL (identity)  (flows.py:612).
 * log_sigma: [B,D] or NULL (= zeros, what LeanNPE passes, lean_npe.py:315).
 * Any of z / logdet / nll may be NULL.  batch may be 0 (no-op). */
int pf_flow_forward(const PfFlowDesc* desc, const void* packed,
                    const float* x, const float* ctx, const int32_t* ar_perm,
                    const float* log_sigma, int64_t batch,
                    float* z, float* logdet, float* nll,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- training: forward that keeps what the backward needs, and the spline backward ----------
 * pf_flow_forward_train = pf_flow_forward that additionally writes layer_inputs[L][batch][D]: the
 * input of every layer's conditioner (after the layer's ReversePermutation), i.e. what nflows'
 * autograd graph would have kept alive under flows.py:615-617.  layer_inputs may be NULL.
 * pf_flow_rqs_backward: hand-derived backward of the rational-quadratic spline with linear tails
 * for rows x D (input, raw parameters [rows][D][3K-1] in nflows order) pairs: given dL/dy [rows][D]
 * and dL/dlogabsdet [rows] returns dL/d(raw parameters) and the direct part of dL/du.  Uses only
 * features, num_bins, tail_bound and the min_* fields of desc. */
int pf_flow_forward_train(const PfFlowDesc* desc, const void* packed,
                          const float* x, const float* ctx, const int32_t* ar_perm,
                          const float* log_sigma, int64_t batch,
                          float* z, float* logdet, float* nll, float* layer_inputs,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* pf_flow_forward_train_dropout = pf_flow_forward_train in TRAIN mode of a flow built with dropout_probability > 0
 * (src/ahsd/models/flows.py:522 passes it to nflows; create_flow_model defaults to 0.15, flows.py:1008): nflows'
 * MaskedResidualBlock applies dropout after the second activation of every residual block.  The keep decision of
 * (row, layer, block, hidden unit) is a counter hash of dropout_seed -- kept values are scaled by 1 / (1 - p) -- so the
 * backward regenerates the factors with pf_flow_dropout_mask: mask[2][L][batch][H] (block, layer, row, unit in nflows
 * order), each 0 or 1 / (1 - p).  dropout_p = 0 is pf_flow_forward_train.  Not available with PF_FLAG_WIDE. */
int pf_flow_forward_train_dropout(const PfFlowDesc* desc, const void* packed,
                                  const float* x, const float* ctx, const int32_t* ar_perm,
                                  const float* log_sigma, int64_t batch,
                                  float* z, float* logdet, float* nll, float* layer_inputs,
                                  float dropout_p, uint64_t dropout_seed,
                                  void* workspace, int64_t workspace_bytes, void* stream);
int pf_flow_dropout_mask(const PfFlowDesc* desc, float dropout_p, uint64_t dropout_seed, int64_t batch,
                         float* mask, void* stream);
/* pf_flow_forward_reduce: pf_flow_forward for a loss -- nll (may be NULL) as there, and the kernel adds (sum of nll over
 * the batch, batch) to the accumulator nll_sum_count = float[PF_REDUCE_SLOTS][2]: workgroup b adds its rows' pair to slot
 * b mod PF_REDUCE_SLOTS (float atomics after a wave shuffle reduction; spread over slots because 256 workgroups ending
 * together on one address pair serialise), so (sum, count) = the column sums over the slots.  The 128-byte vector is what a
 * data-parallel rank all-reduces (train_lean_npe.py:108-127 computes sum / count on the host).  The accumulator must be
 * zero before the launch: either the caller zeroes it, or an EARLIER launch did through zero_pair (an accumulator of the
 * same size or NULL, set to zero by this launch; must differ from nll_sum_count) -- with three rotating accumulators a step
 * of a data-parallel loop is exactly one kernel launch and one asynchronous all-reduce. */
#define PF_REDUCE_SLOTS 16
int pf_flow_forward_reduce(const PfFlowDesc* desc, const void* packed,
                           const float* x, const float* ctx, const int32_t* ar_perm,
                           const float* log_sigma, int64_t batch, float* nll, float* nll_sum_count,
                           float* zero_pair, void* workspace, int64_t workspace_bytes, void* stream);
int pf_flow_rqs_backward(const PfFlowDesc* desc, const float* u, const float* params,
                         const float* grad_y, const float* grad_logabsdet, int64_t rows,
                         float* grad_params, float* grad_u, void* stream);

/* ---- backward: the data-gradient chain in one launch (fp32) -----------------------------------------
 * Replaces the per-layer walk of the flow's backward (experiments/train_lean_npe.py:363-368 runs autograd through
 * nflows' MADE + spline): for every layer, last first: spline backward, gh = Wf^T Gp, through the two residual blocks
 * (GLU gates), Gh0, gu += W0^T gh, ReversePermutation.  The caller supplies the forward quantities of every layer (the
 * layer inputs pf_flow_forward_train kept and the layer-batched re-evaluation of the conditioners) and the TRANSPOSED
 * masked weight matrices; the kernel returns the per-layer gradients the weight-gradient GEMMs need and dL/dx.
 * Plain (GLU) conditioner with 2 residual blocks only; context-free flows pass t2s = gates = pc = Gc = NULL.
 * All tensors fp32, contiguous.  H in {64, 128, 192, 256}, D <= 16, K <= 16. */
typedef struct PfFlowBwdChainArgs {
    int64_t batch;
    const float* WfT;    /* [L][H][PM]: (final_layer.weight * mask)^T, PM = D (3K-1) rounded up to 16, zero padded */
    const float* W2T;    /* [2][L][H][H]: (blocks[j].linear_layers[1].weight * mask)^T  ([in][out]) */
    const float* W1T;    /* [2][L][H][H]: (blocks[j].linear_layers[0].weight * mask)^T */
    const float* W0T;    /* [L][16][H]: (initial_layer.weight * mask)^T, rows >= D zero */
    const float* U;      /* [L][B][D]      layer inputs (pf_flow_forward_train) */
    const float* params; /* [L][B][D (3K-1)] raw spline parameters */
    const float* hs;     /* [2][L][B][H]   residual state before block j */
    const float* t1s;    /* [2][L][B][H]   W0_j relu(h) + b (pre-activation of the block's second linear) */
    const float* t2s;    /* [2][L][B][H]   second linear's output (before the gate), or NULL */
    const float* gates;  /* [2][L][B][H]   sigmoid of the block's context projection, or NULL */
    const float* pc;     /* [L][B][H]      context_layer projection (before its ReLU), or NULL */
    const float* g_z;    /* [B][D]  dL/dz (layer order, as the forward returns z before un-permuting) */
    const float* g_lad;  /* [B]     dL/dlogdet */
    /* ... or, for the loss of compute_psd_aware_nll (flows.py:727-779), g_z = g_lad = NULL and the three below: the kernel
     * forms dL/dz = g_nll z exp(-2 log_sigma) and dL/dlogdet = -g_nll itself (log_sigma NULL: the N(0, I) base) */
    const float* g_nll;      /* [B] dL/dnll, or NULL */
    const float* nll_z;      /* [B][D] z as the forward returned it */
    const float* log_sigma;  /* [B][D] or NULL */
    float* Gp;           /* [L][B][D (3K-1)]  dL/d(raw spline parameters) */
    float* Gh0;          /* [L][B][H]  dL/d(initial layer output) */
    float* Gt1;          /* [2][L][B][H] */
    float* Gt2;          /* [2][L][B][H] */
    float* Gc;           /* [L][3][B][H]  dL/d(context projections): context_layer, gate of block 0, of block 1 (masked-context
                          * descs: the additive projection of block 0, of block 1); or NULL */
    float* g_x;          /* [B][D]  dL/d(x[:, ar_perm]) */
    const float* drop;   /* [2][L][B][H] dropout factors of the forward (pf_flow_dropout_mask), or NULL:
                          * gt1 = (W2^T gt2) . drop . [t1 > 0] */
    uint32_t compact;    /* bf16 descs only, 1: the activation tensors are the bf16 ones pf_flow_reevaluate writes in its compact
                          * mode (hs = relu(h_j), t1s = relu(t1_j) . dropout factor, t2s, gates, pc = relu(pc): the chain only
                          * needs their signs and products) and Gp, Gh0, Gt1, Gt2, Gc are written as bf16 -- the operands of
                          * bf16 weight-gradient GEMMs; `drop` is not read (its factor is drop_scale where t1s > 0).  0: fp32. */
    float drop_scale;    /* compact: 1 / (1 - p) of the forward's dropout, 1 without */
    const void* packed;  /* bf16 descs: the PF_FLAG_BWD stream (W*T above are ignored): the transposed GEMMs run on bf16
                          * MFMA with bf16-rounded gradient vectors as their second operand; the spline, the accumulators,
                          * the gate / ReLU algebra and every output stay fp32.  fp32 descs: ignored. */
    uint32_t gp_ld;      /* row stride of Gp in elements; 0: D (3K-1).  A stride that is a multiple of 8 (compact) lets the rows
                          * leave as 16-byte stores and be the operand of pf_dense_tn without a padding copy; elements
                          * D (3K-1) .. gp_ld - 1 of every row are written as zeros */
} PfFlowBwdChainArgs;
int pf_flow_backward_chain(const PfFlowDesc* desc, const PfFlowBwdChainArgs* args, void* stream);

/* ---- backward: the conditioner re-evaluation in one launch -------------------------------------------------------------
 * Recomputes, from the layer inputs pf_flow_forward_train kept, the activations of every layer's MADE that the chain and the
 * weight-gradient GEMMs read -- what autograd keeps alive for the reference under flows.py:615-617.  Grid = 16-row blocks x
 * layers (the layers are independent given their inputs); bf16 operands, x as hi + lo, fp32 accumulate: the arithmetic of the
 * bf16 forward kernel.  `packed` is the PF_FLAG_BWD stream (its forward region).  Plain conditioner, bf16 desc, H % 32 == 0.
 * Outputs fp32, nflows unit order; context-free flows pass ctx = t2s = gates = pc = NULL.
 * fp32 descs (the parity mode): the same function in exact-fp32 MFMA arithmetic by the generic kernel's conditioner
 * (csrc/pf_flow_generic.hip); `packed` is then the packed buffer of the desc with PF_FLAG_GENERIC set, `compact` must be 0. */
typedef struct PfFlowReevalArgs {
    int64_t batch;
    const void* packed;  /* PF_FLAG_BWD stream */
    const float* U;      /* [L][B][D] layer inputs */
    const float* ctx;    /* [B][C] or NULL */
    float* hs;           /* [2][L][B][H] residual state before block j */
    float* t1s;          /* [2][L][B][H] */
    float* t2s;          /* [2][L][B][H] or NULL */
    float* gates;        /* [2][L][B][H] or NULL */
    float* pc;           /* [L][B][H] or NULL */
    float* h2;           /* [L][B][H] residual state after the last block (input of the final layer) */
    float* params;       /* [L][B][D (3K-1)] raw spline parameters */
    const float* drop;   /* [2][L][B][H] dropout factors the training forward applied (pf_flow_dropout_mask), or NULL:
                          * the second linear of block j sees relu(t1_j) . drop[j] */
    uint32_t compact;    /* 1: hs, t1s, t2s, gates, pc, h2 are written as bf16 and already in the form the backward uses them:
                          * hs = relu(h_j), t1s = relu(t1_j) . drop[j], pc = relu(pc) (params stay fp32): half the bytes, and
                          * the operands of bf16 weight-gradient GEMMs without a cast.  0: fp32 raw values as described above. */
} PfFlowReevalArgs;
int pf_flow_reevaluate(const PfFlowDesc* desc, const PfFlowReevalArgs* args, void* stream);

/* ---- backward: the context gradient's weight operand ---------------------------------------------------------------------
 * All layers' context weights transposed and packed as ONE matrix P[c][(layer, j, unit)] = W_{layer, j}[unit][c] (j = 0: MADE
 * context_layer, 1 / 2: the residual blocks' gates) in pf_dense_nt's fragment format, so that
 *   g_ctx[row][c] = sum_{layer, j, unit} Gc[layer][j][row][unit] W_{layer, j}[unit][c]
 * is pf_dense_nt over the 3 L slabs of pf_flow_backward_chain's Gc (PfDenseArgs.a_chunk_stride = batch * H, KC = H).
 * Plain conditioner, C % 16 == 0, L <= 16; -1 / PF_ERR_UNSUPPORTED otherwise.  desc->precision selects the operand type. */
int64_t pf_flow_ctx_transposed_bytes(const PfFlowDesc* desc);
int pf_flow_pack_ctx_transposed(const PfFlowDesc* desc, const float* raw, void* out, void* stream);

/* ---- inverse / sampling -----------------------------------------------------
 * x = transform^-1(z, ctx)[:, ar_inv_perm], logdet of the inverse map
 * (nflows returns the log-det of the last autoregressive pass of each layer,
 * which is the exact inverse log-det; flows.py:637).
 * ctx_rows == batch: one context row per sample; ctx_rows divides batch:
 * sample i uses context row i / (batch / ctx_rows)  (the expand().reshape()
 * pattern of lean_npe.py:328, pipeline.py:171).
 * fail_flags (uint32[batch] or NULL): bit 0 set where the quadratic
 * discriminant was negative (the reference's AssertionError, flows.py:638). */
int pf_flow_inverse(const PfFlowDesc* desc, const void* packed,
                    const float* z, const float* ctx, int64_t ctx_rows,
                    const int32_t* ar_inv_perm, int64_t batch,
                    float* x, float* logdet, uint32_t* fail_flags,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- incremental inverse -------------------------------------------------------------------------
 * Same result as pf_flow_inverse (the transform.inverse call of flows.py:637), with ONE masked conditioner
 * evaluation per layer instead of D dense ones: pass i computes only the hidden units of degree i (2-3 tiles per
 * hidden stage, hidden units sorted by degree) from the activations of the earlier passes, which stay in LDS.
 * 16 draws per workgroup, three workgroups per CU.  Returns PF_ERR_UNSUPPORTED when more than 8 sixteen-unit tiles hold
 * the hidden units of one degree (H / (D - 1) > 112: use pf_flow_inverse).  The caller prepares, per layer l (bytes:
 * pf_flow_inc_layer_bytes):
 *   A fragments (pf_pack_bf16_frags of the masked weights, rows / columns of hidden units in degree-sorted
 *   order): W0 [H][32] (columns f and 16 + f both = initial_layer.weight[:, f]: the input enters as a bf16
 *   hi | lo pair), W1 / W2 of block 0, W1 / W2 of block 1 [H][H], final layer [D * 48][H] (per feature 16
 *   width rows, 16 height rows, 16 derivative rows, zero padded), then fp32: b0 | b1_0 | b2_0 | b1_1 | b2_1
 *   (H each, sorted) | final bias [D][48];
 *   ctx_proj [ctx_rows][L][3][H] fp32 = context_layer / block-0 / block-1 context projections (with their
 *   biases, before ReLU / sigmoid), sorted-unit order, or NULL for a context-free flow;
 *   units_upto_degree (host int32[D + 1]): number of hidden units with degree <= i.
 * PF_PREC_F32 (parity mode, v_mfma_f32_16x16x4_f32): the same blocks with fp32 fragments of 16 units x 16 k --
 *   fragment (tile, q), lane (r = lane & 15, kq = lane >> 4): W[16 tile + r][16 q + 4 kq .. + 3] -- , W0 [H][16]
 *   (column f = initial_layer.weight[:, f]); the caller packs them (a reshape / permute, posteriflow_amd/flows.py).
 * z, x, logdet, fail_flags, ar_inv_perm, ctx_rows as in pf_flow_inverse. */
int pf_pack_bf16_frags(const float* src, int32_t n_rows, int32_t k, void* out, void* stream);
int64_t pf_flow_inc_layer_bytes(const PfFlowDesc* desc);
/* ctx_proj above, made on the GPU: out[r][n] = sum_c ctx[r][c] W[n][c] + bias[n] (fp32, row-major [rows][n_units]) for
 * W [n_units][context_features] packed by pf_dense_pack_matrix(precision, W, 0, context_features, n_units, K) with K =
 * context_features rounded up to 32 (bf16) / 16 (fp32) and zero columns beyond.  PF_PREC_BF16: context and weights enter the
 * MFMA as bf16 (the forward kernel's operand rounding), fp32 accumulation.  The kernel is the hoisted-projection GEMM of
 * the D-pass inverse (csrc/pf_flow_ctx.hip) with a row-major epilogue; replaces the three context_layer calls nflows makes
 * per MADE pass (reference call site src/ahsd/models/flows.py:637). */
int pf_flow_ctx_project_rows(int32_t precision, const void* wfrags, const float* bias, const float* ctx, int64_t rows,
                             int32_t context_features, int32_t n_units, float* out, void* stream);
int pf_flow_inverse_inc(const PfFlowDesc* desc, const int32_t* units_upto_degree, const void* packed,
                        const float* ctx_proj, int64_t ctx_rows, const float* z, const int32_t* ar_inv_perm,
                        int64_t batch, float* x, float* logdet, uint32_t* fail_flags, void* stream);

/* ---- strain-embedding stem ------------------------------------------------------
 * tokens[N,61,192], log_energy[N,16] = stem(strain[N,16384]) for N = batch * n_detectors
 * sequences: replaces lean_npe.py:207 (sanitise), :210-212 (window log-energy) and :216-217
 * (asinh + 4 strided Conv1d + GELU, shared across detectors) of LeanStrainEncoder.
 * Raw parameters: flat fp32 [stem.0.weight, stem.0.bias, stem.2.weight, stem.2.bias,
 * stem.4.*, stem.6.*] (Conv1d layout [cout][cin][k]).  precision: PF_PREC_*.
 * workspace: pf_embed_stem_workspace_bytes(precision, N) bytes of device scratch. */
int64_t pf_embed_stem_raw_param_count(void);
int64_t pf_embed_stem_packed_bytes(int32_t precision);
int64_t pf_embed_stem_pack_map_len(int32_t precision);
int pf_embed_stem_build_pack_map(int32_t precision, int32_t* map_host);
int pf_embed_stem_pack(int32_t precision, const float* raw, const int32_t* map, void* packed, void* stream);
int64_t pf_embed_stem_workspace_bytes(int32_t precision, int64_t n_sequences);
int pf_embed_stem_forward(int32_t precision, const void* packed, const float* strain,
                          int64_t n_sequences, float* tokens, float* log_energy,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ---- strain-embedding token mixer ---------------------------------------------------------
 * The 3 pre-norm Transformer layers (d_model 192, 6 heads, FFN 768, GELU) over an event's tokens and
 * the key/value side of the 8-query attention pool, eval mode: replaces `self.fusion(tok)` and the
 * attention of `self.pool_attn(queries, tok, tok)` in LeanStrainEncoder._compute_feats
 * (src/ahsd/models/lean_npe.py:226-229; modules built at :167-176).  bf16 MFMA, fp32 accumulate.
 * tokens [n_events][n_tokens][192] fp32 (stem output, geometry tokens prepended for the coherent
 * encoder) is UPDATED IN PLACE to the Transformer output; token_bias [n_tokens][192] (or NULL) is added to
 * every event's tokens first: the positional + detector embedding of lean_npe.py:218-222;
 * pool_queries [8][192] = (pool_queries W_q^T + b_q) / sqrt(32) (input-independent, computed by the
 * caller); pooled [n_events][8][192] = concatenated heads of the pool attention BEFORE its out_proj.
 * Raw parameters: flat fp32, per layer l = 0..2: norm1.weight, norm1.bias, self_attn.in_proj_weight
 * [576,192], self_attn.in_proj_bias, self_attn.out_proj.weight [192,192], .bias, norm2.weight, norm2.bias,
 * linear1.weight [768,192], .bias, linear2.weight [192,768], .bias; then pool_attn.in_proj_weight[192:576]
 * ([384,192], the K and V rows) and pool_attn.in_proj_bias[192:576]. */
int64_t pf_embed_fusion_raw_param_count(void);
int64_t pf_embed_fusion_packed_bytes(void);
int pf_embed_fusion_pack(const float* raw, void* packed, void* stream);
int pf_embed_fusion_forward(const void* packed, float* tokens, int32_t n_tokens, const float* token_bias,
                            const float* pool_queries, int64_t n_events, float* pooled, void* stream);

/* ---- strain embedding: training path (forward that keeps what the backward needs, and the backward) --------------------
 * One C call each for a differentiable / train()-mode evaluation of LeanStrainEncoder._compute_feats up to the pooled
 * features (src/ahsd/models/lean_npe.py:199-233: stem, positional + detector embedding, optional extra tokens, 3 pre-norm
 * TransformerEncoderLayers with dropout, K / V side of the attention pool) and for its backward -- what autograd records
 * under experiments/train_lean_npe.py:363-364 for the reference.  Every kernel is enqueued on `stream`; nothing returns to
 * the host in between.
 * Raw parameters (fp32, flat; gradients come back in the same layout): stem.{0,2,4,6}.{weight,bias}; per layer l = 0..2:
 * norm1.{weight,bias}, self_attn.in_proj_{weight,bias}, self_attn.out_proj.{weight,bias}, norm2.{weight,bias},
 * linear1.{weight,bias}, linear2.{weight,bias}; pool_attn.in_proj_weight [576][192] and in_proj_bias [576] WHOLE (the query
 * rows are not read; their gradient is zero).
 * Dropout (nn.TransformerEncoderLayer: attention probabilities, both residual branches, the FFN's hidden layer) is a counter
 * hash of (dropout_seed, site, element): the backward regenerates the factors, nothing is stored.
 * extra_tokens [n_events][n_extra_tokens][192] or NULL; token_bias [n_extra + 61 n_det][192] or NULL (rows of extra tokens
 * are not read); pool_q [8][192] = projected, scaled pool queries as for pf_embed_fusion_forward;
 * pooled [n_events][8][192]; log_energy [n_events * n_det][16].
 * The workspace written by the forward must reach the backward unchanged.  grad_raw is zeroed and filled by the backward
 * (float atomics across workgroups: the last bits vary from run to run); grad_extra / grad_token_bias / grad_pool_q may be
 * NULL. */
typedef struct PfEmbedTrainDesc {
    int32_t precision;        /* PF_PREC_* (bf16: bf16 activations in HBM, bf16 MFMA operands, fp32 accumulate / residual / LayerNorm) */
    int32_t n_detectors;      /* 1..3 */
    int32_t n_extra_tokens;   /* tokens prepended per event (CoherentEncoder: 4), n_extra + 61 n_det <= 192 */
    int32_t training;         /* 1: dropout active */
    float dropout_p;
    int32_t forward_only;     /* 1: no backward will follow (a no-grad call): the workspace holds ONE layer's activations (the
                                 layers reuse them) and none of the backward's temporaries -- ~5.5 MB per 3-detector event in
                                 fp32 instead of ~16 MB; pf_embed_train_backward refuses such a desc */
    uint64_t dropout_seed;
} PfEmbedTrainDesc;
int64_t pf_embed_train_raw_param_count(void);
int64_t pf_embed_train_packed_bytes(int32_t precision);
int pf_embed_train_pack(int32_t precision, const float* raw, void* packed, void* stream);
int64_t pf_embed_train_workspace_bytes(const PfEmbedTrainDesc* desc, int64_t n_events);
int pf_embed_train_forward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* strain,
                           const float* extra_tokens, const float* token_bias, const float* pool_q, int64_t n_events,
                           float* pooled, float* log_energy, void* workspace, int64_t workspace_bytes, void* stream);
int pf_embed_train_backward(const PfEmbedTrainDesc* desc, const void* packed, const float* raw, const float* pool_q,
                            const float* grad_pooled, int64_t n_events, void* workspace, int64_t workspace_bytes,
                            float* grad_raw, float* grad_extra_tokens, float* grad_token_bias, float* grad_pool_q,
                            void* stream);

/* ---- the building blocks of the training path (also used by the flow's weight gradients), exported for tests ------------
 * "act type" = bf16 (PF_PREC_BF16) or fp32 (PF_PREC_F32).  Sequence-strided rows: row m of an operand = sequence
 * m / rows_per_seq, position m % rows_per_seq, at base + seq * seq_stride + pos * ld (elements) -- with ld < K the rows are
 * OVERLAPPING windows of a position-major activation array: a convolution's im2col matrix without materialising it. */
#define PF_EPI_PLAIN 0   /* out = act(v),                         v = sum_k A[m,k] W[n,k] + bias[n]                   */
#define PF_EPI_GELU 1    /* out = act(D gelu(v)), dact = act(D gelu'(v)),  D = dropout factor (1 when drop_p = 0)     */
#define PF_EPI_RESID 2   /* out (fp32) = resid + D v                                                                  */
#define PF_EPI_MUL 3     /* out = act(v * mul)                                                                        */
typedef struct PfDenseArgs {
    const void* A;            /* activations, act type */
    int64_t M;                /* rows */
    int64_t rows_per_seq;     /* = M for a plain matrix */
    int64_t a_seq_stride;
    int32_t lda;
    int32_t K, N, KC;         /* reduction length, output units (multiple of 16), k-chunk staged in LDS (divides K, multiple of 64,
                               * <= 256 after the library's own halving for LDS; K / KC > 1 needs N <= 256) */
    const void* wfrags;       /* W as MFMA A-fragments: pf_dense_pack_matrix */
    const float* bias;        /* [N] or NULL */
    void* out;                /* row m at out + seq * o_seq_stride + pos * ldo */
    int64_t o_seq_stride;
    int32_t ldo;
    int64_t o_valid_per_seq;  /* > 0: elements of a sequence's output that exist (transposed convolution: Lin * Cin) */
    int64_t x_seq_stride;     /* sequence stride of dact / resid / mul (0: o_seq_stride) */
    void* dact;               /* PF_EPI_GELU, or NULL */
    const float* resid;       /* PF_EPI_RESID */
    const void* mul;          /* PF_EPI_MUL, act type */
    float drop_p;
    uint32_t seed, site;      /* dropout factor of element (m, n) = hash(seed, site, m * N + n) */
    int32_t out_f32;          /* PF_EPI_PLAIN: fp32 output in either precision */
    int64_t a_chunk_stride;   /* > 0: k-chunk c of every row starts at A + c * a_chunk_stride (+ the row's offset) instead of
                               * c * KC elements into the row: the reduction runs over K / KC separate [rows][KC] slabs */
    int32_t a_slab_chunks;    /* > 1 (with a_chunk_stride): a slab holds this many consecutive chunks of KC -- chunk c starts at
                               * (c / a_slab_chunks) a_chunk_stride + (c % a_slab_chunks) KC (slabs wider than the LDS image allows) */
    int32_t k_splits;         /* > 1 (PF_EPI_PLAIN, out_f32, K / KC chunks): the chunks are divided over k_splits workgroups per
                               * strip which ADD their partial sums into out with float atomics -- out must hold zeros (or the
                               * value to accumulate onto); for few-row, long-reduction products that would leave CUs idle */
    int32_t n_group;          /* PF_EPI_PLAIN without dropout / o_valid_per_seq: the N output units are divided into groups of
                               * n_group (a multiple of 16; <= 256 when K / KC > 1) handled by separate workgroups of ONE launch,
                               * which lifts the N <= 256 limit of a chunked reduction and fills the chip when there are few
                               * rows -- deterministic (no atomics), at the price of staging the rows once per group.
                               * 0: chosen by the library (no grouping when the strips alone fill the chip) */
} PfDenseArgs;
typedef struct PfDenseTnArgs {  /* dW[n1][n2] += sum_m G[m][n1] A[m][n2],  db[n1] += sum_m G[m][n1]  (float atomics) */
    const void* G; int64_t g_seq_stride; int32_t ldg;
    const void* A; int64_t a_seq_stride; int32_t lda;
    int64_t M, rows_per_seq;
    int32_t N1, N2;           /* multiples of 8 (bf16) / 4 (fp32) */
    float* dW; int32_t ldw;
    int32_t conv_cin, conv_kw;   /* > 0: column n2 = tap * cin + ch lands at ch * kw + tap (Conv1d weight [cout][cin][kw]) */
    float* db;                /* or NULL */
    int32_t splits;           /* workgroups along M per output tile; <= 0: chosen by the library */
    /* batched form (e.g. the layers of the flow): problem z = 0 .. batch - 1 reads G + z g_batch_stride, A + z a_batch_stride
     * (elements) and accumulates into dW + z w_batch_stride, db + z b_batch_stride (floats); batch <= 1: one problem */
    int32_t batch;
    int64_t g_batch_stride, a_batch_stride, w_batch_stride, b_batch_stride;
    int32_t n1_rows, n2_cols; /* > 0: only rows < n1_rows / columns < n2_cols of dW (and db) exist -- N1 / N2 then describe the
                               * (zero-padded) operand widths only */
    const float* mask;        /* or NULL: dW[n1][col] += mask[n1 * ldw + col] * (the sum) -- one mask for every problem of a batch
                               * (the autoregressive masks of the flow's weights) */
} PfDenseTnArgs;
/* mode 0: W[n][k] = src[n * ld + k]; 1: W[n][k] = src[k * ld + n]; out: pf_dense_frag_bytes(precision, N, K) bytes */
int64_t pf_dense_frag_bytes(int32_t precision, int32_t N, int32_t K);
int pf_dense_pack_matrix(int32_t precision, const float* src, int32_t mode, int32_t ld, int32_t N, int32_t K, void* out, void* stream);
/* an nn.Linear weight [n][k] (row stride ld) zero-extended to [n_padded][k_padded] as fragments, in ONE launch and without
 * a padded copy: the forward form (mode 0) at out and, when with_transposed, W^T [k_padded][n_padded] (the data gradient's
 * operand) behind it at out + pf_dense_frag_bytes(precision, n_padded, k_padded) */
int pf_dense_pack_linear(int32_t precision, const float* weight, int32_t ld, int32_t n, int32_t k, int32_t n_padded,
                         int32_t k_padded, int32_t with_transposed, void* out, void* stream);
int pf_dense_nt(int32_t precision, int32_t epilogue, const PfDenseArgs* args, void* stream);
int pf_dense_tn(int32_t precision, const PfDenseTnArgs* args, void* stream);
/* dropout factor of the training path, for tests that rebuild the masks: 0 or 1 / (1 - p) */
float pf_dropout_factor(float p, uint32_t seed, uint32_t site, uint32_t index);

typedef struct PfLnArgs {       /* LayerNorm over 192 features, eps 1e-5 */
    const float* x; const float* gamma; const float* beta; int64_t M;
    void* y; float* mean; float* rstd;                      /* forward outputs: y act type [M][192] */
    const void* dy; const float* dres; float* dx;           /* backward: dx = dres + LN'(dy) */
    void* gout;                                             /* act(dx * dropout factor(seed, site, m * 192 + c)) or NULL */
    float* dgamma; float* dbeta;                            /* += (float atomics) */
    float drop_p; uint32_t seed, site;
} PfLnArgs;
int pf_enc_ln_forward(int32_t precision, const PfLnArgs* args, void* stream);
int pf_enc_ln_backward(int32_t precision, const PfLnArgs* args, void* stream);
typedef struct PfAttnArgs {     /* 6-head self-attention over T <= 192 tokens, head dimension 32 */
    const void* qkv;          /* [B * T][576] act type: q | k | v */
    int64_t B; int32_t T;
    void* out;                /* [B * T][192] act type: forward output (read by the backward) */
    float* lse;               /* [B][6][T] */
    float drop_p; uint32_t seed, site;   /* factor of (event e, head h, query q, key k) = hash(seed, site, ((e 6 + h) T + q) 192 + k) */
    const void* dout; void* dqkv;
} PfAttnArgs;
int pf_enc_attn_forward(int32_t precision, const PfAttnArgs* args, void* stream);
int pf_enc_attn_backward(int32_t precision, const PfAttnArgs* args, void* stream);
typedef struct PfPoolArgs {     /* 8 learned queries per head over the tokens of an event */
    const void* kv; const float* q; int64_t B; int32_t T;
    float* pooled; const float* dpooled; void* dkv; float* dq;
} PfPoolArgs;
int pf_enc_pool_forward(int32_t precision, const PfPoolArgs* args, void* stream);
int pf_enc_pool_backward(int32_t precision, const PfPoolArgs* args, void* stream);

/* ---- coherent encoder: frequency-domain geometry features ------------------------------------------------
 * CoherentEncoder._geometry_rel (src/ahsd/models/coherent_encoder.py:79-116) for a batch of events: rfft (ortho) of every
 * detector's 16 384 sanitised samples, kept bins [band_lo, band_lo + nf); per detector n_bands log band energies
 * log(mean_band |X|^2 + 1e-8); per detector pair (i < j, lexicographic) and band the power-weighted coherence
 * (|g| + 1e-8, Re g / |g|, Im g / |g|), g = sum X_i conj X_j / (sum |X_i| |X_j| + 1e-8) with |X| = sqrt(|X|^2 + 1e-12); the
 * peak of |irfft(X_i conj X_j on the band)| over the lags -maxlag .. maxlag as (lag / maxlag, max / (mean + 1e-8)); and
 * log(sum |X_i|^2 + 1e-8) - log(sum |X_j|^2 + 1e-8).  Row layout of rel = the reference's torch.cat order:
 * [n_det][n_bands] energies, then per pair [n_bands] |g| | [n_bands] cos | [n_bands] sin | lag | sharpness | log ratio.
 * Both transforms run in LDS (csrc/pf_geom.hip); fp32.  Limits: n_det <= 8, n_bands <= 16, 1 <= band_lo, band_lo + nf <= 4096,
 * 1 <= maxlag <= 127; band b = kept bins [band_edge[b], band_edge[b+1]) (non-decreasing, within [0, nf]).
 * twiddle: device copy of the table pf_geom_twiddles writes (host memory, 8192 x (cos, sin) of -2 pi m / 16384).
 * spec / etot: caller-owned workspaces. */
typedef struct PfGeomArgs {
    const float* clean;        /* [batch][n_det][16384] */
    int64_t batch;
    int32_t n_det, band_lo, nf, n_bands, maxlag;
    int32_t band_edge[17];
    const float* twiddle;      /* [8192][2] */
    float* spec;               /* [batch][n_det][nf][2] workspace: band spectra */
    float* etot;               /* [batch][n_det] workspace: total band power */
    float* rel;                /* [batch][n_det n_bands + npairs (3 n_bands + 3)] */
    int32_t sanitize;          /* 1: `clean` is the RAW strain; the samples are sanitised on load as lean_npe.py:207 does
                                * (nan -> 0, +-inf -> +-100, clamp to +-100), saving the caller a pass over the strain */
} PfGeomArgs;
int pf_geom_twiddles(float* host_table);
int pf_geom_features(const PfGeomArgs* args, void* stream);

/* ---- training-example remix ------------------------------------------
 * The deterministic half of RemixDataset.__getitem__ (experiments/remix_data.py:218-299) for a
 * batch of examples whose random decisions have already been drawn:
 *   strain[b,d,:] = f32(noise_pool[noise_row[b], d, :])
 *                   + sum_{k < nsig[b]} scale[b,k] * roll(f32(signal_pool[sig_start[b]+k, d, :]), shift[b,k])
 * (fp32, summed in k order, bit-identical to the numpy arithmetic of remix_data.py:232-260),
 * detector d of example b replaced by fill[fill_row[b,d], :] where fill_row[b,d] >= 0
 * (detector dropout, :262-279), net_snr[b] = ||sig_sum over kept detectors||_2 (:286).
 * Pools are fp16 [rows][3][16384] (the memmap cache layout, remix_data.py:49-111); noise_row[b] < 0
 * = no pool noise (the caller adds its own, e.g. real-noise crops).  Rows outside a pool contribute 0.
 * sig_sum (fp32 [batch][3][16384]) and net_snr may be NULL; fill_row may be NULL (no dropout).
 * workspace: pf_remix_workspace_bytes(batch) bytes, 8-byte aligned. */
int64_t pf_remix_workspace_bytes(int64_t batch);
int pf_remix_forward(const void* noise_pool, int64_t n_noise, const void* signal_pool, int64_t n_signals,
                     const int64_t* noise_row, const int64_t* sig_start, const int32_t* nsig,
                     const float* scale, const int32_t* shift, const int32_t* fill_row, const float* fill,
                     int64_t n_fill, int64_t batch, float* strain, float* sig_sum, float* net_snr,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* ---- measurement aid (not part of the path) ------------------------------------
 * The compute-free read that bounds the 16-row flow kernel: `workgroups` workgroups of 8 waves each read the SAME `bytes`
 * of `buf` (16-byte loads, every wave its own contiguous eighth, `in_flight` = 2 | 4 | 8 | 16 one-KiB loads outstanding per
 * wave, an xor per load) and write one word each to sink[workgroups].  bench.py times it over the packed weight stream of
 * the flow it benchmarks, on the box it runs on: `roofline.ceiling` (csrc/pf_diag.hip). */
int pf_diag_stream_ingest(const void* buf, int64_t bytes, int32_t workgroups, int32_t in_flight, uint32_t* sink, void* stream);

/* ---- introspection ------------------------------------------------------------ */
const char* pf_last_error(void);
const char* pf_version(void);
/* name (as rocprofv3 prints it) of the kernel pf_flow_forward dispatches for `batch` rows; thread-local
 * storage, NULL for an unsupported desc (bench / profiling) */
const char* pf_flow_forward_kernel_name(const PfFlowDesc* desc, int64_t batch);
/* FLOP one batch row costs in the MFMAs pf_flow_forward actually ISSUES for this desc: 2 x 512 (fp32: 256) multiply-adds per
 * 1-KiB fragment of the packed stream -- all-zero fragments of the autoregressive masks are neither stored nor multiplied,
 * partly masked fragments are multiplied whole.  Lies between the mask-aware useful count and the dense-GEMM count of
 * SURVEY 8d (bench.py reports all three). */
int64_t pf_flow_issued_flop_per_row(const PfFlowDesc* desc);
/* rows of the batch one workgroup processes for a given batch size (bench / tests): 16 / 32 / 48 (16-row kernel), 64
 * (mid-batch kernel) or 128 (large-batch kernel) */
int32_t pf_flow_rows_per_workgroup(const PfFlowDesc* desc, int64_t batch);

#ifdef __cplusplus
}
#endif
#endif /* PF_HIP_H */
