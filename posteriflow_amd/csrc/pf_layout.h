// pf_layout.h -- packed-weight layout ("plan") shared by the host pack-map builder
// and the gfx950 flow kernels.  See DESIGN.md "Data layout in HBM".
//
// Hidden units are taken in DEGREE-SORTED order (stable sort by MADE degree):
// position p <-> unit perm[p]; tile t = positions 16t..16t+15, T = H/16 tiles.
// With sorted units the autoregressive masks are block lower-triangular: tile t
// of a masked H x H layer only needs its first kH[t] k-steps, spline feature f of
// the final layer only its first kO[f] k-steps.
//
// One workgroup = NW = T/2 waves.  Wave w owns the tile PAIR (w, T-1-w) and the
// feature pair (w, D-1-w): kH[w] + kH[T-1-w] and kO[w] + kO[D-1-w] are (nearly)
// the same for every wave, so all waves run ONE static schedule with no
// per-wave predicates, padded to the plan-wide maxima KHS / KOS.  All-zero
// fragments are neither stored nor fetched.
//
// The packed buffer holds, per wave, one linear stream [layer][NF frags] of MFMA
// A-fragments ("frags", 1 KiB = 64 lanes x 16 B) in consumption order, then a
// bias region [layer][tile][12 slots][16] fp32.
//
// Hoisted-context plans (PF_FLAG_HOIST_CTX): the context GEMMs are not part of the
// per-wave streams (CKM = 0); their weights form a third region, [(layer*3 + j)*NT +
// tile][CK frags] (j = 0 MADE context layer, 1/2 the gates of block 0/1) followed by
// [(layer*3 + j)*NT + tile][16] fp32 biases, consumed by the context-projection kernel,
// whose output (in MFMA C-fragment order, [row/64][(layer*3+j)*NT+tile][4][64][4]) = relu /
// sigmoid of the projections is read by the layer chain instead of recomputing it per layer (and, in the
// inverse, per autoregressive pass).
//
//   bf16 mode  (v_mfma_f32_16x16x32_bf16): a frag is one MFMA A operand,
//              lane = 16*g + r16 holds A[row r16][k = 8*g + j], j = 0..7
//   f32 mode   (v_mfma_f32_16x16x4_f32): a frag feeds 4 MFMAs (a 16-wide k-group),
//              lane = 16*g + r16 holds, in element e, A[row r16][k-slot g] of MFMA e
//
// k index -> source column (the activation layout in LDS fixes it):
//   hidden input   bf16: pos = 16*(2*ks + (j>>2)) + 4*g + (j&3)     (sorted position)
//                  f32 : pos = 16*q + 4*g + e
//   context input  bf16: col = 32*ks + 8*g + j        f32: col = 16*q + 4*g + e
//   x input        bf16: d   = (8*g + j) & 15  (k<16: bf16 hi part of x, k>=16: lo part)
//                  f32 : d   = 4*g + e
//
// Per-layer stream of a wave (tile A = w, tile B = T-1-w; feature A = w, B = D-1-w):
//   in:   A, B                                            2
//   ctx:  for ks < CKM: A, B                              2*CKM
//   block b (x2):  W0: KHS entries: entry i < kH[A] is k-step i of tile A; entry
//                      i >= KHS - kH[B] is k-step KHS-1-i of tile B (REVERSE order, so the
//                      k-step of every entry is a compile-time constant); zero pad between
//                  W1: KHS entries, same
//                  gate: for ks < CKM: A, B               2*KHS + 2*CKM
//   out:  for q in widths|heights|derivs: KOS entries, feature A forward / feature B reversed
#pragma once
#include <stdint.h>

#include "../../include/pf_hip.h"
#include "pf_wide_layout.h"

#if defined(__HIPCC__)
#define PF_HD __host__ __device__
#else
#define PF_HD
#endif

namespace pf {

constexpr int kFragBytes = 1024;
constexpr int kBiasSlots = 12;            // per tile: in, ctx, 2 x (W0, W1, gate), out w|h|d, spare
constexpr int kBiasFloatsPerTile = kBiasSlots * 16;
constexpr int kSlotIn = 0, kSlotCtx = 1, kSlotBlk = 2 /* +3*b: W0, W1, gate */, kSlotOut = 8;
constexpr int kWindowPad = 16;            // frags of zero padding behind each wave's stream
constexpr int kMaxWaves = 8;
constexpr int kMaxTiles = 16;

template <bool BF16> constexpr int kPadH = BF16 ? 2 : 5;
template <bool BF16> constexpr int kPadO = BF16 ? 1 : 2;

// compile-time schedule (frag offsets inside one layer of a wave's stream)
template <bool BF16, int NT, int CKM, int DENSE>   // DENSE: 0 masked stream, 1 dense stream, 2 masked with one more H x H entry
struct Sched {
    static constexpr int HK = BF16 ? NT / 2 : NT;          // frags per full hidden tile row
    // masked streams: a tile pair (w, T-1-w) needs kH[w] + kH[T-1-w] <= HK + pad entries.  32-wide bf16 k-steps pair up
    // within HK + 2 / HK + 1; the 16-wide fp32 k-steps are finer, the sums of LeanNPE-sized flows reach HK + 5 / HK + 2
    // (D = 11: 9 + 12 = 21 of 32 dense entries) -- still a third fewer fragments and MFMAs than the dense stream
    static constexpr int KHS = DENSE == 1 ? 2 * HK : HK + kPadH<BF16> + (DENSE == 2 ? 1 : 0);   // entries of a masked H x H GEMM (tile pair)
    static constexpr int KOS = DENSE == 1 ? 2 * HK : HK + kPadO<BF16>;   // entries per spline-parameter tile (feature pair)
    static constexpr int E_IN = 0;
    static constexpr int E_CTX = 2;
    static constexpr int E_BLK = E_CTX + 2 * CKM;
    static constexpr int BLK = 2 * KHS + 2 * CKM;           // W0 | W1 | gate
    static constexpr int E_OUT = E_BLK + 2 * BLK;
    static constexpr int NF = E_OUT + 3 * KOS;
};

struct FlowPlan {
    int D, C, H, K, L, M, NB;   // M = 3K-1 params per feature, NB = residual blocks
    int bf16;                   // 1: bf16 frags, 0: f32 frags
    int NT, NW;                 // tiles = H/16, waves = NT/2
    int kstep;                  // k extent of one frag: 32 (bf16) or 16 (f32)
    int CK, CKM, HK;            // needed / scheduled context frags per tile; frags per full hidden row
    int hoist;                  // 1: context projections hoisted (CKM = 0 in the streams)
    int additive;               // 1: masked-context conditioner: additive context, no reverse permutation
    int dense;                  // 0: masked stream, KHS / KOS = HK + kPadH / HK + kPadO; 1: KHS = KOS = 2*HK (masks too coarse to
                                // pair up); 2: masked, KHS = HK + kPadH + 1 (bf16, NT = 16)
    int wide;                   // 1: the large-batch kernel's single common stream (pf_wide_layout.h); CKS in CKM
    int bwd;                    // 1: PF_FLAG_BWD -- bf16 A-fragments for the backward: transposed matrices (chain), forward matrices + biases (re-evaluation)
    int generic;                // 1: a shape none of the scheduled kernels is built for (e.g. H = 384, K = 24): dense masked
                                // matrices as plain [tile][k-step] fragment arrays (unit order: see gsorted), pf_flow_generic.hip
    int gKx, gKc, gKh, gTf;     // generic: k-steps of the x / context / hidden operands, tiles of the final layer
    int gsorted;                // generic: hidden units stored in degree order (sorted_units): the masked matrices are then block
                                // lower-triangular and the kernel skips the k-steps beyond a tile's last non-zero column.  0 for
                                // the PF_FLAG_GENERIC layout the re-evaluation reads (its outputs are in nflows unit order)
    int KHS, KOS, NF;
    int kH[kMaxTiles];          // active k-steps of tile t in a masked H x H GEMM
    int kO[kMaxTiles];          // active k-steps of feature f in the final layer
    int featA[kMaxWaves], featB[kMaxWaves];   // features of wave w (-1: none)
    int64_t fragsPerWave;       // L * NF + kWindowPad
    int64_t fragsTotal;
    int64_t weightBytes;
    int64_t biasFloats;         // L * NT * kBiasFloatsPerTile
    int64_t ctxFrags;           // hoisted: 3 * L * NT * CK frags of context weights, else 0
    int64_t ctxBiasFloats;      // hoisted: 3 * L * NT * 16
    int64_t rawPerLayer;

    PF_HD int64_t packed_bytes() const {
        return weightBytes + (biasFloats + ctxBiasFloats) * (int64_t)sizeof(float) + ctxFrags * kFragBytes;
    }
    // byte offsets of the regions inside the packed buffer
    PF_HD int64_t ctx_frag_offset() const { return weightBytes + biasFloats * (int64_t)sizeof(float); }
    PF_HD int64_t ctx_bias_offset() const { return ctx_frag_offset() + ctxFrags * kFragBytes; }

    // PF_FLAG_BWD stream, in 1-KiB fragments: per layer [WfT][W2T_0][W1T_0][W2T_1][W1T_1][W0T]
    PF_HD int bwd_ksf() const { return (D * M + 31) / 32; }                   // k-steps of the final layer's transpose
    PF_HD int bwd_wf_frags() const { return NT * bwd_ksf(); }
    PF_HD int bwd_hh_frags() const { return NT * (H / 32); }
    PF_HD int bwd_layer_frags() const { return bwd_wf_frags() + 4 * bwd_hh_frags() + H / 32; }
    PF_HD int64_t bwd_w2t(int layer, int j) const { return (int64_t)layer * bwd_layer_frags() + bwd_wf_frags() + (2 * j) * bwd_hh_frags(); }
    PF_HD int64_t bwd_w1t(int layer, int j) const { return bwd_w2t(layer, j) + bwd_hh_frags(); }
    PF_HD int64_t bwd_w0t(int layer) const { return (int64_t)layer * bwd_layer_frags() + bwd_wf_frags() + 4 * bwd_hh_frags(); }
    // ... followed by the FORWARD matrices in the same fragment form (nflows unit order, for the conditioner re-evaluation
    // kernel, pf_flow_reeval.hip): per layer [Win: NT x 1 (x as hi | lo)][Wc, Wg0, Wg1: NT x CKB each, if C > 0]
    // [W1_0, W2_0, W1_1, W2_1: NT x H/32][Wf: NTF x H/32], then fp32 biases per layer
    // [b_in H][bc, bg0, bg1: H each, if C > 0][b1_0, b2_0, b1_1, b2_1: H each][bf: 16 NTF]
    PF_HD int bwd_ckb() const { return (C + 31) / 32; }
    PF_HD int bwd_ntf() const { return (D * M + 15) / 16; }
    PF_HD int fwd_layer_frags() const { return NT + (C > 0 ? 3 * NT * bwd_ckb() : 0) + 4 * bwd_hh_frags() + bwd_ntf() * (H / 32); }
    PF_HD int fwd_layer_bias() const { return H + (C > 0 ? 3 * H : 0) + 4 * H + 16 * bwd_ntf(); }
    PF_HD int64_t fwd_region() const { return (int64_t)L * bwd_layer_frags(); }          // first forward fragment

    // generic layout: per layer [Win: NT x gKx][Wc, Wg0, Wg1: NT x gKc each, if C > 0][W1_0, W2_0, W1_1, W2_1: NT x gKh]
    // [Wf: gTf x gKh] fragments; after the fragments of ALL layers the fp32 biases, per layer
    // [b_in H][bc, bg0, bg1: H each, if C > 0][b1_0, b2_0, b1_1, b2_1: H each][bf: 16 gTf]
    PF_HD int gen_layer_frags() const { return NT * gKx + (C > 0 ? 3 * NT * gKc : 0) + 4 * NT * gKh + gTf * gKh; }
    PF_HD int gen_layer_bias() const { return H + (C > 0 ? 3 * H : 0) + 4 * H + 16 * gTf; }
    PF_HD int gen_xh() const { return (D + 15) / 16 * 16; }                  // bf16: x enters as hi | lo halves of this width
    PF_HD int64_t gen_lds_bytes() const {
        const int esz = bf16 ? 2 : 4;
        return (int64_t)16 * ((int64_t)gKx * kstep * esz + 16 + (int64_t)gKc * kstep * esz + 16 + 2 * ((int64_t)gKh * kstep * esz + 16)
                              + (int64_t)(H + 4) * 4 + (int64_t)(16 * gTf + 4) * 4 + 4 * 32 * 4) + 256;
    }

    PF_HD int64_t bias_index(int layer, int tile) const {
        return ((int64_t)layer * NT + tile) * kBiasFloatsPerTile;
    }
};

inline int hid_degree(int D, int u) {
    const int hi = D - 1 > 1 ? D - 1 : 1, lo = D - 1 < 1 ? D - 1 : 1;
    return u % hi + lo;
}

// scheduled context frags per tile for a given C: smallest supported CKM covering it
// (the set of built kernels: NT 4/8 have {0, small, mid}, NT 12 {0, mid}, NT 16 {0, mid, large})
inline int pick_ckm(bool bf16, int NT, int C) {
    if (C == 0) return 0;
    const int ck = (C + (bf16 ? 31 : 15)) / (bf16 ? 32 : 16);
    const int small = bf16 ? 3 : 6, mid = bf16 ? 9 : 18, large = bf16 ? 18 : 36;
    if (NT <= 8 && ck <= small) return small;
    if (ck <= mid) return mid;
    if (NT == 16 && ck <= large) return large;
    return -1;
}

// sorted position -> hidden unit (stable sort by degree)
inline void sorted_units(int D, int H, int* perm) {
    int n = 0;
    const int maxdeg = D > 1 ? D - 1 : 1;
    for (int deg = 0; deg <= maxdeg; ++deg)
        for (int u = 0; u < H; ++u)
            if (hid_degree(D, u) == deg) perm[n++] = u;
}

// Shapes outside the scheduled set (the reference also builds 12 x 384 x 24 heads, experiments/frozen_context_heads.py:159-163):
// one generic kernel, plain conditioner, forward and D-pass inverse in both precisions.  H a multiple of 16 up to 512, D <= 32,
// K <= 32, and the workgroup's LDS image within 160 KB.
inline int make_generic_plan(const PfFlowDesc& d, FlowPlan& o) {
    if (d.reserved & (PF_FLAG_WIDE | PF_FLAG_BWD)) return PF_ERR_UNSUPPORTED;
    if (d.num_bins * 3 - 1 < 1 || d.num_layers < 1) return PF_ERR_UNSUPPORTED;
    if (d.hidden_features < 16 || d.hidden_features > 512 || d.hidden_features % 16) return PF_ERR_UNSUPPORTED;
    if (d.features < 1 || d.features > 32 || d.num_bins < 2 || d.num_bins > 32) return PF_ERR_UNSUPPORTED;
    o = FlowPlan{};
    o.generic = 1;
    o.D = d.features; o.C = d.context_features; o.H = d.hidden_features;
    o.K = d.num_bins; o.L = d.num_layers; o.M = 3 * d.num_bins - 1; o.NB = d.num_blocks;
    o.bf16 = d.precision == PF_PREC_BF16;
    o.NT = o.H / 16; o.NW = 8;
    o.kstep = o.bf16 ? 32 : 16;
    const int xw = o.bf16 ? 2 * o.gen_xh() : o.D;                       // bf16: hi | lo halves
    o.gKx = (xw + o.kstep - 1) / o.kstep;
    o.gKc = (o.C + o.kstep - 1) / o.kstep;
    o.gKh = (o.H + o.kstep - 1) / o.kstep;
    o.gTf = (o.D * o.M + 15) / 16;
    o.CK = o.gKc; o.CKM = 0; o.HK = o.gKh; o.hoist = 0;
    o.additive = (d.reserved & PF_FLAG_MASKED_CONTEXT) && o.C > 0 ? 1 : 0;     // masked-context conditioner (flows.py:186-234)
    o.gsorted = ((d.reserved & PF_FLAG_GENERIC) || o.D < 2) ? 0 : 1;
    if (o.gen_lds_bytes() > 160 * 1024) return PF_ERR_UNSUPPORTED;
    o.fragsPerWave = 0;
    o.fragsTotal = (int64_t)o.L * o.gen_layer_frags();
    o.weightBytes = o.fragsTotal * kFragBytes;
    o.biasFloats = (int64_t)o.L * o.gen_layer_bias();
    o.ctxFrags = 0; o.ctxBiasFloats = 0;
    const int64_t ctxp = o.C > 0 ? ((int64_t)o.H * o.C + o.H) : 0;
    o.rawPerLayer = (int64_t)o.H * o.D + o.H + ctxp + (int64_t)o.NB * (ctxp + 2 * ((int64_t)o.H * o.H + o.H))
                  + (int64_t)o.D * o.M * o.H + (int64_t)o.D * o.M;
    return PF_OK;
}

// returns 0 on success, PF_ERR_* otherwise
inline int make_plan(const PfFlowDesc& d, FlowPlan& o) {
    if (d.num_layers < 1 || d.context_features < 0 || d.num_blocks != 2) return PF_ERR_UNSUPPORTED;
    if (d.precision != PF_PREC_F32 && d.precision != PF_PREC_BF16) return PF_ERR_BAD_ARG;
    o.generic = 0;
    const bool scheduled_shape = (d.hidden_features == 64 || d.hidden_features == 128 || d.hidden_features == 192 ||
                                  d.hidden_features == 256) &&
                                 d.features >= 1 && d.features <= d.hidden_features / 16 && d.num_bins >= 2 && d.num_bins <= 16;
    if (!scheduled_shape || (d.reserved & PF_FLAG_GENERIC)) return make_generic_plan(d, o);
    o.D = d.features; o.C = d.context_features; o.H = d.hidden_features;
    o.K = d.num_bins; o.L = d.num_layers; o.M = 3 * d.num_bins - 1; o.NB = d.num_blocks;
    o.bf16 = d.precision == PF_PREC_BF16;
    o.NT = o.H / 16; o.NW = o.NT / 2;
    o.kstep = o.bf16 ? 32 : 16;
    o.CK = (o.C + o.kstep - 1) / o.kstep;
    o.additive = (d.reserved & PF_FLAG_MASKED_CONTEXT) ? 1 : 0;
    o.wide = 0;
    if (d.reserved & PF_FLAG_WIDE) {
        // the large-batch kernel: bf16, H = 256, plain (GLU) conditioner, context in-layer; built for (D, C) pairs only
        if (!o.bf16 || o.H != wide::kHidden || o.K != 16 || (d.reserved & (PF_FLAG_MASKED_CONTEXT | PF_FLAG_HOIST_CTX)) ||
            !wide::built(o.D, o.C))
            return PF_ERR_UNSUPPORTED;
        o.wide = 1;
    }
    o.bwd = 0;
    if (d.reserved & PF_FLAG_BWD) {
        if (!o.bf16 || (d.reserved & ~PF_FLAG_BWD) || o.H % 32) return PF_ERR_UNSUPPORTED;
        o.bwd = 1;
    }
    o.hoist = (d.reserved & (PF_FLAG_HOIST_CTX | PF_FLAG_MASKED_CONTEXT)) && o.C > 0 ? 1 : 0;
    o.CKM = (o.hoist || o.bwd) ? 0 : pick_ckm(o.bf16, o.NT, o.C);      // (the backward stream has no scheduled context slots)
    if (o.CKM < 0 || o.CK > 64) return PF_ERR_UNSUPPORTED;
    o.HK = o.H / o.kstep;
    int perm[256], deg_sorted[256];
    sorted_units(o.D, o.H, perm);
    for (int p = 0; p < o.H; ++p) deg_sorted[p] = hid_degree(o.D, perm[p]);
    for (int t = 0; t < kMaxTiles; ++t) o.kH[t] = o.kO[t] = 0;
    for (int t = 0; t < o.NT; ++t) {
        // hidden mask: deg_out >= deg_in -> inputs with degree <= the tile's largest degree
        const int tile_max = deg_sorted[16 * t + 15];
        int cnt = 0;
        while (cnt < o.H && deg_sorted[cnt] <= tile_max) ++cnt;
        o.kH[t] = (cnt + o.kstep - 1) / o.kstep;
    }
    for (int f = 0; f < o.D; ++f) {
        // output mask: deg_out (= f+1) > deg_in
        int cnt = 0;
        while (cnt < o.H && deg_sorted[cnt] < f + 1) ++cnt;
        o.kO[f] = (cnt + o.kstep - 1) / o.kstep;
    }
    int khs = 0, kos = 0;
    for (int w = 0; w < kMaxWaves; ++w) o.featA[w] = o.featB[w] = -1;
    for (int w = 0; w < o.NW; ++w) {
        const int fa = w, fb = o.D - 1 - w;
        o.featA[w] = (fa < o.D && fa <= fb) ? fa : -1;
        o.featB[w] = (fb > fa && fb >= 0) ? fb : -1;
        const int h = o.kH[w] + o.kH[o.NT - 1 - w];
        const int q = (o.featA[w] >= 0 ? o.kO[o.featA[w]] : 0) + (o.featB[w] >= 0 ? o.kO[o.featB[w]] : 0);
        if (h > khs) khs = h;
        if (q > kos) kos = q;
    }
    const int padh = o.bf16 ? kPadH<true> : kPadH<false>, pado = o.bf16 ? kPadO<true> : kPadO<false>;
    o.dense = (khs > o.HK + padh || kos > o.HK + pado || o.HK + padh > 2 * o.HK) ? 1 : 0;
    // LeanNPE's own shape (D = 11, H = 256) in bf16: tile-pair sums reach HK + 3 -- a variant with one more entry per
    // H x H GEMM (11 of 16 dense entries) instead of the dense stream
    if (o.dense && o.bf16 && o.NT == 16 && khs <= o.HK + padh + 1 && kos <= o.HK + pado) o.dense = 2;
    o.KHS = o.dense == 1 ? 2 * o.HK : o.HK + padh + (o.dense == 2 ? 1 : 0);
    o.KOS = o.dense == 1 ? 2 * o.HK : o.HK + pado;
    o.NF = 2 + 2 * o.CKM + o.NB * (2 * o.KHS + 2 * o.CKM) + 3 * o.KOS;
    o.fragsPerWave = (int64_t)o.L * o.NF + kWindowPad;
    o.fragsTotal = o.fragsPerWave * o.NW;
    o.weightBytes = o.fragsTotal * kFragBytes;
    o.biasFloats = (int64_t)o.L * o.NT * kBiasFloatsPerTile;
    o.ctxFrags = o.hoist ? (int64_t)3 * o.L * o.NT * o.CK : 0;
    o.ctxBiasFloats = o.hoist ? (int64_t)3 * o.L * o.NT * 16 : 0;
    if (o.wide) {
        const int cks = (o.C + 15) / 16;
        o.CKM = cks; o.hoist = 0; o.dense = 0;
        o.NF = wide::n_frags_padded(o.D, cks);
        o.fragsPerWave = 0;
        o.fragsTotal = wide::stream_frags(o.D, cks, o.L);
        o.weightBytes = o.fragsTotal * kFragBytes;
        o.biasFloats = (int64_t)o.L * wide::kBiasFloats;
        o.ctxFrags = 0; o.ctxBiasFloats = 0;
    }
    if (o.bwd) {
        o.fragsPerWave = 0;
        o.fragsTotal = (int64_t)o.L * (o.bwd_layer_frags() + o.fwd_layer_frags());
        o.weightBytes = o.fragsTotal * kFragBytes;
        o.biasFloats = (int64_t)o.L * o.fwd_layer_bias(); o.ctxFrags = 0; o.ctxBiasFloats = 0;
    }
    const int64_t ctxp = o.C > 0 ? ((int64_t)o.H * o.C + o.H) : 0;
    o.rawPerLayer = (int64_t)o.H * o.D + o.H + ctxp
                  + (int64_t)o.NB * (ctxp + 2 * ((int64_t)o.H * o.H + o.H))
                  + (int64_t)o.D * o.M * o.H + (int64_t)o.D * o.M;
    return PF_OK;
}

}  // namespace pf
