// pf_layout.h -- packed-weight layout ("plan") shared by the host pack-map builder
// and the gfx950 flow kernels.  See DESIGN.md "Data layout in HBM".
//
// One workgroup = NW = H/16 waves.  Hidden units are taken in DEGREE-SORTED order
// (stable sort by MADE degree): position p <-> unit perm[p].  Wave w owns sorted
// positions 16w..16w+15 of every H-wide layer, and spline feature NW-1-w of the
// final layer (light hidden tiles carry the heavy output features, so every
// wave streams about the same number of fragments per layer).  With sorted units the autoregressive masks are block lower-triangular,
// so a wave's tile only needs the first kmaxH[w] k-steps of a masked H x H GEMM
// and feature w only the first kmaxO[w] k-steps of the final layer: all-zero
// fragments are neither stored nor fetched.
//
// The packed buffer holds, per wave, ONE linear stream over all layers of MFMA
// A-fragments ("frags", 1 KiB = 64 lanes x 16 B) in the exact order the wave
// consumes them, then a bias region (fp32, 256 floats per (layer, wave)).
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
// Per-layer schedule of a wave (entries; an entry is one frag slot, fetched only if
// active for this wave):
//   in(1) | ctx(CKM) | block0: W0(HK) W1(HK) gate(CKM) | block1: ... | out: 3 x HK | pad
// kinds:  ALWAYS      CTX (ks < CK)      HID (ks < kmaxH[w])            OUT (ks < kmaxO[w])
#pragma once
#include <stdint.h>

#include "../../include/pf_hip.h"

#if defined(__HIPCC__) || defined(__CUDACC__)
#define PF_HD __host__ __device__
#else
#define PF_HD
#endif

namespace pf {

constexpr int kFragBytes = 1024;
constexpr int kBiasFloatsPerWave = 256;   // slots of 16 floats
constexpr int kSlotIn = 0, kSlotCtx = 1, kSlotBlk = 2 /* +3*b: W0, W1, gate */, kSlotOut = 8;
constexpr int kWindow = 12;               // max frags a wave keeps in flight (= stream overrun pad)
constexpr int kMaxWaves = 16;

enum EntryKind : int { kAlways = 0, kCtx = 1, kHid = 2, kOut = 3, kNever = 4 };

// compile-time schedule shared by kernel (template) and host (runtime mirror below)
template <bool BF16, int NW, int CKM, int WIN>
struct Sched {
    static constexpr int HK = BF16 ? NW / 2 : NW;      // frags per full hidden tile row
    static constexpr int E_IN = 0;
    static constexpr int E_CTX = 1;
    static constexpr int E_BLK = E_CTX + CKM;
    static constexpr int BLK = 2 * HK + CKM;           // W0 | W1 | gate
    static constexpr int E_OUT = E_BLK + 2 * BLK;
    static constexpr int NE_RAW = E_OUT + 3 * HK;
    static constexpr int NE = (NE_RAW + WIN - 1) / WIN * WIN;
    static constexpr int kind(int e) {
        if (e < E_CTX) return kAlways;
        if (e < E_BLK) return kCtx;
        if (e < E_OUT) { const int r = (e - E_BLK) % BLK; return r < 2 * HK ? kHid : kCtx; }
        if (e < NE_RAW) return kOut;
        return kNever;
    }
    static constexpr int ks(int e) {
        if (e < E_CTX) return 0;
        if (e < E_BLK) return e - E_CTX;
        if (e < E_OUT) { const int r = (e - E_BLK) % BLK; return r < 2 * HK ? r % HK : r - 2 * HK; }
        if (e < NE_RAW) return (e - E_OUT) % HK;
        return 0;
    }
};

struct FlowPlan {
    int D, C, H, K, L, M, NB;   // M = 3K-1 params per feature, NB = residual blocks
    int bf16;                   // 1: bf16 frags, 0: f32 frags
    int NW;                     // waves per workgroup = H/16
    int kstep;                  // k extent of one frag: 32 (bf16) or 16 (f32)
    int CK, CKM, HK;            // active / scheduled context frags per tile; hidden frags per tile
    int kmaxH[kMaxWaves];       // active k-steps of this wave's tile in a masked H x H GEMM
    int kmaxO[kMaxWaves];       // active k-steps of this wave's feature in the final layer (0: none)
    int feat[kMaxWaves];        // spline feature owned by the wave (NW-1-w), -1 if >= D
    int fragsPerLayer[kMaxWaves];
    int64_t waveBase[kMaxWaves];   // first frag of the wave's stream
    int64_t fragsTotal;            // incl. kWindow pad frags per wave (prefetch overrun)
    int64_t weightBytes;
    int64_t biasFloats;            // L * NW * 256
    int64_t rawPerLayer;

    PF_HD int64_t bias_index(int layer, int wave) const {
        return ((int64_t)layer * NW + wave) * kBiasFloatsPerWave;
    }
};

inline int hid_degree(int D, int u) {
    const int hi = D - 1 > 1 ? D - 1 : 1, lo = D - 1 < 1 ? D - 1 : 1;
    return u % hi + lo;
}

// scheduled context frags for a given C: smallest supported CKM covering it
inline int pick_ckm(bool bf16, int C) {
    const int ck = (C + (bf16 ? 31 : 15)) / (bf16 ? 32 : 16);
    const int small = bf16 ? 9 : 18, large = bf16 ? 18 : 36;
    if (ck <= small) return small;
    if (ck <= large) return large;
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

// returns 0 on success, PF_ERR_* otherwise
inline int make_plan(const PfFlowDesc& d, FlowPlan& o) {
    if (d.hidden_features != 64 && d.hidden_features != 128 && d.hidden_features != 192 &&
        d.hidden_features != 256)
        return PF_ERR_UNSUPPORTED;
    if (d.features < 1 || d.features > d.hidden_features / 16) return PF_ERR_UNSUPPORTED;
    if (d.num_bins < 2 || d.num_bins > 16) return PF_ERR_UNSUPPORTED;
    if (d.num_layers < 1 || d.context_features < 0 || d.num_blocks != 2) return PF_ERR_UNSUPPORTED;
    if (d.precision != PF_PREC_F32 && d.precision != PF_PREC_BF16) return PF_ERR_BAD_ARG;
    o.D = d.features; o.C = d.context_features; o.H = d.hidden_features;
    o.K = d.num_bins; o.L = d.num_layers; o.M = 3 * d.num_bins - 1; o.NB = d.num_blocks;
    o.bf16 = d.precision == PF_PREC_BF16;
    o.NW = o.H / 16;
    o.kstep = o.bf16 ? 32 : 16;
    o.CK = (o.C + o.kstep - 1) / o.kstep;
    o.CKM = pick_ckm(o.bf16, o.C);
    if (o.CKM < 0) return PF_ERR_UNSUPPORTED;
    o.HK = o.H / o.kstep;
    int perm[256], deg_sorted[256];
    sorted_units(o.D, o.H, perm);
    for (int p = 0; p < o.H; ++p) deg_sorted[p] = hid_degree(o.D, perm[p]);
    for (int w = 0; w < kMaxWaves; ++w) { o.feat[w] = -1; o.kmaxH[w] = o.kmaxO[w] = 0; o.fragsPerLayer[w] = 0; o.waveBase[w] = 0; }
    int64_t base = 0;
    for (int w = 0; w < o.NW; ++w) {
        // hidden mask: deg_out >= deg_in -> inputs with degree <= the tile's largest degree
        const int tile_max = deg_sorted[16 * w + 15];
        int cnt = 0;
        while (cnt < o.H && deg_sorted[cnt] <= tile_max) ++cnt;
        o.kmaxH[w] = (cnt + o.kstep - 1) / o.kstep;
        // output mask: deg_out (= feature+1) > deg_in
        const int f = o.NW - 1 - w;
        o.feat[w] = f < o.D ? f : -1;
        cnt = 0;
        if (f < o.D) while (cnt < o.H && deg_sorted[cnt] < f + 1) ++cnt;
        o.kmaxO[w] = (cnt + o.kstep - 1) / o.kstep;
        o.fragsPerLayer[w] = 1 + o.CK + o.NB * (2 * o.kmaxH[w] + o.CK) + 3 * o.kmaxO[w];
        o.waveBase[w] = base;
        base += (int64_t)o.L * o.fragsPerLayer[w] + kWindow;
    }
    o.fragsTotal = base;
    o.weightBytes = o.fragsTotal * kFragBytes;
    o.biasFloats = (int64_t)o.L * o.NW * kBiasFloatsPerWave;
    const int64_t ctxp = o.C > 0 ? ((int64_t)o.H * o.C + o.H) : 0;
    o.rawPerLayer = (int64_t)o.H * o.D + o.H + ctxp
                  + (int64_t)o.NB * (ctxp + 2 * ((int64_t)o.H * o.H + o.H))
                  + (int64_t)o.D * o.M * o.H + (int64_t)o.D * o.M;
    return PF_OK;
}

}  // namespace pf
