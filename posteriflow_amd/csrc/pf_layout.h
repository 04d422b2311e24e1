// pf_layout.h -- packed-weight layout shared by the host pack-map builder and
// the gfx950 kernels.  See DESIGN.md "Data layout in HBM".
//
// One workgroup = NW = H/16 waves.  Wave w owns hidden tile w (16 hidden units)
// of every masked/dense H-wide layer and spline feature w of the final layer.
// The packed buffer holds, per (layer, wave), a linear stream of MFMA
// A-fragments ("frags", 1 KiB = 64 lanes x 16 B) in the exact order the wave
// consumes them, followed by a bias region (fp32, 256 floats per (layer, wave)).
//
//   bf16 mode  (v_mfma_f32_16x16x32_bf16): a frag is one MFMA A operand,
//              lane = 16*g + r16 holds A[row r16][k = 8*g + j], j = 0..7
//   f32 mode   (v_mfma_f32_16x16x4_f32): a frag feeds 4 MFMAs (a 16-wide k-group),
//              lane = 16*g + r16 holds, in element e, A[row r16][k-slot g] of MFMA e
//
// k index -> source column (the activation layout in LDS fixes it):
//   hidden input   bf16: unit = 16*(2*ks + (j>>2)) + 4*g + (j&3)
//                  f32 : unit = 16*q + 4*g + e
//   context input  bf16: col  = 32*ks + 8*g + j        f32: col = 16*q + 4*g + e
//   x input        bf16: d    = (8*g + j) & 15  (k<16: bf16 hi part of x, k>=16: lo part)
//                  f32 : d    = 4*g + e
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
// bias slots
constexpr int kSlotIn = 0, kSlotCtx = 1, kSlotBlk = 2 /* +3*b: W0, W1, gate */, kSlotOut = 8;

struct FlowLayout {
    int D, C, H, K, L, M, NB;   // M = 3K-1 params per feature, NB = residual blocks
    int bf16;                   // 1: bf16 frags, 0: f32 frags
    int NW;                     // waves per workgroup = H/16
    int kstep;                  // k extent of one frag: 32 (bf16) or 16 (f32)
    int CK, HK;                 // frags per tile for a context / hidden GEMM
    int Cpad;                   // CK * kstep
    // frag offsets inside a (layer, wave) stream
    int oIn, oCtx, oBlk0, blkStride, oW0, oW1, oGate, oOut, NF;
    int64_t fragsTotal;         // L * NW * NF
    int64_t weightBytes;        // fragsTotal * 1024
    int64_t biasFloats;         // L * NW * 256
    int64_t rawPerLayer;        // raw fp32 parameters per layer

    PF_HD int64_t frag_index(int layer, int wave, int f) const {
        return ((int64_t)layer * NW + wave) * NF + f;
    }
    PF_HD int64_t bias_index(int layer, int wave) const {
        return ((int64_t)layer * NW + wave) * kBiasFloatsPerWave;
    }
};

// returns 0 on success, PF_ERR_* otherwise
inline PF_HD int make_layout(const PfFlowDesc& d, FlowLayout& o) {
    if (d.features < 1 || d.features > 16) return PF_ERR_UNSUPPORTED;
    if (d.hidden_features != 64 && d.hidden_features != 128 && d.hidden_features != 192 &&
        d.hidden_features != 256)
        return PF_ERR_UNSUPPORTED;
    if (d.num_bins < 2 || d.num_bins > 16) return PF_ERR_UNSUPPORTED;
    if (d.num_layers < 1 || d.context_features < 0 || d.num_blocks != 2) return PF_ERR_UNSUPPORTED;
    if (d.precision != PF_PREC_F32 && d.precision != PF_PREC_BF16) return PF_ERR_BAD_ARG;
    o.D = d.features; o.C = d.context_features; o.H = d.hidden_features;
    o.K = d.num_bins; o.L = d.num_layers; o.M = 3 * d.num_bins - 1; o.NB = d.num_blocks;
    o.bf16 = d.precision == PF_PREC_BF16;
    o.NW = o.H / 16;
    if (o.D > o.NW) return PF_ERR_UNSUPPORTED;        // one spline feature per wave
    o.kstep = o.bf16 ? 32 : 16;
    o.CK = (o.C + o.kstep - 1) / o.kstep;
    o.HK = o.H / o.kstep;
    o.Cpad = o.CK * o.kstep;
    o.oIn = 0;
    o.oCtx = 1;
    o.oBlk0 = o.oCtx + o.CK;
    o.oW0 = 0; o.oW1 = o.HK; o.oGate = 2 * o.HK;
    o.blkStride = 2 * o.HK + o.CK;
    o.oOut = o.oBlk0 + o.NB * o.blkStride;
    o.NF = o.oOut + 3 * o.HK;
    o.fragsTotal = (int64_t)o.L * o.NW * o.NF;
    o.weightBytes = o.fragsTotal * kFragBytes;
    o.biasFloats = (int64_t)o.L * o.NW * kBiasFloatsPerWave;
    int64_t ctxp = o.C > 0 ? ((int64_t)o.H * o.C + o.H) : 0;
    o.rawPerLayer = (int64_t)o.H * o.D + o.H + ctxp
                  + (int64_t)o.NB * (ctxp + 2 * ((int64_t)o.H * o.H + o.H))
                  + (int64_t)o.D * o.M * o.H + (int64_t)o.D * o.M;
    return PF_OK;
}

}  // namespace pf
