// pf_wide_layout.h -- schedule and packed-weight layout of the large-batch ("wide") forward kernel
// (pf_flow_wide_kernel.h), shared by the kernel (compile-time) and the host pack-map builder (run-time).
//
// The wide kernel turns the work split of pf_flow_fwd_kernel.h around: a workgroup is 4 waves, one per SIMD with
// the whole 512-register budget, and a wave OWNS 32 batch rows (one v_mfma_f32_32x32x16_bf16 column tile) through
// every layer -- its hidden activations never leave its registers, because a 32x32 accumulator tile converted to
// bf16 IS the B operand of the next GEMM when that GEMM's k order is permuted to the accumulator layout
// (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand").  All four waves consume the SAME
// stream of weight fragments, so the stream is fetched from L2 ONCE per workgroup (128 rows) by LDS-DMA into a
// two-half LDS ring and read from there by every wave: per-CU ingest per row drops 8x against the 16-row kernel,
// which is what bounded it.
//
// Units: hidden width H = 256 = 8 tiles of 32 units (degree-sorted positions, pf_layout.h) = 16 k-steps of 16.
// A "frag" is one MFMA A operand: 32 output units x 16 k, 1 KiB, lane (r = lane & 31, hf = lane >> 5) element j
// holds W[unit r of the tile][k = 8 hf + j of the k-step].
//
// k index -> source column of a frag:
//   hidden input (W0, W1, final layer), k-step ks = 2 Ti + s:  sorted position 32 Ti + 16 s + 8 (j >> 2) + 4 hf + (j & 3)
//                                                               (= accumulator register 8 s + j of input tile Ti)
//   context input, k-step ks:                                   column 16 ks + 8 hf + j
//   x input: k-step 0 = bf16 hi part of x[d], k-step 1 = lo part, d = 8 hf + j (both frags hold W_in[:, d])
//
// Per-layer stream (all waves; NF frags, padded with zero frags to NFP = a multiple of the ring size):
//   stage 1  for T < 8:  W_in tile T (2 frags) | context-layer tile T (CKS frags)
//   block b  W0: for T < 8: kH16(T) frags
//            W1 + gate: for T < 8: kH16(T) frags of W1 | CKS frags of the block's context (gate) layer
//   final    for batch m < (D + 1) / 2 (features fa = 2m, fb = 2m + 1):
//              WH(fa): kO16(fa) frags   rows u < 16 raw width u, u >= 16 raw height u - 16
//              WH(fb): kO16(fb) frags   (absent when fb == D)
//              DD(m):  kO16(fb or fa) frags   rows u < 16 raw derivative u of fa, u >= 16 of fb
// Masked (all-zero) k-steps are neither stored nor fetched.  Biases: fp32 [layer][kBiasFloats] in natural unit
// order per tile (accumulators are initialised with them).
#pragma once
#include <stdint.h>

namespace pf {
namespace wide {

constexpr int kHidden = 256;
constexpr int kTiles = 8;            // 32-unit tiles
constexpr int kKSteps = 16;          // 16-wide k-steps over the hidden width
constexpr int kFrag = 1024;          // bytes
constexpr int kEpoch = 36;           // frags per ring half
constexpr int kRing = 2 * kEpoch;    // frags in the LDS ring
constexpr int kRowsPerWave = 32;
constexpr int kWaves = 4;
constexpr int kRowsPerWG = kRowsPerWave * kWaves;
// bias offsets (floats) inside one layer's block
constexpr int kBiasIn = 0, kBiasCtx = 256, kBiasBlk = 512 /* + 768 b: W0, W1 (+256), gate (+512) */, kBiasOut = 2048;
constexpr int kBiasFloats = kBiasOut + 8 * 96;   // 8 batches x (WH a | WH b | DD) x 32
constexpr int kParStride = 52;       // floats per (row, feature) pair in the spline transpose (48 used)
constexpr int kXStride = 20;         // floats per row in the x / z exchange buffers (16 used)

constexpr int imin(int a, int b) { return a < b ? a : b; }

// number of hidden units with MADE degree <= d  (degree of unit u: u % (D - 1) + 1, D >= 2)
constexpr int cnt_upto(int D, int d) {
    if (d <= 0) return 0;
    const int m = D - 1;
    if (d >= m) return kHidden;
    return (kHidden / m) * d + imin(kHidden % m, d);
}
// degree of the unit at degree-sorted position p
constexpr int deg_at(int D, int p) {
    for (int d = 1; d < D - 1; ++d)
        if (cnt_upto(D, d) > p) return d;
    return D - 1;
}
// active 16-wide k-steps of hidden tile T in a masked H x H layer (mask: deg_out >= deg_in)
constexpr int kH16(int D, int T) { return (cnt_upto(D, deg_at(D, 32 * T + 31)) + 15) / 16; }
// ... of spline feature f in the final layer (mask: deg_out = f + 1 > deg_in)
constexpr int kO16(int D, int f) { return (cnt_upto(D, f) + 15) / 16; }

constexpr int n_batches(int D) { return (D + 1) / 2; }
constexpr int kDD(int D, int m) { return kO16(D, 2 * m + 1 < D ? 2 * m + 1 : 2 * m); }
constexpr int kWHb(int D, int m) { return 2 * m + 1 < D ? kO16(D, 2 * m + 1) : 0; }

constexpr int w0_off(int D, int T) { int s = 0; for (int t = 0; t < T; ++t) s += kH16(D, t); return s; }
constexpr int w0_len(int D) { return w0_off(D, kTiles); }
constexpr int w1_off(int D, int CKS, int T) { return w0_off(D, T) + CKS * T; }
constexpr int blk_len(int D, int CKS) { return 2 * w0_len(D) + CKS * kTiles; }
constexpr int e_in(int CKS, int T) { return T * (2 + CKS); }
constexpr int e_blk(int D, int CKS, int b) { return kTiles * (2 + CKS) + b * blk_len(D, CKS); }
constexpr int e_out(int D, int CKS) { return e_blk(D, CKS, 2); }
constexpr int out_off(int D, int m) {
    int s = 0;
    for (int i = 0; i < m; ++i) s += kO16(D, 2 * i) + kWHb(D, i) + kDD(D, i);
    return s;
}
constexpr int n_frags(int D, int CKS) { return e_out(D, CKS) + out_off(D, n_batches(D)); }
constexpr int n_frags_padded(int D, int CKS) { return (n_frags(D, CKS) + kRing - 1) / kRing * kRing; }

// total frags of the packed stream: L layers + one ring of zero frags (the DMA runs up to two epochs ahead)
constexpr int64_t stream_frags(int D, int CKS, int L) { return (int64_t)L * n_frags_padded(D, CKS) + kRing; }
constexpr int64_t packed_bytes(int D, int CKS, int L) {
    return stream_frags(D, CKS, L) * kFrag + (int64_t)L * kBiasFloats * (int64_t)sizeof(float);
}

// shapes the wide kernel is built for: (D, context k-steps)
constexpr bool built(int D, int C) { return (D == 15 || D == 11) && C == 288; }

constexpr int lds_bytes() {
    return kRing * kFrag + kWaves * 64 * kParStride * 4 + kWaves * 2 * kRowsPerWave * kXStride * 4 + kBiasFloats * 4;
}

}  // namespace wide
}  // namespace pf
