// pf_pack.hip -- host pack-map builder + device gather that produces the packed
// (pre-masked, fragment-ordered) weights the flow kernels stream.
//
// Replaces the per-call `self.weight * self.mask` of nflows MaskedLinear
// (executed by the reference at src/ahsd/models/flows.py:615-617, 637): the
// autoregressive masks are folded into the index map (masked entry -> -1 -> 0),
// computed from the same degree rule:
//   in_deg[d] = d+1;  hid_deg[u] = u % max(1,D-1) + min(1,D-1);  out_deg[f] = f+1
//   hidden mask: deg_out >= deg_in;  output mask: deg_out > deg_in.
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <algorithm>

#include "pf_flow_params.h"

namespace pf {

int64_t pack_map_len(const FlowPlan& L);

struct RawOffsets {
    int64_t in_w, in_b, c_w, c_b;
    int64_t g_w[2], g_b[2], w0_w[2], w0_b[2], w1_w[2], w1_b[2];
    int64_t out_w, out_b, total;
};

static RawOffsets raw_offsets(const FlowPlan& L) {
    RawOffsets r{};
    int64_t o = 0;
    const int64_t H = L.H, D = L.D, C = L.C;
    r.in_w = o; o += H * D; r.in_b = o; o += H;
    if (C > 0) { r.c_w = o; o += H * C; r.c_b = o; o += H; } else { r.c_w = r.c_b = -1; }
    for (int b = 0; b < L.NB; ++b) {
        if (C > 0) { r.g_w[b] = o; o += H * C; r.g_b[b] = o; o += H; } else { r.g_w[b] = r.g_b[b] = -1; }
        r.w0_w[b] = o; o += H * H; r.w0_b[b] = o; o += H;
        r.w1_w[b] = o; o += H * H; r.w1_b[b] = o; o += H;
    }
    r.out_w = o; o += D * (int64_t)L.M * H; r.out_b = o; o += D * (int64_t)L.M;
    r.total = o;
    return r;
}

// spline parameter row for (tile q of the feature, row r16): widths / heights / derivs
static inline int out_param(const FlowPlan& L, int q, int r16) {
    if (q == 0) return r16 < L.K ? r16 : -1;
    if (q == 1) return r16 < L.K ? L.K + r16 : -1;
    return r16 < L.K - 1 ? 2 * L.K + r16 : -1;
}

// Frag f of wave w's per-layer stream -> what it holds (runtime mirror of Sched<>).
struct Entry { int phase, ks, blk, q, tile, feat; bool pad; };   // phase: 0 in, 1 ctx, 2 W0, 3 W1, 4 gate, 5 out
static Entry stream_entry(const FlowPlan& L, int w, int f) {
    const int tA = w, tB = L.NT - 1 - w;
    const int eCtx = 2, eBlk = eCtx + 2 * L.CKM, blkLen = 2 * L.KHS + 2 * L.CKM, eOut = eBlk + L.NB * blkLen;
    Entry x{0, 0, 0, 0, tA, -1, false};
    // pair GEMM of n entries: i < nA -> k-step i of A; i >= n - nB -> k-step n-1-i of B; else pad
    auto pair_entry = [&](int i, int n, int nA, int nB, bool& isA, int& ks, bool& pad) {
        isA = i < nA; pad = !isA && i < n - nB;
        ks = isA ? i : n - 1 - i;
    };
    if (f < eCtx) { x.phase = 0; x.tile = f == 0 ? tA : tB; }
    else if (f < eBlk) { const int r = f - eCtx; x.phase = 1; x.ks = r / 2; x.tile = (r & 1) ? tB : tA; x.pad = x.ks >= L.CK; }
    else if (f < eOut) {
        int r = f - eBlk; x.blk = r / blkLen; r %= blkLen;
        if (r < 2 * L.KHS) {
            x.phase = r < L.KHS ? 2 : 3;
            bool isA; pair_entry(r % L.KHS, L.KHS, L.kH[tA], L.kH[tB], isA, x.ks, x.pad);
            x.tile = isA ? tA : tB;
        } else { r -= 2 * L.KHS; x.phase = 4; x.ks = r / 2; x.tile = (r & 1) ? tB : tA; x.pad = x.ks >= L.CK; }
    } else {
        const int r = f - eOut; x.phase = 5; x.q = r / L.KOS;
        const int fA = L.featA[w], fB = L.featB[w];
        bool isA; pair_entry(r % L.KOS, L.KOS, fA >= 0 ? L.kO[fA] : 0, fB >= 0 ? L.kO[fB] : 0, isA, x.ks, x.pad);
        x.feat = isA ? fA : fB;
        if (x.feat < 0) x.pad = true;
    }
    return x;
}

// ---- the large-batch kernel's packed layout (pf_wide_layout.h) -----------------------------------------------------
// One common stream [layer][NFP frags] (+ one ring of zero frags), then biases [layer][kBiasFloats].
static int build_wide_pack_map(const FlowPlan& L, int32_t* map) {
    namespace W = wide;
    const RawOffsets ro = raw_offsets(L);
    const int D = L.D, C = L.C, K = L.K, H = L.H, M = L.M, CKS = L.CKM;
    const int NF = W::n_frags(D, CKS), NFP = W::n_frags_padded(D, CKS), NB = W::n_batches(D);
    int perm[256];
    sorted_units(D, H, perm);
    // spline parameter of row u of a WH tile (u < 16: width u, else height u - 16) / a DD tile (derivative u & 15)
    auto wh_param = [&](int u) { const int i = u & 15; return i < K ? (u < 16 ? i : K + i) : -1; };
    auto dd_param = [&](int u) { const int i = u & 15; return i < K - 1 ? 2 * K + i : -1; };
    int64_t idx = 0;
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        for (int E = 0; E < NFP; ++E) {
            // decode the frag: stage 0 in, 1 ctx, 2 W0, 3 W1, 4 gate, 5 WH, 6 DD, -1 pad
            int stage = -1, T = 0, ks = 0, blk = 0, feat = 0, m = 0;
            if (E < W::e_blk(D, CKS, 0)) {
                T = E / (2 + CKS);
                const int r = E % (2 + CKS);
                if (r < 2) { stage = 0; ks = r; } else { stage = 1; ks = r - 2; }
            } else if (E < W::e_out(D, CKS)) {
                int r = E - W::e_blk(D, CKS, 0);
                blk = r / W::blk_len(D, CKS);
                r %= W::blk_len(D, CKS);
                if (r < W::w0_len(D)) {
                    stage = 2;
                    while (T + 1 < W::kTiles && W::w0_off(D, T + 1) <= r) ++T;
                    ks = r - W::w0_off(D, T);
                } else {
                    r -= W::w0_len(D);
                    while (T + 1 < W::kTiles && W::w1_off(D, CKS, T + 1) <= r) ++T;
                    const int rr = r - W::w1_off(D, CKS, T);
                    if (rr < W::kH16(D, T)) { stage = 3; ks = rr; } else { stage = 4; ks = rr - W::kH16(D, T); }
                }
            } else if (E < NF) {
                const int r = E - W::e_out(D, CKS);
                while (m + 1 < NB && W::out_off(D, m + 1) <= r) ++m;
                int rr = r - W::out_off(D, m);
                if (rr < W::kO16(D, 2 * m)) { stage = 5; feat = 2 * m; ks = rr; }
                else {
                    rr -= W::kO16(D, 2 * m);
                    if (rr < W::kWHb(D, m)) { stage = 5; feat = 2 * m + 1; ks = rr; }
                    else { stage = 6; ks = rr - W::kWHb(D, m); }
                }
            }
            for (int within = 0; within < 512; ++within, ++idx) {
                const int lane = within >> 3, j = within & 7;
                const int r32 = lane & 31, hf = lane >> 5;
                int64_t src = -1;
                const int pin = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hf + (j & 3);   // hidden-input k
                if (stage >= 0 && stage <= 4) {
                    const int u = perm[32 * T + r32];
                    if (stage == 0) {
                        const int d = 8 * hf + j;
                        if (d < D && hid_degree(D, u) >= d + 1) src = ro.in_w + (int64_t)u * D + d;
                    } else if (stage == 1 || stage == 4) {
                        const int col = 16 * ks + 8 * hf + j;
                        if (col < C) src = (stage == 1 ? ro.c_w : ro.g_w[blk]) + (int64_t)u * C + col;
                    } else {
                        const int uin = perm[pin];
                        if (hid_degree(D, u) >= hid_degree(D, uin))
                            src = (stage == 2 ? ro.w0_w[blk] : ro.w1_w[blk]) + (int64_t)u * H + uin;
                    }
                } else if (stage == 5) {
                    const int pm = wh_param(r32), uin = perm[pin];
                    if (pm >= 0 && feat + 1 > hid_degree(D, uin)) src = ro.out_w + ((int64_t)feat * M + pm) * H + uin;
                } else if (stage == 6) {
                    const int f = 2 * m + (r32 >> 4), pm = dd_param(r32), uin = perm[pin];
                    if (f < D && pm >= 0 && f + 1 > hid_degree(D, uin)) src = ro.out_w + ((int64_t)f * M + pm) * H + uin;
                }
                map[idx] = src < 0 ? -1 : (int32_t)(base + src);
            }
        }
    }
    for (int64_t k = 0; k < (int64_t)W::kRing * 512; ++k) map[idx++] = -1;       // the DMA's run-ahead
    if (idx != L.fragsTotal * 512) return PF_ERR_BAD_ARG;
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        for (int s = 0; s < W::kBiasFloats; ++s, ++idx) {
            int64_t src = -1;
            if (s < W::kBiasOut) {
                const int u = perm[s & 255];
                const int which = s >> 8;                      // 0 in, 1 ctx, 2.. : block (W0, W1, gate)
                if (which == 0) src = ro.in_b + u;
                else if (which == 1) { if (C > 0) src = ro.c_b + u; }
                else {
                    const int b = (which - 2) / 3, w = (which - 2) % 3;
                    if (w == 0) src = ro.w0_b[b] + u;
                    else if (w == 1) src = ro.w1_b[b] + u;
                    else if (C > 0) src = ro.g_b[b] + u;
                }
            } else {
                const int r = s - W::kBiasOut, m = r / 96, t = (r % 96) / 32, u = r % 32;
                if (t < 2) {
                    const int f = 2 * m + t, pm = wh_param(u);
                    if (f < D && pm >= 0) src = ro.out_b + (int64_t)f * M + pm;
                } else {
                    const int f = 2 * m + (u >> 4), pm = dd_param(u);
                    if (f < D && pm >= 0) src = ro.out_b + (int64_t)f * M + pm;
                }
            }
            map[idx] = src < 0 ? -1 : (int32_t)(base + src);
        }
    }
    return PF_OK;
}

// ---- PF_FLAG_BWD: transposed masked matrices of the backward chain as bf16 A-fragments (pf_flow_bwd_chain.hip) -------
// out^T[unit, row] = sum_k A[unit][k] in^T[k, row]; lane (i = lane & 15, g = lane >> 4) of fragment (tile t, k-step ks)
// holds A[16 t + i][32 ks + 8 g + j], j = 0..7.  Units in nflows order (the re-evaluated activations are).
static int build_bwd_pack_map(const FlowPlan& L, int32_t* map) {
    const RawOffsets ro = raw_offsets(L);
    const int D = L.D, H = L.H, M = L.M, NT = L.NT, HK = H / 32, KSF = L.bwd_ksf();
    int64_t idx = 0;
    auto frag = [&](auto&& src_of) {       // src_of(i, k) -> raw offset inside the layer or -1
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) map[idx++] = (int32_t)src_of(lane & 15, 8 * (lane >> 4) + j);
    };
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        auto at = [&](int64_t off) { return off < 0 ? (int64_t)-1 : base + off; };
        // gh[u] = sum_p Wf[p][u] Gp[p]            (output mask: deg_out(feature) > deg(u))
        for (int t = 0; t < NT; ++t)
            for (int ks = 0; ks < KSF; ++ks)
                frag([&](int i, int kk) {
                    const int u = 16 * t + i, p = 32 * ks + kk;
                    if (p >= D * M || !(p / M + 1 > hid_degree(D, u))) return (int64_t)-1;
                    return at(ro.out_w + (int64_t)p * H + u);
                });
        for (int j = 0; j < 2; ++j) {
            // gt1[u] = sum_k W2_j[k][u] gt2[k], then gr[u] = sum_k W1_j[k][u] gt1[k]   (hidden mask: deg(k) >= deg(u))
            for (int which = 0; which < 2; ++which) {
                const int64_t w = which == 0 ? ro.w1_w[j] : ro.w0_w[j];   // linear_layers[1] first (the chain walks backwards)
                for (int t = 0; t < NT; ++t)
                    for (int ks = 0; ks < HK; ++ks)
                        frag([&](int i, int kk) {
                            const int u = 16 * t + i, k = 32 * ks + kk;
                            if (hid_degree(D, k) < hid_degree(D, u)) return (int64_t)-1;
                            return at(w + (int64_t)k * H + u);
                        });
            }
        }
        // gu[f] = sum_k W0[k][f] gh[k]              (input mask: deg(k) >= f + 1)
        for (int ks = 0; ks < HK; ++ks)
            frag([&](int f, int kk) {
                const int k = 32 * ks + kk;
                if (f >= D || hid_degree(D, k) < f + 1) return (int64_t)-1;
                return at(ro.in_w + (int64_t)k * D + f);
            });
    }
    // forward matrices, same fragment form: A[16 t + i][32 ks + 8 g + j] = W[out unit 16 t + i][in k]
    const int C = L.C, CKB = L.bwd_ckb(), NTF = L.bwd_ntf();
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        auto at = [&](int64_t off) { return off < 0 ? (int64_t)-1 : base + off; };
        for (int t = 0; t < NT; ++t)         // x enters as hi | lo halves of one k-step: the input weights twice
            frag([&](int i, int kk) {
                const int u = 16 * t + i, d = kk & 15;
                if (d >= D || hid_degree(D, u) < d + 1) return (int64_t)-1;
                return at(ro.in_w + (int64_t)u * D + d);
            });
        if (C > 0)
            for (int m = 0; m < 3; ++m) {
                const int64_t w = m == 0 ? ro.c_w : ro.g_w[m - 1];
                for (int t = 0; t < NT; ++t)
                    for (int ks = 0; ks < CKB; ++ks)
                        frag([&](int i, int kk) {
                            const int col = 32 * ks + kk;
                            return col < C ? at(w + (int64_t)(16 * t + i) * C + col) : (int64_t)-1;
                        });
            }
        for (int j = 0; j < 2; ++j)
            for (int which = 0; which < 2; ++which) {
                const int64_t w = which == 0 ? ro.w0_w[j] : ro.w1_w[j];
                for (int t = 0; t < NT; ++t)
                    for (int ks = 0; ks < HK; ++ks)
                        frag([&](int i, int kk) {
                            const int u = 16 * t + i, k = 32 * ks + kk;
                            if (hid_degree(D, u) < hid_degree(D, k)) return (int64_t)-1;
                            return at(w + (int64_t)u * H + k);
                        });
            }
        for (int t = 0; t < NTF; ++t)
            for (int ks = 0; ks < HK; ++ks)
                frag([&](int i, int kk) {
                    const int p = 16 * t + i, k = 32 * ks + kk;
                    if (p >= D * M || !(p / M + 1 > hid_degree(D, k))) return (int64_t)-1;
                    return at(ro.out_w + (int64_t)p * H + k);
                });
    }
    if (idx != L.fragsTotal * 512) return PF_ERR_BAD_ARG;
    for (int l = 0; l < L.L; ++l) {          // biases
        const int64_t base = (int64_t)l * ro.total;
        auto run = [&](int64_t off, int n, int padded) {
            for (int i = 0; i < padded; ++i) map[idx++] = i < n ? (int32_t)(base + off + i) : -1;
        };
        run(ro.in_b, H, H);
        if (C > 0) { run(ro.c_b, H, H); run(ro.g_b[0], H, H); run(ro.g_b[1], H, H); }
        for (int j = 0; j < 2; ++j) { run(ro.w0_b[j], H, H); run(ro.w1_b[j], H, H); }
        run(ro.out_b, D * M, 16 * NTF);
    }
    return idx == L.fragsTotal * 512 + L.biasFloats ? PF_OK : PF_ERR_BAD_ARG;
}

// generic plan (pf_layout.h): every matrix as [tile][k-step][lane][elements] fragments in nflows unit order, masks from the
// degree rule, then the biases
static int build_generic_pack_map(const FlowPlan& L, int32_t* map) {
    const RawOffsets ro = raw_offsets(L);
    const int per = L.bf16 ? 8 : 4, ks_w = L.kstep;
    const int xh = L.gen_xh();
    int64_t idx = 0;
    // position p of the hidden dimension holds nflows unit unit_of[p]: degree order when the plan is sorted (the masks are then
    // block lower-triangular, csrc/pf_flow_generic.hip skips the zero k-steps), identity otherwise
    int unit_of[512];
    if (L.gsorted) sorted_units(L.D, L.H, unit_of);
    else for (int u = 0; u < L.H; ++u) unit_of[u] = u;
    // kind: 0 x input (initial layer), 1 context (unmasked), 2 hidden -> hidden, 3 hidden -> spline parameters
    auto matrix = [&](int kind, int64_t w_off, int n_rows, int n_tiles, int nks, int64_t base) {
        for (int t = 0; t < n_tiles; ++t)
            for (int ks = 0; ks < nks; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < per; ++e, ++idx) {
                        const int row = 16 * t + (lane & 15), k = ks_w * ks + per * (lane >> 4) + e;
                        int64_t src = -1;
                        if (row < n_rows) {
                            const int ur = kind == 3 ? row : unit_of[row];                        // hidden output unit of this row
                            if (kind == 0) {
                                const int d = L.bf16 ? (k < 2 * xh ? k % xh : L.D) : k;          // bf16: hi | lo halves of x
                                if (d < L.D && hid_degree(L.D, ur) >= d + 1) src = w_off + (int64_t)ur * L.D + d;
                            } else if (kind == 1) {
                                if (k < L.C) src = w_off + (int64_t)ur * L.C + k;
                            } else if (kind == 2) {
                                if (k < L.H && hid_degree(L.D, ur) >= hid_degree(L.D, unit_of[k])) src = w_off + (int64_t)ur * L.H + unit_of[k];
                            } else {
                                const int f = row / L.M;                                          // output unit row = f M + j
                                if (k < L.H && f + 1 > hid_degree(L.D, unit_of[k])) src = w_off + (int64_t)row * L.H + unit_of[k];
                            }
                        }
                        map[idx] = src < 0 ? -1 : (int32_t)(base + src);
                    }
    };
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        matrix(0, ro.in_w, L.H, L.NT, L.gKx, base);
        if (L.C > 0) {
            matrix(1, ro.c_w, L.H, L.NT, L.gKc, base);
            matrix(1, ro.g_w[0], L.H, L.NT, L.gKc, base);
            matrix(1, ro.g_w[1], L.H, L.NT, L.gKc, base);
        }
        for (int b = 0; b < 2; ++b) {
            matrix(2, ro.w0_w[b], L.H, L.NT, L.gKh, base);
            matrix(2, ro.w1_w[b], L.H, L.NT, L.gKh, base);
        }
        matrix(3, ro.out_w, L.D * L.M, L.gTf, L.gKh, base);
    }
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        auto vec = [&](int64_t off, int n, int padded, bool hidden) {
            for (int i = 0; i < padded; ++i, ++idx) map[idx] = i < n ? (int32_t)(base + off + (hidden ? unit_of[i] : i)) : -1;
        };
        vec(ro.in_b, L.H, L.H, true);
        if (L.C > 0) { vec(ro.c_b, L.H, L.H, true); vec(ro.g_b[0], L.H, L.H, true); vec(ro.g_b[1], L.H, L.H, true); }
        for (int b = 0; b < 2; ++b) { vec(ro.w0_b[b], L.H, L.H, true); vec(ro.w1_b[b], L.H, L.H, true); }
        vec(ro.out_b, L.D * L.M, 16 * L.gTf, false);
    }
    return idx == pack_map_len(L) ? PF_OK : PF_ERR_BAD_ARG;
}

int build_pack_map(const FlowPlan& L, int32_t* map) {
    if (L.generic) return build_generic_pack_map(L, map);
    if (L.bwd) return build_bwd_pack_map(L, map);
    if (L.wide) return build_wide_pack_map(L, map);
    const RawOffsets ro = raw_offsets(L);
    const int fragElems = L.bf16 ? 512 : 256;
    const int per = L.bf16 ? 8 : 4;   // elements per lane
    int perm[256];
    sorted_units(L.D, L.H, perm);
    int64_t idx = 0;
    for (int w = 0; w < L.NW; ++w) {
        for (int l = 0; l < L.L; ++l) {
            const int64_t base = (int64_t)l * ro.total;
            for (int f = 0; f < L.NF; ++f) {
                const Entry en = stream_entry(L, w, f);
                const int phase = en.phase, ks = en.ks, blk = en.blk;
                for (int within = 0; within < fragElems; ++within, ++idx) {
                    if (en.pad) { map[idx] = -1; continue; }
                    const int lane = within / per, el = within % per;
                    const int g = lane >> 4, r16 = lane & 15;
                    int col;   // k -> source column / sorted position
                    if (phase == 0) col = L.bf16 ? ((8 * g + el) & 15) : (4 * g + el);
                    else if (phase == 1 || phase == 4) col = L.bf16 ? (32 * ks + 8 * g + el) : (16 * ks + 4 * g + el);
                    else col = L.bf16 ? (16 * (2 * ks + (el >> 2)) + 4 * g + (el & 3)) : (16 * ks + 4 * g + el);
                    int64_t src = -1;
                    const int u = perm[16 * en.tile + r16];          // hidden output unit of this row
                    switch (phase) {
                    case 0:
                        if (col < L.D && hid_degree(L.D, u) >= col + 1) src = ro.in_w + (int64_t)u * L.D + col;
                        break;
                    case 1:
                        if (col < L.C) src = ro.c_w + (int64_t)u * L.C + col;
                        break;
                    case 4:
                        if (col < L.C) src = ro.g_w[blk] + (int64_t)u * L.C + col;
                        break;
                    case 2:
                    case 3: {
                        const int uin = perm[col];
                        if (hid_degree(L.D, u) >= hid_degree(L.D, uin))
                            src = (phase == 2 ? ro.w0_w[blk] : ro.w1_w[blk]) + (int64_t)u * L.H + uin;
                        break;
                    }
                    case 5: {
                        const int m = out_param(L, en.q, r16);
                        const int uin = perm[col];
                        if (m >= 0 && (en.feat + 1) > hid_degree(L.D, uin))
                            src = ro.out_w + ((int64_t)en.feat * L.M + m) * L.H + uin;
                        break;
                    }
                    }
                    map[idx] = src < 0 ? -1 : (int32_t)(base + src);
                }
            }
        }
        for (int64_t k = 0; k < (int64_t)kWindowPad * fragElems; ++k) map[idx++] = -1;   // prefetch overrun pad
    }
    if (idx != L.fragsTotal * fragElems) return PF_ERR_BAD_ARG;
    // (bias region follows; the hoisted context region comes after it)
    // bias region, [layer][tile][slot][r16]; out slots of tile t hold the biases of the
    // feature whose spline the owning wave evaluates as "A" (tile < NW) or "B" (tile >= NW)
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        for (int t = 0; t < L.NT; ++t) {
            const int w = t < L.NW ? t : L.NT - 1 - t;
            const int feat = t < L.NW ? L.featA[w] : L.featB[w];
            for (int s = 0; s < kBiasFloatsPerTile; ++s, ++idx) {
                const int slot = s >> 4, r16 = s & 15;
                const int u = perm[16 * t + r16];
                int64_t src = -1;
                if (slot == kSlotIn) src = ro.in_b + u;
                else if (slot == kSlotCtx) { if (L.C > 0 && !L.hoist) src = ro.c_b + u; }
                else if (slot >= kSlotBlk && slot < kSlotBlk + 3 * L.NB) {
                    const int b = (slot - kSlotBlk) / 3, which = (slot - kSlotBlk) % 3;
                    if (which == 0) src = ro.w0_b[b] + u;
                    else if (which == 1) src = ro.w1_b[b] + u;
                    else if (L.C > 0 && !L.hoist) src = ro.g_b[b] + u;
                } else if (slot >= kSlotOut && slot < kSlotOut + 3) {
                    const int m = out_param(L, slot - kSlotOut, r16);
                    if (feat >= 0 && m >= 0) src = ro.out_b + (int64_t)feat * L.M + m;
                }
                map[idx] = src < 0 ? -1 : (int32_t)(base + src);
            }
        }
    }
    // hoisted context projections: frags [(l*3 + j)*NT + t][ks], then biases [(l*3 + j)*NT + t][16]
    if (L.hoist) {
        for (int l = 0; l < L.L; ++l) {
            const int64_t base = (int64_t)l * ro.total;
            for (int j = 0; j < 3; ++j) {
                const int64_t wsrc = j == 0 ? ro.c_w : ro.g_w[j - 1];
                for (int t = 0; t < L.NT; ++t)
                    for (int ks = 0; ks < L.CK; ++ks)
                        for (int within = 0; within < fragElems; ++within, ++idx) {
                            const int lane = within / per, el = within % per;
                            const int g = lane >> 4, r16 = lane & 15;
                            const int col = L.bf16 ? (32 * ks + 8 * g + el) : (16 * ks + 4 * g + el);
                            const int u = perm[16 * t + r16];
                            map[idx] = col < L.C ? (int32_t)(base + wsrc + (int64_t)u * L.C + col) : -1;
                        }
            }
        }
        for (int l = 0; l < L.L; ++l) {
            const int64_t base = (int64_t)l * ro.total;
            for (int j = 0; j < 3; ++j) {
                const int64_t bsrc = j == 0 ? ro.c_b : ro.g_b[j - 1];
                for (int t = 0; t < L.NT; ++t)
                    for (int r16 = 0; r16 < 16; ++r16, ++idx)
                        map[idx] = (int32_t)(base + bsrc + perm[16 * t + r16]);
            }
        }
    }
    return PF_OK;
}

int64_t pack_map_len(const FlowPlan& L) {
    const int64_t fe = L.bf16 ? 512 : 256;
    return L.fragsTotal * fe + L.biasFloats + L.ctxFrags * fe + L.ctxBiasFloats;
}

int64_t raw_param_count(const FlowPlan& L) { return raw_offsets(L).total * L.L; }

// ---- device gather ------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ raw,
                                                        const int32_t* __restrict__ map,
                                                        __bf16* __restrict__ out, int64_t n) {
    // 8 elements per thread -> one 16-B store
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const int4 m0 = *reinterpret_cast<const int4*>(map + i);
    const int4 m1 = *reinterpret_cast<const int4*>(map + i + 4);
    const int m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    bf16x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__bf16)(m[k] < 0 ? 0.0f : raw[m[k]]);
    *reinterpret_cast<bf16x8*>(out + i) = v;
}

__global__ __launch_bounds__(256) void pack_f32_kernel(const float* __restrict__ raw,
                                                       const int32_t* __restrict__ map,
                                                       float* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int4 m = *reinterpret_cast<const int4*>(map + i);
    float4 v;
    v.x = m.x < 0 ? 0.0f : raw[m.x];
    v.y = m.y < 0 ? 0.0f : raw[m.y];
    v.z = m.z < 0 ? 0.0f : raw[m.z];
    v.w = m.w < 0 ? 0.0f : raw[m.w];
    *reinterpret_cast<float4*>(out + i) = v;
}

// generic gather used by other packers: n elements of operand type (bf16) or fp32
int launch_gather(bool bf16, const float* raw, const int32_t* map, void* out, int64_t n, hipStream_t stream) {
    if (n == 0) return PF_OK;
    if (bf16) {
        hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, raw, map,
                           reinterpret_cast<__bf16*>(out), n);
    } else {
        hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, raw, map,
                           reinterpret_cast<float*>(out), n);
    }
    return launch_status();
}

int launch_pack(const FlowPlan& L, const float* raw, const int32_t* map, void* packed,
                hipStream_t stream) {
    const int64_t fe = L.bf16 ? 512 : 256;
    auto gather_w = [&](const int32_t* m, void* dst, int64_t n) {
        if (n == 0) return;
        if (L.bf16) {
            hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, raw, m,
                               reinterpret_cast<__bf16*>(dst), n);
        } else {
            hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, raw, m,
                               reinterpret_cast<float*>(dst), n);
        }
    };
    auto gather_f = [&](const int32_t* m, void* dst, int64_t n) {
        if (n == 0) return;
        hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, raw, m,
                           reinterpret_cast<float*>(dst), n);
    };
    char* out = reinterpret_cast<char*>(packed);
    const int64_t nW = L.fragsTotal * fe, nB = L.biasFloats, nCW = L.ctxFrags * fe, nCB = L.ctxBiasFloats;
    gather_w(map, out, nW);
    gather_f(map + nW, out + L.weightBytes, nB);
    gather_w(map + nW + nB, out + L.ctx_frag_offset(), nCW);
    gather_f(map + nW + nB + nCW, out + L.ctx_bias_offset(), nCB);
    return launch_status();
}

}  // namespace pf
