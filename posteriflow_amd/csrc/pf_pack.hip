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

#include <algorithm>

#include "pf_flow_params.h"

namespace pf {

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

// runtime mirror of Sched<> (pf_layout.h)
struct Entry { int phase, ks, blk, tile; bool active; };   // phase: 0 in, 1 ctx, 2 W0, 3 W1, 4 gate, 5 out
static int sched_len(const FlowPlan& L) {
    return 1 + L.CKM + 2 * (2 * L.HK + L.CKM) + 3 * L.HK;   // pad entries are never active
}
static Entry sched_entry(const FlowPlan& L, int w, int e) {
    const int eBlk = 1 + L.CKM, blkLen = 2 * L.HK + L.CKM, eOut = eBlk + 2 * blkLen, raw = eOut + 3 * L.HK;
    Entry x{0, 0, 0, 0, false};
    if (e < 1) { x.phase = 0; x.active = true; }
    else if (e < eBlk) { x.phase = 1; x.ks = e - 1; x.active = x.ks < L.CK; }
    else if (e < eOut) {
        int r = e - eBlk; x.blk = r / blkLen; r %= blkLen;
        if (r < L.HK) { x.phase = 2; x.ks = r; x.active = x.ks < L.kmaxH[w]; }
        else if (r < 2 * L.HK) { x.phase = 3; x.ks = r - L.HK; x.active = x.ks < L.kmaxH[w]; }
        else { x.phase = 4; x.ks = r - 2 * L.HK; x.active = x.ks < L.CK; }
    } else if (e < raw) {
        const int r = e - eOut; x.phase = 5; x.tile = r / L.HK; x.ks = r % L.HK; x.active = x.ks < L.kmaxO[w];
    }
    return x;
}

int build_pack_map(const FlowPlan& L, int32_t* map) {
    const RawOffsets ro = raw_offsets(L);
    const int fragElems = L.bf16 ? 512 : 256;
    const int per = L.bf16 ? 8 : 4;   // elements per lane
    int perm[256];
    sorted_units(L.D, L.H, perm);
    const int NE = sched_len(L);
    int64_t idx = 0;
    for (int w = 0; w < L.NW; ++w) {
        if (idx != L.waveBase[w] * fragElems) return PF_ERR_BAD_ARG;
        for (int l = 0; l < L.L; ++l) {
            const int64_t base = (int64_t)l * ro.total;
            for (int e = 0; e < NE; ++e) {
                const Entry en = sched_entry(L, w, e);
                if (!en.active) continue;
                const int phase = en.phase, ks = en.ks, blk = en.blk;
                for (int within = 0; within < fragElems; ++within, ++idx) {
                    const int lane = within / per, el = within % per;
                    const int g = lane >> 4, r16 = lane & 15;
                    int col;   // k -> source column / sorted position
                    if (phase == 0) col = L.bf16 ? ((8 * g + el) & 15) : (4 * g + el);
                    else if (phase == 1 || phase == 4) col = L.bf16 ? (32 * ks + 8 * g + el) : (16 * ks + 4 * g + el);
                    else col = L.bf16 ? (16 * (2 * ks + (el >> 2)) + 4 * g + (el & 3)) : (16 * ks + 4 * g + el);
                    int64_t src = -1;
                    const int u = perm[16 * w + r16];          // hidden output unit of this row
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
                        const int m = out_param(L, en.tile, r16);
                        const int uin = perm[col];
                        const int f = L.feat[w];
                        if (f >= 0 && m >= 0 && (f + 1) > hid_degree(L.D, uin))
                            src = ro.out_w + ((int64_t)f * L.M + m) * L.H + uin;
                        break;
                    }
                    }
                    map[idx] = src < 0 ? -1 : (int32_t)(base + src);
                }
            }
        }
        for (int64_t k = 0; k < (int64_t)kWindow * fragElems; ++k) map[idx++] = -1;   // prefetch overrun pad
    }
    if (idx != L.fragsTotal * fragElems) return PF_ERR_BAD_ARG;
    // bias region, [layer][wave][slot][r16]
    for (int l = 0; l < L.L; ++l) {
        const int64_t base = (int64_t)l * ro.total;
        for (int w = 0; w < L.NW; ++w) {
            for (int s = 0; s < kBiasFloatsPerWave; ++s, ++idx) {
                const int slot = s >> 4, r16 = s & 15;
                const int u = perm[16 * w + r16];
                int64_t src = -1;
                if (slot == kSlotIn) src = ro.in_b + u;
                else if (slot == kSlotCtx) { if (L.C > 0) src = ro.c_b + u; }
                else if (slot >= kSlotBlk && slot < kSlotBlk + 3 * L.NB) {
                    const int b = (slot - kSlotBlk) / 3, which = (slot - kSlotBlk) % 3;
                    if (which == 0) src = ro.w0_b[b] + u;
                    else if (which == 1) src = ro.w1_b[b] + u;
                    else if (L.C > 0) src = ro.g_b[b] + u;
                } else if (slot >= kSlotOut && slot < kSlotOut + 3) {
                    const int m = out_param(L, slot - kSlotOut, r16);
                    if (L.feat[w] >= 0 && m >= 0) src = ro.out_b + (int64_t)L.feat[w] * L.M + m;
                }
                map[idx] = src < 0 ? -1 : (int32_t)(base + src);
            }
        }
    }
    return PF_OK;
}

int64_t pack_map_len(const FlowPlan& L) {
    return L.fragsTotal * (L.bf16 ? 512 : 256) + L.biasFloats;
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

int launch_pack(const FlowPlan& L, const float* raw, const int32_t* map, void* packed,
                hipStream_t stream) {
    const int64_t nW = L.fragsTotal * (L.bf16 ? 512 : 256);
    const int64_t nB = L.biasFloats;
    if (L.bf16) {
        const int64_t blocks = (nW / 8 + 255) / 256;
        hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, raw, map,
                           reinterpret_cast<__bf16*>(packed), nW);
    } else {
        const int64_t blocks = (nW / 4 + 255) / 256;
        hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, raw, map,
                           reinterpret_cast<float*>(packed), nW);
    }
    float* bias_out = reinterpret_cast<float*>(reinterpret_cast<char*>(packed) + L.weightBytes);
    const int64_t bblocks = (nB / 4 + 255) / 256;
    hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)bblocks), dim3(256), 0, stream, raw, map + nW,
                       bias_out, nB);
    return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

}  // namespace pf
