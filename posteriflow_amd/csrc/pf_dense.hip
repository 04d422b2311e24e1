// pf_dense.hip -- dense_nt / dense_tn / dense_pack: see pf_dense.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <type_traits>

#include "pf_dense.h"
#include <cstdlib>
#include "pf_math.h"
#include "pf_status.h"

namespace pf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// =====================================================================================================================
// packing: raw fp32 parameters -> MFMA A-fragments [tile][k-step][lane][8 bf16 | 4 fp32]
// =====================================================================================================================
int64_t dense_frag_count(bool bf16, int N, int K) { return (int64_t)(N / 16) * (K / (bf16 ? 32 : 16)) * 64; }

template <bool BF16>
__global__ __launch_bounds__(256) void dense_pack_kernel(const float* __restrict__ raw, const DensePackTable tab, u32x4* __restrict__ packed) {
    const DensePackEntry e = tab.e[blockIdx.y];
    constexpr int KSTEP = BF16 ? 32 : 16, PER = BF16 ? 8 : 4;
    const int nks = e.K / KSTEP;
    const int64_t total = (int64_t)(e.N / 16) * nks * 64;           // fragments-lanes (one u32x4 each)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const int64_t f = i >> 6;
        const int ks = (int)(f % nks), tile = (int)(f / nks);
        const int g = lane >> 4, r16 = lane & 15;
        const int n = tile * 16 + r16;
        float v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = KSTEP * ks + PER * g + j;
            int64_t src;
            if (e.mode == 0) src = e.src_off + (int64_t)n * e.ld + k;
            else if (e.mode == 1) src = e.src_off + (int64_t)k * e.ld + n;
            else if (e.mode == 3) {                      // Conv1d weight [cout][cin][kw] as im2col matrix: k = tap * cin + ch
                const int tap = k / e.cin, ci = k - tap * e.cin;
                src = e.src_off + ((int64_t)n * e.cin + ci) * e.kw + tap;
            } else {
                const int t = n / e.cin, ci = n - t * e.cin;
                const int u = k / e.cout, co = k - u * e.cout;
                src = e.src_off + ((int64_t)co * e.cin + ci) * e.kw + e.s * (e.kw / e.s - 1 - u) + t;
            }
            const bool outside = (e.n_valid > 0 && n >= e.n_valid) || (e.k_valid > 0 && k >= e.k_valid);
            v[j] = outside ? 0.f : raw[src];
        }
        u32x4 o;
        if constexpr (BF16) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
            o = __builtin_bit_cast(u32x4, b);
        } else {
            o = __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]});
        }
        if (e.nks_total > 0) packed[e.dst_off + ((int64_t)tile * e.nks_total + e.ks_off + ks) * 64 + lane] = o;
        else packed[e.dst_off + i] = o;
    }
}

int dense_pack(bool bf16, const float* raw, const DensePackTable& tab, void* packed, hipStream_t s) {
    if (tab.n <= 0) return PF_OK;
    const dim3 grid(64, (unsigned)tab.n);
    if (bf16) hipLaunchKernelGGL(dense_pack_kernel<true>, grid, dim3(256), 0, s, raw, tab, reinterpret_cast<u32x4*>(packed));
    else hipLaunchKernelGGL(dense_pack_kernel<false>, grid, dim3(256), 0, s, raw, tab, reinterpret_cast<u32x4*>(packed));
    return launch_status();
}

// =====================================================================================================================
// dense_nt: the strip kernel
// =====================================================================================================================
// RD = depth of the weight-fragment register ring in k-steps: fragment (tile i, k-step t + RD) is requested the moment
// (tile i, k-step t) has been multiplied -- across chunk, pass and epilogue boundaries -- so a wave always has RD * TP
// one-KiB loads in flight.  (First version: one k-step ahead = 24 MFMAs = 0.16 us of cover for a ~1 us L2 round trip; the
// QKV projection of 187 K rows took 180 us against 17 us of MFMA time and 58 us of HBM time.)
#ifndef PF_DENSE_BM64_MIN_ROUNDS
#define PF_DENSE_BM64_MIN_ROUNDS 1   // 64-row strips from this many strips per CU (measured better from 23 K rows up: -16 .. -26 % at 47 K / 94 K rows, -4 .. -9 % at 187 K)
#endif
#ifndef PF_DENSE_BM64_OCC
#define PF_DENSE_BM64_OCC 3     // waves per SIMD the 64-row variant is compiled for (registers: 512 / that; at 4 it spills 9-77 VGPRs and loses)
#endif
template <bool BF16, int EPI, int TP, int RD, int BM>
__global__ __launch_bounds__(256, BM == 64 ? PF_DENSE_BM64_OCC : (TP <= 3 ? 2 : 1)) void dense_strip_kernel(const DenseArgs p_in) {
    // BM rows per workgroup: 128, or 64 (half the accumulators and half the LDS image: four workgroups per CU instead of two --
    // a strip is load, multiply, store one after the other, and what overlaps them is the other workgroups of the CU)
    constexpr int CG = BM / 16;
    static_assert(BM == 128 || BM == 64, "rows per workgroup");
    constexpr int KSTEP = BF16 ? 32 : 16, ESZ = BF16 ? 2 : 4, EPC = 16 / ESZ;
    DenseArgs p = p_in;
    if constexpr (EPI == kEpiPlain) {
        // column groups (grid.z): this workgroup owns output units [col0, col0 + n_group) -- their fragments are a
        // contiguous run of tiles, their bias and output columns an offset; everything below sees a narrower matrix
        if (p.n_group > 0) {
            const int col0 = (int)blockIdx.z * p.n_group;
            p.wfrags = reinterpret_cast<const u32x4*>(p.wfrags) + (size_t)(col0 >> 4) * (p.K / KSTEP) * 64;
            if (p.bias) p.bias += col0;
            p.out = reinterpret_cast<char*>(p.out) + (size_t)col0 * ((!BF16 || p.out_f32) ? 4 : 2);
            p.N = p.N - col0 < p.n_group ? p.N - col0 : p.n_group;
        }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* s_aoff = reinterpret_cast<int64_t*>(smem);                // [BM] element offset of the row in A, -1 beyond M
    int64_t* s_ooff = s_aoff + BM;                                      // [BM] element offset of the row in out
    int64_t* s_xoff = s_ooff + BM;                                      // [BM] ... in the auxiliary operands (dact / resid / mul)
    int32_t* s_vlim = reinterpret_cast<int32_t*>(s_xoff + BM);          // [BM] output elements of the row that exist (o_valid_per_seq)
    char* img = smem + 3 * BM * sizeof(int64_t) + BM * sizeof(int32_t);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t m0 = (int64_t)blockIdx.x * BM;

    const int ntiles = p.N >> 4;
    const int nks_total = p.K / KSTEP;
    const int nks = p.KC / KSTEP;                      // k-steps per chunk (a multiple of RD: host-checked)
    const int nchunks = p.K / p.KC;
    const int n_pass = (ntiles + 4 * TP - 1) / (4 * TP);
    const int cpr = p.KC / EPC;                        // 16-byte slots per image row
    const int rowbytes = p.KC * ESZ;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    constexpr int PC = 4 * TP * 16;                    // output columns of one pass
    char* stg = img + (size_t)BM * rowbytes;           // epilogue staging: 2 x (1 | 2 outputs) x 16 rows (sized by the host)

    // ---- the fragment ring: requested before anything else so that the first chunk's staging runs under it ----------
    int f_pass = 0, f_kk = 0;
    const u32x4* fwf[TP];
    auto set_fetch_pass = [&](int ps) {
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            const int t = ps * 4 * TP + wave * TP + i;
            const int tcl = t < ntiles ? t : ntiles - 1;                 // tail: a valid tile is fetched, nothing is stored
            fwf[i] = reinterpret_cast<const u32x4*>(p.wfrags) + (size_t)tcl * nks_total * 64 + lane;
        }
    };
    set_fetch_pass(0);
    // split reduction: workgroup y of a strip takes chunks [ch_lo, ch_hi) and adds its partial sums into out
    const int ksp = p.k_splits > 1 ? p.k_splits : 1;
    const int cps = (nchunks + ksp - 1) / ksp;
    const int ch_lo = (int)blockIdx.y * cps, ch_hi = ch_lo + cps < nchunks ? ch_lo + cps : nchunks;
    if (ch_lo >= nchunks) return;      // (the launcher sizes grid.y to the non-empty splits; an empty one would prefetch weight
                                       // fragments from beyond the stream: 25 splits of 30 chunks = 15 of 2, ten empty)
    f_kk = ch_lo * nks;
    u32x4 ring[RD][TP];
    auto fetch = [&](u32x4 (&dst)[TP]) {
#pragma unroll
        for (int i = 0; i < TP; ++i) dst[i] = fwf[i][(size_t)f_kk * 64];
        if (++f_kk == nks_total) {
            f_kk = 0;
            ++f_pass;
            set_fetch_pass(f_pass < n_pass ? f_pass : n_pass - 1);
        }
    };
#pragma unroll
    for (int j = 0; j < RD; ++j) fetch(ring[j]);

    if (tid < BM) {
        const int64_t m = m0 + tid;
        int64_t ao = -1, oo = -1, xo = -1;
        int32_t vl = 0x7fffffff;
        if (m < p.M) {
            const int64_t n = m / p.rows_per_seq, pos = m - n * p.rows_per_seq;
            ao = n * p.a_seq_stride + pos * p.lda;
            oo = n * p.o_seq_stride + pos * p.ldo;
            xo = n * (p.x_seq_stride ? p.x_seq_stride : p.o_seq_stride) + pos * p.ldo;
            if (p.o_valid_per_seq > 0) {                         // (one division per row here, none per stored piece)
                const int64_t left = p.o_valid_per_seq - pos * p.ldo;
                vl = left < 0 ? 0 : (left > 0x7fffffff ? 0x7fffffff : (int32_t)left);
            }
        }
        s_aoff[tid] = ao;
        s_ooff[tid] = oo;
        s_xoff[tid] = xo;
        s_vlim[tid] = vl;
    }
    __syncthreads();

    // staging: 8 threads per row (128 contiguous bytes of one row per wave instruction group), 32 rows per sweep; RB
    // sweeps are requested before the first store (the first chunk, with no accumulators alive yet, takes all four)
    const int st_sub = tid & 7, st_row = tid >> 3;
    auto stage = [&](int ch, auto rb_tag) {
        constexpr int RB = decltype(rb_tag)::value;
        constexpr int U = BF16 ? 4 : 8;                         // slots per thread and sweep (KC <= 256)
        int64_t k0 = (int64_t)ch * p.KC;
        if (p.a_chunk_stride > 0) {
            const int sc = p.a_slab_chunks > 1 ? p.a_slab_chunks : 1;
            k0 = (int64_t)(ch / sc) * p.a_chunk_stride + (int64_t)(ch % sc) * p.KC;
        }
#pragma unroll 1
        for (int rr0 = 0; rr0 < BM / 32; rr0 += RB) {
            u32x4 v[RB][U];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int row = st_row + 32 * (rr0 + r);
                const int64_t ao = s_aoff[row];
                const char* src = Abase + (ao + k0) * ESZ;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc = st_sub + 8 * u;
                    v[r][u] = u32x4{0u, 0u, 0u, 0u};
                    if (cc < cpr && ao >= 0) v[r][u] = *reinterpret_cast<const u32x4*>(src + (size_t)cc * 16);
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int row = st_row + 32 * (rr0 + r);
                char* dst = img + (size_t)row * rowbytes;
                const int sw = row & 7;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc = st_sub + 8 * u;
                    if (cc < cpr) *reinterpret_cast<u32x4*>(dst + ((cc ^ sw) << 4)) = v[r][u];
                }
            }
        }
    };

    const int swl = c & 7;                                               // (16 cg + c) & 7
    const char* brow = img + (size_t)c * rowbytes;

#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
        f32x4 acc[TP][CG];
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) acc[i][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        int tile[TP];
#pragma unroll
        for (int i = 0; i < TP; ++i) tile[i] = pass * 4 * TP + wave * TP + i;
#pragma unroll 1
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            if (pass == 0) {                                             // (nchunks > 1 implies n_pass == 1: host-checked)
                if (ch == ch_lo) stage(ch, std::integral_constant<int, BF16 ? 2 : 1>{});
                else { __syncthreads(); stage(ch, std::integral_constant<int, 1>{}); }
                __syncthreads();
            }
#pragma unroll 1
            for (int ks0 = 0; ks0 < nks; ks0 += RD) {
#pragma unroll
                for (int j = 0; j < RD; ++j) {
                    const int ks = ks0 + j;
                    const int slot = ((4 * ks + g) ^ swl) << 4;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {                     // B fragments in two halves: 16 registers, not 32
                        u32x4 b[CG / 2];
#pragma unroll
                        for (int cg = 0; cg < CG / 2; ++cg)
                            b[cg] = *reinterpret_cast<const u32x4*>(brow + (size_t)(16 * (cg + (CG / 2) * hf)) * rowbytes + slot);
#pragma unroll
                        for (int i = 0; i < TP; ++i)
#pragma unroll
                            for (int cg = 0; cg < CG / 2; ++cg) {
                                f32x4& ac = acc[i][cg + (CG / 2) * hf];
                                if constexpr (BF16) {
                                    ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, ring[j][i]), __builtin_bit_cast(bf16x8, b[cg]), ac, 0, 0, 0);
                                } else {
                                    const f32x4 af = __builtin_bit_cast(f32x4, ring[j][i]), bf = __builtin_bit_cast(f32x4, b[cg]);
#pragma unroll
                                    for (int q = 0; q < 4; ++q) ac = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], ac, 0, 0, 0);
                                }
                            }
                    }
                    fetch(ring[j]);                                      // k-step t + RD of the flattened (pass, chunk) walk
                }
            }
        }
        // ---- epilogue: lane (g, c) holds units 16 tile + 4 g .. + 3 of row 16 cg + c -------------------------------
        // The results of a 16-row group go through a small LDS buffer and leave as WHOLE ROWS of the pass's 4 TP 16
        // columns (16 bytes per lane, consecutive lanes on consecutive bytes).  Stored straight from the accumulator
        // layout -- 8 bytes per lane, 16 rows x 32-byte pieces per instruction -- the kernels were bound by their store
        // instructions: 192 -> 768 plain 183 us, with GELU + its derivative (two outputs) 438 us, for 17 us of MFMAs.
        const uint32_t thr = enc_drop_threshold(p.drop_p);
        const float dscale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
        const int pcol0 = pass * PC;
        f32x4 b4[TP];
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            b4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && tile[i] < ntiles && blockIdx.y == 0) b4[i] = *reinterpret_cast<const f32x4*>(p.bias + tile[i] * 16 + 4 * g);
        }
        if constexpr (EPI == kEpiResid || EPI == kEpiMul) {
            // Epilogues with a second global operand (the residual stream / the gelu' factor): it is combined in the ROW phase,
            // from whole-row 16-byte loads requested Q groups ahead.  (First version: 8-byte loads in the accumulator layout,
            // each used the moment it was requested -- one exposed global round trip per 16-row group and pass: the FFN2 data
            // gradient 192 -> 768 took 106 us per workgroup for 4.6 us of MFMAs, conv2's 28 us for 1 us.)
            constexpr int OESZ = EPI == kEpiResid ? 4 : (BF16 ? 2 : 4);      // output (= operand) element size
            constexpr int EPC = 16 / OESZ, CPROW = PC * OESZ / 16, NPC = (16 * CPROW + 255) / 256;
            constexpr int Q = NPC >= 3 ? 2 : 3;                              // groups ahead (12 registers per group at NPC = 3)
            constexpr int SROW = PC * 4 + 16;                                // staged rows are fp32
            const char* auxb = EPI == kEpiResid ? reinterpret_cast<const char*>(p.resid) : reinterpret_cast<const char*>(p.mul);
            u32x4 aux[Q][NPC];
            auto fetch = [&](int cgx, u32x4 (&dst)[NPC]) {
#pragma unroll
                for (int np = 0; np < NPC; ++np) {
                    const int idx = tid + 256 * np;
                    const int r16 = idx / CPROW, ch = idx - r16 * CPROW;
                    dst[np] = u32x4{0u, 0u, 0u, 0u};
                    if (idx < 16 * CPROW) {
                        const int64_t xo = s_xoff[16 * cgx + r16];
                        const int col = pcol0 + ch * EPC;
                        const bool ok = xo >= 0 && col < p.N && col + EPC <= s_vlim[16 * cgx + r16];   // (the operand ends where the output does)
                        if (ok) dst[np] = *reinterpret_cast<const u32x4*>(auxb + (xo + col) * OESZ);
                    }
                }
            };
#pragma unroll
            for (int q = 0; q < Q; ++q) fetch(q, aux[q]);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                char* buf = stg + (cg & 1) * 16 * SROW;
                const int row = 16 * cg + c;
#pragma unroll
                for (int i = 0; i < TP; ++i) {
                    const int n0 = tile[i] * 16 + 4 * g;
                    f32x4 v = acc[i][cg] + b4[i];
                    if constexpr (EPI == kEpiResid) {
                        if (p.drop_p > 0.f) {
                            f32x4 dfac;
                            enc_drop4(p.seed, p.site, (uint32_t)((m0 + row) * p.N + n0), thr, dscale, dfac);
                            v = v * dfac;
                        }
                    }
                    *reinterpret_cast<f32x4*>(buf + c * SROW + ((wave * TP + i) * 16 + 4 * g) * 4) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int np = 0; np < NPC; ++np) {
                    const int idx = tid + 256 * np;
                    const int r16 = idx / CPROW, ch = idx - r16 * CPROW;
                    const int grow = 16 * cg + r16;
                    if (idx >= 16 * CPROW) continue;
                    const int64_t oo = s_ooff[grow];
                    const int col = pcol0 + ch * EPC;
                    if (oo < 0 || col >= p.N || col + EPC > s_vlim[grow]) continue;
                    const u32x4 a = aux[cg % Q][np];
                    char* dst = reinterpret_cast<char*>(p.out) + (oo + col) * OESZ;
                    if constexpr (OESZ == 4) {
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(buf + r16 * SROW + ch * 16);
                        const f32x4 av = __builtin_bit_cast(f32x4, a);
                        *reinterpret_cast<f32x4*>(dst) = EPI == kEpiResid ? av + sv : sv * av;
                    } else {
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(buf + r16 * SROW + ch * 32);
                        const f32x4 s1 = *reinterpret_cast<const f32x4*>(buf + r16 * SROW + ch * 32 + 16);
                        const bf16x8 mv = __builtin_bit_cast(bf16x8, a);
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (__bf16)(s0[e] * (float)mv[e]); o[4 + e] = (__bf16)(s1[e] * (float)mv[4 + e]); }
                        *reinterpret_cast<bf16x8*>(dst) = o;
                    }
                }
                if (cg + Q < CG) fetch(cg + Q, aux[cg % Q]);
            }
        } else {
        const bool f32o = !BF16 || p.out_f32;
        const int oesz = f32o ? 4 : 2;
        const int srow = PC * oesz + 16;                                 // staged row (bytes), 16-byte aligned
        const int sbuf = 16 * srow;                                      // one output's 16 rows
        constexpr int NOUT = EPI == kEpiGelu ? 2 : 1;
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            char* buf = stg + (cg & 1) * NOUT * sbuf;
            const int row = 16 * cg + c;
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                const int n0 = tile[i] * 16 + 4 * g;
                f32x4 v = acc[i][cg] + b4[i];
                f32x4 dfac = {1.f, 1.f, 1.f, 1.f};
                if constexpr (EPI == kEpiGelu) {
                    if (p.drop_p > 0.f) {
                        enc_drop4(p.seed, p.site, (uint32_t)((m0 + row) * p.N + n0), thr, dscale, dfac);
                    }
                }
                f32x4 v2 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == kEpiGelu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float ye, de;
                        if constexpr (BF16) gelu_fast_pair(v[e], ye, de);
                        else { ye = gelu_f32(v[e]); de = gelu_grad_f32(v[e]); }
                        v[e] = ye * dfac[e]; v2[e] = de * dfac[e];
                    }
                }
                char* dst = buf + c * srow + ((wave * TP + i) * 16 + 4 * g) * oesz;
                if (f32o) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                    if constexpr (NOUT == 2) *reinterpret_cast<f32x4*>(dst + sbuf) = v2;
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x4*>(dst) = o;
                    if constexpr (NOUT == 2) {
                        bf16x4 o2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o2[e] = (__bf16)v2[e];
                        *reinterpret_cast<bf16x4*>(dst + sbuf) = o2;
                    }
                }
            }
            // LDS only: the global stores of the previous group (and the weight ring) stay in flight across the barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int cprow = PC * oesz / 16;                            // 16-byte pieces per staged row
            const int epc = 16 / oesz;
            for (int idx = tid; idx < 16 * cprow; idx += 256) {
                const int r16 = idx / cprow, ch = idx - r16 * cprow;
                const int grow = 16 * cg + r16;
                const int64_t oo = s_ooff[grow];
                const int col = pcol0 + ch * epc;
                if (oo < 0 || col >= p.N || col + epc > s_vlim[grow]) continue;
                const u32x4 val = *reinterpret_cast<const u32x4*>(buf + r16 * srow + ch * 16);
                if constexpr (EPI == kEpiPlain) {
                    if (ksp > 1) {                                       // (fp32 output: host-checked)
                        const f32x4 fv = __builtin_bit_cast(f32x4, val);
                        float* o = reinterpret_cast<float*>(p.out) + oo + col;
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(o + e, fv[e]);
                        continue;
                    }
                }
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.out) + (oo + col) * oesz) = val;
                if constexpr (NOUT == 2) {
                    if (p.dact) {
                        const u32x4 val2 = *reinterpret_cast<const u32x4*>(buf + sbuf + r16 * srow + ch * 16);
                        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.dact) + (s_xoff[grow] + col) * oesz) = val2;
                    }
                }
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the last group's LDS reads, before the next pass
        __builtin_amdgcn_s_barrier();                                    // (or chunk) may touch LDS again
    }
}

template <bool BF16, int EPI, int TP, int RD, int BM>
static int launch_strip_bm(const DenseArgs& a, hipStream_t s) {
    const int oesz = (EPI == kEpiResid || EPI == kEpiMul || !BF16 || a.out_f32) ? 4 : 2;      // staged element: fp32 for the two-operand epilogues
    const size_t stage = (size_t)2 * (EPI == kEpiGelu ? 2 : 1) * 16 * (4 * TP * 16 * oesz + 16);
    const size_t lds = 3 * BM * sizeof(int64_t) + BM * sizeof(int32_t) + (size_t)BM * a.KC * (BF16 ? 2 : 4) + stage;
    auto k = dense_strip_kernel<BF16, EPI, TP, RD, BM>;
    if (lds > 160 * 1024) return PF_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
    const unsigned grid = (unsigned)((a.M + BM - 1) / BM);
    if (grid == 0) return PF_OK;
    // splits: ceil(nchunks / k_splits) chunks each -- only the splits that own a chunk are launched
    const int nchunks = a.K / a.KC, ksp = a.k_splits > 1 ? a.k_splits : 1, cps = (nchunks + ksp - 1) / ksp;
    const unsigned ny = (unsigned)((nchunks + cps - 1) / cps);
    const unsigned nz = a.n_group > 0 ? (unsigned)((a.N + a.n_group - 1) / a.n_group) : 1u;
    hipLaunchKernelGGL(k, dim3(grid, ny, nz), dim3(256), lds, s, a);
    return launch_status();
}

// 64-row strips: bf16, one chunk, three tiles per wave (the encoder's big-M GEMMs), enough strips to fill the chip twice over.
// $PF_DENSE_BM (test knob, read once): 128 never, 64 wherever built.
static int strip_rows(bool bf16, int tp, const DenseArgs& a) {
    static const int forced = [] { const char* e = getenv("PF_DENSE_BM"); return e ? atoi(e) : 0; }();
    const bool built = bf16 && tp == 3 && a.k_splits <= 1 && a.n_group == 0;
    if (!built || forced == 128) return 128;
    if (forced == 64) return 64;
    return a.M >= (int64_t)64 * 256 * PF_DENSE_BM64_MIN_ROUNDS ? 64 : 128;
}
template <bool BF16, int EPI, int TP, int RD>
static int launch_strip(const DenseArgs& a, hipStream_t s) {
    // (the two-operand epilogues lose with 64 rows -- four 16-row groups leave their operand prefetch no room: 180 -> 257 us --
    // and are not built)
    if constexpr (BF16 && TP == 3 && (EPI == kEpiPlain || EPI == kEpiGelu)) {
        if (strip_rows(true, TP, a) == 64) return launch_strip_bm<BF16, EPI, TP, RD, 64>(a, s);
    }
    return launch_strip_bm<BF16, EPI, TP, RD, 128>(a, s);
}

template <bool BF16, int EPI, int TP>
static int launch_strip_rd(const DenseArgs& a, hipStream_t s) {
    const int nks = a.KC / (BF16 ? 32 : 16);
    if constexpr (BF16) {
        // ring depth: 3 k-steps = 72 MFMAs (1150 cycles) of cover per wave, two waves per SIMD; 6 spilled 30-40 registers
        if (nks % 3 == 0) return launch_strip<BF16, EPI, TP, 3>(a, s);
        if (nks % 4 == 0) return launch_strip<BF16, EPI, TP, 4>(a, s);
        return launch_strip<BF16, EPI, TP, 2>(a, s);
    } else {
        return launch_strip<BF16, EPI, TP, 4>(a, s);          // KC % 64 == 0: nks % 4 == 0
    }
}

// Tiles per wave and pass (TP) of a strip launch -- ONE rule for the launcher and for dense_nt's LDS budget (they once
// differed for a chunked reduction over 13..15 tiles).  A chunked reduction keeps its accumulators across chunks: one pass
// over all tiles (<= 16); otherwise 3 tiles per wave and pass, 4 where that divides the tile count evenly.  Column groups
// (n_group: few-row problems whose time is one workgroup's latency) of <= 4 / <= 8 tiles take 1 / 2 tiles per wave, so that
// all four waves multiply: out_proj's 1024 x 1728 -> 512 in fp32, 64 workgroups of 4 tiles, 195 us with two waves working.
static int strip_tiles_per_pass(int ntiles, int nchunks, bool grouped) {
    if (grouped && ntiles <= 4) return 1;
    if (grouped && ntiles <= 8) return 2;
    return ((nchunks > 1 && ntiles > 12) || (nchunks == 1 && ntiles % 16 == 0 && ntiles % 12 != 0)) ? 4 : 3;
}

template <bool BF16, int EPI>
static int launch_strip_tp(const DenseArgs& a, hipStream_t s) {
    const int ntiles = (a.n_group > 0 ? a.n_group : a.N) / 16, nchunks = a.K / a.KC;      // tiles ONE workgroup owns
    if (nchunks > 1 && ntiles > 16) return PF_ERR_UNSUPPORTED;
    const int tp = strip_tiles_per_pass(ntiles, nchunks, a.n_group > 0);
    if constexpr (EPI == kEpiPlain) {                                                     // (grouping exists for the plain epilogue only)
        if (tp == 1) return launch_strip_rd<BF16, EPI, 1>(a, s);
        if (tp == 2) return launch_strip_rd<BF16, EPI, 2>(a, s);
    }
    return tp == 4 ? launch_strip_rd<BF16, EPI, 4>(a, s) : launch_strip_rd<BF16, EPI, 3>(a, s);
}

int dense_nt(bool bf16, int epilogue, const DenseArgs& a0, hipStream_t s) {
    if (a0.N % 16 || a0.KC % 64 || a0.K % a0.KC || a0.KC <= 0 || a0.M < 0) return PF_ERR_BAD_ARG;
    if (a0.M == 0) return PF_OK;
    DenseArgs a = a0;
    if (epilogue == kEpiMul && bf16 && a.out_f32) return PF_ERR_BAD_ARG;         // (operand and output share one type)
    if (a.k_splits > 1 && (epilogue != kEpiPlain || !(a.out_f32 || !bf16) || a.k_splits > a.K / a.KC)) return PF_ERR_BAD_ARG;
    // column groups (n_group): one launch, grid.z groups of output units -- plain epilogue only
    const bool groupable = epilogue == kEpiPlain && !(a.drop_p > 0.f) && a.o_valid_per_seq <= 0 && a.k_splits <= 1 && !a.dact;
    if (a.n_group < 0 || a.n_group % 16 || (a.n_group > 0 && !groupable)) return PF_ERR_BAD_ARG;
    if (a.n_group >= a.N) a.n_group = 0;
    if (a.n_group == 0 && groupable) {
        // chosen here: few strips are grouped until ~3/4 of the CUs have a workgroup (the rows are staged once per group, so
        // not beyond that); when the strips alone fill the chip nothing is grouped
        const int64_t strips = (a.M + 127) / 128;
        const int cand[4] = {256, 192, 128, 64};
        for (int i = 0; i < 4 && strips < 192; ++i) {
            if (cand[i] >= a.N) continue;
            a.n_group = cand[i];
            if (strips * ((a.N + cand[i] - 1) / cand[i]) >= 192) break;
        }
    }
    auto n_eff = [&] { return a.n_group > 0 ? a.n_group : a.N; };                 // output units ONE workgroup owns
    // the strip image (128 rows x KC) + the epilogue's staging must fit 160 KiB of LDS: halve the chunk while it does not
    // (fp32 with KC = 256: 128 KiB of image alone)
    auto lds_of = [&](int kc) {
        const int tp = strip_tiles_per_pass(n_eff() / 16, a.K / kc, a.n_group > 0);
        const int oesz = (epilogue == kEpiResid || epilogue == kEpiMul || !bf16 || a.out_f32) ? 4 : 2;
        return (size_t)3 * 128 * 8 + 128 * 4 + (size_t)128 * kc * (bf16 ? 2 : 4) + (size_t)2 * (epilogue == kEpiGelu ? 2 : 1) * 16 * (4 * tp * 16 * oesz + 16);
    };
    while (lds_of(a.KC) > 160 * 1024 && a.KC % 128 == 0 && a.a_chunk_stride <= 0 && a.k_splits <= 1) a.KC /= 2;
    // two workgroups per CU where one pass covers the output (<= 12 tiles): a 128 x 256 image + staging is 90 KB = ONE
    // workgroup of 4 waves per CU, whose staging, MFMA and store phases do not overlap with anything (conv2's data gradient:
    // 720 us at KC = 256, 500 us at KC = 128)
    if (n_eff() / 16 <= 12 && lds_of(a.KC) > 80 * 1024 && a.KC % 128 == 0 && a.a_chunk_stride <= 0 && a.k_splits <= 1) a.KC /= 2;
    if (a.KC > 256) return PF_ERR_UNSUPPORTED;                                    // the staging sweeps cover 256 operands per row
    // a chunked reduction keeps its accumulators across chunks: at most 256 units per workgroup
    if (a.K / a.KC > 1 && n_eff() > 256) {
        if (!groupable) return PF_ERR_UNSUPPORTED;
        a.n_group = 256;
    }
    switch (epilogue) {
    case kEpiPlain: return bf16 ? launch_strip_tp<true, kEpiPlain>(a, s) : launch_strip_tp<false, kEpiPlain>(a, s);
    case kEpiGelu: return bf16 ? launch_strip_tp<true, kEpiGelu>(a, s) : launch_strip_tp<false, kEpiGelu>(a, s);
    case kEpiResid: return bf16 ? launch_strip_tp<true, kEpiResid>(a, s) : launch_strip_tp<false, kEpiResid>(a, s);
    case kEpiMul: return bf16 ? launch_strip_tp<true, kEpiMul>(a, s) : launch_strip_tp<false, kEpiMul>(a, s);
    }
    return PF_ERR_BAD_ARG;
}

// =====================================================================================================================
// dense_tn: dW[n1, n2] += sum_m G[m, n1] A[m, n2] (+ db[n1] += sum_m G[m, n1]), split over m, float atomics
// =====================================================================================================================
// bf16: both operands are K-major in memory (the reduction index m is the ROW), so the MFMA operands (8 consecutive k for
// one output index per lane) are transposed reads of row-major LDS tiles: ds_read_b64_tr_b16 (a 4 x 16 block per 16-lane
// group, delivered column-major).  Tiles are [64 rows][128 columns] bf16 = 256-byte rows with the 16-byte slot XOR of the
// CDNA guide's "one image for row reads and transposed reads (b)": slot ^= ((row & 3) << 2) | ((row >> 2) & 3).
__device__ __forceinline__ int tn_off(int row, int slot) { return 256 * row + 16 * (slot ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

struct SeqCursor {                 // row m -> (sequence, position) advanced incrementally (no division in the loop)
    int64_t seq, pos;
    __device__ void init(int64_t m, int64_t rps) { seq = m / rps; pos = m - seq * rps; }
    __device__ void advance(int64_t d, int64_t rps) { pos += d; while (pos >= rps) { pos -= rps; ++seq; } }
};

// WG tile = (16 TI WR) x (16 TJ WC) outputs, WR x WC waves of TI x TJ MFMA tiles.  Every operand byte is read once per
// output tile of the OTHER dimension, so the tile decides the traffic: at 128 x 128 the FFN weight gradients (192 x 768)
// moved 1.0 GB for 360 MB of operands; 192 x 256 / 256 x 192 with eight waves move 504 MB.
//
// Staging: LDS-DMA into a ring of NS stages.  (First version: staged through registers, ONE chunk in flight -- a chunk's
// MFMAs, a few hundred cycles, cannot cover a global round trip, and a workgroup took 2.0-2.8 us per 64-row chunk whatever
// its tile; the mixer's 768 x 192 weight gradient 155 us, 126 us in this form, of which ~85 us are the 504 MB of operand
// reads at 5.9 TB/s and the rest the atomics of the tail.)  The operands go global -> LDS directly (global_load_lds_dwordx4:
// one wave instruction = 1 KiB = 4 rows of a [BK][128] sub-image) into a ring of NS stages, NS - 1 chunks in flight across
// the (raw) barriers, with a counted vmcnt.  The DMA writes lane-linearly, so the XOR swizzle of tn_off sits on the SOURCE
// address: the lane that fills physical slot s of row r fetches logical slot s ^ f(r) of that row (the same involution as
// the reads).  Rows beyond M and columns beyond the matrix read a 16-byte page of zeros instead.
__device__ __attribute__((aligned(16))) const uint32_t tn_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ uint32_t tn_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WR, int WC, int TI, int TJ, int BK, int NS>
__global__ __launch_bounds__(64 * WR * WC) void dense_tn_bf16_dma_kernel(const DenseTnArgs p) {
    constexpr int NW = WR * WC;
    constexpr int BN1 = 16 * TI * WR, BN2 = 16 * TJ * WC;
    constexpr int SUB1 = (BN1 + 127) / 128, SUB2 = (BN2 + 127) / 128, NSUB = SUB1 + SUB2;
    constexpr int SUBB = BK * 256;                     // one [BK][128] bf16 sub-image
    constexpr int STB = NSUB * SUBB;                   // one stage
    constexpr int RG = BK / 4;                         // 4-row pieces per sub-image
    constexpr int PPS = NSUB * RG;                     // 1-KiB pieces per stage
    constexpr int PW = PPS / NW;                       // pieces per wave and stage
    static_assert(PPS % NW == 0 && BK % 32 == 0 && NS >= 2 && PW * (NS - 2) < 64, "piece schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave - wr * WC;
    const int g = lane >> 4, lam = lane & 15, q = lam >> 2, pq = lam & 3;
    const int tiles2 = (p.N2 + BN2 - 1) / BN2;
    const int t1 = blockIdx.x / tiles2, t2 = blockIdx.x - t1 * tiles2;
    const int n1_0 = t1 * BN1, n2_0 = t2 * BN2;

    const int64_t total_chunks = (p.M + BK - 1) / BK;
    const int64_t per = (total_chunks + p.splits - 1) / p.splits;
    const int64_t c_lo = (int64_t)blockIdx.y * per, c_hi = c_lo + per < total_chunks ? c_lo + per : total_chunks;
    if (c_lo >= c_hi) return;
    const int nch = (int)(c_hi - c_lo);

    const int64_t bz = blockIdx.z;
    const char* Gb = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(p.G) + bz * p.g_batch_stride + n1_0);
    const char* Ab = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(p.A) + bz * p.a_batch_stride + n2_0);
    float* const dWz = p.dW + bz * p.w_batch_stride;
    float* const dbz = p.db ? p.db + bz * p.b_batch_stride : nullptr;
    const int n1_lim = p.n1_rows > 0 ? p.n1_rows : p.N1, n2_lim = p.n2_cols > 0 ? p.n2_cols : p.N2;

    // this lane's part of piece k of a stage (piece index wave + NW k: sub-image, 4-row group): the byte offset of its 16
    // bytes inside a row of the operand (or -1: outside the tile / the matrix) and the row's (sequence, position) cursor
    int coff[PW];
    SeqCursor base, cur[PW];
    base.init(c_lo * BK, p.rows_per_seq);              // (wave-uniform: one scalar division)
    int64_t mrow[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        const int pi = wave + NW * k, sub = pi / RG, rg = pi - sub * RG;
        const int logical = lam ^ ((g << 2) | (rg & 3));
        const bool isg = sub < SUB1;
        const int col = 128 * (isg ? sub : sub - SUB1) + 8 * logical;
        const bool ok = isg ? (col < BN1 && n1_0 + col + 8 <= p.N1) : (col < BN2 && n2_0 + col + 8 <= p.N2);
        coff[k] = ok ? 2 * col : -1;
        mrow[k] = c_lo * BK + 4 * rg + g;
        cur[k] = base;
        cur[k].advance(4 * rg + g, p.rows_per_seq);
    }
    const uint32_t lds0 = tn_lds_addr(smem);
    auto issue = [&](int stage) {                      // the next chunk in order (the cursors advance)
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int pi = wave + NW * k;
            const bool isg = pi / RG < SUB1;
            const char* src = reinterpret_cast<const char*>(tn_zero16);
            if (coff[k] >= 0 && mrow[k] < p.M)
                src = (isg ? Gb + 2 * (cur[k].seq * p.g_seq_stride + cur[k].pos * p.ldg)
                           : Ab + 2 * (cur[k].seq * p.a_seq_stride + cur[k].pos * p.lda)) + coff[k];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             reinterpret_cast<__attribute__((address_space(3))) void*>(lds0 + stage * STB + pi * 1024),
                                             16, 0, 0);
            mrow[k] += BK;
            cur[k].advance(BK, p.rows_per_seq);
        }
    };

    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // db = column sums of G: one more MFMA per G fragment against a vector of ones, in the waves of tile column 0
    f32x4 accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_db = p.db && t2 == 0 && wc == 0;
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    // The fragment reads are inline asm: in front of an LDS read it can see, the compiler waits vmcnt(0) for every DMA in
    // flight (it cannot tell which stage a DMA writes), which would drain the ring at every chunk.
    // fragment (column block col0, k-step kk): rows 32 kk + 8 g .. + 7 of column col0 + lam = two 4 x 16 blocks
    uint32_t aoff[BK / 32][TI][2], boff[BK / 32][TJ][2];
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
        const int r0 = 32 * kk + 8 * g;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int col0 = 16 * TI * wr + 16 * i, slot = ((col0 & 127) >> 3) + (pq >> 1);
            aoff[kk][i][0] = (col0 >> 7) * SUBB + tn_off(r0 + q, slot) + 8 * (pq & 1);
            aoff[kk][i][1] = (col0 >> 7) * SUBB + tn_off(r0 + 4 + q, slot) + 8 * (pq & 1);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col0 = 16 * TJ * wc + 16 * j, slot = ((col0 & 127) >> 3) + (pq >> 1);
            boff[kk][j][0] = SUB1 * SUBB + (col0 >> 7) * SUBB + tn_off(r0 + q, slot) + 8 * (pq & 1);
            boff[kk][j][1] = SUB1 * SUBB + (col0 >> 7) * SUBB + tn_off(r0 + 4 + q, slot) + 8 * (pq & 1);
        }
    }
    auto rd_tr = [&](uint32_t addr) -> u32x2 {
        u32x2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };

#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nch) issue(s0);
    int stage = 0, fill = NS - 1;                      // stage read by this iteration / filled by it
#pragma unroll 1
    for (int it = 0; it < nch; ++it) {
        // this wave's pieces of chunk `it` have landed when at most the later chunks' pieces are outstanding
        const int later = nch - 1 - it;
        if (NS >= 4 && later >= 2) wait_vmcnt<PW * (NS >= 4 ? 2 : 0)>();
        else if (NS >= 3 && later >= 1) wait_vmcnt<PW * (NS >= 3 ? 1 : 0)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // everyone's pieces; and everyone is done reading stage `fill`
        if (it + NS - 1 < nch) issue(fill);
        const uint32_t sbase = lds0 + stage * STB;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            u32x2 al[TI][2], bl[TJ][2];
#pragma unroll
            for (int i = 0; i < TI; ++i) { al[i][0] = rd_tr(sbase + aoff[kk][i][0]); al[i][1] = rd_tr(sbase + aoff[kk][i][1]); }
#pragma unroll
            for (int j = 0; j < TJ; ++j) { bl[j][0] = rd_tr(sbase + boff[kk][j][0]); bl[j][1] = rd_tr(sbase + boff[kk][j][1]); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bf16x8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                asm volatile("" : "+v"(al[i][0]), "+v"(al[i][1]));        // (uses stay behind the wait)
                af[i] = __builtin_bit_cast(bf16x8, u32x4{al[i][0][0], al[i][0][1], al[i][1][0], al[i][1][1]});
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                asm volatile("" : "+v"(bl[j][0]), "+v"(bl[j][1]));
                bfr[j] = __builtin_bit_cast(bf16x8, u32x4{bl[j][0][0], bl[j][0][1], bl[j][1][0], bl[j][1][1]});
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (want_db) {
#pragma unroll
                for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
            }
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }
    // lane (g, lam): dW[n1_0 + 16 TI wr + 16 i + 4 g + r][n2_0 + 16 TJ wc + 16 j + lam]; the mask values of a tile are
    // requested together, ahead of its atomics (one load + wait per element otherwise)
    auto flush = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int n2 = n2_0 + 16 * TJ * wc + 16 * j + lam;
                int col = n2;
                if (p.conv_cin > 0) { const int tap = n2 / p.conv_cin, chn = n2 - tap * p.conv_cin; col = chn * p.conv_kw + tap; }
                float mk[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (MASKED) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n1 = n1_0 + 16 * TI * wr + 16 * i + 4 * g + r;
                        if (n2 < n2_lim && n1 < n1_lim) mk[r] = p.mask[(int64_t)n1 * p.ldw + col];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n1 = n1_0 + 16 * TI * wr + 16 * i + 4 * g + r;
                    if (n2 < n2_lim && n1 < n1_lim) atomicAdd(dWz + (int64_t)n1 * p.ldw + col, mk[r] * acc[i][j][r]);
                }
            }
    };
    if (p.mask) flush(std::true_type{});
    else flush(std::false_type{});
    if (want_db && lam == 0) {                         // every column of accb holds the sums: lane (g, 0) adds rows 4 g + r
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + 16 * TI * wr + 16 * i + 4 * g + r;
                if (n1 < n1_lim) atomicAdd(dbz + n1, accb[i][r]);
            }
    }
}

// fp32: v_mfma_f32_16x16x4_f32 takes ONE value per lane (A[i = lane & 15][k = lane >> 4]), so plain ds_read_b32 of
// row-major tiles [32 rows][64 + 16 columns] serve as the transposed reads (row stride = 16 mod 32 banks: conflict-free)
__global__ __launch_bounds__(256) void dense_tn_f32_kernel(const DenseTnArgs p) {
    constexpr int BK = 32, BN = 64, LD = BN + 16;
    __shared__ __attribute__((aligned(16))) float sG[BK * LD];
    __shared__ __attribute__((aligned(16))) float sA[BK * LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, lam = lane & 15;
    const int tiles2 = (p.N2 + BN - 1) / BN;
    const int t1 = blockIdx.x / tiles2, t2 = blockIdx.x - t1 * tiles2;
    const int n1_0 = t1 * BN, n2_0 = t2 * BN;
    const int64_t total_chunks = (p.M + BK - 1) / BK;
    const int64_t per = (total_chunks + p.splits - 1) / p.splits;
    const int64_t c_lo = (int64_t)blockIdx.y * per, c_hi = c_lo + per < total_chunks ? c_lo + per : total_chunks;
    if (c_lo >= c_hi) return;

    // staging: 16 slots of 4 floats per row, 16 rows per sweep, 2 sweeps
    const int st_slot = tid & 15, st_row = tid >> 4;
    const bool g_ok = n1_0 + 4 * st_slot + 4 <= p.N1, a_ok = n2_0 + 4 * st_slot + 4 <= p.N2;
    const int64_t bz = blockIdx.z;
    const float* Gb = reinterpret_cast<const float*>(p.G) + bz * p.g_batch_stride + n1_0 + 4 * st_slot;
    const float* Ab = reinterpret_cast<const float*>(p.A) + bz * p.a_batch_stride + n2_0 + 4 * st_slot;
    float* const dWz = p.dW + bz * p.w_batch_stride;
    float* const dbz = p.db ? p.db + bz * p.b_batch_stride : nullptr;
    const int n1_lim = p.n1_rows > 0 ? p.n1_rows : p.N1, n2_lim = p.n2_cols > 0 ? p.n2_cols : p.N2;
    SeqCursor cur[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) cur[i].init(c_lo * BK + st_row + 16 * i, p.rows_per_seq);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    f32x4 vg[2], va[2];
    auto fetch = [&](int64_t chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = chunk * BK + st_row + 16 * i;
            vg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            va[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < p.M) {
                if (g_ok) vg[i] = *reinterpret_cast<const f32x4*>(Gb + cur[i].seq * p.g_seq_stride + cur[i].pos * p.ldg);
                if (a_ok) va[i] = *reinterpret_cast<const f32x4*>(Ab + cur[i].seq * p.a_seq_stride + cur[i].pos * p.lda);
            }
            cur[i].advance(BK, p.rows_per_seq);
        }
    };
    fetch(c_lo);
#pragma unroll 1
    for (int64_t chunk = c_lo; chunk < c_hi; ++chunk) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(sG + (st_row + 16 * i) * LD + 4 * st_slot) = vg[i];
            *reinterpret_cast<f32x4*>(sA + (st_row + 16 * i) * LD + 4 * st_slot) = va[i];
        }
        __syncthreads();
        if (chunk + 1 < c_hi) fetch(chunk + 1);
        if (p.db && t2 == 0 && tid < BN) {
            float sacc = 0.f;
#pragma unroll 8
            for (int r = 0; r < BK; ++r) sacc += sG[r * LD + tid];
            bsum += sacc;
        }
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = sG[(4 * ks + g) * LD + 32 * wr + 16 * i + lam];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = sA[(4 * ks + g) * LD + 32 * wc + 16 * j + lam];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n2 = n2_0 + 32 * wc + 16 * j + lam;
            if (n2 >= n2_lim) continue;
            int col = n2;
            if (p.conv_cin > 0) { const int tap = n2 / p.conv_cin, chn = n2 - tap * p.conv_cin; col = chn * p.conv_kw + tap; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + 32 * wr + 16 * i + 4 * g + r;
                if (n1 < n1_lim) {
                    const int64_t o = (int64_t)n1 * p.ldw + col;
                    atomicAdd(dWz + o, p.mask ? p.mask[o] * acc[i][j][r] : acc[i][j][r]);
                }
            }
        }
    if (dbz && t2 == 0 && tid < BN) {
        const int n1 = n1_0 + tid;
        if (n1 < n1_lim) atomicAdd(dbz + n1, bsum);
    }
}

template <int WR, int WC, int TI, int TJ, int BK = 32, int NS = 4>
static int launch_tn_bf16(DenseTnArgs a, hipStream_t s) {
    constexpr int BN1 = 16 * TI * WR, BN2 = 16 * TJ * WC;
    constexpr int NSUB = (BN1 + 127) / 128 + (BN2 + 127) / 128;
    constexpr int bk = BK;
    constexpr int lds = NS * NSUB * BK * 256;
    auto k = dense_tn_bf16_dma_kernel<WR, WC, TI, TJ, BK, NS>;
    if (lds > 160 * 1024) return PF_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 - 256 && !opt_in_lds(reinterpret_cast<const void*>(k), lds)) return PF_ERR_HIP;
    const int tiles = ((a.N1 + BN1 - 1) / BN1) * ((a.N2 + BN2 - 1) / BN2);
    const int64_t chunks = (a.M + bk - 1) / bk;
    const int nb = a.batch > 1 ? a.batch : 1;
    if (a.splits <= 0) {
        // ONE round of workgroups: every workgroup ends in float atomics of its whole tile (1.3 TB/s chip-wide against
        // 5 TB/s of operand reads), so the split count is the atomic volume: 170 splits of the FFN weight gradient were
        // 92 MB of atomics = 70 of its 160 us; and never a few workgroups more than a whole round
        const int per_cu = lds > 80 * 1024 ? 1 : 2;
        int64_t sp = (256 * per_cu) / ((int64_t)tiles * nb);
        sp = std::min<int64_t>(sp, std::max<int64_t>(1, chunks * bk / 256));
        a.splits = (int)std::max<int64_t>(1, sp);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)tiles, (unsigned)a.splits, (unsigned)nb), dim3(64 * WR * WC), lds, s, a);
    return launch_status();
}

int dense_tn(bool bf16, const DenseTnArgs& a0, hipStream_t s) {
    if (a0.M <= 0) return PF_OK;
    if (a0.N1 <= 0 || a0.N2 <= 0 || a0.N1 % (bf16 ? 8 : 4) || a0.N2 % (bf16 ? 8 : 4)) return PF_ERR_BAD_ARG;
    DenseTnArgs a = a0;
    if (bf16) {
        // operand traffic of a tiling: every G byte is read once per N2 tile, every A byte once per N1 tile
        auto cost = [&](int bn1, int bn2) {
            return (int64_t)a.N1 * ((a.N2 + bn2 - 1) / bn2) + (int64_t)a.N2 * ((a.N1 + bn1 - 1) / bn1);
        };
        const int64_t c128 = cost(128, 128), ca = cost(192, 256), cb = cost(256, 192);
        // narrow gradients against wide (overlapping-window) inputs = the convolutions' weight gradients: a tile as wide as
        // the im2col row, so that the 4-8x redundant window reads happen once (conv2: 64 x 512 outputs, 600 us with
        // 128 x 128 tiles of which half the rows were empty and the input was read four times)
        static const int forced_cfg = [] { const char* e = getenv("PF_TN_CFG"); return e ? atoi(e) : -1; }();   // tuning runs only
        if (forced_cfg >= 0) {                           // (scripts/time_tn_flow.py; read once per process)
            switch (forced_cfg) {
            case 0: return launch_tn_bf16<1, 4, 2, 1>(a, s);
            case 1: return launch_tn_bf16<2, 2, 2, 2>(a, s);
            case 2: return launch_tn_bf16<2, 2, 4, 4>(a, s);
            case 3: return launch_tn_bf16<2, 4, 6, 4>(a, s);
            case 4: return launch_tn_bf16<4, 2, 4, 6>(a, s);
            case 5: return launch_tn_bf16<1, 8, 4, 4, 32, 3>(a, s);
            case 6: return launch_tn_bf16<2, 2, 2, 4>(a, s);
            }
        }
        if (a.N1 <= 32 && a.N2 <= 64) return launch_tn_bf16<1, 4, 2, 1>(a, s);
        // few rows (the flow's weight gradients: 2048 rows x 10 layers per launch): a workgroup's time is its chain of
        // chunk round trips (~2 us each, one chunk prefetched), so many small tiles with few chunks each beat the
        // traffic-optimal large ones (256 x 256 x 10 at 2048 rows: 64 x 128 tiles 20.8 us, 192 x 256 tiles 30.9 us)
        if (a.M <= 4096) return a.N2 <= 64 ? launch_tn_bf16<2, 2, 2, 2>(a, s) : launch_tn_bf16<2, 2, 2, 4>(a, s);
        if (a.N1 <= 128 && a.N2 > 256) return launch_tn_bf16<1, 8, 4, 4, 32, 3>(a, s);  // (128 x 512 per workgroup spills 160 registers)
        if (a.N1 <= 128) return launch_tn_bf16<2, 2, 4, 4>(a, s);
        // one large tile = every workgroup adds the WHOLE matrix at the end: 192 x 192 as one 192 x 256 tile 84 us (256 x 147 KB
        // of atomics), as four 128 x 128 tiles 72 us
        if (a.N1 <= 192 && a.N2 <= 256) return launch_tn_bf16<2, 2, 4, 4>(a, s);
        if (ca <= cb && ca < c128) return launch_tn_bf16<2, 4, 6, 4>(a, s);
        if (cb < c128) return launch_tn_bf16<4, 2, 4, 6>(a, s);
        return launch_tn_bf16<2, 2, 4, 4>(a, s);
    }
    const int bn = 64, bk = 32;
    const int tiles = ((a.N1 + bn - 1) / bn) * ((a.N2 + bn - 1) / bn);
    const int64_t chunks = (a.M + bk - 1) / bk;
    const int nb = a.batch > 1 ? a.batch : 1;
    if (a.splits <= 0) {
        int64_t sp = std::max<int64_t>(1, 768 / ((int64_t)tiles * nb));
        sp = std::min<int64_t>(sp, std::max<int64_t>(1, chunks / 4));
        a.splits = (int)std::max<int64_t>(1, sp);
    }
    hipLaunchKernelGGL(dense_tn_f32_kernel, dim3((unsigned)tiles, (unsigned)a.splits, (unsigned)nb), dim3(256), 0, s, a);
    return launch_status();
}

}  // namespace pf
