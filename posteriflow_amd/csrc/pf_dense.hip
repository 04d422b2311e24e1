// pf_dense.hip -- dense_nt / dense_tn / dense_pack: see pf_dense.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "pf_dense.h"
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
            v[j] = raw[src];
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
        packed[e.dst_off + i] = o;
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
template <bool BF16, int EPI, int TP>
__global__ __launch_bounds__(256) void dense_strip_kernel(const DenseArgs p) {
    constexpr int BM = 128, CG = 8;
    constexpr int KSTEP = BF16 ? 32 : 16, ESZ = BF16 ? 2 : 4, EPC = 16 / ESZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* s_aoff = reinterpret_cast<int64_t*>(smem);                // [BM] element offset of the row in A, -1 beyond M
    int64_t* s_ooff = s_aoff + BM;                                      // [BM] element offset of the row in out
    int64_t* s_xoff = s_ooff + BM;                                      // [BM] ... in the auxiliary operands (dact / resid / mul)
    char* img = smem + 3 * BM * sizeof(int64_t);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t m0 = (int64_t)blockIdx.x * BM;

    if (tid < BM) {
        const int64_t m = m0 + tid;
        int64_t ao = -1, oo = -1, xo = -1;
        if (m < p.M) {
            const int64_t n = m / p.rows_per_seq, pos = m - n * p.rows_per_seq;
            ao = n * p.a_seq_stride + pos * p.lda;
            oo = n * p.o_seq_stride + pos * p.ldo;
            xo = n * (p.x_seq_stride ? p.x_seq_stride : p.o_seq_stride) + pos * p.ldo;
        }
        s_aoff[tid] = ao;
        s_ooff[tid] = oo;
        s_xoff[tid] = xo;
    }
    __syncthreads();

    const int ntiles = p.N >> 4;
    const int nks_total = p.K / KSTEP;
    const int nks = p.KC / KSTEP;                      // k-steps per chunk
    const int nchunks = p.K / p.KC;
    const int n_pass = (ntiles + 4 * TP - 1) / (4 * TP);
    const int cpr = p.KC / EPC;                        // 16-byte slots per image row
    const int rowbytes = p.KC * ESZ;
    const char* Abase = reinterpret_cast<const char*>(p.A);

    // staging: 8 threads per row (128 contiguous bytes of one row per wave instruction group), 32 rows per sweep
    const int st_sub = tid & 7, st_row = tid >> 3;
    auto stage = [&](int ch) {
        const int64_t k0 = (int64_t)ch * p.KC;
#pragma unroll 1
        for (int rr = 0; rr < BM / 32; ++rr) {
            const int row = st_row + 32 * rr;
            const int64_t ao = s_aoff[row];
            const char* src = Abase + (ao + k0) * ESZ;
            char* dst = img + (size_t)row * rowbytes;
            const int sw = row & 7;
            for (int cc0 = 0; cc0 < cpr; cc0 += 32) {          // up to 4 slots in flight per thread
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = cc0 + st_sub + 8 * u;
                    v[u] = u32x4{0u, 0u, 0u, 0u};
                    if (cc < cpr && ao >= 0) v[u] = *reinterpret_cast<const u32x4*>(src + (size_t)cc * 16);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = cc0 + st_sub + 8 * u;
                    if (cc < cpr) *reinterpret_cast<u32x4*>(dst + ((cc ^ sw) << 4)) = v[u];
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
        const u32x4* wf[TP];
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            tile[i] = pass * 4 * TP + wave * TP + i;
            const int tcl = tile[i] < ntiles ? tile[i] : ntiles - 1;    // tail: a valid tile is computed, nothing stored
            wf[i] = reinterpret_cast<const u32x4*>(p.wfrags) + (size_t)tcl * nks_total * 64 + lane;
        }
#pragma unroll 1
        for (int ch = 0; ch < nchunks; ++ch) {
            if (pass == 0) {                                             // (nchunks > 1 implies n_pass == 1: host-checked)
                if (ch > 0) __syncthreads();
                stage(ch);
                __syncthreads();
            }
            const int ksb = ch * nks;
            u32x4 a_cur[TP], a_nxt[TP];
#pragma unroll
            for (int i = 0; i < TP; ++i) a_cur[i] = wf[i][(size_t)ksb * 64];
            for (int ks = 0; ks < nks; ++ks) {
                const int kn = ks + 1 < nks ? ks + 1 : ks;
#pragma unroll
                for (int i = 0; i < TP; ++i) a_nxt[i] = wf[i][(size_t)(ksb + kn) * 64];
                const int slot = ((4 * ks + g) ^ swl) << 4;
                u32x4 b[CG];
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
                    b[cg] = *reinterpret_cast<const u32x4*>(brow + (size_t)(16 * cg) * rowbytes + slot);
#pragma unroll
                for (int i = 0; i < TP; ++i)
#pragma unroll
                    for (int cg = 0; cg < CG; ++cg) {
                        if constexpr (BF16) {
                            acc[i][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, a_cur[i]), __builtin_bit_cast(bf16x8, b[cg]), acc[i][cg], 0, 0, 0);
                        } else {
                            const f32x4 af = __builtin_bit_cast(f32x4, a_cur[i]), bf = __builtin_bit_cast(f32x4, b[cg]);
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[i][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], acc[i][cg], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int i = 0; i < TP; ++i) a_cur[i] = a_nxt[i];
            }
        }
        // ---- epilogue: lane (g, c) holds units 16 tile + 4 g .. + 3 of row 16 cg + c -------------------------------
        const uint32_t thr = enc_drop_threshold(p.drop_p);
        const float dscale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            if (tile[i] >= ntiles) continue;
            const int n0 = tile[i] * 16 + 4 * g;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + n0);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const int row = 16 * cg + c;
                const int64_t oo = s_ooff[row];
                if (oo < 0) continue;
                if (p.o_valid_per_seq > 0) {
                    const int64_t m = m0 + row;
                    const int64_t pos = m - (m / p.rows_per_seq) * p.rows_per_seq;
                    if (pos * p.ldo + n0 + 4 > p.o_valid_per_seq) continue;
                }
                const int64_t off = oo + n0, xoff = s_xoff[row] + n0;
                f32x4 v = acc[i][cg] + b4;
                f32x4 dfac = {1.f, 1.f, 1.f, 1.f};
                if constexpr (EPI == kEpiGelu || EPI == kEpiResid) {
                    if (p.drop_p > 0.f) {
                        const uint32_t idx = (uint32_t)((m0 + row) * p.N + n0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dfac[e] = enc_drop_hash(p.seed, p.site, idx + e) >= thr ? dscale : 0.f;
                    }
                }
                if constexpr (EPI == kEpiGelu) {
                    f32x4 y, dy;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float ye, de;
                        if constexpr (BF16) gelu_fast_pair(v[e], ye, de);
                        else { ye = gelu_f32(v[e]); de = gelu_grad_f32(v[e]); }
                        y[e] = ye; dy[e] = de;
                    }
                    y = y * dfac; dy = dy * dfac;
                    if constexpr (BF16) {
                        bf16x4 o, d;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (__bf16)y[e]; d[e] = (__bf16)dy[e]; }
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + off) = o;
                        if (p.dact) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.dact) + xoff) = d;
                    } else {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + off) = y;
                        if (p.dact) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dact) + xoff) = dy;
                    }
                } else if constexpr (EPI == kEpiResid) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + xoff);
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + off) = r + v * dfac;
                } else {
                    if constexpr (EPI == kEpiMul) {
                        if constexpr (BF16) {
                            const bf16x4 mv = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.mul) + xoff);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= (float)mv[e];
                        } else {
                            v = v * *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.mul) + xoff);
                        }
                    }
                    if (BF16 && !p.out_f32) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + off) = o;
                    } else {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + off) = v;
                    }
                }
            }
        }
    }
}

template <bool BF16, int EPI, int TP>
static int launch_strip(const DenseArgs& a, hipStream_t s) {
    const size_t lds = 3 * 128 * sizeof(int64_t) + (size_t)128 * a.KC * (BF16 ? 2 : 4);
    auto k = dense_strip_kernel<BF16, EPI, TP>;
    if (lds > 160 * 1024) return PF_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && !opt_in_lds(reinterpret_cast<const void*>(k), (int)lds)) return PF_ERR_HIP;
    const unsigned grid = (unsigned)((a.M + 127) / 128);
    if (grid == 0) return PF_OK;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    return launch_status();
}

template <bool BF16, int EPI>
static int launch_strip_tp(const DenseArgs& a, hipStream_t s) {
    const int ntiles = a.N / 16, nchunks = a.K / a.KC;
    // a chunked reduction keeps its accumulators across chunks: one pass over all tiles; otherwise the tile count per
    // wave and pass that leaves the fewest waves idle in the last pass
    int tp;
    if (nchunks > 1) {
        tp = (ntiles + 3) / 4;
        if (tp > 4) return PF_ERR_UNSUPPORTED;
    } else if (ntiles % 12 == 0) tp = 3;
    else if (ntiles % 16 == 0) tp = 4;
    else tp = std::min(4, (ntiles + 3) / 4);
    switch (tp) {
    case 1: return launch_strip<BF16, EPI, 1>(a, s);
    case 2: return launch_strip<BF16, EPI, 2>(a, s);
    case 3: return launch_strip<BF16, EPI, 3>(a, s);
    default: return launch_strip<BF16, EPI, 4>(a, s);
    }
}

int dense_nt(bool bf16, int epilogue, const DenseArgs& a, hipStream_t s) {
    if (a.N % 16 || a.KC % 64 || a.K % a.KC || a.KC <= 0 || a.M < 0) return PF_ERR_BAD_ARG;
    if (a.M == 0) return PF_OK;
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

__global__ __launch_bounds__(256) void dense_tn_bf16_kernel(const DenseTnArgs p) {
    constexpr int BK = 64, BN = 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * BK * BN * 2];
    char* sG = smem;
    char* sA = smem + BK * BN * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, lam = lane & 15, q = lam >> 2, pq = lam & 3;
    const int tiles2 = (p.N2 + BN - 1) / BN;
    const int t1 = blockIdx.x / tiles2, t2 = blockIdx.x - t1 * tiles2;
    const int n1_0 = t1 * BN, n2_0 = t2 * BN;

    const int64_t total_chunks = (p.M + BK - 1) / BK;
    const int64_t per = (total_chunks + p.splits - 1) / p.splits;
    const int64_t c_lo = (int64_t)blockIdx.y * per, c_hi = c_lo + per < total_chunks ? c_lo + per : total_chunks;
    if (c_lo >= c_hi) return;

    // staging map: 16 slots per row, 16 rows per sweep, 4 sweeps
    const int st_slot = tid & 15, st_row = tid >> 4;
    const bool g_ok = n1_0 + 8 * st_slot + 8 <= p.N1, a_ok = n2_0 + 8 * st_slot + 8 <= p.N2;
    const __bf16* Gb = reinterpret_cast<const __bf16*>(p.G) + n1_0 + 8 * st_slot;
    const __bf16* Ab = reinterpret_cast<const __bf16*>(p.A) + n2_0 + 8 * st_slot;
    SeqCursor cur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i].init(c_lo * BK + st_row + 16 * i, p.rows_per_seq);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                  // db: thread = column tid & 127, rows half tid >> 7

    u32x4 vg[4], va[4];
    auto fetch = [&](int64_t chunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = chunk * BK + st_row + 16 * i;
            vg[i] = u32x4{0u, 0u, 0u, 0u};
            va[i] = u32x4{0u, 0u, 0u, 0u};
            if (m < p.M) {
                if (g_ok) vg[i] = *reinterpret_cast<const u32x4*>(Gb + cur[i].seq * p.g_seq_stride + cur[i].pos * p.ldg);
                if (a_ok) va[i] = *reinterpret_cast<const u32x4*>(Ab + cur[i].seq * p.a_seq_stride + cur[i].pos * p.lda);
            }
            cur[i].advance(BK, p.rows_per_seq);
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = st_row + 16 * i;
            *reinterpret_cast<u32x4*>(sG + tn_off(row, st_slot)) = vg[i];
            *reinterpret_cast<u32x4*>(sA + tn_off(row, st_slot)) = va[i];
        }
    };
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto tr_frag = [&](const char* tile, int col0, int kk) -> bf16x8 {
        // rows 32 kk + 8 g .. + 7 of column col0 + lam: two 4 x 16 blocks
        const int r0 = 32 * kk + 8 * g;
        const int slot = (col0 >> 3) + (pq >> 1);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(tile + tn_off(r0 + q, slot) + 8 * (pq & 1)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(tile + tn_off(r0 + 4 + q, slot) + 8 * (pq & 1)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    fetch(c_lo);
#pragma unroll 1
    for (int64_t chunk = c_lo; chunk < c_hi; ++chunk) {
        __syncthreads();                               // everyone is done reading the previous tiles
        put();
        __syncthreads();
        if (chunk + 1 < c_hi) fetch(chunk + 1);        // in flight under the MFMAs below
        if (p.db && t2 == 0) {
            const int col = tid & 127, half = tid >> 7;
            float sacc = 0.f;
#pragma unroll 8
            for (int r = 32 * half; r < 32 * half + 32; ++r)
                sacc += (float)*reinterpret_cast<const __bf16*>(sG + tn_off(r, col >> 3) + 2 * (col & 7));
            bsum += sacc;
        }
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = tr_frag(sG, 64 * wr + 16 * i, kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(sA, 64 * wc + 16 * j, kk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    // lane (g, lam): dW[n1_0 + 64 wr + 16 i + 4 g + r][n2_0 + 64 wc + 16 j + lam]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n2 = n2_0 + 64 * wc + 16 * j + lam;
            if (n2 >= p.N2) continue;
            int col = n2;
            if (p.conv_cin > 0) { const int tap = n2 / p.conv_cin, chn = n2 - tap * p.conv_cin; col = chn * p.conv_kw + tap; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + 64 * wr + 16 * i + 4 * g + r;
                if (n1 < p.N1) atomicAdd(p.dW + (int64_t)n1 * p.ldw + col, acc[i][j][r]);
            }
        }
    if (p.db && t2 == 0) {
        const int n1 = n1_0 + (tid & 127);
        if (n1 < p.N1) atomicAdd(p.db + n1, bsum);
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
    const float* Gb = reinterpret_cast<const float*>(p.G) + n1_0 + 4 * st_slot;
    const float* Ab = reinterpret_cast<const float*>(p.A) + n2_0 + 4 * st_slot;
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
            if (n2 >= p.N2) continue;
            int col = n2;
            if (p.conv_cin > 0) { const int tap = n2 / p.conv_cin, chn = n2 - tap * p.conv_cin; col = chn * p.conv_kw + tap; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + 32 * wr + 16 * i + 4 * g + r;
                if (n1 < p.N1) atomicAdd(p.dW + (int64_t)n1 * p.ldw + col, acc[i][j][r]);
            }
        }
    if (p.db && t2 == 0 && tid < BN) {
        const int n1 = n1_0 + tid;
        if (n1 < p.N1) atomicAdd(p.db + n1, bsum);
    }
}

int dense_tn(bool bf16, const DenseTnArgs& a0, hipStream_t s) {
    if (a0.M <= 0) return PF_OK;
    if (a0.N1 <= 0 || a0.N2 <= 0 || a0.N1 % (bf16 ? 8 : 4) || a0.N2 % (bf16 ? 8 : 4)) return PF_ERR_BAD_ARG;
    DenseTnArgs a = a0;
    const int bn = bf16 ? 128 : 64, bk = bf16 ? 64 : 32;
    const int tiles = ((a.N1 + bn - 1) / bn) * ((a.N2 + bn - 1) / bn);
    const int64_t chunks = (a.M + bk - 1) / bk;
    if (a.splits <= 0) {
        // ~3 workgroups per CU over all tiles, each at least 4 row chunks deep
        int64_t sp = (768 + tiles - 1) / tiles;
        sp = std::min<int64_t>(sp, std::max<int64_t>(1, chunks / 4));
        a.splits = (int)std::max<int64_t>(1, sp);
    }
    const dim3 grid((unsigned)tiles, (unsigned)a.splits);
    if (bf16) hipLaunchKernelGGL(dense_tn_bf16_kernel, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(dense_tn_f32_kernel, grid, dim3(256), 0, s, a);
    return launch_status();
}

}  // namespace pf
