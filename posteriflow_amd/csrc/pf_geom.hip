// pf_geom.hip -- frequency-domain geometry features of the coherent encoder (reference: CoherentEncoder._geometry_rel,
// src/ahsd/models/coherent_encoder.py:79-116): per event, from the sanitised strain of its detectors,
//   * rfft (ortho) of every detector's 16 384 samples, bins [band_lo, band_lo + nf) kept (20-1024 Hz: 80 .. 4095);
//   * K log band energies per detector:              log(mean_band |X|^2 + 1e-8);
//   * per detector pair (i, j), per band: the power-weighted coherence g = sum X_i conj(X_j) / (sum |X_i| |X_j| + 1e-8)
//     as (|g| + 1e-8, Re g / |g|, Im g / |g|);
//   * per pair: the generalised cross-correlation cc = irfft(X_i conj(X_j) on the band, 0 elsewhere) on the lags
//     -maxlag .. maxlag (+-30 ms): argmax |cc| as a normalised lag, max |cc| / (mean |cc| + 1e-8);
//   * per pair: log(sum |X_i|^2 + 1e-8) - log(sum |X_j|^2 + 1e-8).
// Two kernels, both built on ONE in-LDS FFT:
//   a real transform of length T = 16 384 is a complex transform of length N = 8192 of z[n] = x[2n] + i x[2n+1] followed by
//   X[k] = (Z[k] + conj Z[N-k]) / 2 - i e^{-2 pi i k / T} (Z[k] - conj Z[N-k]) / 2; the inverse real transform of a
//   Hermitian spectrum F runs the same complex transform over Y[k] = conj((F[k] + conj F[N-k]) + i e^{2 pi i k / T} (F[k] -
//   conj F[N-k])): x[2n] = Re y[n] / T, x[2n+1] = -Im y[n] / T.  F is zero outside the band (and the band ends at N / 2), so
//   only Y[k] and Y[N-k] of the band's k are non-zero.
//   The complex transform: decimation in frequency, natural order in, bit-reversed order out (the consumers read single
//   bins through __brev, nothing is reordered): four radix-4 passes over the 64-KB LDS array (sizes 8192 .. 64, twiddles
//   from a 64-KB table of e^{-2 pi i m / T} made in double precision on the host) and the last five stages on 32
//   consecutive values per thread in registers with literal twiddles.  LDS index i lives at i + (i >> 5): a thread's 32
//   consecutive values are then 33 slots from its neighbour's (conflict-free), the strided passes stay unit-stride.
//   geom_spectrum_kernel: one workgroup per (event, detector): FFT, recombination, band spectrum -> workspace (L2-resident
//   for the pair kernel), band energies, total band power.  66 KB of LDS: two workgroups per CU.
//   geom_pair_kernel: one workgroup per (event, pair): cross spectrum -> Y in LDS -> FFT -> 2 maxlag + 1 lags -> argmax /
//   sharpness; the band sums by 16 threads per band (fixed order: deterministic).
// HBM: 64 KB of strain in and 32 KB of band spectrum out per sequence; everything else stays in LDS.  fp32 throughout (the
// features feed an MLP in fp32: coherent_encoder.py:118-123 under the reference's autocast(enabled=False)).
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <cmath>
#include <cstdint>

#include "../../include/pf_hip.h"

namespace pf {
namespace {
constexpr int kT = 16384, kN = 8192, kThreads = 256, kSlots = kN + kN / 32;
struct c32 { float x, y; };
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c32 mul_neg_i(c32 a) { return {a.y, -a.x}; }          // a . (-i)
__device__ __forceinline__ int slot(int i) { return i + (i >> 5); }
__device__ __forceinline__ int brev13(int k) { return (int)(__brev((unsigned)k) >> 19); }

// e^{-2 pi i m / 32}, m = 0 .. 15
__device__ constexpr float kCos32[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                                         0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f,
                                         -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                                         -0.92387953251128674f, -0.98078528040323043f};
__device__ constexpr float kSin32[16] = {0.f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f,
                                         -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f, -1.f, -0.98078528040323043f,
                                         -0.92387953251128674f, -0.83146961230254524f, -0.70710678118654752f, -0.55557023301960218f,
                                         -0.38268343236508977f, -0.19509032201612825f};

// complex DFT of the kN values in s (natural order) -> s (bit-reversed order); tw[m] = e^{-2 pi i m / kT}, m < kN.
// Entered and left with the workgroup synchronised.
__device__ void fft_dif(c32* s, const c32* __restrict__ tw, int tid) {
#pragma unroll 1
    for (int ls = 13; ls >= 7; ls -= 2) {                  // stages of size 2^ls and 2^(ls-1) in one pass
        const int q = 1 << (ls - 2);
        // the pass's 16 twiddles of this thread in one batch: one L2 round trip per pass instead of one per butterfly group
        constexpr int kIt = kN / 4 / kThreads;
        c32 w1s[kIt], w2s[kIt];
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int j = (tid + it * kThreads) & (q - 1);
            w1s[it] = tw[j << (14 - ls)];                  // W_s^j;  W_s^(j+q) = -i W_s^j
            w2s[it] = tw[j << (15 - ls)];                  // W_(s/2)^j
        }
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int gi = tid + it * kThreads;
            const int j = gi & (q - 1), base = (gi >> (ls - 2)) << ls;
            const int i0 = base + j;
            const c32 a0 = s[slot(i0)], a1 = s[slot(i0 + q)], a2 = s[slot(i0 + 2 * q)], a3 = s[slot(i0 + 3 * q)];
            const c32 w1 = w1s[it], w2 = w2s[it];
            const c32 b0 = a0 + a2, b2 = cmul(a0 - a2, w1), b1 = a1 + a3, b3 = mul_neg_i(cmul(a1 - a3, w1));
            s[slot(i0)] = b0 + b1;
            s[slot(i0 + q)] = cmul(b0 - b1, w2);
            s[slot(i0 + 2 * q)] = b2 + b3;
            s[slot(i0 + 3 * q)] = cmul(b2 - b3, w2);
        }
        __syncthreads();
    }
    c32 r[32];                                             // sizes 32 .. 2 on this thread's 32 consecutive values
    c32* mine = s + 33 * tid;                              // slot(32 tid + e) = 33 tid + e
#pragma unroll
    for (int e = 0; e < 32; ++e) r[e] = mine[e];
#pragma unroll
    for (int sz = 32; sz >= 2; sz >>= 1) {
        const int h = sz >> 1;
#pragma unroll
        for (int base = 0; base < 32; base += sz)
#pragma unroll
            for (int j = 0; j < h; ++j) {
                const c32 u = r[base + j], v = r[base + j + h];
                r[base + j] = u + v;
                const int m = j * (32 / sz);
                r[base + j + h] = m == 0 ? u - v : (m == 8 ? mul_neg_i(u - v) : cmul(u - v, c32{kCos32[m], kSin32[m]}));
            }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) mine[e] = r[e];
    __syncthreads();
}

struct GeomParams {
    PfGeomArgs a;
    int npairs, rel_dim;
};

__device__ __forceinline__ float row16_sum(float v) {      // over the 16 lanes that share tid >> 4
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}

__global__ __launch_bounds__(kThreads) void geom_spectrum_kernel(const GeomParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32* const s = reinterpret_cast<c32*>(smem);
    float* const pw = reinterpret_cast<float*>(smem);               // [nf] band power: over the transform's array once it has been read
    float* const red = reinterpret_cast<float*>(smem + (size_t)kSlots * sizeof(c32));       // [4]
    const PfGeomArgs& A = p.a;
    const int tid = threadIdx.x;
    const int64_t seq = blockIdx.x;
    const c32* const tw = reinterpret_cast<const c32*>(A.twiddle);
    const float4* x4 = reinterpret_cast<const float4*>(A.clean + seq * kT);
    float4 xv[kT / 4 / kThreads];                               // all 16 loads of the thread in flight together
#pragma unroll
    for (int it = 0; it < kT / 4 / kThreads; ++it) xv[it] = x4[tid + it * kThreads];
#pragma unroll
    for (int it = 0; it < kT / 4 / kThreads; ++it) {
        const int i = tid + it * kThreads;
        if (A.sanitize) {                                       // lean_npe.py:207
            auto san = [](float x) { x = (x != x) ? 0.f : x; return fminf(fmaxf(x, -100.f), 100.f); };
            xv[it] = float4{san(xv[it].x), san(xv[it].y), san(xv[it].z), san(xv[it].w)};
        }
        s[slot(2 * i)] = c32{xv[it].x, xv[it].y};
        s[slot(2 * i + 1)] = c32{xv[it].z, xv[it].w};
    }
    __syncthreads();
    fft_dif(s, tw, tid);
    c32* const out = reinterpret_cast<c32*>(A.spec) + seq * A.nf;
    float tot = 0.f, pr[kN / 2 / kThreads];
#pragma unroll
    for (int it = 0; it < kN / 2 / kThreads; ++it) {
        const int kk = tid + it * kThreads;
        pr[it] = 0.f;
        if (kk >= A.nf) continue;
        const int k = A.band_lo + kk;
        const c32 zk = s[slot(brev13(k))];
        c32 zm = s[slot(brev13(kN - k))];
        zm.y = -zm.y;
        const c32 ev = zk + zm, od = cmul(zk - zm, tw[k]);      // X = ev / 2 - i od / 2
        const float sc = 0.5f / 128.f;                          // ortho: 1 / sqrt(kT)
        const c32 X = {(ev.x + od.y) * sc, (ev.y - od.x) * sc};
        out[kk] = X;
        pr[it] = X.x * X.x + X.y * X.y;
        tot += pr[it];
    }
    __syncthreads();                                            // every bin of the transform has been read
#pragma unroll
    for (int it = 0; it < kN / 2 / kThreads; ++it) pw[tid + it * kThreads] = pr[it];
    // total band power: lanes, then waves, in a fixed order
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = tot;
    __syncthreads();
    if (tid == 0) A.etot[seq] = (red[0] + red[1]) + (red[2] + red[3]);
    const int b = tid >> 4, r = tid & 15;
    float acc = 0.f;
    if (b < A.n_bands)
        for (int kk = A.band_edge[b] + r; kk < A.band_edge[b + 1]; kk += 16) acc += pw[kk];
    acc = row16_sum(acc);
    if (b < A.n_bands && r == 0) {
        const int cnt = A.band_edge[b + 1] - A.band_edge[b];
        const int64_t ev_i = seq / A.n_det;
        const int det = (int)(seq - ev_i * A.n_det);
        A.rel[ev_i * p.rel_dim + det * A.n_bands + b] = logf(acc / (float)(cnt > 0 ? cnt : 1) + 1e-8f);
    }
}

__global__ __launch_bounds__(kThreads) void geom_pair_kernel(const GeomParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c32* const s = reinterpret_cast<c32*>(smem);
    float* const sa = reinterpret_cast<float*>(smem + (size_t)kSlots * sizeof(c32));        // [256] |cc| on the lag window
    const PfGeomArgs& A = p.a;
    const int tid = threadIdx.x;
    const int64_t ev_i = blockIdx.x / p.npairs;
    int pr = (int)(blockIdx.x - ev_i * p.npairs), di = 0, dj = 1;
    for (int i = 0, q = 0; i < A.n_det; ++i)                    // pair index -> (i < j), lexicographic (coherent_encoder.py:74)
        for (int j = i + 1; j < A.n_det; ++j, ++q)
            if (q == pr) { di = i; dj = j; }
    const c32* const tw = reinterpret_cast<const c32*>(A.twiddle);
    const c32* const Si = reinterpret_cast<const c32*>(A.spec) + (ev_i * A.n_det + di) * A.nf;
    const c32* const Sj = reinterpret_cast<const c32*>(A.spec) + (ev_i * A.n_det + dj) * A.nf;
    // cross spectrum of this thread's bins (kk = tid + 256 it) in registers; a copy + |X_i| |X_j| through LDS for the band sums
    // (the bands are 22 .. 893 bins long: summing them from global memory by 16 threads per band was 56 dependent L2
    // round trips for the last band -- 40 % of this kernel)
    constexpr int kIt = kN / 2 / kThreads;
    c32 F[kIt], wv[kIt];
    float* const fr = reinterpret_cast<float*>(smem);             // [3][4096]: Re F | Im F | |X_i| |X_j| (before the transform's array is built)
    {
        c32 av[kIt], bv[kIt];
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int kk = tid + it * kThreads, kc = kk < A.nf ? kk : A.nf - 1;
            av[it] = Si[kc]; bv[it] = Sj[kc]; wv[it] = tw[A.band_lo + kc];
        }
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int kk = tid + it * kThreads;
            const c32 a = av[it], b = bv[it];
            F[it] = c32{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};     // X_i conj(X_j)
            fr[kk] = F[it].x;
            fr[4096 + kk] = F[it].y;
            fr[8192 + kk] = sqrtf(a.x * a.x + a.y * a.y + 1e-12f) * sqrtf(b.x * b.x + b.y * b.y + 1e-12f);
        }
    }
    __syncthreads();
    {
        const int b = tid >> 4, r = tid & 15;
        float sr = 0.f, si = 0.f, sd = 0.f;
        if (b < A.n_bands)
            for (int kk = A.band_edge[b] + r; kk < A.band_edge[b + 1]; kk += 16) { sr += fr[kk]; si += fr[4096 + kk]; sd += fr[8192 + kk]; }
        sr = row16_sum(sr); si = row16_sum(si); sd = row16_sum(sd);
        if (b < A.n_bands && r == 0) {
            const float den = sd + 1e-8f, gr = sr / den, gi = si / den, gm = sqrtf(gr * gr + gi * gi) + 1e-8f;
            float* o = A.rel + ev_i * p.rel_dim + A.n_det * A.n_bands + pr * (3 * A.n_bands + 3);
            o[b] = gm;
            o[A.n_bands + b] = gr / gm;
            o[2 * A.n_bands + b] = gi / gm;
        }
    }
    __syncthreads();
    for (int i = tid; i < kSlots; i += kThreads) s[i] = c32{0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int kk = tid + it * kThreads;
        if (kk >= A.nf) continue;
        const int k = A.band_lo + kk;
        const c32 w = wv[it];                                               // e^{-2 pi i k / T}
        s[slot(k)] = cmul(c32{F[it].x, -F[it].y}, c32{1.f + w.y, -w.x});   // conj(F) (1 - i w)
        s[slot(kN - k)] = cmul(F[it], c32{1.f - w.y, -w.x});               // F (1 - i conj w)
    }
    __syncthreads();
    fft_dif(s, tw, tid);
    const int nl = 2 * A.maxlag + 1;
    float av = -1.f;
    if (tid < nl) {
        const int idx = (tid - A.maxlag) & (kT - 1);
        const c32 y = s[slot(brev13(idx >> 1))];
        av = fabsf(((idx & 1) ? -y.y : y.x) * (1.f / (float)kT));
    }
    sa[tid] = av;
    __syncthreads();
    if (tid < 64) {           // first maximum and the sum over the window, one wave
        float best = -1.f, sum = 0.f;
        int bi = 0;
        for (int t = tid; t < nl; t += 64) {
            const float v = sa[t];
            sum += v;
            if (v > best) { best = v; bi = t; }
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            sum += __shfl_xor(sum, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) {
            float* o = A.rel + ev_i * p.rel_dim + A.n_det * A.n_bands + pr * (3 * A.n_bands + 3) + 3 * A.n_bands;
            o[0] = (float)(bi - A.maxlag) / (float)A.maxlag;
            o[1] = best / (sum / (float)nl + 1e-8f);
            o[2] = logf(A.etot[ev_i * A.n_det + di] + 1e-8f) - logf(A.etot[ev_i * A.n_det + dj] + 1e-8f);
        }
    }
}
}  // namespace

void geom_twiddles(float* table) {
    for (int m = 0; m < kN; ++m) {
        const double ang = -2.0 * M_PI * (double)m / (double)kT;
        table[2 * m] = (float)std::cos(ang);
        table[2 * m + 1] = (float)std::sin(ang);
    }
}

int geom_features(const PfGeomArgs& a, hipStream_t st) {
    GeomParams p{};
    p.a = a;
    p.npairs = a.n_det * (a.n_det - 1) / 2;
    p.rel_dim = a.n_det * a.n_bands + p.npairs * (3 * a.n_bands + 3);
    const int lds_a = kSlots * (int)sizeof(c32) + 4 * 4, lds_b = kSlots * (int)sizeof(c32) + kThreads * 4;
    if (!opt_in_lds(reinterpret_cast<const void*>(geom_spectrum_kernel), lds_a) ||
        !opt_in_lds(reinterpret_cast<const void*>(geom_pair_kernel), lds_b))
        return PF_ERR_HIP;
    hipLaunchKernelGGL(geom_spectrum_kernel, dim3((unsigned)(a.batch * a.n_det)), dim3(kThreads), lds_a, st, p);
    int rc = launch_status();
    if (rc != PF_OK || p.npairs == 0) return rc;
    hipLaunchKernelGGL(geom_pair_kernel, dim3((unsigned)(a.batch * p.npairs)), dim3(kThreads), lds_b, st, p);
    return launch_status();
}
}  // namespace pf
