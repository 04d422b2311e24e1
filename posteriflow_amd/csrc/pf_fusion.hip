// pf_fusion.hip -- the strain encoder's token mixer on gfx950: the 3 pre-norm Transformer layers
// (LayerNorm -> 6-head self-attention -> residual, LayerNorm -> 192-768-192 GELU FFN -> residual)
// and the key/value side of the 8-query attention pool, fused in ONE kernel, ONE workgroup per event
// (reference: LeanStrainEncoder.fusion / pool_attn, src/ahsd/models/lean_npe.py:167-176, 226-229,
// executed through nn.TransformerEncoder / nn.MultiheadAttention in eval mode).
//
// An event is T <= 192 tokens x 192 features (183 for three detectors, +4 geometry tokens for the
// coherent encoder): 12 token tiles of 16.  Everything is computed TRANSPOSED like the flow kernel,
//   out^T[feature, token] = W[feature, k] . act^T[k, token]          (v_mfma_f32_16x16x32_bf16)
// so the weights are the A operand, read straight from their packed fragment order in L2 (1 KiB per
// fragment), and the activations are the
// B operand, read from LDS where they live as bf16 [token][feature] rows, XOR-swizzled so that every
// ds_read_b128 lane group hits 16 distinct 4-bank groups.  The fp32 residual stream lives in REGISTERS for
// the whole kernel, in the accumulator layout of the wave that owns the block (72 registers per lane): the
// out-projection and the second FFN matmul accumulate straight into it, LayerNorm reads it from there (token
// statistics cross the 4 feature-block waves through 6 KB of LDS), and global memory is touched only when the
// tokens are loaded and when the result is stored.
//   8 waves, two per SIMD (the 4-wave / one-per-SIMD layout of the first version overlapped nothing: DESIGN.md
//   4.7); dense layers: wave = (feature block of 3 tiles) x (token half of 6 tiles): 18 MFMAs per 3 weight + 6
//   activation fragment loads, a dense weight fragment is read by 2 waves, a QKV fragment by 4.
//   attention, per (head, 16-query tile): S^T = K . Q^T is 12 MFMAs (the head dimension is one k-step),
//   softmax over keys in registers (columns = queries: 4 rows per lane x 12 tiles, then two xor-shuffles),
//   O^T = V^T . P^T with P taken from the S accumulators WITHOUT leaving registers: the k-step's key
//   order is permuted to the accumulator layout (slot 8g+j <- key 32s+4g+j | 32s+16+4g+j-4) and V is kept
//   token-major like K ([token][32 dims], 8-byte stores from the accumulator layout) and read TRANSPOSED in
//   that order by ds_read_b64_tr_b16 (a 16-lane group fetches keys 4g .. 4g+3 x 16 dims and each lane gets its
//   dim's four keys; round 4: the V^T image written by 2-byte scatter stores was 25 % of the LDS-active cycles
//   in bank conflicts).  Each head's output goes through LDS once into the out-projection,
//   which accumulates over heads in registers; the FFN hidden layer goes through LDS in 4 chunks of 192
//   and never exists in full.  Two heads are in flight per iteration (24 (head, query tile) units = 3 per wave; O goes over
//   the Q tile its wave consumed).  LDS: 73 728 (normalised tokens) + 73 728 (2 x (Q/O|K|V) or hidden chunk) + 6 144.
// bf16 operands, fp32 accumulation, fp32 LayerNorm / softmax / GELU (erf) / residual.
#include <hip/hip_runtime.h>

#include "pf_status.h"

#include <cstdint>

#include "../../include/pf_hip.h"
#include "pf_math.h"

#ifndef PF_FUSION_WAVES
#define PF_FUSION_WAVES 8       // 8: two waves per SIMD, 256 registers each (the build); 4: one per SIMD with 512 registers and a
#endif                          // deeper weight pipeline (experiment)
#ifndef PF_FUSION_ABLATE
#define PF_FUSION_ABLATE 0      // timing experiments only: 1 no GELU, 2 no attention, 4 no dense matmuls, 8 no QKV projection, 16 no LN
#endif
namespace pf {
namespace {
constexpr int kAbl = PF_FUSION_ABLATE;
constexpr int kWaves = PF_FUSION_WAVES, kThreads = 64 * kWaves;
constexpr int kDepth = kWaves == 4 ? 4 : 2;            // weight fragments of kDepth - 1 k-steps in flight ahead of the MFMAs
static_assert(kWaves == 8 || kWaves == 4, "4 feature blocks x 1 or 2 token halves");
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int kE = 192, kHeads = 6, kHd = 32, kFF = 768, kLayers = 3, kPoolQ = 8;
constexpr int kTok = 192, kTT = 12;                   // padded tokens, token tiles
constexpr int kXS = 384, kQS = 64;                      // LDS byte strides: [tok][192], [tok][32] (XOR-swizzled)
constexpr int kBuf = kTok * kXS;                       // 73 728
constexpr int kHeadSet = 3 * kTok * kQS;               // one head's Q (later O) | K | V, each [tok][32]: 36 864
constexpr int kBufB = 2 * kHeadSet > kBuf ? 2 * kHeadSet : kBuf;       // two heads in flight / an FFN hidden chunk: 73 728
constexpr int kLds = kBuf + kBufB + 4 * kTok * 8;     // + per-wave token statistics
constexpr int kFrag = 1024;

// ---- packed parameter block --------------------------------------------------------------------
// per layer: Wqkv 36 tiles x 6 k-steps | Wo 12 x 6 | W1 48 x 6 | W2 12 x 24 fragments, then fp32
// ln1_g ln1_b bqkv bo ln2_g ln2_b b1 b2; after the layers: pool Wkv 24 x 6 fragments, bkv fp32.
constexpr int64_t kWqkv = 0, kWo = kWqkv + 36 * 6 * kFrag, kW1 = kWo + 12 * 6 * kFrag, kW2 = kW1 + 48 * 6 * kFrag;
constexpr int64_t kVec = kW2 + 12 * 24 * kFrag;
constexpr int kVecFloats = 192 + 192 + 576 + 192 + 192 + 192 + 768 + 192;      // 2496
constexpr int kLn1g = 0, kLn1b = 192, kBqkv = 384, kBo = 960, kLn2g = 1152, kLn2b = 1344, kB1 = 1536, kB2 = 2304;
constexpr int64_t kLayerBytes = kVec + kVecFloats * 4;
constexpr int64_t kPoolW = kLayers * kLayerBytes, kPoolB = kPoolW + 24 * 6 * kFrag;
constexpr int64_t kPackedBytes = kPoolB + 384 * 4;
// raw fp32 parameter order (state_dict order of the modules involved), per layer:
//   norm1.weight norm1.bias self_attn.in_proj_weight[576,192] self_attn.in_proj_bias
//   self_attn.out_proj.weight[192,192] .bias norm2.weight norm2.bias linear1.weight[768,192] .bias
//   linear2.weight[192,768] .bias;  then pool_attn.in_proj_weight[192:576] and in_proj_bias[192:576]
constexpr int64_t kRawLayer = 192 + 192 + 576 * 192 + 576 + 192 * 192 + 192 + 192 + 192 + 768 * 192 + 768 + 192 * 768 + 192;
constexpr int64_t kRawCount = kLayers * kRawLayer + 384 * 192 + 384;

struct FusionParams {
    const char* packed;
    float* x;                // [n_events][T][192] fp32, updated in place
    const float* tok_bias;   // [T][192] added to every event's tokens on load (positional + detector), or null
    const float* pool_q;     // [8][192] projected queries, already divided by sqrt(32)
    float* pooled;           // [n_events][8][192] attention-pool output before its out-projection
    int T;
};

template <int N> struct ic { static constexpr int value = N; };

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 to_bf16(f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    return o;
}
// LDS images read by ds_read_b128 are XOR-swizzled: that instruction's 16-lane groups are
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. 8 lanes of fragment group g and
// 8 of g+1, and no linear row stride keeps their sixteen 16-byte slots on distinct 4-bank groups.
// [tok][192] rows (24 slots): slot ^= (row >> 1) & 7;  [tok][32] rows (4 slots): slot ^= (4 - (row >> 2)) & 3.
__device__ __forceinline__ int xoff(int row, int byte) {
    return row * kXS + ((((byte >> 4) ^ ((row >> 1) & 7)) << 4) | (byte & 15));
}
__device__ __forceinline__ int qoff(int row, int byte) {
    return row * kQS + ((((byte >> 4) ^ ((4 - ((row >> 2) & 3)) & 3)) << 4) | (byte & 15));
}

__global__ __launch_bounds__(kThreads) void fusion_kernel(FusionParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xn = smem;                       // normalised tokens, bf16 [192][384 B]
    char* const hb = smem + kBuf;                // FFN hidden chunk, bf16 [192][384 B]   (aliases the head sets below)
    // two heads are in flight at a time (24 (head, query tile) units = 3 per wave; with one head the 12 tiles left
    // half the waves idle for a third of the attention phase): head slot hs at hb + hs * kHeadSet holds
    // Q [192][64 B] -- overwritten tile by tile with O by the wave that consumed the tile -- | K [192][64 B] | V [192][64 B]
    char* const qb = smem + kBuf;
    char* const kb = qb + kTok * kQS;
    char* const vt = kb + kTok * kQS;
    float2* const stats = reinterpret_cast<float2*>(smem + kBuf + kBufB);   // [4 waves][192 tokens] (sum, sum of squares)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
    const int T = p.T;
    float* const Xg = p.x + static_cast<int64_t>(blockIdx.x) * T * kE;
    const int fblk = w & 3, th = w >> 2;         // dense layers: 4 feature blocks (3 tiles) x 2 token halves (6 tiles)
    const int fb2 = w & 1, tb4 = w >> 1;         // per-head projections: 2 feature blocks x 4 token blocks (3 tiles)
    constexpr float kQScale = 0.17677669529663687f;      // 1 / sqrt(head dim), applied where torch applies it
    constexpr float kLog2e = 1.4426950408889634f;
    const int gsw = g ^ ((c >> 1) & 7);
    const int xlane_even = c * kXS + (gsw << 4), xlane_odd = c * kXS + ((gsw ^ 4) << 4);
    const int qlane = c * kQS + ((g ^ ((4 - (c >> 2)) & 3)) << 4);
    // transposed reads of V (ds_read_b64_tr_b16): the 16 lanes of group g fetch the block keys 4 g .. 4 g + 3 (+ 32 s, + 16
    // for the second half of the k-step) x dims 16 dt .. 16 dt + 15 -- lane 4 q + p supplies the address of key row q, dims
    // 4 p .. 4 p + 3 -- and lane c receives dim 16 dt + c of the four keys: the A fragment O^T = V^T . P^T wants.  The rows'
    // swizzle term (4 - ((row >> 2) & 3)) & 3 is (4 - g) & 3 for every row of the lane's blocks (32 s, 16 are multiples of 16)
    int vtr[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
        vtr[dt] = (4 * g + (c >> 2)) * kQS + (((2 * dt + ((c & 3) >> 1)) ^ ((4 - g) & 3)) << 4) + 8 * (c & 1);
    // per-lane parts of the swizzled STORE offsets (accumulator layout: 4 features 16 t + 4 g .. of token c)
    int wx[3], qw[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) wx[i] = xoff(c, (16 * (3 * fblk + i) + 4 * g) * 2);     // + 16 j rows = j * 16 * kXS
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) qw[dt] = qoff(c, (16 * dt + 4 * g) * 2);             // + 16 tt rows = tt * 16 * kQS

    // ---- the fp32 residual stream lives in registers for the whole kernel: X[i][j] = features
    // 16 (3 w + i) + 4 g .. +3 of token 16 j + c (the MFMA accumulator layout of this wave's block), so
    // that the out-projection and the second FFN matmul accumulate straight into it
    constexpr int kHT = kTT / (kWaves / 4);      // token tiles per wave
    constexpr int kQT = kTT / (kWaves / 2);      // token tiles per wave in the per-head projections
    f32x4 X[3][kHT];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f0 = 16 * (3 * fblk + i) + 4 * g;
#pragma unroll
        for (int j = 0; j < kHT; ++j) {
            // (unconditional loads from a clamped row: a load under a per-lane condition is a branch of its own, and 36
            // of them wait for one another -- 18 HBM round trips per event before the first MFMA)
            const int tok = 16 * (kHT * th + j) + c, tokc = tok < T ? tok : T - 1;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Xg + tokc * kE + f0);
            X[i][j] = tok < T ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (p.tok_bias) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f0 = 16 * (3 * fblk + i) + 4 * g;
#pragma unroll
            for (int j = 0; j < kHT; ++j) {
                const int tok = 16 * (kHT * th + j) + c, tokc = tok < T ? tok : T - 1;
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.tok_bias + tokc * kE + f0);
                X[i][j] = X[i][j] + (tok < T ? b : f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    }

    // residual (registers) -> LDS bf16 [token][feature], LayerNorm-ed when gamma is given; rows >= T zero.
    // Token statistics need all 192 features = all 4 waves: per-wave partial (sum, sum of squares) through LDS.
    auto stage_tokens = [&](const float* gamma, const float* beta) {
        float mean[kHT], rstd[kHT];
        // (the scale / shift vectors are requested before the statistics exchange: their L2 round trip runs under it)
        f32x4 ga[3], be[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int f0 = 16 * (3 * fblk + i) + 4 * g;
            ga[i] = f32x4{1.f, 1.f, 1.f, 1.f}; be[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gamma) { ga[i] = *reinterpret_cast<const f32x4*>(gamma + f0); be[i] = *reinterpret_cast<const f32x4*>(beta + f0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gamma && !(kAbl & 16)) {
#pragma unroll
            for (int j = 0; j < kHT; ++j) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s += X[i][j][e]; q += X[i][j][e] * X[i][j][e]; }
                s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
                s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                if (g == 0) stats[fblk * kTok + 16 * (kHT * th + j) + c] = make_float2(s, q);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kHT; ++j) {
                const int tok = 16 * (kHT * th + j) + c;
                const float2 a0 = stats[tok], a1 = stats[kTok + tok], a2 = stats[2 * kTok + tok], a3 = stats[3 * kTok + tok];
                const float m = ((a0.x + a1.x) + (a2.x + a3.x)) * (1.f / kE);
                const float var = ((a0.y + a1.y) + (a2.y + a3.y)) * (1.f / kE) - m * m;
                mean[j] = m;
                rstd[j] = rsqrtf(fmaxf(var, 0.f) + 1e-5f);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < kHT; ++j) {
                const int tok = 16 * (kHT * th + j) + c;
                f32x4 v = X[i][j];
                if (gamma && !(kAbl & 16)) v = (v - mean[j]) * rstd[j] * ga[i] + be[i];
                if (tok >= T) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<bf16x4*>(xn + (kHT * th + j) * (16 * kXS) + wx[i]) = to_bf16(v);
            }
        }
        __syncthreads();
    };
    auto add_bias = [&](const float* bias) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 16 * (3 * fblk + i) + 4 * g);
#pragma unroll
            for (int j = 0; j < kHT; ++j) X[i][j] = X[i][j] + b4;
        }
    };
    auto afrag = [&](const char* wbase, int tile, int ksteps, int ks) {
        return *reinterpret_cast<const bf16x8*>(wbase + (static_cast<int64_t>(tile * ksteps + ks) * 64 + lane) * 16);
    };
    auto bx = [&](const char* buf, int tt, int ks) {           // [tok][192] buffers
        // = xoff(16 tt + c, 64 ks + 16 g) with the per-lane part hoisted: (4 ks + g) ^ sw = 4 ks ^ (g ^ sw)
        return *reinterpret_cast<const bf16x8*>(buf + tt * (16 * kXS) + (ks >> 1) * 128 + ((ks & 1) ? xlane_odd : xlane_even));
    };
    auto bq = [&](const char* buf, int tt) {                   // [tok][32] buffers (also the A operand K)
        return *reinterpret_cast<const bf16x8*>(buf + tt * (16 * kQS) + qlane);       // = qoff(16 tt + c, 16 g)
    };
    // dense block over NT token tiles starting at tile t0: acc[3][NT] += W[tiles ft0..ft0+2][k-steps] . B
    auto dense = [&](auto& acc, int t0, const char* wbase, int ft0, int ksteps, int ks0, auto nks, auto&& bload) {
        constexpr int NT = sizeof(acc[0]) / sizeof(f32x4);
        constexpr int KS = decltype(nks)::value;
        if constexpr (kAbl & 4) return;
        // weight fragments (L2 latency) double-buffered in registers: those of k-step ks+1 are issued before
        // the MFMAs of k-step ks; activation fragments (LDS latency) are loaded per k-step
        bf16x8 a[kDepth][3];
#pragma unroll
        for (int d = 0; d < kDepth - 1; ++d)
            if (d < KS) {
#pragma unroll
                for (int i = 0; i < 3; ++i) a[d][i] = afrag(wbase, ft0 + i, ksteps, ks0 + d);
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + kDepth - 1 < KS) {
#pragma unroll
                for (int i = 0; i < 3; ++i) a[(ks + kDepth - 1) % kDepth][i] = afrag(wbase, ft0 + i, ksteps, ks0 + ks + kDepth - 1);
            }
            // keep the next k-step's weight loads ahead of this k-step's MFMAs: under register pressure the scheduler
            // sinks them to ~6 MFMAs before their use and every k-step then waits out an L2 round trip
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 b[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = bload(t0 + j, ks);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma(a[ks % kDepth][i], b[j], acc[i][j]);
        }
    };
    // one (head, 16-query tile): softmax(K Q^T) V with keys >= T masked; result O^T tiles (2 x f32x4).
    // exp(s - m) = exp2(s log2e - m log2e): one packed fma + a bare v_exp_f32 per score.
    // mask_from: first key tile that can hold padded keys (compile time: 11 when T > 176, else 0) -- selects instead of a
    // uniform branch per tile, so that one call is one basic block and the scheduler can run the MFMAs of one (head, query
    // tile) unit under the softmax of another
    auto attend = [&](auto mask_from, const char* qb, const char* kb, const char* vt, int qt, f32x4 (&o)[2]) {
        const bf16x8 qf = bq(qb, qt);
        f32x4 s[kTT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < kTT; ++kt) {
            s[kt] = mfma(bq(kb, kt), qf, f32x4{0.f, 0.f, 0.f, 0.f});
            if (kt >= decltype(mask_from)::value) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[kt][e] = 16 * kt + 4 * g + e >= T ? -INFINITY : s[kt][e];
            }
            m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < kTT; ++kt) {
            s[kt] = s[kt] * kLog2e - m * kLog2e;
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kt][e] = __builtin_amdgcn_exp2f(s[kt][e]);
            sum4 = sum4 + s[kt];
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        o[0] = o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < kTT / 2; ++ss) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pf[e] = (__bf16)s[2 * ss][e]; pf[4 + e] = (__bf16)s[2 * ss + 1][e]; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const char* blk = vt + 32 * ss * kQS + vtr[dt];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + 16 * kQS));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[dt] = mfma(__builtin_bit_cast(bf16x8, v8), pf, o[dt]);
            }
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
        o[0] = o[0] * inv;
        o[1] = o[1] * inv;
    };
    // projection epilogues: a 16 x 16 tile of Q / K (token-major rows) or V (transposed)
    auto put_qk = [&](char* buf, int dt, int tt, f32x4 v) {
        *reinterpret_cast<bf16x4*>(buf + tt * (16 * kQS) + qw[dt]) = to_bf16(v);
    };

    for (int l = 0; l < kLayers; ++l) {
        const char* lw = p.packed + l * kLayerBytes;
        const float* vec = reinterpret_cast<const float*>(lw + kVec);
        // ================= self-attention block =================
        stage_tokens(vec + kLn1g, vec + kLn1b);
        add_bias(vec + kBo);
        for (int hp = 0; hp < kHeads / 2; ++hp) {
            {   // Q | K | V of heads 2 hp, 2 hp + 1: this wave's 3 of a head's 6 tiles x its 3 token tiles, the two heads' 12
                // k-steps as ONE pipelined sequence (the weight fragments of a step are requested one step ahead, across the
                // head boundary too)
                f32x4 t[3][kQT];
                auto tile_of = [&](int hs, int i) {
                    // tile index in the 36-tile in_proj: Q 2h,2h+1 | K 12+2h,12+2h+1 | V 24+2h,24+2h+1
                    const int h = 2 * hp + hs;
                    return i == 0 ? (fb2 == 0 ? 2 * h : 12 + 2 * h + 1) : i == 1 ? (fb2 == 0 ? 2 * h + 1 : 24 + 2 * h)
                                                                                  : (fb2 == 0 ? 12 + 2 * h : 24 + 2 * h + 1);
                };
                bf16x8 a[2][3];
                if constexpr (!(kAbl & 8)) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) a[0][i] = afrag(lw + kWqkv, tile_of(0, i), 6, 0);
                }
#pragma unroll
                for (int step = 0; step < 12; ++step) {
                    const int hs = step / 6, ks = step % 6;
                    if (ks == 0) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < kQT; ++j) t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if constexpr (!(kAbl & 8)) {
                        if (step + 1 < 12) {
#pragma unroll
                            for (int i = 0; i < 3; ++i)
                                a[(step + 1) & 1][i] = afrag(lw + kWqkv, tile_of((step + 1) / 6, i), 6, (step + 1) % 6);
                        }
                        __builtin_amdgcn_sched_barrier(0);          // (see dense)
                        bf16x8 b[kQT];
#pragma unroll
                        for (int j = 0; j < kQT; ++j) b[j] = bx(xn, kQT * tb4 + j, ks);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < kQT; ++j) t[i][j] = mfma(a[step & 1][i], b[j], t[i][j]);
                    }
                    if (ks == 5) {
                        char* const q_ = qb + hs * kHeadSet;
                        char* const k_ = kb + hs * kHeadSet;
                        char* const v_ = vt + hs * kHeadSet;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec + kBqkv + 16 * tile_of(hs, i) + 4 * g);
#pragma unroll
                            for (int j = 0; j < kQT; ++j) {
                                const int tt = kQT * tb4 + j;
                                const f32x4 v = t[i][j] + b4;
                                if (fb2 == 0) {
                                    if (i == 0) put_qk(q_, 0, tt, v * kQScale);
                                    else if (i == 1) put_qk(q_, 1, tt, v * kQScale);
                                    else put_qk(k_, 0, tt, v);
                                } else {
                                    if (i == 0) put_qk(k_, 1, tt, v);
                                    else put_qk(v_, i - 1, tt, v);
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            auto units = [&](auto mask_from) {              // 24 (head slot, query tile) units: 3 per wave
#pragma unroll 1
                for (int it = 0; it < 2 * kTT / kWaves; ++it) {
                    const int u = w + kWaves * it;
                    const int hs = u >= kTT ? 1 : 0, qt = u - hs * kTT;
                    char* const q_ = qb + hs * kHeadSet;
                    f32x4 o[2];
                    if constexpr (kAbl & 2) { o[0] = o[1] = f32x4{0.f, 0.f, 0.f, 0.f}; } else
                    attend(mask_from, q_, kb + hs * kHeadSet, vt + hs * kHeadSet, qt, o);
                    put_qk(q_, 0, qt, o[0]);                 // O over the Q tile this wave alone has read
                    put_qk(q_, 1, qt, o[1]);
                }
            };
            if (T > 16 * (kTT - 1)) units(ic<kTT - 1>{}); else units(ic<0>{});
            __syncthreads();
            // out-projection straight into the residual: k-steps 2 hp, 2 hp + 1 of Wo against the two heads' O
            dense(X, kHT * th, lw + kWo, 3 * fblk, 6, 2 * hp, ic<2>{},
                  [&](int tt, int ks) { return bq(qb + ks * kHeadSet, tt); });
            __syncthreads();                                   // O lives where the next pair's Q goes
        }
        // ================= feed-forward block =================
        stage_tokens(vec + kLn2g, vec + kLn2b);
        add_bias(vec + kB2);
        for (int ch = 0; ch < kFF / kE; ++ch) {
            {
                f32x4 t[3][kHT];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < kHT; ++j) t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                dense(t, kHT * th, lw + kW1, 12 * ch + 3 * fblk, 6, 0, ic<6>{}, [&](int tt, int ks) { return bx(xn, tt, ks); });
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int f0 = 16 * (3 * fblk + i) + 4 * g;              // feature within the chunk
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec + kB1 + kE * ch + f0);
#pragma unroll
                    for (int j = 0; j < kHT; ++j)
                        *reinterpret_cast<bf16x4*>(hb + (kHT * th + j) * (16 * kXS) + wx[i]) = to_bf16((kAbl & 1) ? t[i][j] + b4 : gelu_erf_fast4(t[i][j] + b4));
                }
            }
            __syncthreads();
            dense(X, kHT * th, lw + kW2, 3 * fblk, 24, 6 * ch, ic<6>{}, [&](int tt, int ks) { return bx(hb, tt, ks); });
            __syncthreads();
        }
    }
    // the Transformer output (in-place contract of the entry point)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f0 = 16 * (3 * fblk + i) + 4 * g;
#pragma unroll
        for (int j = 0; j < kHT; ++j) {
            const int tok = 16 * (kHT * th + j) + c;
            if (tok < T) *reinterpret_cast<f32x4*>(Xg + tok * kE + f0) = X[i][j];
        }
    }

    // ================= attention pool: keys / values of the final tokens, 8 projected queries ========
    stage_tokens(nullptr, nullptr);
    const float* bkv = reinterpret_cast<const float*>(p.packed + kPoolB);
    for (int h = 0; h < kHeads; ++h) {
        {   // K | V of head h: 2 tiles per wave x 6 token tiles (tile space: K 0..11 | V 12..23)
            f32x4 t[2][kQT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < kQT; ++j) t[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int tile0 = (fb2 == 0 ? 0 : 12) + 2 * h;
            // all 12 weight fragments of the head in one batch (the residual registers are free here): one L2 round
            // trip per head instead of one per k-step
            bf16x8 a[6][2];
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) a[ks][i] = afrag(p.packed + kPoolW, tile0 + i, 6, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                bf16x8 b[kQT];
#pragma unroll
                for (int j = 0; j < kQT; ++j) b[j] = bx(xn, kQT * tb4 + j, ks);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < kQT; ++j) t[i][j] = mfma(a[ks][i], b[j], t[i][j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bkv + 16 * (tile0 + i) + 4 * g);
#pragma unroll
                for (int j = 0; j < kQT; ++j) {
                    if (fb2 == 0) put_qk(kb, i, kQT * tb4 + j, t[i][j] + b4);
                    else put_qk(vt, i, kQT * tb4 + j, t[i][j] + b4);
                }
            }
            // the head's 8 queries (rows 8..15 of the tile are zero)
            for (int i = tid; i < 16 * kHd; i += kThreads) {
                const int q = i >> 5, d = i & 31;
                const float v = q < kPoolQ ? p.pool_q[q * kE + kHd * h + d] : 0.f;
                *reinterpret_cast<__bf16*>(qb + qoff(q, d * 2)) = (__bf16)v;
            }
        }
        __syncthreads();
        if (w == 0) {
            f32x4 o[2];
            attend(ic<0>{}, qb, kb, vt, 0, o);
            if (c < kPoolQ) {
                float* dst = p.pooled + (static_cast<int64_t>(blockIdx.x) * kPoolQ + c) * kE + kHd * h + 4 * g;
                *reinterpret_cast<f32x4*>(dst) = o[0];
                *reinterpret_cast<f32x4*>(dst + 16) = o[1];
            }
        }
        __syncthreads();
    }
}

// raw fp32 [N][K] row-major -> bf16 A fragments: fragment (tile, ks), lane (g, r): W[16 tile + r][32 ks + 8 g ..+7]
__global__ void pack_frags_kernel(const float* wsrc, int n_rows, int k, __bf16* out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<int64_t>(n_rows) * k) return;
    const int ksteps = k / 32;
    const int e = i & 7, lane = (i >> 3) & 63;
    const int64_t f = i >> 9;
    const int tile = static_cast<int>(f / ksteps), ks = static_cast<int>(f % ksteps);
    out[i] = (__bf16)wsrc[static_cast<int64_t>(16 * tile + (lane & 15)) * k + 32 * ks + 8 * (lane >> 4) + e];
}
}  // namespace

int64_t fusion_raw_count() { return kRawCount; }
int64_t fusion_packed_bytes() { return kPackedBytes; }

int fusion_pack(const float* raw, char* packed, hipStream_t s) {
    auto frags = [&](const float* src, int n, int k, int64_t off) {
        const int64_t tot = static_cast<int64_t>(n) * k;
        pack_frags_kernel<<<dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s>>>(
            src, n, k, reinterpret_cast<__bf16*>(packed + off));
    };
    auto vec = [&](const float* src, int n, int64_t off) {
        (void)hipMemcpyAsync(packed + off, src, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, s);
    };
    const float* r = raw;
    for (int l = 0; l < kLayers; ++l) {
        const int64_t base = l * kLayerBytes, vb = base + kVec;
        vec(r, 192, vb + 4 * kLn1g); r += 192;
        vec(r, 192, vb + 4 * kLn1b); r += 192;
        frags(r, 576, 192, base + kWqkv); r += 576 * 192;
        vec(r, 576, vb + 4 * kBqkv); r += 576;
        frags(r, 192, 192, base + kWo); r += 192 * 192;
        vec(r, 192, vb + 4 * kBo); r += 192;
        vec(r, 192, vb + 4 * kLn2g); r += 192;
        vec(r, 192, vb + 4 * kLn2b); r += 192;
        frags(r, 768, 192, base + kW1); r += 768 * 192;
        vec(r, 768, vb + 4 * kB1); r += 768;
        frags(r, 192, 768, base + kW2); r += 192 * 768;
        vec(r, 192, vb + 4 * kB2); r += 192;
    }
    frags(r, 384, 192, kPoolW); r += 384 * 192;
    vec(r, 384, kPoolB);
    return launch_status();
}

int fusion_forward(const char* packed, float* tokens, int n_tokens, const float* tok_bias, const float* pool_q,
                   int64_t n_events, float* pooled, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        if (!opt_in_lds(reinterpret_cast<const void*>(fusion_kernel), kLds))
            return PF_ERR_HIP;
        configured = true;
    }
    FusionParams p{packed, tokens, tok_bias, pool_q, pooled, n_tokens};
    fusion_kernel<<<dim3(static_cast<unsigned>(n_events)), dim3(kThreads), kLds, s>>>(p);
    return launch_status();
}
}  // namespace pf
