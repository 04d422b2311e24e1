// pf_dense.h -- the dense building blocks of the strain embedding's TRAINING path on gfx950 (csrc/pf_dense.hip):
//
//   dense_nt   out[m, n] = epilogue( sum_k A[m, k] W[n, k] + bias[n] )         (every Linear / transposed convolution
//              of the token mixer's forward and of the data-gradient chain; lean_npe.py:157-176, 226-243)
//   dense_tn   dW[n1, n2] += sum_m G[m, n1] A[m, n2],  db[n1] += sum_m G[m, n1]   (every weight / bias gradient)
//
// Both take "sequence-strided" row addressing so that a convolution's im2col matrix is never materialised: row m of an
// operand = sequence m / rows_per_seq, position m % rows_per_seq, at base + seq * seq_stride + pos * ld -- with
// ld < K the rows are OVERLAPPING windows of a position-major activation array (ld = stride * Cin, K = kernel * Cin).
//
// As everywhere in this library the products are formed transposed, out^T[n, m] = W . A^T: the weights are the MFMA A
// operand, pre-packed as 1-KiB fragments ([16-unit tile][k-step][64 lanes]: dense_pack) and streamed from L2 straight
// into registers; a strip of 128 activation rows is staged once in LDS (16-byte slots XOR-swizzled by the row) and is the
// B operand, so a lane ends up with 4 consecutive output units of one row = one 8- or 16-byte store.
//   bf16 mode: v_mfma_f32_16x16x32_bf16, activations bf16 in HBM;   f32 mode: v_mfma_f32_16x16x4_f32, fp32 throughout.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pf_hip.h"

namespace pf {

typedef __attribute__((ext_vector_type(4))) unsigned int dn_u32x4;

enum DenseEpilogue : int {
    kEpiPlain = 0,   // out = act_t(v)                                                      v = acc + bias
    kEpiGelu = 1,    // out = act_t(D . gelu(v)),  dact = act_t(D . gelu'(v))               D = dropout factor (1 if p = 0)
    kEpiResid = 2,   // out(fp32) = resid + D . v
    kEpiMul = 3,     // out = act_t(v . mul)
};

using DenseArgs = PfDenseArgs;        // include/pf_hip.h
using DenseTnArgs = PfDenseTnArgs;

// packing descriptors: one entry per packed matrix, all packed by ONE launch
struct DensePackEntry {
    int64_t src_off;          // element offset into the raw fp32 parameter buffer
    int64_t dst_off;          // 16-byte fragments offset (in units of dn_u32x4) into the packed buffer
    int32_t mode;             // 0: P[n][k] = raw[off + n ld + k]; 1: P[n][k] = raw[off + k ld + n] (transposed);
                              // 2: transposed convolution, P[t cin + ci][u cout + co] = raw[off + (co cin + ci) kw + s (kw/s - 1 - u) + t]
                              // 3: convolution as im2col GEMM, P[co][tap cin + ci] = raw[off + (co cin + ci) kw + tap]
    int32_t ld, N, K;
    int32_t cin, cout, kw, s;
    int32_t nks_total, ks_off;    // > 0: this matrix is k-steps ks_off .. of a wider packed matrix with nks_total k-steps per tile
    int32_t n_valid, k_valid;     // > 0 (modes 0 / 1): P[n][k] with n >= n_valid or k >= k_valid is zero and the source is not read
                                  // there -- a matrix zero-extended to the kernels' multiples without a padded copy
};
constexpr int kMaxPackEntries = 48;
struct DensePackTable { int32_t n; DensePackEntry e[kMaxPackEntries]; };

int64_t dense_frag_count(bool bf16, int N, int K);      // dn_u32x4 elements of one packed matrix
int dense_pack(bool bf16, const float* raw, const DensePackTable& tab, void* packed, hipStream_t s);
int dense_nt(bool bf16, int epilogue, const DenseArgs& a, hipStream_t s);
int dense_tn(bool bf16, const DenseTnArgs& a, hipStream_t s);

// counter-hash dropout shared by every kernel of the training path.  `site` numbers the dropout layers of the model,
// `idx` the element inside the layer's tensor.  Elements 2 j and 2 j + 1 share ONE 32-bit hash (lowbias32 of j ^ key) and
// take its low / high 16 bits: keep <=> bits >= p * 2^16, kept values are scaled by 1 / (1 - p).  (The first version spent
// five 32-bit multiplies per element; in the attention kernels the hash was two thirds of the vector instructions.)
__host__ __device__ inline uint32_t enc_drop_key(uint32_t seed, uint32_t site) {
    uint32_t k = seed * 0x9E3779B9u + site * 0x85EBCA6Bu + 0x7F4A7C15u;
    k ^= k >> 15; k *= 0x2C1B3C6Du; k ^= k >> 13;
    return k;
}
__host__ __device__ inline uint32_t enc_drop_pair(uint32_t key, uint32_t pair) {
    uint32_t x = pair ^ key;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t enc_drop_hash(uint32_t seed, uint32_t site, uint32_t idx) {       // 16 bits
    const uint32_t x = enc_drop_pair(enc_drop_key(seed, site), idx >> 1);
    return (idx & 1u) ? (x >> 16) : (x & 0xFFFFu);
}
__host__ __device__ inline uint32_t enc_drop_threshold(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
// the four factors of elements idx .. idx + 3, idx a multiple of 4: two hashes
template <typename V4>
__device__ inline void enc_drop4(uint32_t seed, uint32_t site, uint32_t idx, uint32_t thr, float scale, V4& fac) {
    const uint32_t key = enc_drop_key(seed, site);
    const uint32_t a = enc_drop_pair(key, idx >> 1), b = enc_drop_pair(key, (idx >> 1) + 1);
    fac[0] = (a & 0xFFFFu) >= thr ? scale : 0.f;
    fac[1] = (a >> 16) >= thr ? scale : 0.f;
    fac[2] = (b & 0xFFFFu) >= thr ? scale : 0.f;
    fac[3] = (b >> 16) >= thr ? scale : 0.f;
}

}  // namespace pf
