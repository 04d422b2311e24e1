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
};
constexpr int kMaxPackEntries = 48;
struct DensePackTable { int32_t n; DensePackEntry e[kMaxPackEntries]; };

int64_t dense_frag_count(bool bf16, int N, int K);      // dn_u32x4 elements of one packed matrix
int dense_pack(bool bf16, const float* raw, const DensePackTable& tab, void* packed, hipStream_t s);
int dense_nt(bool bf16, int epilogue, const DenseArgs& a, hipStream_t s);
int dense_tn(bool bf16, const DenseTnArgs& a, hipStream_t s);

// counter-hash dropout shared by every kernel of the training path: keep <=> hash >= p * 2^24 (24 bits), kept values are
// scaled by 1 / (1 - p).  `site` numbers the dropout layers of the model, `idx` the element inside the layer's tensor.
__host__ __device__ inline uint32_t enc_drop_hash(uint32_t seed, uint32_t site, uint32_t idx) {
    uint32_t h = seed ^ (site * 0x9E3779B9u);
    h ^= idx + 0x7F4A7C15u + (h << 6) + (h >> 2);
    h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += idx * 0x27D4EB2Fu; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h >> 8;
}
__host__ __device__ inline uint32_t enc_drop_threshold(float p) { return (uint32_t)(p * 16777216.0f + 0.5f); }

}  // namespace pf
