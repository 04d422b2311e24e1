// pf_enc_ops.h -- the non-GEMM kernels of the strain embedding's training path (csrc/pf_enc_ops.hip): LayerNorm,
// multi-head self-attention and the learned-query attention pool, each forward and backward, plus the token assembly.
// Reference: nn.TransformerEncoderLayer(d_model 192, 6 heads, FFN 768, GELU, norm_first, dropout 0.05) x 3 and
// nn.MultiheadAttention pooling, src/ahsd/models/lean_npe.py:167-176, 226-233.  d_model = 192, head dim 32 throughout.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/pf_hip.h"

namespace pf {

constexpr int kEncD = 192, kEncHeads = 6, kEncHd = 32, kEncFF = 768, kEncLayers = 3, kEncPoolQ = 8;
constexpr int kEncMaxTokens = 192;

using LnArgs = PfLnArgs;          // include/pf_hip.h
int ln_forward(bool bf16, const LnArgs& a, hipStream_t s);
int ln_backward(bool bf16, const LnArgs& a, hipStream_t s);

using AttnArgs = PfAttnArgs;
int attn_forward(bool bf16, const AttnArgs& a, hipStream_t s);
int attn_backward(bool bf16, const AttnArgs& a, hipStream_t s);

using PoolArgs = PfPoolArgs;
int pool_forward(bool bf16, const PoolArgs& a, hipStream_t s);
int pool_backward(bool bf16, const PoolArgs& a, hipStream_t s);

// x0[e][t] = t < n_extra ? extra[e][t] : stem_tokens[e][t - n_extra] + token_bias[t]      (lean_npe.py:218-228)
int tok_assemble(const float* stem_tokens, const float* extra, const float* token_bias, int64_t B, int n_extra, int n_stem,
                 float* x0, hipStream_t s);
// backward of the above + the last GELU of the stem: gstem[e][j] = act(dx0[e][n_extra + j] . dact4[e][j]) written with row
// stride / offset of the padded gradient image, dextra[e][t] = dx0[e][t], dbias[t] += sum_e dx0[e][t]
int tok_backward(bool bf16, const float* dx0, const void* dact, int64_t B, int n_extra, int n_det, void* gpad, int64_t gpad_seq_stride,
                 int64_t gpad_offset, float* dextra, float* dbias, hipStream_t s);
int cast_rows(bool bf16, const float* src, void* dst, int64_t n, hipStream_t s);      // fp32 -> act type (copy in fp32 mode)

}  // namespace pf
