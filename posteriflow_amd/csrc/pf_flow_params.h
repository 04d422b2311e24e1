// pf_flow_params.h -- launch parameter blocks and internal entry points shared by
// pf_api.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "pf_layout.h"

namespace pf {

struct FwdParams {
    const char* packed;      // weights then biases
    const float* x;          // [B, D]
    const float* ctx;        // [B, C]
    const int32_t* ar_perm;  // [D] or null (forward: ar_perm; inverse: ar_inv_perm)
    const float* log_sigma;  // [B, D] or null (PSDScaledNormal log-scale, flows.py:56-85)
    float* z;                // [B, D] or null
    float* logdet;           // [B] or null
    float* nll;              // [B] or null
    int64_t batch;
    int64_t ctx_rows;        // inverse: context rows (divides batch); forward: == batch
    uint32_t* fail_flags;    // inverse: [B] or null, bit 0 = negative discriminant
    float* nll_sum;          // forward: float[2], (sum of nll, rows) accumulated with atomics, or null
    float* zero_pair;        // forward: float[2 PF_REDUCE_SLOTS] set to zero by the kernel (the accumulator of a LATER launch), or null
    float* u_save;           // forward: [L, B, D] input of every layer's conditioner (training), or null
    const void* cproj;       // hoisted plans: fp32 projections in fragment order (else null)
    FlowPlan plan;
    float tail_bound, min_w, min_h, min_d;
    float deriv_const;       // log(exp(1 - min_d) - 1), computed in double on the host
    // training dropout of the residual blocks (flow_train_kernel): keep <=> (hash >> 8) >= drop_thresh, kept values are
    // multiplied by drop_scale = 1 / (1 - p); drop_thresh = 0: none
    uint32_t drop_thresh, drop_seed;
    float drop_scale;
    int ablate;              // timing experiments only ($PF_ABLATE): 1 no spline, 2 no weight traffic, 4 no MFMA, 8 no barriers
};

// Dropout decision of (row, layer-block lb = 2 l + b, degree-sorted position pos): a counter hash, so that the forward kernel
// and the mask kernel the backward uses (pf_flow_dropout_mask) agree without a stored mask.
__host__ __device__ inline uint32_t drop_row_hash(uint32_t seed, uint32_t row) {
    uint32_t x = seed ^ (row * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t drop_hash(uint32_t row_hash, uint32_t lb, uint32_t pos) {
    uint32_t x = row_hash + (lb * 256u + pos + 1u) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}
__host__ __device__ inline float drop_factor(uint32_t row_hash, uint32_t lb, uint32_t pos, uint32_t thresh, float scale) {
    return (drop_hash(row_hash, lb, pos) >> 8) >= thresh ? scale : 0.f;
}

int build_pack_map(const FlowPlan& L, int32_t* map);
int64_t pack_map_len(const FlowPlan& L);
int64_t raw_param_count(const FlowPlan& L);
int launch_pack(const FlowPlan& L, const float* raw, const int32_t* map, void* packed, hipStream_t s);
int rows_per_workgroup(const FlowPlan& L, int64_t batch);
int launch_flow_forward(const FwdParams& p, hipStream_t s);
void forward_kernel_name(const FlowPlan& L, int64_t batch, char* out, size_t n);
int launch_flow_inverse(const FwdParams& p, hipStream_t s);
int launch_dropout_mask(const FlowPlan& L, uint32_t thresh, uint32_t seed, float scale, int64_t batch, float* mask, hipStream_t s);
int launch_ctx_project(const FlowPlan& L, const char* packed, const float* ctx, int64_t ctx_rows,
                       void* out, hipStream_t s);
int64_t ctx_project_bytes(const FlowPlan& L, int64_t ctx_rows);

}  // namespace pf
