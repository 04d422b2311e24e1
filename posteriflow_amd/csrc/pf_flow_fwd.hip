// pf_flow_fwd.hip -- host dispatch of the fused flow forward kernels
// (pf_flow_fwd_kernel.h, instantiated per (precision, NT) by pf_flow_fwd_inst.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "pf_flow_params.h"
#include "pf_status.h"

namespace pf {

#define PF_DECL(P, N)                                                           \
    int launch_flow_forward_p##P##_nt##N(const FwdParams&, int, hipStream_t); \
    int launch_flow_inverse_p##P##_nt##N(const FwdParams&, int, hipStream_t); \
    int launch_flow_train_p##P##_nt##N(const FwdParams&, int, hipStream_t);
PF_DECL(0, 4) PF_DECL(0, 8) PF_DECL(0, 12) PF_DECL(0, 16)
PF_DECL(1, 4) PF_DECL(1, 8) PF_DECL(1, 12) PF_DECL(1, 16)
#undef PF_DECL

int launch_flow_wide_d15(const FwdParams&, hipStream_t);
int launch_flow_wide_d11(const FwdParams&, hipStream_t);
int launch_flow_mid_d15(const FwdParams&, hipStream_t);
int launch_flow_mid_d11(const FwdParams&, hipStream_t);

// A PF_FLAG_WIDE plan is served by two kernels over the same packed stream: the mid-batch kernel (64 rows per workgroup, 8
// waves = two per SIMD: pf_flow_mid_kernel.h) and the large-batch kernel (128 rows per workgroup, one wave per SIMD).  A launch
// costs first round + (rounds - 1) x following round (measured on one box, us: 256 64-row workgroups 178 / 150 -- consecutive
// rounds overlap at their ends --, 256 128-row workgroups 316 / 286): the mid kernel wins where the large-batch kernel's last
// round would be mostly empty (8 193 - 16 384 rows, 32 769 - 49 152, 65 537 - 81 920, ...), the large-batch kernel elsewhere,
// by a few per cent (measured at 32 768 / 49 152 / 65 536 rows: 316 / ~575 / 601 us against 331 / 478 / 628).
// $PF_FLOW_MID (test knob, read per call): 0 never the mid kernel, 1 always.
constexpr double kMidFirstUs = 178.0, kMidNextUs = 150.0, kWideFirstUs = 316.0, kWideNextUs = 286.0;
static bool use_mid(const FlowPlan& L, int64_t batch) {
    if (!L.wide) return false;
    if (const char* e = getenv("PF_FLOW_MID")) return atoi(e) != 0;
    const int64_t mid_rounds = ((batch + 63) / 64 + 255) / 256, wide_rounds = ((batch + 127) / 128 + 255) / 256;
    return kMidFirstUs + (mid_rounds - 1) * kMidNextUs < kWideFirstUs + (wide_rounds - 1) * kWideNextUs;
}

size_t fwd_lds_bytes_host(const FlowPlan& L, int R) {
    const size_t par = (size_t)L.D * 16 * 52 * sizeof(float);
    const size_t pb = std::max((size_t)L.HK * R * kFragBytes, par);
    return (size_t)L.CKM * R * kFragBytes + (size_t)L.HK * R * kFragBytes + pb
         + (size_t)(3 * 16 * 16 * R + 2 * L.NT * kBiasFloatsPerTile + L.D * 16 * R) * sizeof(float);
}

static int rows_per_workgroup_dir(const FlowPlan& L, int64_t batch, bool inverse) {
    if (L.wide) return use_mid(L, batch) ? 64 : wide::kRowsPerWG;
    if (L.generic) return 16;
    // A workgroup streams the whole weight set whatever its row count, so more rows per workgroup (R groups
    // of 16) amortise the stream -- but a launch costs ceil(workgroups / 256 CUs) rounds, and a round of
    // R = 1 / 2 / 3 groups takes 108 / 142 / 195 us (forward, measured): pick the R with the cheapest launch.
    // R = 3 exists for the bf16, H = 256, in-layer-context forward kernel only (registers).
    int R = 1;
    if (const char* f = getenv("PF_FORCE_R")) {          // test knob
        R = atoi(f);
        R = R < 1 ? 1 : (R > 3 ? 3 : R);
    } else if (inverse) {
        if (batch > 256 * 16) R = 2;
    } else {
        const bool r3 = L.bf16 && L.NT == 16 && !L.hoist && L.dense == 0;
        const double t[4] = {0.0, 1.0, 1.32, 1.81};
        double best = 1e300;
        for (int r = 1; r <= (r3 ? 3 : 2); ++r) {
            const int64_t wgs = (batch + 16 * r - 1) / (16 * r);
            const double cost = (double)((wgs + 255) / 256) * t[r];
            if (cost < best - 1e-9) { best = cost; R = r; }
        }
    }
    const bool r3_built = !inverse && L.bf16 && L.NT == 16 && !L.hoist && L.dense == 0;
    if (R == 3 && !r3_built) R = 2;
    if (L.dense == 1) R = 1;
    while (R > 1 && fwd_lds_bytes_host(L, R) > 160 * 1024) --R;
    return 16 * R;
}
int rows_per_workgroup(const FlowPlan& L, int64_t batch) { return rows_per_workgroup_dir(L, batch, false); }

// the kernel launch_flow_forward picks for (plan, batch), as rocprofv3 prints it
void forward_kernel_name(const FlowPlan& L, int64_t batch, char* out, size_t n) {
    if (L.wide) { snprintf(out, n, use_mid(L, batch) ? "pf::flow_mid_kernel<%d, %d>" : "pf::flow_wide_kernel<%d, %d>", L.D, L.CKM); return; }
    if (L.generic) { snprintf(out, n, "pf::flow_generic_kernel<%s, false>", L.bf16 ? "true" : "false"); return; }
    const int R = L.dense == 1 ? 1 : rows_per_workgroup(L, batch) / 16;
    snprintf(out, n, "pf::flow_kernel<%s, %d, %d, %d, %d, false>", L.bf16 ? "true" : "false", L.NT, R, L.CKM, L.dense);
}

int launch_flow_generic(const FwdParams& p, bool inverse, hipStream_t s);       // pf_flow_generic.hip

int launch_flow_forward(const FwdParams& p_in, hipStream_t s) {
    if (p_in.batch == 0) return PF_OK;
    if (p_in.plan.generic) return p_in.drop_thresh ? (int)PF_ERR_UNSUPPORTED : launch_flow_generic(p_in, false, s);
    if (p_in.plan.wide) {
        if (p_in.drop_thresh) return PF_ERR_UNSUPPORTED;   // the large-batch kernels are evaluation kernels
        if (use_mid(p_in.plan, p_in.batch)) {
            if (p_in.plan.D == 15) return launch_flow_mid_d15(p_in, s);
            if (p_in.plan.D == 11) return launch_flow_mid_d11(p_in, s);
            return PF_ERR_UNSUPPORTED;
        }
        if (p_in.plan.D == 15) return launch_flow_wide_d15(p_in, s);
        if (p_in.plan.D == 11) return launch_flow_wide_d11(p_in, s);
        return PF_ERR_UNSUPPORTED;
    }
    FwdParams p = p_in;
    static const int ablate = [] { const char* a = getenv("PF_ABLATE"); return a ? atoi(a) : 0; }();   // (read once per process;
    if (ablate) p.ablate = ablate;                                                                  // -DPF_ABLATE_BUILD builds only)
    const int R = rows_per_workgroup(p.plan, p.batch) / 16;
#define PF_CASE(P, N) case N: return p.drop_thresh ? launch_flow_train_p##P##_nt##N(p, R, s) : launch_flow_forward_p##P##_nt##N(p, R, s);
    if (p.plan.bf16) {
        switch (p.plan.NT) { PF_CASE(1, 4) PF_CASE(1, 8) PF_CASE(1, 12) PF_CASE(1, 16) }
    } else {
        switch (p.plan.NT) { PF_CASE(0, 4) PF_CASE(0, 8) PF_CASE(0, 12) PF_CASE(0, 16) }
    }
#undef PF_CASE
    return PF_ERR_UNSUPPORTED;
}

// The dropout factors flow_train_kernel applied, for the backward: mask[j][l][row][unit] in {0, 1 / (1 - p)}, units in
// nflows order (the kernel hashes the degree-sorted position; sorted_units maps it back).
namespace {
struct DropMaskArgs {
    float* mask;
    int64_t batch;
    int L, H;
    uint32_t thresh, seed;
    float scale;
    uint8_t pos_of_unit[256];
};
__global__ __launch_bounds__(256) void dropout_mask_kernel(const DropMaskArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, n = 2 * (int64_t)a.L * a.batch * a.H;
    if (idx >= n) return;
    const int u = (int)(idx % a.H);
    const int64_t t = idx / a.H, row = t % a.batch, jl = t / a.batch;
    const int l = (int)(jl % a.L), j = (int)(jl / a.L);
    a.mask[idx] = drop_factor(drop_row_hash(a.seed, (uint32_t)row), 2 * l + j, a.pos_of_unit[u], a.thresh, a.scale);
}
}  // namespace

int launch_dropout_mask(const FlowPlan& L, uint32_t thresh, uint32_t seed, float scale, int64_t batch, float* mask, hipStream_t s) {
    if (L.H > 256) return PF_ERR_UNSUPPORTED;
    DropMaskArgs a{mask, batch, L.L, L.H, thresh, seed, scale, {}};
    int perm[256];
    sorted_units(L.D, L.H, perm);
    for (int p = 0; p < L.H; ++p) a.pos_of_unit[perm[p]] = (uint8_t)p;
    const int64_t n = 2 * (int64_t)L.L * batch * L.H;
    if (n == 0) return PF_OK;
    dropout_mask_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
    return launch_status();
}

int launch_flow_inverse(const FwdParams& p, hipStream_t s) {
    if (p.batch == 0) return PF_OK;
    if (p.plan.generic) return launch_flow_generic(p, true, s);
    if (p.plan.wide) return PF_ERR_UNSUPPORTED;             // the large-batch layout is forward-only
    const int R = rows_per_workgroup_dir(p.plan, p.batch, true) / 16;
#define PF_CASE(P, N) case N: return launch_flow_inverse_p##P##_nt##N(p, R, s);
    if (p.plan.bf16) {
        switch (p.plan.NT) { PF_CASE(1, 4) PF_CASE(1, 8) PF_CASE(1, 12) PF_CASE(1, 16) }
    } else {
        switch (p.plan.NT) { PF_CASE(0, 4) PF_CASE(0, 8) PF_CASE(0, 12) PF_CASE(0, 16) }
    }
#undef PF_CASE
    return PF_ERR_UNSUPPORTED;
}

}  // namespace pf
