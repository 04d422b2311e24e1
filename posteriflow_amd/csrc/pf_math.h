// pf_math.h -- cheap fp32 forms of the encoder's element-wise functions for the bf16 paths, whose results
// are rounded to bf16 (relative spacing 2^-8) right after: a few packed fma + one or two hardware
// transcendental ops instead of the ~30-40 instruction libm expansions.  The fp32 parity paths keep libm.
#pragma once
#include <hip/hip_runtime.h>

namespace pf {
typedef __attribute__((ext_vector_type(4))) float pf_f32x4;

// GELU in its erf form, 0.5 v (1 + erf(v / sqrt 2)), erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7).
__device__ __forceinline__ pf_f32x4 gelu_erf_fast4(pf_f32x4 v) {
    pf_f32x4 x, t, e, r;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = fabsf(v[k]);
    x = x * 0.70710678118654752f;
    const pf_f32x4 d = x * 0.3275911f + 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_rcpf(d[k]);
    const pf_f32x4 poly = t * (t * (t * (t * (t * 1.061405429f - 1.453152027f) + 1.421413741f) - 0.284496736f) + 0.254829592f);
    const pf_f32x4 a = x * x * -1.4426950408889634f;
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f(a[k]);
    const pf_f32x4 erf_abs = 1.f - poly * e;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = copysignf(erf_abs[k], v[k]);
    return v * 0.5f * (r + 1.f);
}

// asinh(x) = sign(x) log(|x| + sqrt(x^2 + 1)); below 1e-3 the identity (error x^3 / 6) avoids the
// cancellation in log(1 + small).  Relative error < 1e-4 over |x| <= 100.
__device__ __forceinline__ float asinh_fast(float x) {
    const float a = fabsf(x);
    const float big = __logf(a + __builtin_amdgcn_sqrtf(a * a + 1.f));
    return copysignf(a < 1e-3f ? a : big, x);
}
}  // namespace pf
