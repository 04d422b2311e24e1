// pf_math.h -- cheap fp32 forms of the encoder's element-wise functions for the bf16 paths, whose results
// are rounded to bf16 (relative spacing 2^-8) right after: a few packed fma + one or two hardware
// transcendental ops instead of the ~30-40 instruction libm expansions.  The fp32 parity paths keep libm.
#pragma once
#include <hip/hip_runtime.h>

namespace pf {
typedef __attribute__((ext_vector_type(4))) float pf_f32x4;

// GELU in its erf form, 0.5 v (1 + erf(v / sqrt 2)), with erf by Abramowitz & Stegun 7.1.28:
//   erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16,  x >= 0,  |error| <= 3e-7
// -- a degree-6 Horner, four squarings and ONE hardware reciprocal per value, all on packed fp32
// instructions (7.1.26, used first, needs a reciprocal AND an exponential: the two quarter-rate
// transcendental ops were 2/3 of its cost).  p^16 overflows to +inf for |v| > ~13, where 1 / inf = 0 gives erf = 1.
// Evaluated as  gelu(v) = max(v, 0) - (|v| / 2) r,  r = p(|v|)^-16  with the 2^(-k/2) of x = |v| / sqrt 2 folded into
// the coefficients: no copysign, no 1 - r, no 1 + erf (12.5 issue slots per value instead of 17 with the bias add
// and the accumulator reads counted, LABLOG R4.12), and the negative tail keeps its relative precision (v r / 2 instead
// of v (1 - (1 - r)) / 2).  max(v, 0) is 0.5 v + 0.5 |v|: exact, and a packed fma.
// (One reciprocal shared by the four values -- r_i = rcp(p0 p1 p2 p3) x the other three, |v| clamped at 6 -- is 10.75 slots
// on paper and measured SLOWER in both users, stem 1.82 vs 1.74 ms, mixer 4.36 vs 4.20: one long dependency chain per four
// values instead of four short ones.  Not kept.)
__device__ __forceinline__ pf_f32x4 gelu_erf_fast4(pf_f32x4 v) {
    pf_f32x4 x, r;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = fabsf(v[k]);
    pf_f32x4 p = x * (0.0000430638f * 0.125f) + (0.0002765672f * 0.17677669529663688f);
    p = p * x + (0.0001520143f * 0.25f);
    p = p * x + (0.0092705272f * 0.35355339059327376f);
    p = p * x + (0.0422820123f * 0.5f);
    p = p * x + (0.0705230784f * 0.70710678118654752f);
    p = p * x + 1.f;
    p = p * p; p = p * p; p = p * p; p = p * p;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_rcpf(p[k]);
    const pf_f32x4 h = x * 0.5f;
    const pf_f32x4 m = v * 0.5f + h;
    return m - h * r;
}

// GELU (erf form) and its derivative, exact fp32 (parity mode) ...
__device__ __forceinline__ float gelu_f32(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f32(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
// ... and the cheap pair for the bf16 training paths (results rounded to bf16 = relative 4e-3 right after): erf by
// Abramowitz & Stegun 7.1.25, erf(a) = 1 - (a1 t + a2 t^2 + a3 t^3) e^(-a^2), t = 1 / (1 + 0.47047 a), |error| <= 2.5e-5 --
// with a = |x| / sqrt 2 its exponential IS the one phi(x) needs, e^(-x^2 / 2): one hardware exponential and one reciprocal
// serve both outputs (14 vector instructions against 25 for the 7.1.28 form; the FFN's GELU epilogue was bound by them)
__device__ __forceinline__ void gelu_fast_pair(float x, float& y, float& dy) {
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.33267253f * a);               // 0.47047 / sqrt 2
    const float e = __builtin_amdgcn_exp2f(-0.72134752f * x * x);               // e^(-x^2 / 2)
    const float poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
    const float half_erfc = 0.5f * poly * e;                                    // 0.5 erfc(|x| / sqrt 2)
    const float cdf = x >= 0.f ? 1.f - half_erfc : half_erfc;
    y = x * cdf;
    dy = cdf + x * 0.3989422804014327f * e;
}

// asinh(x) = sign(x) log(|x| + sqrt(x^2 + 1)); below 1e-3 the identity (error x^3 / 6) avoids the
// cancellation in log(1 + small).  Relative error < 1e-4 over |x| <= 100.
__device__ __forceinline__ float asinh_fast(float x) {
    const float a = fabsf(x);
    // the argument of the logarithm is >= 1: the bare v_log_f32 (no denormal scaling, no compensated ln 2 product: 12 of
    // the library form's 14 instructions)
    const float big = 0.69314718055994531f * __builtin_amdgcn_logf(a + __builtin_amdgcn_sqrtf(a * a + 1.f));
    return copysignf(a < 1e-3f ? a : big, x);
}
}  // namespace pf
