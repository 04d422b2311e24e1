// pf_rqs_bwd.h -- backward of the rational-quadratic spline with linear tails for ONE (row, feature) pair (device
// function shared by pf_flow_rqs_backward and the fused backward chain, pf_flow_bwd_chain.hip).  Derivation and
// reference lines: pf_flow_bwd.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pf {

constexpr int kRqsMaxBins = 16;      // the scheduled kernels' bin capacity (register arrays); the wide variant below takes 32

struct RqsConsts {
    int K;
    float tb, min_w, min_h, min_d, deriv_const;
};

__device__ __forceinline__ float rqs_softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }
__device__ __forceinline__ float rqs_sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

// One (row, feature) pair: par = its 3K-1 raw parameters (nflows order), gp receives dL/d(raw parameters) (may alias
// par: every parameter is read before the first write), returns the direct part of dL/du.
template <int MAXB = kRqsMaxBins>
__device__ __forceinline__ float rqs_backward_pair(const float* par, float* gp, float x, float gy, float gl, const RqsConsts& a) {
    const int K = a.K, P = 3 * K - 1;
    const float tb = a.tb;
    if (!(x >= -tb && x <= tb)) {          // linear tail: identity
        for (int i = 0; i < P; ++i) gp[i] = 0.f;
        return gy;
    }
    float sw[MAXB], sh[MAXB], ud[MAXB];
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        sw[i] = i < K ? par[i] : -INFINITY;
        sh[i] = i < K ? par[K + i] : -INFINITY;
        ud[i] = i < K - 1 ? par[2 * K + i] : 0.f;
        mw = fmaxf(mw, sw[i]); mh = fmaxf(mh, sh[i]);
    }
    float tw = 0.f, th_ = 0.f;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        sw[i] = i < K ? expf(sw[i] - mw) : 0.f; tw += sw[i];
        sh[i] = i < K ? expf(sh[i] - mh) : 0.f; th_ += sh[i];
    }
    const float cw = 1.f - a.min_w * static_cast<float>(K), ch = 1.f - a.min_h * static_cast<float>(K);
#pragma unroll
    for (int i = 0; i < MAXB; ++i) { sw[i] /= tw; sh[i] /= th_; }      // softmax probabilities

    // bin search, as the forward does it (right knot of the last bin carries the 1e-6 of searchsorted)
    const float span = 2.f * tb;
    float cumw = 0.f, cumh = 0.f, xl = -tb, xr = tb, yl = -tb, yr = tb;
    float dl_raw = a.deriv_const, dr_raw = a.deriv_const;
    int b = 0;
    bool prev_ge = true;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        if (i < K) {
            cumw += a.min_w + cw * sw[i];
            cumh += a.min_h + ch * sh[i];
            const bool last = i == K - 1;
            const float kr = last ? tb : span * cumw - tb;
            const float hr = last ? tb : span * cumh - tb;
            const float dr = last ? a.deriv_const : ud[i];
            const bool ge = x >= (last ? tb + 1e-6f : kr);
            const bool sel = prev_ge && !ge;
            xr = sel ? kr : xr; yr = sel ? hr : yr; dr_raw = sel ? dr : dr_raw; b = sel ? i : b;
            xl = ge ? kr : xl;  yl = ge ? hr : yl;  dl_raw = ge ? dr : dl_raw;
            prev_ge = ge;
        }
    }
    const float W = xr - xl, Hh = yr - yl;
    const float dl = a.min_d + rqs_softplus_f(dl_raw), dr = a.min_d + rqs_softplus_f(dr_raw);
    const float delta = Hh / W, th = (x - xl) / W, omt = 1.f - th, tt = th * omt;
    const float A = delta * th * th + dl * tt;
    const float Dn = delta + (dl + dr - 2.f * delta) * tt;
    const float E = dr * th * th + 2.f * delta * tt + dl * omt * omt;
    const float iDn = 1.f / Dn, iDn2 = iDn * iDn, iE = 1.f / E, iW = 1.f / W;

    const float y_th = Hh * delta * E * iDn2;
    const float y_de = Hh * (th * th * Dn - A * (1.f - 2.f * tt)) * iDn2;
    const float y_dl = Hh * tt * (Dn - A) * iDn2;
    const float y_dr = -Hh * A * tt * iDn2;
    const float E_th = 2.f * dr * th + 2.f * delta * (1.f - 2.f * th) - 2.f * dl * omt;
    const float Dn_th = (dl + dr - 2.f * delta) * (1.f - 2.f * th);
    const float l_th = E_th * iE - 2.f * Dn_th * iDn;
    const float l_de = 2.f / delta + 2.f * tt * iE - 2.f * (1.f - 2.f * tt) * iDn;
    const float l_dl = omt * omt * iE - 2.f * tt * iDn;
    const float l_dr = th * th * iE - 2.f * tt * iDn;

    const float G_th = gy * y_th + gl * l_th;
    const float G_de = gy * y_de + gl * l_de;
    const float G_dl = gy * y_dl + gl * l_dl;
    const float G_dr = gy * y_dr + gl * l_dr;
    const float g_W = -(G_th * th + G_de * delta) * iW;
    const float g_H = gy * A * iDn + G_de * iW;
    const float g_kxb = -G_th * iW - g_W, g_kxb1 = g_W;       // d/d kx_b, d/d kx_{b+1}
    const float g_kyb = gy - g_H, g_kyb1 = g_H;
    const float gu_direct = G_th * iW;

    // knots -> softmax probabilities: kx_j = span * sum_{i<j} (min + c s_i) - tb for 1 <= j <= K-1
    const bool right_free = b + 1 <= K - 1;                    // knot b+1 is interior
    float dot_w = 0.f, dot_h = 0.f;
    float gsw[MAXB], gsh[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const float below = i < b ? 1.f : 0.f;                 // i < b  (knot b; b = 0 -> never)
        const float upto = (i <= b && right_free) ? 1.f : 0.f; // i < b+1 (knot b+1)
        gsw[i] = span * cw * (below * g_kxb + upto * g_kxb1);
        gsh[i] = span * ch * (below * g_kyb + upto * g_kyb1);
        dot_w += sw[i] * gsw[i];
        dot_h += sh[i] * gsh[i];
    }
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        if (i < K) {
            gp[i] = sw[i] * (gsw[i] - dot_w);
            gp[K + i] = sh[i] * (gsh[i] - dot_h);
        }
        if (i < K - 1)        // raw derivative i belongs to interior knot i+1
            gp[2 * K + i] = ((i == b - 1) ? G_dl : 0.f) * rqs_sigmoid_f(ud[i]) + ((i == b) ? G_dr : 0.f) * rqs_sigmoid_f(ud[i]);
    }
    return gu_direct;
}

}  // namespace pf
