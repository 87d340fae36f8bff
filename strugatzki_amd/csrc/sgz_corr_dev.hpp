// Device code shared by the correlation kernels: the per-window epilogue
// (window mean / std-dev from prefix sums, boost, gate, mix).
#pragma once

#include "sgz_internal.hpp"

namespace sgz {

// Order-preserving integer key of a float (for atomicMax over floats of either sign); its own inverse.
__device__ __forceinline__ int bound_key(float x)
{
    const int b = __float_as_int(x);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float bound_from_key(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// Prefix sums of the per-frame moments: P(q, i) = sum over frames < i.
struct Epilogue {
    const double *pre;    // 4 arrays of `frames` in-tile inclusive sums
    const double *tbase;  // 4 arrays of `nptiles` tile bases
    long long frames, nptiles;
    int tileShift;        // log2(frames per prep tile)
    __device__ __forceinline__ double P(int q, long long i) const
    {
        if (i == 0) return 0.0;
        return pre[q * frames + (i - 1)] + tbase[q * nptiles + ((i - 1) >> tileShift)];
    }
};

// The similarity and boost of window o as the reference computes them from the cross terms
// (impl/FeatureCorrelationImpl.scala:193-232): fp64 statistics, fp32 mix.
__device__ __forceinline__ void window_result(const CorrParams &cp, const Epilogue &ep, long long o,
                                              double crossT, double crossS, float &simOut, float &boostOut)
{
    const double nT = (double)cp.L;
    const double nS = (double)cp.L * (double)(cp.numCh - 1);
    const double dmT = (ep.P(0, o + cp.L) - ep.P(0, o)) / nT;  // window mean - kT
    const double avgD = cp.kT + dmT;
    const float avgF = (float)avgD;  // MathUtil.avg returns Float
    const double lnAvgB = log((double)avgF);
    const float boost = (float)exp((cp.lnAvgIn - lnAvgB) / 0.6);
    float sim = 0.0f;
    if (boost <= cp.maxBoost) {
        float temporal = 0.0f, spectral = 0.0f;
        if (cp.w > 0.0f) {
            double var = (ep.P(1, o + cp.L) - ep.P(1, o)) / nT - dmT * dmT;
            if (var < 0.0) var = 0.0;
            const double bStd = sqrt(var);
            const double cross = crossT - dmT * cp.sumAT;
            temporal = (float)(cross / (cp.aStdT * bStd * nT));
        }
        if (cp.w < 1.0f) {
            const double dmS = (ep.P(2, o + cp.L) - ep.P(2, o)) / nS;
            double var = (ep.P(3, o + cp.L) - ep.P(3, o)) / nS - dmS * dmS;
            if (var < 0.0) var = 0.0;
            const double bStd = sqrt(var);
            const double cross = crossS - dmS * cp.sumAS;
            spectral = (float)(cross / (cp.aStdS * bStd * nS));
        }
        sim = temporal * cp.w + spectral * (1.0f - cp.w);  // fp32, no contraction (-ffp-contract=off)
    }
    simOut = sim;
    boostOut = boost;
}

// Screening variant for cross terms formed from reduced-precision products of relative error <= relErr:
// an upper and a lower bound of the similarity window_result would give, and the boost to about 1e-6.
// By Cauchy-Schwarz the error of a normalised cross term is at most relErr * sqrt(1 + (dm / sd)^2), dm the
// offset of the window mean from the centring constant, sd the window std-dev.  Differences of prefix sums
// and the variance are taken in fp64, everything after them in fp32: that adds a few 1e-7, covered by the
// absolute slack of 1e-5.  A window without spread gets (-inf, +inf); a boost within 1e-5 of the gate makes
// the bounds include 0, the value of a gated window.
// The part that does not depend on the cross terms is evaluated ahead of them (window_coeffs) and folded into
// four numbers: sim ~ a * crossT + b * crossS - c, error bound |e|; e < 0 marks a boost next to the gate.
struct WindowCoeffs {
    float a, b, c, e;
};

__device__ __forceinline__ WindowCoeffs window_coeffs(const CorrParams &cp, const Epilogue &ep, long long o,
                                                      float relErr, float &boostOut)
{
    const double nTd = (double)cp.L, nSd = (double)cp.L * (double)(cp.numCh - 1);
    const double rnT = 1.0 / nTd, rnS = 1.0 / nSd;  // wave-uniform: one division each, not one per window
    const double dmTd = (ep.P(0, o + cp.L) - ep.P(0, o)) * rnT;
    const double varT = (ep.P(1, o + cp.L) - ep.P(1, o)) * rnT - dmTd * dmTd;
    const double dmSd = (ep.P(2, o + cp.L) - ep.P(2, o)) * rnS;
    const double varS = (ep.P(3, o + cp.L) - ep.P(3, o)) * rnS - dmSd * dmSd;
    const float avgF = (float)(cp.kT + dmTd);
    const float boost = expf(((float)cp.lnAvgIn - logf(avgF)) * (1.0f / 0.6f));
    boostOut = boost;
    WindowCoeffs wc = {0.0f, 0.0f, 0.0f, 0.0f};  // a gated window: sim = 0 exactly
    if (boost <= cp.maxBoost * 1.00001f) {
        float errT = 0.0f, errS = 0.0f;
        if (cp.w > 0.0f) {
            const float dm = (float)dmTd, sd = sqrtf(fmaxf((float)varT, 0.0f));
            const float r = dm / sd;
            wc.a = cp.w / ((float)cp.aStdT * sd * (float)nTd);
            wc.c = wc.a * dm * (float)cp.sumAT;
            errT = sd > 0.0f ? relErr * sqrtf(1.0f + r * r) : INFINITY;
        }
        if (cp.w < 1.0f) {
            const float dm = (float)dmSd, sd = sqrtf(fmaxf((float)varS, 0.0f));
            const float r = dm / sd;
            wc.b = (1.0f - cp.w) / ((float)cp.aStdS * sd * (float)nSd);
            wc.c += wc.b * dm * (float)cp.sumAS;
            errS = sd > 0.0f ? relErr * sqrtf(1.0f + r * r) : INFINITY;
        }
        wc.e = errT * cp.w + errS * (1.0f - cp.w) + 1e-5f;
        if (boost > cp.maxBoost * 0.99999f) wc.e = -wc.e;  // the exact path may gate this window
    }
    return wc;
}

__device__ __forceinline__ void window_bounds(const WindowCoeffs &wc, float crossT, float crossS, float &upOut,
                                              float &loOut)
{
    const float sim = wc.a * crossT + wc.b * crossS - wc.c;
    const float e = fabsf(wc.e);
    float up, lo;
    const bool finite = e < INFINITY && fabsf(sim) < INFINITY;  // false for NaN as well
    if (sim != sim && e < INFINITY) {  // NaN data: stays NaN, as in the exact path
        up = lo = sim;
    } else {
        lo = finite ? sim - e : -INFINITY;
        up = finite ? sim + e : INFINITY;
    }
    if (wc.e < 0.0f) {
        up = fmaxf(up, 0.0f);
        lo = fminf(lo, 0.0f);
    }
    if (wc.e == 0.0f) up = lo = 0.0f;  // gated
    upOut = up;
    loOut = lo;
}

}  // namespace sgz
