// Device code shared by the correlation kernels: the per-window epilogue
// (window mean / std-dev from prefix sums, boost, gate, mix).
#pragma once

#include "sgz_internal.hpp"

namespace sgz {

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

// kUpper = false: the similarity as the reference computes it.
// kUpper = true (screening with reduced-precision products of relative error <= relErr): simOut is an upper
// bound of it and *loOut (if given) a lower bound.  By Cauchy-Schwarz the error of a normalised cross term
// is at most relErr * sqrt(1 + (dm / sd)^2), dm the offset of the window mean from the centring constant,
// sd the window std-dev.  A window without spread gets (-inf, +inf); a boost within 1e-5 of the gate makes
// the bounds include 0, the value of a gated window.
template <bool kUpper = false>
__device__ __forceinline__ void window_result(const CorrParams &cp, const Epilogue &ep, long long o,
                                              double crossT, double crossS, float &simOut, float &boostOut,
                                              float relErr = 0.0f, float *loOut = nullptr)
{
    const double nT = (double)cp.L;
    const double nS = (double)cp.L * (double)(cp.numCh - 1);
    const double dmT = (ep.P(0, o + cp.L) - ep.P(0, o)) / nT;  // window mean - kT
    const double avgD = cp.kT + dmT;
    const float avgF = (float)avgD;  // MathUtil.avg returns Float
    const double lnAvgB = log((double)avgF);
    const float boost = (float)exp((cp.lnAvgIn - lnAvgB) / 0.6);
    float sim = 0.0f, lo = 0.0f;
    const bool open = kUpper ? boost <= cp.maxBoost * 1.00001f : boost <= cp.maxBoost;
    if (open) {
        float temporal = 0.0f, spectral = 0.0f, errT = 0.0f, errS = 0.0f;
        if (cp.w > 0.0f) {
            double var = (ep.P(1, o + cp.L) - ep.P(1, o)) / nT - dmT * dmT;
            if (var < 0.0) var = 0.0;
            const double bStd = sqrt(var);
            const double cross = crossT - dmT * cp.sumAT;
            temporal = (float)(cross / (cp.aStdT * bStd * nT));
            if (kUpper) errT = bStd > 0.0 ? relErr * (float)sqrt(1.0 + (dmT / bStd) * (dmT / bStd)) : INFINITY;
        }
        if (cp.w < 1.0f) {
            const double dmS = (ep.P(2, o + cp.L) - ep.P(2, o)) / nS;
            double var = (ep.P(3, o + cp.L) - ep.P(3, o)) / nS - dmS * dmS;
            if (var < 0.0) var = 0.0;
            const double bStd = sqrt(var);
            const double cross = crossS - dmS * cp.sumAS;
            spectral = (float)(cross / (cp.aStdS * bStd * nS));
            if (kUpper) errS = bStd > 0.0 ? relErr * (float)sqrt(1.0 + (dmS / bStd) * (dmS / bStd)) : INFINITY;
        }
        sim = temporal * cp.w + spectral * (1.0f - cp.w);  // fp32, no contraction (-ffp-contract=off)
        if (kUpper) {
            const float e = errT * cp.w + errS * (1.0f - cp.w) + 1e-5f;
            const bool finite = e < INFINITY && fabsf(sim) < INFINITY;  // false for NaN as well
            const bool isNan = sim != sim && e < INFINITY;               // NaN data: stays NaN, as in the exact path
            if (!isNan) {
                lo = finite ? sim - e : -INFINITY;
                sim = finite ? sim + e : INFINITY;
            } else {
                lo = sim;
            }
            if (boost > cp.maxBoost * 0.99999f) {  // the exact path may gate this window
                sim = fmaxf(sim, 0.0f);
                lo = fminf(lo, 0.0f);
            }
        }
    }
    simOut = sim;
    boostOut = boost;
    if (kUpper && loOut) *loOut = lo;
}

}  // namespace sgz
