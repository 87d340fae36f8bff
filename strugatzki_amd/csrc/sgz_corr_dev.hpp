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
// The variance comes from differences of prefix sums; where it is so small that the rounding of those sums
// (a few 2^-52 of their magnitude) could move it by more than one part in 10^6 -- stretches of constant
// features, digital silence -- `suspect` is set: the reference's two-pass statistics give exactly 0 there
// (and a NaN similarity), and only a two-pass evaluation from the frames reproduces that.
__device__ __forceinline__ void window_result(const CorrParams &cp, const Epilogue &ep, long long o,
                                              double crossT, double crossS, float &simOut, float &boostOut,
                                              bool &suspect)
{
    constexpr double kRound = 8.0 * 2.220446049250313e-16, kTrust = 1.0e6;
    const double nT = (double)cp.L;
    const double nS = (double)cp.L * (double)(cp.numCh - 1);
    const double p0h = ep.P(0, o + cp.L), p0l = ep.P(0, o);
    const double dmT = (p0h - p0l) / nT;  // window mean - kT
    const double avgD = cp.kT + dmT;
    const float avgF = (float)avgD;  // MathUtil.avg returns Float
    const double lnAvgB = log((double)avgF);
    const float boost = (float)exp((cp.lnAvgIn - lnAvgB) / 0.6);
    float sim = 0.0f;
    suspect = false;
    if (boost <= cp.maxBoost) {
        float temporal = 0.0f, spectral = 0.0f;
        if (cp.w > 0.0f) {
            const double p1h = ep.P(1, o + cp.L), p1l = ep.P(1, o);
            double var = (p1h - p1l) / nT - dmT * dmT;
            const double err = kRound * ((fabs(p1h) + fabs(p1l)) + 2.0 * fabs(dmT) * (fabs(p0h) + fabs(p0l))) / nT;
            suspect |= !(var >= kTrust * err);
            if (var < 0.0) var = 0.0;
            const double bStd = sqrt(var);
            const double cross = crossT - dmT * cp.sumAT;
            temporal = (float)(cross / (cp.aStdT * bStd * nT));
        }
        if (cp.w < 1.0f) {
            const double p2h = ep.P(2, o + cp.L), p2l = ep.P(2, o);
            const double p3h = ep.P(3, o + cp.L), p3l = ep.P(3, o);
            const double dmS = (p2h - p2l) / nS;
            double var = (p3h - p3l) / nS - dmS * dmS;
            const double err = kRound * ((fabs(p3h) + fabs(p3l)) + 2.0 * fabs(dmS) * (fabs(p2h) + fabs(p2l))) / nS;
            suspect |= !(var >= kTrust * err);
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
// offset of the window mean from the centring constant, sd the window std-dev.
//
// The screening preparation stores the in-tile prefix sums of the four frame moments as one fp32 record per
// frame (HalfPrefix: 16 bytes fetched per window end instead of four doubles); the tile bases stay fp64.  A
// stored value p differs from the fp64 sum by at most 2^-24 |p|, so a window sum is off by at most
// E = 2^-24 (|p_hi| + |p_lo|): that moves the mean by E1 / n and the variance by dv = E2 / n + 2 |dm| E1 / n,
// hence 1 / sd by a relative rho <= dv / var (for dv <= var / 2; beyond that the window counts as having no
// spread).  Both go into the error bound, next to the few 1e-7 of the fp32 arithmetic after the sums (one-ulp
// hardware reciprocal, square root, log2, exp2), which the absolute slack of 1e-5 covers.
// A window without spread gets (-inf, +inf); a boost within 1e-5 of the gate makes the bounds include 0, the
// value of a gated window.
// The part that does not depend on the cross terms is folded into four numbers (window_coeffs):
// sim ~ a * crossT + b * crossS - c, error bound |e|; e < 0 marks a boost next to the gate.
struct WindowCoeffs {
    float a, b, c, e;
};

struct HalfPrefix {
    const float4 *pre;    // per frame: in-tile inclusive sums of (t, t^2, sum s, sum s^2)
    const double *tbase;  // 4 arrays of `nptiles` tile bases
    long long nptiles;
    int tileShift;
};

// The two prefix records of window o (no branch around the loads; fetched ahead of their use).
struct WindowRecords {
    float4 hi, lo;
};
__device__ __forceinline__ WindowRecords window_records(const HalfPrefix &hp, unsigned o, int L)
{
    return WindowRecords{hp.pre[o + (unsigned)L - 1], hp.pre[o > 0 ? o - 1 : 0]};
}

// What window_coeffs needs of a window that depends on the database, the norms and the target *length* only --
// kept per window (32 bytes) from the first search on and reused by every later one, so that the epilogue of the
// matrix-core kernel is a handful of fp32 operations on two 16-byte loads instead of fp64 statistics from
// prefix records and tile bases:
//   rsd = 1 / std-dev and dm = mean - centring constant of the two channel groups, lnAvg = ln(window's average
//   loudness), g = dv / var (how far the fp32 prefix records can move 1 / sd; +inf where the window has no
//   usable spread), h = rsd * E1 / n (what they can move the mean by, times rsd; the two h as half-precision
//   numbers rounded up, in one word).
struct WindowStatic {
    float rsdT, dmT, gT, lnAvg;
    float rsdS, dmS, gS;
    unsigned int hTS;
};

__device__ __forceinline__ unsigned int half_bits_up(float x)  // a half-precision number >= x (x >= 0)
{
    _Float16 h = (_Float16)x;
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    if ((float)h < x) b++;  // next representable (also subnormal -> normal, max -> inf)
    return b;
}
__device__ __forceinline__ float half_from_bits(unsigned int b)
{
    const unsigned short u = (unsigned short)b;
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}

// cT: centring constant of the loudness channel of this file's image.
__device__ __forceinline__ WindowStatic window_static(int L, int numCh, const HalfPrefix &hp, unsigned o,
                                                      const WindowRecords &rec, double cT)
{
    const double nTd = (double)L, nSd = (double)L * (double)(numCh - 1);
    const double rnT = 1.0 / nTd, rnS = 1.0 / nSd;
    const unsigned l1 = o > 0 ? o - 1 : 0, h1 = o + (unsigned)L - 1;
    const float4 ph = rec.hi, pl = rec.lo;
    const unsigned th = h1 >> hp.tileShift, tl = l1 >> hp.tileShift;
    const bool first = o == 0;  // nothing to subtract
    const float phv[4] = {ph.x, ph.y, ph.z, ph.w}, plv[4] = {pl.x, pl.y, pl.z, pl.w};
    double d[4];
    float E[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const double bh = hp.tbase[q * hp.nptiles + th], bl = hp.tbase[q * hp.nptiles + tl];
        d[q] = ((double)phv[q] + bh) - (first ? 0.0 : (double)plv[q] + bl);
        E[q] = 6.0e-8f * (fabsf(phv[q]) + (first ? 0.0f : fabsf(plv[q])));  // 2^-24 = 5.96e-8
    }
    const double dmTd = d[0] * rnT;
    const double varT = d[1] * rnT - dmTd * dmTd;
    const double dmSd = d[2] * rnS;
    const double varS = d[3] * rnS - dmSd * dmSd;
    const float avgF = (float)(cT + dmTd);
    WindowStatic ws;
    ws.lnAvg = __builtin_amdgcn_logf(avgF) * 0.6931471805599453f;  // avg <= 0: NaN / -inf, the window reads as gated
    float hT, hS;
    {
        const float rn = (float)rnT, dm = (float)dmTd, var = fmaxf((float)varT, 0.0f);
        const float rsd = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var));  // sd = 0: inf, and g says so
        const float dv = (E[1] + 2.0f * fabsf(dm) * E[0]) * rn;
        ws.rsdT = rsd;
        ws.dmT = dm;
        ws.gT = var > 0.0f && 2.0f * dv <= var ? dv * (rsd * rsd) : INFINITY;
        hT = rsd * E[0] * rn;
    }
    {
        const float rn = (float)rnS, dm = (float)dmSd, var = fmaxf((float)varS, 0.0f);
        const float rsd = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(var));
        const float dv = (E[3] + 2.0f * fabsf(dm) * E[2]) * rn;
        ws.rsdS = rsd;
        ws.dmS = dm;
        ws.gS = var > 0.0f && 2.0f * dv <= var ? dv * (rsd * rsd) : INFINITY;
        hS = rsd * E[2] * rn;
    }
    // (a NaN or negative h only occurs with g = +inf, which decides the bound)
    ws.hTS = half_bits_up(hT >= 0.0f ? hT : INFINITY) | (half_bits_up(hS >= 0.0f ? hS : INFINITY) << 16);
    return ws;
}

// relErrT / relErrS: relative error of the cross terms of the two channel groups.  Half-precision values below
// 2^-14 round with an absolute error of up to 2^-25 instead of a relative one: over a whole window that moves a
// normalised cross term by at most 2^-25 (1 / sd_b + sqrt(1 + (dm / sd_b)^2) / sd_a), which is added.
//
// kSpectral: cross terms from the fp32 transforms of the spectral screening instead.  Their error is bounded against
// the norm of the whole transformed block, not of the window: absT / absS = eta |a'| |x_block| / (aStd n) of the
// channel group, which times 1 / sd_b bounds the normalised cross term; relErr then only covers the fp32
// rounding of the target values.
template <bool kSpectral = false>
__device__ __forceinline__ WindowCoeffs window_coeffs(const CorrParams &cp, const WindowStatic &ws, float relErrT,
                                                      float relErrS, float &boostOut, float absT = 0.0f,
                                                      float absS = 0.0f)
{
    constexpr float kSubAbs = 2.9802322e-8f;  // 2^-25
    const float nT = (float)cp.L, nS = (float)cp.L * (float)(cp.numCh - 1);
    const float kLog2e = 1.4426950408889634f;
    const float boost = __builtin_amdgcn_exp2f(((float)cp.lnAvgIn - ws.lnAvg) * (kLog2e / 0.6f));
    boostOut = boost;
    WindowCoeffs wc = {0.0f, 0.0f, 0.0f, 0.0f};  // a gated window: sim = 0 exactly
    if (boost <= cp.maxBoost * 1.00001f) {
        float errT = 0.0f, errS = 0.0f;
        if (cp.w > 0.0f) {
            const float rsd = ws.rsdT, dm = ws.dmT;
            const float r = dm * rsd;
            wc.a = (cp.w / ((float)cp.aStdT * nT)) * rsd;
            wc.c = wc.a * dm * (float)cp.sumAT;
            const float q = __builtin_amdgcn_sqrtf(1.0f + r * r);
            const float eh = kSpectral ? relErrT * q + absT * rsd : relErrT * q + kSubAbs * (rsd + q / (float)cp.aStdT);
            errT = eh + (1.0f + eh) * ws.gT + fabsf((float)cp.sumAT) / ((float)cp.aStdT * nT) * half_from_bits(ws.hTS);
        }
        if (cp.w < 1.0f) {
            const float rsd = ws.rsdS, dm = ws.dmS;
            const float r = dm * rsd;
            wc.b = ((1.0f - cp.w) / ((float)cp.aStdS * nS)) * rsd;
            wc.c += wc.b * dm * (float)cp.sumAS;
            const float q = __builtin_amdgcn_sqrtf(1.0f + r * r);
            const float eh = kSpectral ? relErrS * q + absS * rsd : relErrS * q + kSubAbs * (rsd + q / (float)cp.aStdS);
            errS = eh + (1.0f + eh) * ws.gS + fabsf((float)cp.sumAS) / ((float)cp.aStdS * nS) * half_from_bits(ws.hTS >> 16);
        }
        wc.e = errT * cp.w + errS * (1.0f - cp.w) + 1e-5f;
        if (boost > cp.maxBoost * 0.99999f) wc.e = -wc.e;  // the exact path may gate this window
    }
    return wc;
}

// The spectral kernel streams the statics of every window of a block (it is bound by memory traffic), so it
// keeps them in 20 instead of 32 bytes per window: a 16-byte record (one load; 1 / sd of the two channel groups and
// four half-precision numbers rounded up, g and u = h + |dm| / sd of both groups) and, in an array of its own
// because most launches do not read it, ln of the average loudness.  The mean-offset
// term of the similarity, (w sum a' / (aStd n)) dm / sd, is not evaluated: a' is centred, so sum a' is what
// rounding left of zero, and the term's magnitude goes into the error bound instead (it has the same factor as
// h there); sqrt(1 + (dm / sd)^2) is replaced by its bound 1 + u.
struct SpectralStatics {
    const float4 *rec;   // per window: 1 / sd of the loudness channel, of the others, g | u of both as halves (2 words)
    const float *lnAvg;  // per window: ln of the average loudness
};
__host__ __device__ inline long long spectral_statics_stride(long long windows) { return windows; }
__host__ __device__ inline size_t spectral_statics_floats(long long windows) { return 5 * (size_t)spectral_statics_stride(windows); }
// (floats from the base of the buffer to the ln array)
__host__ __device__ inline long long spectral_statics_ln_offset(long long windows) { return 4 * spectral_statics_stride(windows); }
__device__ __forceinline__ SpectralStatics spectral_statics(float *base, long long windows)
{
    return SpectralStatics{reinterpret_cast<const float4 *>(base), base + spectral_statics_ln_offset(windows)};
}
__device__ __forceinline__ void spectral_statics_store(float *base, long long windows, long long w, const WindowStatic &ws)
{
    float uT = half_from_bits(ws.hTS & 0xffffu) + fabsf(ws.dmT * ws.rsdT);
    float uS = half_from_bits(ws.hTS >> 16) + fabsf(ws.dmS * ws.rsdS);
    if (!(uT >= 0.0f)) uT = INFINITY;  // NaN (0 * inf: a window without spread): g is +inf there and decides
    if (!(uS >= 0.0f)) uS = INFINITY;
    const float gT = ws.gT >= 0.0f ? ws.gT : INFINITY, gS = ws.gS >= 0.0f ? ws.gS : INFINITY;
    reinterpret_cast<float4 *>(base)[w] =
        make_float4(ws.rsdT, ws.rsdS, __uint_as_float(half_bits_up(gT) | (half_bits_up(uT * 1.0001f) << 16)),
                    __uint_as_float(half_bits_up(gS) | (half_bits_up(uS * 1.0001f) << 16)));
    base[spectral_statics_ln_offset(windows) + w] = ws.lnAvg;
}

// The boost of a window from ln of its average loudness, as the screening kernels evaluate it (fp32, hardware
// exp2: within 1e-5 of calcBoost, impl/FeatureCorrelationImpl.scala:73-78).
__device__ __forceinline__ float screen_boost(float lnAvgIn, float lnAvg)
{
    const float kLog2e = 1.4426950408889634f;
    return __builtin_amdgcn_exp2f((lnAvgIn - lnAvg) * (kLog2e / 0.6f));
}
// What the boost gate does to a window's bounds: 0 gated (similarity 0 exactly), 1 open, 2 within 1e-5 of maxBoost
// (the exact path may gate it: the bounds then include 0).
__device__ __forceinline__ int screen_gate(float boost, float maxBoost)
{
    if (!(boost <= maxBoost * 1.00001f)) return 0;
    return boost > maxBoost * 0.99999f ? 2 : 1;
}

// absT / absS: every error of a normalised cross term that scales with 1 / sd of the window (the transforms' share,
// eta |a'| |x_block| / (aStd n), and the rounding of the normalised values at the size of the file's level).
__device__ __forceinline__ WindowCoeffs window_coeffs_spectral(const CorrParams &cp, float rsdT, float rsdS, uint2 gu,
                                                               float relErr, float absT, float absS, int gate)
{
    const float nT = (float)cp.L, nS = (float)cp.L * (float)(cp.numCh - 1);
    WindowCoeffs wc = {0.0f, 0.0f, 0.0f, 0.0f};  // a gated window: sim = 0 exactly
    if (gate) {
        float errT = 0.0f, errS = 0.0f;
        if (cp.w > 0.0f) {
            const float g = half_from_bits(gu.x & 0xffffu), u = half_from_bits(gu.x >> 16);
            wc.a = (cp.w / ((float)cp.aStdT * nT)) * rsdT;
            const float eh = relErr * (1.0f + u) + absT * rsdT;
            errT = eh + (1.0f + eh) * g + fabsf((float)cp.sumAT) / ((float)cp.aStdT * nT) * u * 1.0001f;
        }
        if (cp.w < 1.0f) {
            const float g = half_from_bits(gu.y & 0xffffu), u = half_from_bits(gu.y >> 16);
            wc.b = ((1.0f - cp.w) / ((float)cp.aStdS * nS)) * rsdS;
            const float eh = relErr * (1.0f + u) + absS * rsdS;
            errS = eh + (1.0f + eh) * g + fabsf((float)cp.sumAS) / ((float)cp.aStdS * nS) * u * 1.0001f;
        }
        wc.e = errT * cp.w + errS * (1.0f - cp.w) + 1e-5f;
        if (gate == 2) wc.e = -wc.e;  // the exact path may gate this window
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
