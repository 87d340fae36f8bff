// Device code of the target preparation (readInBuffer, impl/FeatureCorrelationImpl.scala:83-98) shared by the
// kernels of sgz_corr.hip and, where the spectral screening is in use, the kernel of sgz_corr_fft.hip that
// transforms the new target while it is being centred.
//
// Every sum is taken in a fixed order that depends on nothing but the target's shape --
// a thread adds its four elements in index order, the 64 lanes of a wave are folded by a butterfly, the four waves
// of a unit of 1024 elements added in wave order, and the units' sums in unit order (the first pass' by every
// workgroup of the second pass for itself, the second pass' by the host) -- so a target always gets the same
// statistics; they differ from MathUtil.stat's sequential sums (impl/MathUtil.scala:29-62) by a few units of the
// last place of a double.
#pragma once

#include <hip/hip_runtime.h>

#include "sgz_internal.hpp"

namespace sgz {

constexpr int kTpPer = 4;     // elements per thread
constexpr int kTpSums = 10;   // doubles a workgroup of the centring pass hands on

// What the centring pass reports to the host, in pinned memory: every workgroup writes the partial sums of its units
// (kTpSums doubles a unit), the first one also the first pass' sums, and then its own sequence word; the host waits
// for all of them and adds the units' sums in order (sgz_corr_set_target) -- a last workgroup that adds them on the
// device costs the launch a fence, an atomic and two more trips to memory.
struct TargetReportLayout {
    double *sum1;               // [2]
    double *rows;               // [units][kTpSums]
    unsigned long long *flags;  // [workgroups]
};

// The sums of a target as the host puts them together.
struct TargetReport {
    double sum1[2];   // sums of the normalised values: loudness row, other rows
    double sum2[2];   // sums of the squared deviations from the group's mean
    double csum[2];   // sums of the centred values (what rounding leaves of zero)
    double fsum[2];   // sums of the centred values rounded to fp32
    double fsq[2];    // sums of their squares
    double fsq1;      // ... of channel 1 alone
    unsigned long long inRange;  // 1: every centred value lies within the half-precision range
    unsigned long long seq;
};

// Device-side state: four doubles that are not in use, then the partial sums of the first pass' workgroups (kTpSums
// doubles a row, two of them used).  The first pass ends with them: whoever needs the means adds them up
// (tp_first_pass_sums), which costs a few hundred cycles where a last workgroup that adds them for all costs the
// launch a fence, an atomic and a trip to memory (measured: 5 of the kernel's 12 us).
#if defined(__HIPCC__)
template <int T, int K>
__device__ __forceinline__ void tp_wg_sums(double (&v)[K], double *red /* (T / 64) * K doubles */, double *__restrict__ out)
{
#pragma unroll
    for (int j = 0; j < K; j++)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[j] += __shfl_xor(v[j], off, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int j = 0; j < K; j++) red[(threadIdx.x >> 6) * K + j] = v[j];
    __syncthreads();
    if (threadIdx.x < K) {
        double s = 0.0;
        for (int w = 0; w < T / 64; w++) s += red[w * K + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// The first pass' sums (loudness row, other rows) from its workgroups' partial sums, added in workgroup order, for
// every thread of the workgroup: fin2[0..1] (LDS).  rows: T doubles of LDS.
template <int T>
__device__ __forceinline__ void tp_first_pass_sums(const double *__restrict__ state, int normWgs, double *rows, double *fin2)
{
    double s = 0.0;
    for (int w0 = 0; w0 < normWgs; w0 += T / 2) {
        const int nrows = min(T / 2, normWgs - w0);
        if ((int)threadIdx.x < 2 * nrows)
            rows[threadIdx.x] = state[4 + (size_t)(w0 + (threadIdx.x >> 1)) * kTpSums + (threadIdx.x & 1)];
        __syncthreads();
        if (threadIdx.x < 2)
            for (int r = 0; r < nrows; r++) s += rows[2 * r + threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x < 2) fin2[threadIdx.x] = s;
    __syncthreads();
}

// The second pass of MathUtil.stat and the centred copies, by workgroup `wg` of `wgCount` workgroups of T threads:
// elements in the order of the centred copies (q = channel * Lpad + frame, the padding included), so the writes of
// a wave are contiguous and the reads gather from what the first pass left in the cache.
// Whatever T is, the work is cut into units of 1024 elements, a unit per 256 threads, and the sums are formed per
// unit and then over the units in order: the statistics of a target do not depend on the launch that computed them.
constexpr int kTpUnitThreads = 256;
constexpr int kTpUnit = kTpUnitThreads * kTpPer;

template <int T>
__device__ __forceinline__ void target_centre_body(int wg, int wgCount, int L, int numCh, int Lpad,
                                                   const float *__restrict__ tnorm, double *__restrict__ aprime,
                                                   float *__restrict__ aprimeF, double *__restrict__ state,
                                                   int normWgs, int *__restrict__ hflags, int numFiles,
                                                   TargetReportLayout report, unsigned long long seq)
{
    static_assert(T % kTpUnitThreads == 0, "whole units per workgroup");
    constexpr int kSub = T / kTpUnitThreads;  // units per workgroup
    __shared__ double red[(T / 64) * kTpSums];
    __shared__ double rows[T];
    __shared__ double sum1[2];
    const unsigned int n = (unsigned int)Lpad * (unsigned int)numCh;
    const int units = (int)((n + kTpUnit - 1) / kTpUnit);
    const int sub = threadIdx.x / kTpUnitThreads, lt = threadIdx.x % kTpUnitThreads;
    const int unit = wg * kSub + sub;
    const unsigned int q0 = (unsigned int)unit * kTpUnit + lt;
    float x[kTpPer];
#pragma unroll
    for (int k = 0; k < kTpPer; k++) {
        const unsigned int q = q0 + k * kTpUnitThreads;
        const unsigned int c = q / (unsigned int)Lpad, i = q - c * (unsigned int)Lpad;
        x[k] = q < n && i < (unsigned int)L ? tnorm[i * (unsigned int)numCh + c] : 0.0f;
    }
    // (behind the loads above, so that both travel together) MathUtil.stat: mean = sum / n
    tp_first_pass_sums<T>(state, normWgs, rows, sum1);
    const double meanT = sum1[0] / (double)L, meanS = sum1[1] / ((double)L * (double)(numCh - 1));
    // sum2 T/S, csum T/S, fsum T/S, fsq T/S, fsq1, count of values outside the half-precision range
    double acc[kTpSums] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kTpPer; k++) {
        const unsigned int q = q0 + k * kTpUnitThreads;
        if (q < n) {
            const unsigned int c = q / (unsigned int)Lpad, i = q - c * (unsigned int)Lpad;
            if (i < (unsigned int)L) {
                const int g = c > 0;
                const double v = (double)x[k] - (g ? meanS : meanT);  // ca(i) + aAdd, impl/MathUtil.scala:190
                aprime[q] = v;
                acc[g] += v * v;
                acc[2 + g] += v;
                if (!(fabs(v) <= 60000.0)) acc[9] += 1.0;  // (NaN counts as outside)
                const float vf = (float)v;
                aprimeF[q] = vf;
                const double w = (double)vf;
                acc[4 + g] += w;
                acc[6 + g] += w * w;
                if (c == 1) acc[8] += w * w;
            } else {  // zero padding behind the target's last frame
                aprime[q] = 0.0;
                aprimeF[q] = 0.0f;
            }
        }
    }
    // the marks the screening kernels left per file for the previous target (bit 2; the range marks of the
    // database's image, bit 1, stay)
    for (long long f = (long long)wg * T + threadIdx.x; f < numFiles; f += (long long)wgCount * T) hflags[f] &= 1;
    // the unit's sums: lanes of a wave by a butterfly, the unit's four waves in order
#pragma unroll
    for (int j = 0; j < kTpSums; j++)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[j] += __shfl_xor(acc[j], off, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int j = 0; j < kTpSums; j++) red[(threadIdx.x >> 6) * kTpSums + j] = acc[j];
    __syncthreads();
    if (lt < kTpSums && unit < units) {
        double s = 0.0;
        for (int w = 0; w < kTpUnitThreads / 64; w++) s += red[(sub * (kTpUnitThreads / 64) + w) * kTpSums + lt];
        report.rows[(size_t)unit * kTpSums + lt] = s;
        __threadfence_system();
    }
    if (wg == 0 && threadIdx.x < 2) {
        report.sum1[threadIdx.x] = sum1[threadIdx.x];
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(report.flags + wg, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

// Arguments of the centring pass, for the launches that carry it along (sgz_corr_fft.hip).
struct TargetCentreArgs {
    int L, numCh, Lpad;
    const float *tnorm;
    double *aprime;
    float *aprimeF;
    double *state;
    int normWgs;     // workgroups of the first pass (rows of partial sums it left)
    int *hflags;
    int numFiles;
    TargetReportLayout report;
    unsigned long long seq;
};

}  // namespace sgz
