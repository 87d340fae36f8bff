// FeatureStats on the GPU: per feature file and channel the 1st and the 99th
// percentile from a 2048-bin, power-law warped histogram
// (impl/FeatureStatsImpl.scala:56-140), merged over the files (:31-54).
//
// Two kernels, one workgroup per file each:
//   stats_range_kernel  min, max (fp32) and sum (fp64) of every channel (first pass, :72-83)
//   stats_hist_kernel   skew from the mean (:86-92), LDS histograms (:98-114), the two
//                       percentile scans (:117-134)
// Thread t of the first (1024 / numCh) * numCh threads owns channel t % numCh:
// consecutive threads read consecutive values of the interleaved file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "sgz_internal.hpp"

using namespace sgz;

namespace {

constexpr int kStThreads = 1024;
constexpr int kBins = 2048;

struct ChanRange {
    float mn, mx;
    double sum;
};

__global__ __launch_bounds__(kStThreads) void stats_range_kernel(const FileDesc *__restrict__ files, int fileBegin,
                                                                int numCh, ChanRange *__restrict__ out)
{
    __shared__ float smn[kStThreads], smx[kStThreads];
    __shared__ double ssum[kStThreads];
    const FileDesc fd = files[fileBegin + blockIdx.x];
    const int tid = threadIdx.x;
    const int rows = kStThreads / numCh, stride = rows * numCh;
    const long long total = fd.frames * numCh;
    float mn = INFINITY, mx = -INFINITY;
    double sum = 0.0;
    if (tid < stride) {
        for (long long e = tid; e < total; e += stride) {
            const float f = fd.raw[e];
            if (f < mn) mn = f;
            if (f > mx) mx = f;
            sum += (double)f;
        }
    }
    smn[tid] = mn;
    smx[tid] = mx;
    ssum[tid] = sum;
    __syncthreads();
    if (tid < numCh) {  // fixed order: reproducible sums
        for (int k = 1; k < rows; k++) {
            const int t = tid + k * numCh;
            if (smn[t] < mn) mn = smn[t];
            if (smx[t] > mx) mx = smx[t];
            sum += ssum[t];
        }
        ChanRange r;
        r.mn = mn;
        r.mx = mx;
        r.sum = sum;
        out[(long long)blockIdx.x * numCh + tid] = r;
    }
}

// chGroup channels per pass share the LDS histogram (chGroup * 2048 ints).
__global__ __launch_bounds__(kStThreads) void stats_hist_kernel(const FileDesc *__restrict__ files, int fileBegin,
                                                               int numCh, int chGroup,
                                                               const ChanRange *__restrict__ ranges,
                                                               double *__restrict__ out /* [file][numCh][2] */)
{
    extern __shared__ int hist[];
    const FileDesc fd = files[fileBegin + blockIdx.x];
    const int tid = threadIdx.x;
    const int rows = kStThreads / numCh, stride = rows * numCh;
    const long long total = fd.frames * numCh;
    const int c = tid % numCh;
    const ChanRange cr = ranges[(long long)blockIdx.x * numCh + c];
    const float mn = cr.mn;
    const float d = cr.mx - cr.mn;                                  // Float, :89
    const double mean = cr.sum / (double)fd.frames;                 // :88
    const double skew = log(0.5) / log((mean - (double)mn) / (double)d);  // :90-91
    for (int c0 = 0; c0 < numCh; c0 += chGroup) {
        for (int k = tid; k < chGroup * kBins; k += kStThreads) hist[k] = 0;
        __syncthreads();
        if (tid < stride && c >= c0 && c < c0 + chGroup) {
            int *h = hist + (c - c0) * kBins;
            for (long long e = tid; e < total; e += stride) {
                const float q = (fd.raw[e] - mn) / d;                // Float arithmetic, :110
                const int bin = (int)(pow((double)q, skew) * 2047 + 0.5);  // NaN -> 0, as the JVM's toInt
                if (bin >= 0 && bin < kBins) atomicAdd(&h[bin], 1);
            }
        }
        __syncthreads();
        if (tid >= c0 && tid < min(c0 + chGroup, numCh)) {  // tid == its own channel here
            const int *h = hist + (tid - c0) * kBins;
            const int p01n = (int)((double)fd.frames * 0.01);  // :119-120
            const int p99n = (int)((double)fd.frames * 0.99);
            const double skewr = 1.0 / skew;
            int cnt = 0, i = 0;
            while (cnt < p01n && i < kBins) cnt += h[i++];
            const double p01 = pow((double)i / 2048, skewr) * (double)d + (double)mn;
            while (cnt < p99n && i < kBins) cnt += h[i++];
            const double p99 = pow((double)i / 2048, skewr) * (double)d + (double)mn;
            out[((long long)blockIdx.x * numCh + tid) * 2] = p01;
            out[((long long)blockIdx.x * numCh + tid) * 2 + 1] = p99;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int sgz_feature_stats(sgz_context *ctx, int file_begin, int file_end, double *minmax_out,
                                 double *per_file_out)
{
    if (!ctx) return SGZ_E_ARG;
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin >= file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    if (!minmax_out) return ctx->fail(SGZ_E_ARG, "minmax_out is NULL");
    const int numCh = ctx->numCh, n = file_end - file_begin;
    for (int f = file_begin; f < file_end; f++)
        if (ctx->files[f].frames <= 0) return ctx->fail(SGZ_E_ARG, "feature file without frames");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    // file table (only raw / frames are read)
    std::vector<FileDesc> fds((size_t)n, FileDesc{});
    for (int k = 0; k < n; k++) {
        fds[(size_t)k].raw = ctx->files[(size_t)(file_begin + k)].d_raw;
        fds[(size_t)k].frames = ctx->files[(size_t)(file_begin + k)].frames;
    }
    FileDesc *d_fds = nullptr;
    ChanRange *d_rng = nullptr;
    double *d_out = nullptr;
    auto cleanup = [&]() {
        if (d_fds) (void)hipFree(d_fds);
        if (d_rng) (void)hipFree(d_rng);
        if (d_out) (void)hipFree(d_out);
    };
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_fds), sizeof(FileDesc) * n);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_rng), sizeof(ChanRange) * (size_t)n * numCh);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(double) * (size_t)n * numCh * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(d_fds, fds.data(), sizeof(FileDesc) * n, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return ctx->fail(SGZ_E_HIP, hipGetErrorString(e));
    }
    const int chGroup = std::max(1, std::min(numCh, (144 * 1024) / (kBins * (int)sizeof(int))));
    const size_t ldsBytes = (size_t)chGroup * kBins * sizeof(int);
    hipLaunchKernelGGL(stats_range_kernel, dim3(n), dim3(kStThreads), 0, ctx->stream, d_fds, 0, numCh, d_rng);
    e = hipGetLastError();
    if (e == hipSuccess && ldsBytes > 48 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(stats_hist_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(stats_hist_kernel, dim3(n), dim3(kStThreads), ldsBytes, ctx->stream, d_fds, 0, numCh, chGroup,
                           d_rng, d_out);
        e = hipGetLastError();
    }
    std::vector<double> per((size_t)n * numCh * 2);
    if (e == hipSuccess)
        e = hipMemcpyAsync(per.data(), d_out, sizeof(double) * per.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return ctx->fail(SGZ_E_HIP, hipGetErrorString(e));
    // :39-49 -- math.min / math.max over the files (NaN propagates)
    for (int ch = 0; ch < numCh; ch++) {
        double lo = per[(size_t)ch * 2], hi = per[(size_t)ch * 2 + 1];
        for (int k = 1; k < n; k++) {
            const double a = per[((size_t)k * numCh + ch) * 2], b = per[((size_t)k * numCh + ch) * 2 + 1];
            lo = (lo != lo || a != a) ? NAN : std::min(lo, a);
            hi = (hi != hi || b != b) ? NAN : std::max(hi, b);
        }
        minmax_out[2 * ch] = lo;
        minmax_out[2 * ch + 1] = hi;
    }
    if (per_file_out) std::copy(per.begin(), per.end(), per_file_out);
    return SGZ_OK;
}
