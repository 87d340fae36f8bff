// FeatureSegmentation and SelfSimilarity on gfx950: both are built on
// MathUtil.correlateHalf (impl/MathUtil.scala:80-99): joint mean / std-dev over a
// window of 2H frames, then the normalised product sum of its two halves.
//
// The window kernels keep the reference's summation order (channel-major, ring
// slot order for the statistics, frame order for the cross term), plain
// multiply-then-add in fp64, so their results agree with a sequential CPU
// evaluation to the last bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "sgz_internal.hpp"

using namespace sgz;

namespace {

// raw interleaved [frames][numCh] -> normalised planar fp32 [numCh][frames]
// (MathUtil.normalize, impl/MathUtil.scala:132-152: (f - min) / (max - min) in fp32)
__global__ __launch_bounds__(256) void planar_norm_kernel(const float *__restrict__ raw, long long frames,
                                                          int numCh, NormParams nrm, float *__restrict__ out)
{
    const long long total = frames * numCh;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long fr = e / numCh;
        const int c = (int)(e - fr * numCh);
        float v = raw[e];
        if (nrm.enabled) v = (v - nrm.mn[c]) / nrm.d[c];
        out[(long long)c * frames + fr] = v;
    }
}

// correlateHalf over channels [c0, c1) of a window made of H frames starting at
// `left` in planar matrix xl and H frames starting at `right` in xr.  `rot` is
// the ring rotation (frameOff of the reference) that fixes the order in which
// MathUtil.stat walks the 2H slots.
__device__ float correlate_half_dev(const float *__restrict__ xl, long long strideL, long long left,
                                    const float *__restrict__ xr, long long strideR, long long right, int H,
                                    int rot, int c0, int c1)
{
    const int W = 2 * H;
    auto slot = [&](int c, int k) -> double {
        // k-th frame of the window in time order
        return k < H ? (double)xl[(long long)c * strideL + left + k]
                     : (double)xr[(long long)c * strideR + right + (k - H)];
    };
    double sum = 0.0;
    for (int c = c0; c < c1; c++) {
        for (int i = 0; i < W; i++) {
            int k = i - rot;
            if (k < 0) k += W;
            sum += slot(c, k);
        }
    }
    const int matSizeAll = W * (c1 - c0);
    const double mean = sum / matSizeAll;
    sum = 0.0;
    for (int c = c0; c < c1; c++) {
        for (int i = 0; i < W; i++) {
            int k = i - rot;
            if (k < 0) k += W;
            const double d = slot(c, k) - mean;
            sum += d * d;
        }
    }
    const double stdDev = sqrt(sum / matSizeAll);
    const double add = -mean;
    const int matSize = (c1 - c0) * H;
    sum = 0.0;
    for (int c = c0; c < c1; c++) {
        for (int k = 0; k < H; k++) {
            sum += (slot(c, k) + add) * (slot(c, k + H) + add);
        }
    }
    return (float)(sum / (stdDev * stdDev * matSize));
}

// impl/FeatureSegmentationImpl.scala:107-120: window o covers frames
// afStart + o .. afStart + o + 2H - 1; the ring rotation is o mod 2H.
__global__ __launch_bounds__(256) void segm_kernel(const float *__restrict__ x, long long frames, int numCh,
                                                   int afStart, long long numWin, int H, float w,
                                                   float *__restrict__ simOut)
{
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= numWin) return;
    const int rot = (int)(o % (2 * H));
    const long long left = afStart + o;
    float temporal = 0.0f, spectral = 0.0f;
    if (w > 0.0f) temporal = correlate_half_dev(x, frames, left, x, frames, left + H, H, rot, 0, 1);
    if (w < 1.0f) spectral = correlate_half_dev(x, frames, left, x, frames, left + H, H, rot, 1, numCh);
    simOut[o] = temporal * w + spectral * (1.0f - w);
}

// Gray colour functions of impl/SelfSimilarityImpl.scala:99-107 on v = pow(max(0, sim), warp) * scale (:150).
__device__ int gray_color(float sim, double warp, float scale, int inv)
{
    const float m0 = (sim != sim) ? sim : (sim > 0.0f ? sim : 0.0f);
    const float v = (float)pow((double)m0, warp) * scale;
    const float s = inv ? (1.0f - v) : v;
    const double d = (double)(s * 255) + 0.5;
    int i;
    if (d != d) i = 0;
    else if (d >= 255.0) i = 255;
    else if (d <= 0.0) i = 0;
    else i = (int)d;
    return (i << 16) | (i << 8) | i;
}

// impl/SelfSimilarityImpl.scala:127-163: pixel (x, y), y >= x, compares the half
// window of file 1 at afStart + x*decim with that of file 2 at afStart + y*decim.
__global__ __launch_bounds__(256) void selfsim_kernel(const float *__restrict__ x1, long long frames1,
                                                      const float *__restrict__ x2, long long frames2, int numCh,
                                                      int afStart, int H, int decim, int imgExt, float w,
                                                      double warp, float scale, int inv,
                                                      float *__restrict__ simOut, int *__restrict__ imgOut)
{
    const int px = blockIdx.x * 16 + (threadIdx.x & 15);
    const int py = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (px >= imgExt || py >= imgExt || py < px) return;
    const long long left = (long long)afStart + (long long)px * decim;
    const long long right = (long long)afStart + (long long)py * decim;
    float temporal = 0.0f, spectral = 0.0f;
    if (w > 0.0f) temporal = correlate_half_dev(x1, frames1, left, x2, frames2, right, H, 0, 0, 1);
    if (w < 1.0f) spectral = correlate_half_dev(x1, frames1, left, x2, frames2, right, H, 0, 1, numCh);
    const float sim = temporal * w + spectral * (1.0f - w);
    if (simOut) {
        simOut[(long long)px * imgExt + py] = sim;
        simOut[(long long)py * imgExt + px] = sim;
    }
    if (imgOut) {
        const int colr = gray_color(sim, warp, scale, inv);
        imgOut[(long long)(imgExt - 1 - py) * imgExt + px] = colr;
        imgOut[(long long)(imgExt - 1 - px) * imgExt + py] = colr;
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

extern "C" {

static NormParams make_norm(const float *norms, int numCh)
{
    NormParams n;
    std::memset(&n, 0, sizeof n);
    if (norms) {
        n.enabled = 1;
        for (int c = 0; c < numCh; c++) {
            n.mn[c] = norms[2 * c];
            n.d[c] = norms[2 * c + 1] - norms[2 * c];
        }
    }
    return n;
}

int sgz_segm_curve(sgz_context *ctx, const float *frames, int64_t num_frames, int num_channels, const float *norms,
                   int af_start, int af_stop, int half_win, float temporal_weight, float *sim_out,
                   int64_t *num_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (!frames || num_frames < 0 || num_channels < 2 || num_channels > SGZ_MAX_CHANNELS)
        return ctx->fail(SGZ_E_ARG, "bad frames / channels");
    if (half_win < 1 || af_start < 0 || af_stop > num_frames || !num_out)
        return ctx->fail(SGZ_E_ARG, "bad window / span");
    const NormParams nrm = make_norm(norms, num_channels);
    const long long afLen = (long long)af_stop - af_start;
    // the reference loop runs once per frame after the first 2H have been read
    // (impl/FeatureSegmentationImpl.scala:101-129)
    if (afLen > 0 && afLen < 2ll * half_win) {
        // a span shorter than the window still gives one iteration: its frames followed by the zeros of the
        // fresh buffer, which normalisation never touches (:86, :107-113) -- here raw values that normalise to 0
        const size_t win = 2 * (size_t)half_win;
        std::vector<float> padded(win * num_channels);
        std::memcpy(padded.data(), frames + (size_t)af_start * num_channels, sizeof(float) * (size_t)afLen * num_channels);
        for (size_t i = (size_t)afLen; i < win; i++)
            for (int c = 0; c < num_channels; c++) padded[i * num_channels + c] = norms ? norms[2 * c] : 0.0f;
        return sgz_segm_curve(ctx, padded.data(), (int64_t)win, num_channels, norms, 0, (int)win, half_win,
                              temporal_weight, sim_out, num_out);
    }
    const long long numWin = afLen >= 2ll * half_win ? afLen - 2ll * half_win + 1 : 0;
    *num_out = numWin;
    if (numWin == 0) return SGZ_OK;
    if (!sim_out) return ctx->fail(SGZ_E_ARG, "sim_out is null");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf raw, planar, sim;
    const size_t elems = (size_t)num_frames * num_channels;
    SGZ_HIP(ctx, hipMalloc(&raw.p, elems * sizeof(float)));
    SGZ_HIP(ctx, hipMalloc(&planar.p, elems * sizeof(float)));
    SGZ_HIP(ctx, hipMalloc(&sim.p, (size_t)numWin * sizeof(float)));
    SGZ_HIP(ctx, hipMemcpyAsync(raw.p, frames, elems * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const int g1 = (int)std::min<size_t>((elems + 255) / 256, 16384);
    hipLaunchKernelGGL(planar_norm_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const float *)raw.p,
                       (long long)num_frames, num_channels, nrm, (float *)planar.p);
    SGZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(segm_kernel, dim3((unsigned)((numWin + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float *)planar.p, (long long)num_frames, num_channels, af_start, numWin, half_win,
                       temporal_weight, (float *)sim.p);
    SGZ_HIP(ctx, hipGetLastError());
    SGZ_HIP(ctx, hipMemcpyAsync(sim_out, sim.p, (size_t)numWin * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SGZ_OK;
}

int sgz_selfsim_run(sgz_context *ctx, const float *frames1, int64_t num_frames1, const float *frames2,
                    int64_t num_frames2, int num_channels, const float *norms, int af_start, int half_win,
                    int decim, int img_ext,
                    float temporal_weight, int colors, float color_warp, float color_ceil, int color_inv,
                    float *sim_out, int32_t *img_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (!frames1 || !frames2 || num_frames1 < 0 || num_frames2 < 0 || num_channels < 2 ||
        num_channels > SGZ_MAX_CHANNELS)
        return ctx->fail(SGZ_E_ARG, "bad frames / channels");
    if (half_win < 1 || decim < 1 || img_ext < 0 || af_start < 0) return ctx->fail(SGZ_E_ARG, "bad geometry");
    if (img_out && colors != 0)
        return ctx->fail(SGZ_E_ARG, "only the gray colour scheme (0) is implemented");
    if (!(color_warp > 0) || !(color_ceil > 0))  // impl/SelfSimilarityImpl.scala:112-113
        return ctx->fail(SGZ_E_ARG, "color warp / ceil must be > 0");
    const NormParams nrm = make_norm(norms, num_channels);
    if (img_ext == 0) return SGZ_OK;
    const long long lastStart = (long long)af_start + (long long)(img_ext - 1) * decim;
    if (lastStart + half_win > num_frames1 || lastStart + half_win > num_frames2)
        return ctx->fail(SGZ_E_ARG, "windows run past the end of the input");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    const bool same = frames1 == frames2 && num_frames1 == num_frames2;
    DevBuf raw, p1, p2, simd, imgd;
    const size_t e1 = (size_t)num_frames1 * num_channels, e2 = (size_t)num_frames2 * num_channels;
    SGZ_HIP(ctx, hipMalloc(&raw.p, std::max(e1, e2) * sizeof(float)));
    SGZ_HIP(ctx, hipMalloc(&p1.p, e1 * sizeof(float)));
    SGZ_HIP(ctx, hipMemcpyAsync(raw.p, frames1, e1 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(planar_norm_kernel, dim3((unsigned)std::min<size_t>((e1 + 255) / 256, 16384)), dim3(256), 0,
                       ctx->stream, (const float *)raw.p, (long long)num_frames1, num_channels, nrm,
                       (float *)p1.p);
    SGZ_HIP(ctx, hipGetLastError());
    if (!same) {
        SGZ_HIP(ctx, hipMalloc(&p2.p, e2 * sizeof(float)));
        SGZ_HIP(ctx, hipMemcpyAsync(raw.p, frames2, e2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(planar_norm_kernel, dim3((unsigned)std::min<size_t>((e2 + 255) / 256, 16384)), dim3(256),
                           0, ctx->stream, (const float *)raw.p, (long long)num_frames2, num_channels, nrm,
                           (float *)p2.p);
        SGZ_HIP(ctx, hipGetLastError());
    }
    const size_t pix = (size_t)img_ext * img_ext;
    if (sim_out) SGZ_HIP(ctx, hipMalloc(&simd.p, pix * sizeof(float)));
    if (img_out) SGZ_HIP(ctx, hipMalloc(&imgd.p, pix * sizeof(int32_t)));
    const unsigned gb = (unsigned)((img_ext + 15) / 16);
    hipLaunchKernelGGL(selfsim_kernel, dim3(gb, gb), dim3(256), 0, ctx->stream, (const float *)p1.p,
                       (long long)num_frames1, (const float *)(same ? p1.p : p2.p), (long long)num_frames2,
                       num_channels, af_start, half_win, decim, img_ext, temporal_weight, (double)color_warp,
                       1.0f / color_ceil, color_inv, (float *)simd.p, (int *)imgd.p);
    SGZ_HIP(ctx, hipGetLastError());
    if (sim_out)
        SGZ_HIP(ctx, hipMemcpyAsync(sim_out, simd.p, pix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (img_out)
        SGZ_HIP(ctx, hipMemcpyAsync(img_out, imgd.p, pix * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SGZ_OK;
}

}  // extern "C"
