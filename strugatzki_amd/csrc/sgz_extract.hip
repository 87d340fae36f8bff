// Feature extraction on gfx950: windowed STFT -> 42 mel bands -> dB -> DCT-II
// (MFCC-like coefficients) + A-weighted loudness in sones.
//
// The reference renders these features with SuperCollider's scsynth
// (impl/FeatureExtractionImpl.scala:40-55, impl/NonRealtimeProcessor.scala:77-117);
// what is kept from it is the framing: channel mix-down (Mix = sum, First, Last),
// Hann window, hop = fftSize / fftOverlap, channel order [loudness / 32, mfcc 0..C-1],
// ceil(S / step) - 1 output frames, output frame i showing the FFT frame that
// completes at time (i + 1) * step (zeros before the first one has completed).
// The spectral model (band edges, dB floor, DCT scaling, loudness weighting) is
// defined here and stated in fp64 by the test
// oracle; scsynth's own tables are not available.
//
// One wave computes one output frame: the fftSize real samples are packed as
// fftSize/2 complex points in LDS, transformed with an in-place radix-2 FFT,
// unpacked to the real spectrum, reduced to band energies (one lane per band,
// fixed summation order), then DCT / loudness with wave shuffles.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "sgz_internal.hpp"

namespace {

constexpr int kBands = 42;
constexpr int kMaxFft = 8192;
constexpr int kMaxFramesPerBlock = 8;  // waves (= frames) per workgroup; fewer for large FFT sizes

struct ExtractParams {
    long long numFrames;   // input sample frames
    long long numOut;      // output feature frames
    int numChannels;
    int channels;          // 0 mix, 1 first, 2 last
    int fftSize, logM;     // M = fftSize / 2 complex points, logM = log2(M)
    int step, overlap;
    int numCoeffs;
    float norm;            // 1 / (fftSize / 4)^2
};

struct BandTable {
    int b0[kBands], cnt[kBands];  // first bin and number of bins of the band's triangle
    float aw[kBands];             // A-weighting at the band centre, dB
    int maxCnt;                   // row length of the weight table
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(64 * kMaxFramesPerBlock) void extract_kernel(const float *__restrict__ pcm,
                                                                       const float2 *__restrict__ twiddle,  // e^{-2 pi i k / fftSize}, k < fftSize/2
                                                                       const float *__restrict__ hann,
                                                                       const float *__restrict__ dct,       // [numCoeffs][kBands]
                                                                       const float *__restrict__ bandW,     // [maxCnt][kBands] triangle weights
                                                                       BandTable bt, ExtractParams p,
                                                                       float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // frames per block x (M + 8), then M twiddles
    const int kFramesPerBlock = blockDim.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int M = p.fftSize >> 1;
    float2 *z = lds + wv * (M + 8);
    float2 *tw = lds + kFramesPerBlock * (M + 8);  // the twiddle table, shared by the workgroup's waves
    for (int n = threadIdx.x; n < M; n += 64 * kFramesPerBlock) tw[n] = twiddle[n];
    float *dctL = reinterpret_cast<float *>(tw + M);  // the DCT table [numCoeffs][kBands]
    for (int n = threadIdx.x; n < p.numCoeffs * kBands; n += 64 * kFramesPerBlock) dctL[n] = dct[n];
    // a wave works on its own frame: between the steps below only its own lanes exchange data through LDS
    auto waveSync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const long long i = (long long)blockIdx.x * kFramesPerBlock + wv;  // output frame
    const bool live = i < p.numOut;
    const long long k = i + 1;
    const long long s0 = (k - p.overlap) * p.step;  // FFT frame j = k - overlap ends exactly at time k * step
    const bool analysed = live && k >= p.overlap && s0 + p.fftSize <= p.numFrames;
    const int numCh = p.numCoeffs + 1;

    // 1. load + mix down + window, bit-reversed complex packing z[rev(n)] = x[2n] + i x[2n+1]
    if (analysed) {
#pragma unroll 8
        for (int n = lane; n < M; n += 64) {
            float xv[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float *fr = pcm + (s0 + 2 * n + h) * p.numChannels;
                float x;
                if (p.channels == 1) x = fr[0];
                else if (p.channels == 2) x = fr[p.numChannels - 1];
                else {
                    x = 0.0f;
                    for (int c = 0; c < p.numChannels; c++) x += fr[c];
                }
                xv[h] = x * hann[2 * n + h];
            }
            const int r = (int)(__brev((unsigned)n) >> (32 - p.logM));
            z[r] = make_float2(xv[0], xv[1]);
        }
    }
    __syncthreads();  // (the twiddle table is complete)
    // 2. radix-2 decimation-in-time stages over M complex points
    for (int s = 1; s <= p.logM; s++) {
        const int half = 1 << (s - 1);
        const int tstride = p.fftSize >> s;  // twiddle index stride: w_len^j = w_fftSize^(j * fftSize / len)
        if (analysed) {
#pragma unroll 4
            for (int b = lane; b < (M >> 1); b += 64) {
                const int j = b & (half - 1);
                const int a = ((b >> (s - 1)) << s) + j;
                const float2 w = tw[j * tstride];
                const float2 u = z[a];
                const float2 t = cmul(z[a + half], w);
                z[a] = make_float2(u.x + t.x, u.y + t.y);
                z[a + half] = make_float2(u.x - t.x, u.y - t.y);
            }
        }
        waveSync();
    }
    // 3. unpack to the spectrum of the real signal, keep |X[k]|^2 * norm in place (as .x), k = 0..M
    //    X[k] = (Z[k] + conj Z[M-k]) / 2 - i w^k (Z[k] - conj Z[M-k]) / 2
    if (analysed) {
        for (int b = lane; b <= (M >> 1); b += 64) {
            const int kk = b, mk = M - b;
            const float2 zk = z[kk & (M - 1)], zm = z[mk & (M - 1)];
            // bin kk
            {
                const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                const float2 w = kk < M ? tw[kk] : make_float2(-1.0f, 0.0f);
                const float2 t = cmul(o, w);  // w^k * o ; X = e - i t
                const float xr = e.x + t.y, xi = e.y - t.x;
                const float pk = (xr * xr + xi * xi) * p.norm;
                // bin M - kk from the same pair
                const float2 e2 = make_float2(e.x, -e.y);
                const float2 o2 = make_float2(-o.x, o.y);
                const float2 w2 = mk < M ? tw[mk] : make_float2(-1.0f, 0.0f);
                const float2 t2 = cmul(o2, w2);
                const float yr = e2.x + t2.y, yi = e2.y - t2.x;
                const float pm = (yr * yr + yi * yi) * p.norm;
                __builtin_amdgcn_wave_barrier();
                if (kk == 0) {
                    // Z[M] == Z[0]: bins 0 and M both come from z[0]
                    z[0].x = pk;
                    z[M].x = pm;
                } else {
                    z[kk].x = pk;
                    z[mk].x = pm;
                }
            }
        }
    }
    waveSync();
    // 4. band energies -> dB, loudness
    float db = 0.0f, sone = 0.0f;
    if (analysed && lane < kBands) {
        const int b0 = bt.b0[lane], cnt = bt.cnt[lane];
        float e = 0.0f;
#pragma unroll 4
        for (int t = 0; t < cnt; t++) {
            const float w = bandW[t * kBands + lane], x = z[b0 + t].x;
            e += w > 0.0f ? w * x : 0.0f;  // (adding 0.0f leaves e >= 0 as it is)
        }
        db = 100.0f + 10.0f * log10f(e + 1e-10f);
        if (db < 0.0f) db = 0.0f;
        float phon = db + bt.aw[lane];
        if (phon < 0.0f) phon = 0.0f;
        sone = exp2f((phon - 40.0f) * 0.1f);
    }
    // loudness: fixed-order tree over the wave
    float loud = sone;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) loud += __shfl_xor(loud, off, 64);
    // 5. DCT-II: lane c sums over the 42 bands in order
    float coef = 0.0f;
    for (int j = 0; j < kBands; j++) {
        const float dj = __shfl(db, j, 64);
        if (lane < p.numCoeffs) coef += dctL[lane * kBands + j] * dj;
    }
    if (live) {
        float *o = out + i * numCh;
        if (analysed) {
            if (lane == 0) o[0] = loud * (1.0f / 32.0f);
            if (lane < p.numCoeffs) o[1 + lane] = 0.5f + 0.5f * (coef * (1.0f / kBands)) * 0.01f;
        } else if (lane < numCh) {
            o[lane] = 0.0f;  // no FFT frame has completed yet
        }
    }
}

double melOfHz(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }
double hzOfMel(double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); }
double aWeightDb(double f)
{
    const double f2 = f * f;
    const double num = 12194.0 * 12194.0 * f2 * f2;
    const double den = (f2 + 20.6 * 20.6) * std::sqrt((f2 + 107.7 * 107.7) * (f2 + 737.9 * 737.9)) * (f2 + 12194.0 * 12194.0);
    if (den == 0.0 || num == 0.0) return -100.0;
    double a = 20.0 * std::log10(num / den) + 2.0;
    return a < -100.0 ? -100.0 : a;
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

int64_t sgz_extract_num_frames(int64_t num_frames, int fft_size, int fft_overlap)
{
    if (num_frames < 0 || fft_overlap < 1 || fft_size < fft_overlap) return -1;
    const int step = fft_size / fft_overlap;
    const int64_t outFrames = (num_frames + step - 1) / step;  // impl/NonRealtimeProcessor.scala:93
    return outFrames > 0 ? outFrames - 1 : 0;                   // first frame is dropped, :107-108
}

int sgz_extract(sgz_context *ctx, const float *pcm, int64_t num_frames, int num_channels, double sample_rate,
                int fft_size, int fft_overlap, int num_coeffs, int channels_behavior, float *features_out,
                int64_t capacity_frames, int64_t *num_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (num_frames < 0 || (num_frames > 0 && !pcm) || num_channels < 1 || !(sample_rate > 0))
        return ctx->fail(SGZ_E_ARG, "bad audio input");
    if (fft_size < 16 || fft_size > kMaxFft || (fft_size & (fft_size - 1)) || fft_overlap < 1 || fft_size % fft_overlap)
        return ctx->fail(SGZ_E_ARG, "fftSize must be a power of two in [16, 8192] and a multiple of fftOverlap");
    if (num_coeffs < 1 || num_coeffs > kBands) return ctx->fail(SGZ_E_ARG, "numCoeffs must be in [1, 42]");
    if (channels_behavior < 0 || channels_behavior > 2) return ctx->fail(SGZ_E_ARG, "bad channels behaviour");
    const int64_t nOut = sgz_extract_num_frames(num_frames, fft_size, fft_overlap);
    if (num_out) *num_out = nOut;
    if (nOut == 0) return SGZ_OK;
    if (!features_out || capacity_frames < nOut) return ctx->fail(SGZ_E_OVERFLOW, "feature buffer too small");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));

    const int M = fft_size / 2;
    ExtractParams p;
    p.numFrames = num_frames;
    p.numOut = nOut;
    p.numChannels = num_channels;
    p.channels = channels_behavior;
    p.fftSize = fft_size;
    p.logM = 0;
    while ((1 << p.logM) < M) p.logM++;
    p.step = fft_size / fft_overlap;
    p.overlap = fft_overlap;
    p.numCoeffs = num_coeffs;
    p.norm = (float)(1.0 / ((double)fft_size * 0.25 * (double)fft_size * 0.25));

    // tables (fp64 on the host, rounded once)
    std::vector<float2> tw((size_t)M);
    std::vector<float> hann((size_t)fft_size), dct((size_t)num_coeffs * kBands);
    for (int k = 0; k < M; k++) {
        const double a = -2.0 * M_PI * k / fft_size;
        tw[(size_t)k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int n = 0; n < fft_size; n++) hann[(size_t)n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / fft_size));
    for (int c = 0; c < num_coeffs; c++)
        for (int j = 0; j < kBands; j++) dct[(size_t)c * kBands + j] = (float)std::cos(M_PI / kBands * (j + 0.5) * c);
    BandTable bt;
    std::vector<float> bandW;
    {
        const double mHi = melOfHz(sample_rate * 0.5);
        double e[kBands + 2];
        for (int j = 0; j < kBands + 2; j++) e[j] = hzOfMel(mHi * j / (kBands + 1)) * fft_size / sample_rate;
        bt.maxCnt = 1;
        for (int j = 0; j < kBands; j++) {
            const float e0 = (float)e[j], e2 = (float)e[j + 2];
            bt.b0[j] = (int)std::ceil(e0);
            const int b1 = std::min((int)std::floor(e2), M);
            bt.cnt[j] = std::max(b1 - bt.b0[j] + 1, 0);
            bt.maxCnt = std::max(bt.maxCnt, bt.cnt[j]);
            bt.aw[j] = (float)aWeightDb(e[j + 1] * sample_rate / fft_size);
        }
        // triangle weights in fp32, bin by bin
        bandW.assign((size_t)bt.maxCnt * kBands, 0.0f);
        for (int j = 0; j < kBands; j++) {
            const float e0 = (float)e[j], e1 = (float)e[j + 1], e2 = (float)e[j + 2];
            for (int t = 0; t < bt.cnt[j]; t++) {
                const float fb = (float)(bt.b0[j] + t);
                bandW[(size_t)t * kBands + j] =
                    fb <= e1 ? (e1 > e0 ? (fb - e0) / (e1 - e0) : 1.0f) : (e2 > e1 ? (e2 - fb) / (e2 - e1) : 0.0f);
            }
        }
    }

    DevBuf dPcm, dTw, dHann, dDct, dOut, dBandW;
    const size_t pcmBytes = sizeof(float) * (size_t)num_frames * num_channels;
    const size_t outBytes = sizeof(float) * (size_t)nOut * (num_coeffs + 1);
    SGZ_HIP(ctx, hipMalloc(&dPcm.p, pcmBytes));
    SGZ_HIP(ctx, hipMalloc(&dTw.p, sizeof(float2) * tw.size()));
    SGZ_HIP(ctx, hipMalloc(&dHann.p, sizeof(float) * hann.size()));
    SGZ_HIP(ctx, hipMalloc(&dDct.p, sizeof(float) * dct.size()));
    SGZ_HIP(ctx, hipMalloc(&dOut.p, outBytes));
    SGZ_HIP(ctx, hipMalloc(&dBandW.p, sizeof(float) * bandW.size()));
    SGZ_HIP(ctx, hipMemcpyAsync(dBandW.p, bandW.data(), sizeof(float) * bandW.size(), hipMemcpyHostToDevice, ctx->stream));
    // (one pageable copy: 27 GB/s; staging it through parallel pinned buffers was slower at this size)
    SGZ_HIP(ctx, hipMemcpyAsync(dPcm.p, pcm, pcmBytes, hipMemcpyHostToDevice, ctx->stream));
    SGZ_HIP(ctx, hipMemcpyAsync(dTw.p, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice, ctx->stream));
    SGZ_HIP(ctx, hipMemcpyAsync(dHann.p, hann.data(), sizeof(float) * hann.size(), hipMemcpyHostToDevice, ctx->stream));
    SGZ_HIP(ctx, hipMemcpyAsync(dDct.p, dct.data(), sizeof(float) * dct.size(), hipMemcpyHostToDevice, ctx->stream));
    {
        const int erc = ctx->next_event_pair();
        if (erc) return erc;
    }
    // frames per workgroup: as many as leave room for two workgroups per CU (160 KB of LDS)
    auto ldsBytes = [&](int nw) {
        return sizeof(float2) * ((size_t)nw * (M + 8) + (size_t)M) + sizeof(float) * (size_t)num_coeffs * kBands;
    };
    int kFramesPerBlock = kMaxFramesPerBlock;
    while (kFramesPerBlock > 1 && ldsBytes(kFramesPerBlock) > 80 * 1024) kFramesPerBlock >>= 1;
    const size_t lds = ldsBytes(kFramesPerBlock);
    SGZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(extract_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((nOut + kFramesPerBlock - 1) / kFramesPerBlock);
    SGZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(extract_kernel, dim3(grid), dim3(64 * kFramesPerBlock), lds, ctx->stream,
                       (const float *)dPcm.p, (const float2 *)dTw.p, (const float *)dHann.p, (const float *)dDct.p, (const float *)dBandW.p, bt,
                       p, (float *)dOut.p);
    SGZ_HIP(ctx, hipGetLastError());
    SGZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    ctx->ev_runs++;
    SGZ_HIP(ctx, hipMemcpyAsync(features_out, dOut.p, outBytes, hipMemcpyDeviceToHost, ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SGZ_OK;
}

}  // extern "C"
