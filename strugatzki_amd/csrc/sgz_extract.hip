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
#include <cstdlib>
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
    int swap16;            // 16-bit samples read in place are big-endian
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

// ---------------------------------------------------------------------------
// extract1024_kernel: the same computation for fftSize 1024 (the reference's default), restructured for
// instruction count -- the general kernel above spends about 3 000 wave instructions per frame, most of them
// in nine radix-2 stages that each go through LDS and in the serial per-band weight loops.
//   * FFT of the 512 packed complex points: three radix-8 passes (Stockham autosort).  A lane keeps 8 points in
//     registers; a pass multiplies them by twiddles, runs an 8-point butterfly in registers and hands the
//     results to the next pass through LDS (two exchanges in all).  The last pass leaves Z[lane + 64 r] in the
//     lane's registers, in natural order.
//   * real-FFT unpack and power per bin from Z[k] (register) and Z[512 - k] (LDS).
//   * band energies by segments between neighbouring band edges: lane s sums, over the bins of segment s, the
//     rising weights (band s) and the falling weights (band s - 1) in one loop, so a band costs half as many
//     serial steps and no weight loads.
//   * DCT: 16 coefficients x 4 quarter sums per sweep over the wave.
// Values agree with the general kernel to fp32 rounding (different order of summation).
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

// Lane s <= 42 walks segment s = the bins in [e_s, e_{s+1}], or the first part of it: the 21 longest segments are
// cut in two and lanes 43 .. 63 take the second parts, so that the longest walk is half as long.
struct SegTable {
    int b0[64], cnt[64];   // first bin and number of bins of the lane's run
    int other[64];         // lanes <= 42: the lane that walks the rest of this segment, or -1
    float aw[kBands];      // A-weighting at the band centre, dB
};

__device__ __forceinline__ f2 cmulv(f2 a, f2 b)
{
    return f2{fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}
__device__ __forceinline__ f2 mul_neg_i(f2 a) { return f2{a.y, -a.x}; }
// 8-point DFT, natural order in and out
__device__ __forceinline__ void fft8(f2 (&v)[8])
{
    constexpr float s = 0.70710678118654752f;
    const f2 a0 = v[0] + v[4], a4 = v[0] - v[4];
    const f2 a1 = v[1] + v[5], d5 = v[1] - v[5];
    const f2 a2 = v[2] + v[6], a6 = mul_neg_i(v[2] - v[6]);
    const f2 a3 = v[3] + v[7], d7 = v[3] - v[7];
    const f2 a5 = f2{d5.x + d5.y, d5.y - d5.x} * s;    // * e^{-i pi / 4}
    const f2 a7 = f2{d7.y - d7.x, -d7.x - d7.y} * s;   // * e^{-3 i pi / 4}
    const f2 b0 = a0 + a2, b2 = a0 - a2;
    const f2 b1 = a1 + a3, b3 = mul_neg_i(a1 - a3);
    const f2 b4 = a4 + a6, b6 = a4 - a6;
    const f2 b5 = a5 + a7, b7 = mul_neg_i(a5 - a7);
    v[0] = b0 + b1;
    v[4] = b0 - b1;
    v[2] = b2 + b3;
    v[6] = b2 - b3;
    v[1] = b4 + b5;
    v[5] = b4 - b5;
    v[3] = b6 + b7;
    v[7] = b6 - b7;
}

constexpr int kFastFft = 1024, kFastM = 512;
constexpr int kFastWaves = 4;                   // frames per workgroup: 6 workgroups of 25 KB LDS per CU (measured: 2: 0.223,
                                                // 3: 0.21, 4: 0.191, 5: 0.213, 6: 0.199, 7: 0.204, 8: 0.199 ms per 20 min of audio)
// A frame's exchange buffer: 8 points of padding per 64 (index i sits at i + 8 (i / 64)), and one more slot
// that holds a second copy of Z[0] (see the unpack step).
constexpr int kFastZ = kFastM + 8 * 8 + 8;

// Tables of the fftSize 1024 kernel, made on the host: laid out so that a lane's eight reads of a pass are one
// base address plus compile-time offsets.
struct FastTables {
    const float2 *tw2;    // [r][kk]   e^{-2 pi i r kk / 64},  r < 8, kk < 8
    const float2 *tw3;    // [r][lane] e^{-2 pi i r lane / 512}
    const float2 *twU;    // [k]       e^{-2 pi i k / 1024},   k < 512
    const float2 *hann2;  // [n]       (hann[2 n], hann[2 n + 1])
    const float2 *wgt;    // [bin]     (rising weight in its segment, falling weight), 513 bins
    const float *dct;     // [numCoeffs][kBands]
};

// kS16: `pcm` holds 16-bit samples as an audio file does, of either byte order (mono: a lane's two samples are one 32-bit
// word), turned into floats as pcm16_decode_kernel does (v / 32768), so the features are bit-identical to those of
// the decoded samples -- without the pass that decodes them and the float copy of the whole signal.
// kCh: 1 = mono, 2 = two interleaved channels (one 16- or 8-byte load per lane, the channels taken as the general
// path takes them: 0 + first + second for the mix), 0 = any number of channels.
template <int kCh, bool kS16 = false>
__global__ __launch_bounds__(64 * kFastWaves) void extract1024_kernel(const float *__restrict__ pcm, FastTables tb,
                                                                    SegTable st, ExtractParams p,
                                                                    float *__restrict__ out)
{
    __shared__ f2 tw2[64];
    __shared__ f2 wgt[kFastM + 8];
    __shared__ f2 zbuf[kFastWaves][kFastZ];
    extern __shared__ __attribute__((aligned(16))) float dctL[];  // [numCoeffs][kBands]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: addresses stay scalar
    if (threadIdx.x < 64) tw2[threadIdx.x] = f2{tb.tw2[threadIdx.x].x, tb.tw2[threadIdx.x].y};
    for (int n = threadIdx.x; n <= kFastM; n += 64 * kFastWaves) wgt[n] = f2{tb.wgt[n].x, tb.wgt[n].y};
    for (int n = threadIdx.x; n < p.numCoeffs * kBands; n += 64 * kFastWaves) dctL[n] = tb.dct[n];
    f2 *z = zbuf[wv];
    auto waveSync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const long long i = (long long)blockIdx.x * kFastWaves + wv;  // output frame
    const bool live = i < p.numOut;
    const long long k = i + 1;
    const long long s0 = (k - p.overlap) * p.step;
    const bool analysed = live && k >= p.overlap && s0 + kFastFft <= p.numFrames;
    const int numCh = p.numCoeffs + 1;

    // 1. load + mix down + window: point n = lane + 64 r holds x[2 n] + i x[2 n + 1]
    f2 v[8];
    if (analysed) {
        const float *base = kS16 ? pcm : pcm + s0 * (kCh ? kCh : p.numChannels);  // scalar
        const short *b16 = reinterpret_cast<const short *>(pcm) + s0 * (kCh ? kCh : p.numChannels);
        const unsigned int *base16 = reinterpret_cast<const unsigned int *>(b16);  // mono: s0 is even
        // two 16-bit samples in one word, each with its bytes in the order the file has them -> little-endian
        auto le2 = [&](unsigned int w) -> unsigned int {
            return p.swap16 ? ((w & 0x00ff00ffu) << 8) | ((w >> 8) & 0x00ff00ffu) : w;
        };
        auto smp = [&](int j) -> float {
            return kS16 ? (float)(short)le2((unsigned short)b16[j]) * (1.0f / 32768.0f) : base[j];
        };
        const f2 *hw = reinterpret_cast<const f2 *>(tb.hann2);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = lane + 64 * r;
            f2 x;
            if (kCh == 2) {  // frames 2 n and 2 n + 1: (first, second) each
                float4 a;
                if (kS16) {
                    uint2 w = *reinterpret_cast<const uint2 *>(b16 + 4 * n);
                    w.x = le2(w.x);
                    w.y = le2(w.y);
                    a = make_float4((float)(short)(w.x & 0xffffu) * (1.0f / 32768.0f), (float)(short)(w.x >> 16) * (1.0f / 32768.0f),
                                    (float)(short)(w.y & 0xffffu) * (1.0f / 32768.0f), (float)(short)(w.y >> 16) * (1.0f / 32768.0f));
                } else {
                    a = *reinterpret_cast<const float4 *>(base + 4 * n);
                }
                if (p.channels == 1) x = f2{a.x, a.z};
                else if (p.channels == 2) x = f2{a.y, a.w};
                else x = f2{(0.0f + a.x) + a.y, (0.0f + a.z) + a.w};
            } else if (kS16 && kCh == 1) {
                const unsigned int w = le2(base16[n]);  // samples 2 n (low half) and 2 n + 1
                x = f2{(float)(short)(w & 0xffffu) * (1.0f / 32768.0f), (float)(short)(w >> 16) * (1.0f / 32768.0f)};
            } else if (kCh == 1) {
                x = *reinterpret_cast<const f2 *>(base + 2 * n);
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int fr = (2 * n + h) * p.numChannels;
                    float xs;
                    if (p.channels == 1) xs = smp(fr);
                    else if (p.channels == 2) xs = smp(fr + p.numChannels - 1);
                    else {
                        xs = 0.0f;
                        for (int c = 0; c < p.numChannels; c++) xs += smp(fr + c);
                    }
                    x[h] = xs;
                }
            }
            v[r] = x * hw[n];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = f2{0.0f, 0.0f};
    }
    // the twiddles of the third pass and of the unpack step come straight from memory into registers (requested
    // now, used much later: the vector memory path is idle otherwise, the LDS is the busiest unit of this kernel)
    f2 t3[8], tU[8];
    {
        const f2 *g3 = reinterpret_cast<const f2 *>(tb.tw3) + lane, *gU = reinterpret_cast<const f2 *>(tb.twU) + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            t3[r] = g3[64 * r];
            tU[r] = gU[64 * r];
        }
    }
    __syncthreads();  // the tables are complete

    // 2. three radix-8 passes: sub-transform lengths 1 -> 8 -> 64 -> 512.
    //    First exchange: result r of lane l is point 8 l + r of the next pass, which lane j reads as point
    //    j + 64 r' = 8 (j / 8 + 8 r') + j % 8.  Stored as [r][l] with a row stride of 68: writes are consecutive
    //    over the lanes, and the reads (row j % 8, column j / 8 + 8 r') fall on 4 (j % 8) + j / 8 modulo 32,
    //    distinct within each half of the wave -- no bank conflicts either way.
    //    Later exchanges, padded positions: lane + 64 r -> lane + 72 r;  64 (lane / 8) + lane % 8 + 8 r -> the
    //    same + 8 (lane / 8).
    f2 *zA = z + lane;
    const f2 *zA2 = z + 68 * (lane & 7) + (lane >> 3);
    f2 *zB = z + lane;
    f2 *zC = z + 72 * (lane >> 3) + (lane & 7);
    fft8(v);
#pragma unroll
    for (int r = 0; r < 8; r++) zA[68 * r] = v[r];
    waveSync();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = zA2[8 * r];
    {
        const f2 *t2 = tw2 + (lane & 7);
#pragma unroll
        for (int r = 1; r < 8; r++) v[r] = cmulv(v[r], t2[8 * r]);
    }
    fft8(v);
    waveSync();  // everyone has read the first exchange
#pragma unroll
    for (int r = 0; r < 8; r++) zC[8 * r] = v[r];
    waveSync();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = zB[72 * r];
#pragma unroll
    for (int r = 1; r < 8; r++) v[r] = cmulv(v[r], t3[r]);
    fft8(v);  // v[r] = Z[lane + 64 r]

    // 3. spectrum of the real signal: X[k] = (Z[k] + conj Z[M-k]) / 2 - i w^k (Z[k] - conj Z[M-k]) / 2; power.
    //    k = lane + 64 r pairs with M - k = (64 - lane) + 64 (7 - r): register 7 - r of lane 64 - lane, fetched
    //    across the lanes; lane 0 pairs 64 r with 64 (8 - r), its own register 8 - r (and Z[0] with itself).
    float pw[8];
    {
        const int src = (64 - lane) & 63;
        f2 zmAll[8];
#pragma unroll
        for (int r = 0; r < 8; r++)  // all sixteen fetches first, one wait for the lot
            zmAll[r] = f2{__shfl(v[7 - r].x, src, 64), __shfl(v[7 - r].y, src, 64)};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const f2 own = v[r == 0 ? 0 : 8 - r];
            const f2 zm = lane == 0 ? own : zmAll[r];
            const f2 zk = v[r];
            const f2 e = f2{zk.x + zm.x, zk.y - zm.y};
            const f2 o = f2{zk.x - zm.x, zk.y + zm.y};
            const f2 t = cmulv(o, tU[r]);
            const float xr = e.x + t.y, xi = e.y - t.x;
            pw[r] = fmaf(xr, xr, xi * xi) * (0.25f * p.norm);  // (the two halves of e and o are folded in here)
        }
    }
    const float z0r = __shfl(v[0].x, 0, 64), z0i = __shfl(v[0].y, 0, 64);
    waveSync();  // all reads of the exchange buffer are done: it now takes the power spectrum
    float *P = reinterpret_cast<float *>(z);
#pragma unroll
    for (int r = 0; r < 8; r++) P[lane + 64 * r] = pw[r];
    if (lane == 0) P[kFastM] = (z0r - z0i) * (z0r - z0i) * p.norm;  // the Nyquist bin
    waveSync();

    // 4. band energies: lane s walks the bins of segment s = [e_s, e_{s+1}] once; they rise in band s and fall in
    //    band s - 1 (weights per bin from the table: a bin that sits exactly on e_s belongs to the segment before
    //    it and has a falling weight of 0 here)
    float rise = 0.0f, fall = 0.0f;
    {
        // SGZ_EX_PROBE (timing builds of tools/extract_phase_probe.sh only; results are wrong): 1 leaves out the walk
        // over the bins, 2 the DCT's sums, 3 both
#if defined(SGZ_EX_PROBE) && (SGZ_EX_PROBE & 1)
        const int b0 = st.b0[lane], cnt = 0;
#else
        const int b0 = st.b0[lane], cnt = st.cnt[lane];
#endif
        const float *Pb = P + b0;
        const f2 *wb = wgt + b0;
#pragma unroll 4
        for (int t = 0; t < cnt; t++) {
            const float x = Pb[t];
            const f2 w = wb[t];
            rise = fmaf(w.x, x, rise);
            fall = fmaf(w.y, x, fall);
        }
        // the rest of a segment that was cut in two (fixed order: first part + second part)
        const int other = lane <= kBands ? st.other[lane] : -1;
        const float r2 = __shfl(rise, other < 0 ? lane : other, 64), f2nd = __shfl(fall, other < 0 ? lane : other, 64);
        if (other >= 0) {
            rise += r2;
            fall += f2nd;
        }
    }
    const float fallNext = __shfl_down(fall, 1, 64);
    float db = 0.0f, sone = 0.0f;
    if (analysed && lane < kBands) {
        const float e = rise + fallNext;
        db = 100.0f + 10.0f * log10f(e + 1e-10f);
        if (db < 0.0f) db = 0.0f;
        float phon = db + st.aw[lane];
        if (phon < 0.0f) phon = 0.0f;
        sone = exp2f((phon - 40.0f) * 0.1f);
    }
    float loud = sone;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) loud += __shfl_xor(loud, off, 64);

    // 5. DCT-II: the dB values go through LDS; lane (c, q) = (lane % 16, lane / 16) sums a quarter of the bands
    waveSync();
    if (lane < kBands) P[kFastM + 8 + lane] = db;
    waveSync();
    float *o = out + i * numCh;
    for (int c0 = 0; c0 < p.numCoeffs; c0 += 16) {
        const int c = c0 + (lane & 15), q = lane >> 4;
        float part = 0.0f;
        if (c < p.numCoeffs) {
#if defined(SGZ_EX_PROBE) && (SGZ_EX_PROBE & 2)
            const int j0 = 11 * q, j1 = j0;
#else
            const int j0 = 11 * q, j1 = min(j0 + 11, kBands);
#endif
            for (int j = j0; j < j1; j++) part = fmaf(dctL[c * kBands + j], P[kFastM + 8 + j], part);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (analysed && lane < 16 && c < p.numCoeffs) o[1 + c] = 0.5f + 0.5f * (part * (1.0f / kBands)) * 0.01f;
    }
    if (live) {
        if (analysed) {
            if (lane == 0) o[0] = loud * (1.0f / 32.0f);
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

namespace {

// Sample decoding on the device: audio files hold 16-, 24- or 32-bit integers or floats of either byte order;
// uploading them as they are halves (16 bit) the transfer that bounds the call and spares the host a pass over
// every sample.  An integer v of b bits becomes v / 2^(b-1), rounded once to float where it does not fit (32 bit),
// as the host-side reader decodes it.
// fixNaNs of the reference's feature writer (impl/NonRealtimeProcessor.scala:176-190, on by default :34): the
// feature stream reaches the output file in chunks of 1024 stream frames, the first of them without stream frame 0
// (:98-108) -- output frames [0, 1023), [1023, 2047), ... -- and within a chunk and channel a NaN is replaced by
// the last value that was none, 0 at the start of every chunk (`lasts` is a fresh array per chunk).
// One workgroup per chunk; a chunk without a NaN (all of them, for sane input) costs one read.
__global__ __launch_bounds__(256) void fix_nans_kernel(float *__restrict__ feat, long long numFrames, int numCh)
{
    const long long f0 = max(0ll, 1024ll * blockIdx.x - 1), f1 = min(numFrames, 1024ll * (blockIdx.x + 1) - 1);
    const int len = (int)(f1 - f0);
    float *chunk = feat + f0 * numCh;
    int bad = 0;
    for (int e = threadIdx.x; e < len * numCh; e += 256) bad |= chunk[e] != chunk[e];
    if (!__syncthreads_or(bad)) return;
    if ((int)threadIdx.x < numCh) {  // a channel per thread, frame after frame (eight loads in flight)
        float last = 0.0f;
        float *p = chunk + threadIdx.x;
        for (int i = 0; i < len; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = i + u < len ? p[(long long)(i + u) * numCh] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i + u < len) {
                    if (v[u] != v[u]) p[(long long)(i + u) * numCh] = last;
                    else last = v[u];
                }
        }
    }
}

__global__ __launch_bounds__(256) void pcm16_decode_kernel(const uint4 *__restrict__ raw, long long count, int bigEndian,
                                                           float *__restrict__ out)
{
    // eight samples per thread: one 16-byte load, two 16-byte stores; the last, partial group sample by sample
    const long long groups = count >> 3;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
        const uint4 q = raw[g];
        const unsigned int w[4] = {q.x, q.y, q.z, q.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned int lo = w[k] & 0xffffu, hi = w[k] >> 16;
            if (bigEndian) {
                lo = ((lo & 0xffu) << 8) | (lo >> 8);
                hi = ((hi & 0xffu) << 8) | (hi >> 8);
            }
            f[2 * k] = (float)(short)lo * (1.0f / 32768.0f);
            f[2 * k + 1] = (float)(short)hi * (1.0f / 32768.0f);
        }
        float4 *dst = reinterpret_cast<float4 *>(out + 8 * g);
        dst[0] = make_float4(f[0], f[1], f[2], f[3]);
        dst[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 7)) {
        const long long i = 8 * groups + threadIdx.x;
        const unsigned char *b = reinterpret_cast<const unsigned char *>(raw) + 2 * i;
        const unsigned int v = bigEndian ? ((unsigned int)b[0] << 8) | b[1] : ((unsigned int)b[1] << 8) | b[0];
        out[i] = (float)(short)v * (1.0f / 32768.0f);
    }
}

// The other layouts byte by byte: bytes = 3 or 4 per sample, integer or (4 bytes only) IEEE float.
__global__ __launch_bounds__(256) void pcm_decode_kernel(const unsigned char *raw, long long count, int bytes,
                                                         int bigEndian, int isFloat, float *out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const unsigned char *b = raw + i * bytes;
        unsigned int u = 0;  // the sample's bytes, most significant first
        for (int k = 0; k < bytes; k++) u = (u << 8) | b[bigEndian ? k : bytes - 1 - k];
        float f;
        if (isFloat) {
            f = __uint_as_float(u);
        } else if (bytes == 3) {
            const int v = (int)(u << 8) >> 8;  // sign extension of 24 bits
            f = (float)v * (1.0f / 8388608.0f);
        } else {
            f = (float)((double)(int)u * (1.0 / 2147483648.0));
        }
        out[i] = f;
    }
}

// 3 or 4 bytes per sample, four samples per thread: three or four 32-bit loads, one 16-byte store; the same values as
// pcm_decode_kernel, which takes the last, partial group.
// For 4-byte layouts `out` may be `raw` itself (a thread reads its words before it writes them: no __restrict__).
template <int kBytes>
__global__ __launch_bounds__(256) void pcm_decode4_kernel(const unsigned int *raw, long long groups, int bigEndian,
                                                          int isFloat, float *out)
{
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
        unsigned int w[kBytes];
#pragma unroll
        for (int k = 0; k < kBytes; k++) w[k] = raw[g * kBytes + k];
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned int u;  // the sample, most significant byte first
            if (kBytes == 4) {
                u = bigEndian ? __builtin_bswap32(w[j]) : w[j];
            } else {
                unsigned int b[3];
#pragma unroll
                for (int k = 0; k < 3; k++) b[k] = (w[(3 * j + k) >> 2] >> (8 * ((3 * j + k) & 3))) & 0xffu;
                u = bigEndian ? (b[0] << 16) | (b[1] << 8) | b[2] : (b[2] << 16) | (b[1] << 8) | b[0];
            }
            if (kBytes == 3) f[j] = (float)((int)(u << 8) >> 8) * (1.0f / 8388608.0f);
            else if (isFloat) f[j] = __uint_as_float(u);
            else f[j] = (float)((double)(int)u * (1.0 / 2147483648.0));
        }
        *reinterpret_cast<float4 *>(out + 4 * g) = make_float4(f[0], f[1], f[2], f[3]);
    }
}

}  // namespace

namespace sgz {

int pcm_format(int sample_format, int *sample_bytes, int *big_endian, int *is_float)
{
    int sampleBytes = 4, bigEndian = 0, isFloat = 0;
    switch (sample_format) {
        case SGZ_PCM_F32: isFloat = 1; break;
        case SGZ_PCM_F32_BE: isFloat = 1; bigEndian = 1; break;
        case SGZ_PCM_S16: sampleBytes = 2; break;
        case SGZ_PCM_S16_BE: sampleBytes = 2; bigEndian = 1; break;
        case SGZ_PCM_S24: sampleBytes = 3; break;
        case SGZ_PCM_S24_BE: sampleBytes = 3; bigEndian = 1; break;
        case SGZ_PCM_S32: break;
        case SGZ_PCM_S32_BE: bigEndian = 1; break;
        default: return SGZ_E_ARG;
    }
    *sample_bytes = sampleBytes;
    *big_endian = bigEndian;
    *is_float = isFloat;
    return SGZ_OK;
}

// `count` samples of the layout `sample_format` in device memory (16-byte aligned, readable up to the next multiple
// of 16 bytes) to fp32, on `stream`.  4-byte layouts may be decoded in place (out == raw).
int pcm_decode_launch(sgz_context *ctx, const void *raw, long long count, int sample_format, float *out, hipStream_t stream)
{
    int sampleBytes, bigEndian, isFloat;
    if (pcm_format(sample_format, &sampleBytes, &bigEndian, &isFloat)) return ctx->fail(SGZ_E_ARG, "unknown sample format");
    if (count <= 0) return SGZ_OK;
    const int grid = (int)std::min<long long>((count / (sampleBytes == 2 ? 8 : 1) + 255) / 256 + 1, 8192);
    if (sampleBytes == 2) {
        hipLaunchKernelGGL(pcm16_decode_kernel, dim3(grid), dim3(256), 0, stream, (const uint4 *)raw, count, bigEndian, out);
    } else {
        const long long groups = count >> 2, tail = count & 3;
        const int grid4 = (int)std::min<long long>((groups + 255) / 256, 16384);
        if (groups && sampleBytes == 3)
            hipLaunchKernelGGL(pcm_decode4_kernel<3>, dim3(grid4), dim3(256), 0, stream, (const unsigned int *)raw, groups,
                               bigEndian, isFloat, out);
        else if (groups)
            hipLaunchKernelGGL(pcm_decode4_kernel<4>, dim3(grid4), dim3(256), 0, stream, (const unsigned int *)raw, groups,
                               bigEndian, isFloat, out);
        if (tail)
            hipLaunchKernelGGL(pcm_decode_kernel, dim3(1), dim3(256), 0, stream,
                               (const unsigned char *)raw + 4 * groups * sampleBytes, tail, sampleBytes, bigEndian, isFloat,
                               out + 4 * groups);
    }
    SGZ_HIP(ctx, hipGetLastError());
    return SGZ_OK;
}

}  // namespace sgz

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
    return sgz_extract_pcm(ctx, pcm, SGZ_PCM_F32, num_frames, num_channels, sample_rate, fft_size, fft_overlap,
                           num_coeffs, channels_behavior, features_out, capacity_frames, num_out);
}

int sgz_extract_pcm(sgz_context *ctx, const void *pcm, int sample_format, int64_t num_frames, int num_channels,
                    double sample_rate, int fft_size, int fft_overlap, int num_coeffs, int channels_behavior,
                    float *features_out, int64_t capacity_frames, int64_t *num_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (num_frames < 0 || (num_frames > 0 && !pcm) || num_channels < 1 || !(sample_rate > 0))
        return ctx->fail(SGZ_E_ARG, "bad audio input");
    int sampleBytes, bigEndian, isFloat;
    if (sgz::pcm_format(sample_format, &sampleBytes, &bigEndian, &isFloat)) return ctx->fail(SGZ_E_ARG, "unknown sample format");
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

    // segments between neighbouring band edges and the other tables of the fftSize 1024 kernel
    SegTable st;
    std::vector<float2> fastTab;  // tw2[64], tw3[512], twU[512], hann2[512], wgt[513 (+7)]
    const bool fast = fft_size == kFastFft && !std::getenv("SGZ_EXTRACT_GENERAL");
    if (fast) {
        const double mHi = melOfHz(sample_rate * 0.5);
        double e[kBands + 2];
        for (int j = 0; j < kBands + 2; j++) e[j] = hzOfMel(mHi * j / (kBands + 1)) * fft_size / sample_rate;
        fastTab.assign(64 + 3 * 512 + 520, make_float2(0.0f, 0.0f));
        float2 *tw2 = fastTab.data(), *tw3 = tw2 + 64, *twU = tw3 + 512, *hann2 = twU + 512, *wgt = hann2 + 512;
        for (int r = 0; r < 8; r++)
            for (int kk = 0; kk < 8; kk++) {
                const double a = -2.0 * M_PI * r * kk / 64.0;
                tw2[8 * r + kk] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        for (int r = 0; r < 8; r++)
            for (int l = 0; l < 64; l++) {
                const double a = -2.0 * M_PI * r * l / 512.0;
                tw3[64 * r + l] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        for (int k = 0; k < 512; k++) {
            twU[k] = tw[(size_t)k];
            hann2[k] = make_float2(hann[(size_t)2 * k], hann[(size_t)2 * k + 1]);
        }
        for (int l = 0; l < 64; l++) {
            st.b0[l] = st.cnt[l] = 0;
            st.other[l] = -1;
        }
        for (int sg = 0; sg <= kBands; sg++) {
            const float e0 = (float)e[sg], e1 = (float)e[sg + 1];
            st.b0[sg] = (int)std::ceil(e0);
            // a bin exactly on e0 is the last bin of the segment before (weight 1 there, none here)
            if (sg > 0 && (float)st.b0[sg] == e0) st.b0[sg]++;
            const int b1 = std::min((int)std::floor(e1), M);
            st.cnt[sg] = std::max(b1 - st.b0[sg] + 1, 0);
            // the weights of the general kernel's table, bin by bin: rising in band sg, falling in band sg - 1
            // (none for a bin exactly on e0: that one belongs to the segment before)
            for (int b = st.b0[sg]; b <= b1; b++) {
                const float fb = (float)b;
                const float wr = e1 > e0 ? (fb - e0) / (e1 - e0) : 0.0f;
                const float wf = (e1 > e0 && fb > e0) ? (e1 - fb) / (e1 - e0) : 0.0f;
                wgt[b] = make_float2(sg < kBands && wr > 0.0f ? wr : 0.0f, sg > 0 && wf > 0.0f ? wf : 0.0f);
            }
        }
        for (int j = 0; j < kBands; j++) st.aw[j] = bt.aw[j];
        // the spare lanes 43 .. 63 take the second halves of the longest segments
        for (int l = kBands + 1; l < 64; l++) {
            int longest = 0;
            for (int sg = 1; sg <= kBands; sg++)
                if (st.other[sg] < 0 && st.cnt[sg] > st.cnt[longest]) longest = sg;
            if (st.other[longest] >= 0 || st.cnt[longest] < 2) break;
            const int first = (st.cnt[longest] + 1) / 2;
            st.b0[l] = st.b0[longest] + first;
            st.cnt[l] = st.cnt[longest] - first;
            st.cnt[longest] = first;
            st.other[longest] = l;
        }
    }

    DevBuf dPcm, dRaw, dTw, dHann, dDct, dOut, dBandW;
    const size_t pcmBytes = sizeof(float) * (size_t)num_frames * num_channels;
    const size_t rawBytes = (size_t)sampleBytes * (size_t)num_frames * num_channels;
    const bool native = sample_format == SGZ_PCM_F32;
    const size_t outBytes = sizeof(float) * (size_t)nOut * (num_coeffs + 1);
    // 16-bit samples at the default FFT size: the extraction kernel reads them as they are
    // (an odd hop -- fftOverlap = fftSize -- would put a lane's 32-bit word of two samples on an odd sample)
    const bool direct16 = fast && sampleBytes == 2 && p.step % 2 == 0;
    p.swap16 = direct16 && bigEndian;
    if (!direct16) SGZ_HIP(ctx, hipMalloc(&dPcm.p, pcmBytes));
    SGZ_HIP(ctx, hipMalloc(&dTw.p, sizeof(float2) * tw.size()));
    SGZ_HIP(ctx, hipMalloc(&dHann.p, sizeof(float) * hann.size()));
    SGZ_HIP(ctx, hipMalloc(&dDct.p, sizeof(float) * dct.size()));
    SGZ_HIP(ctx, hipMalloc(&dOut.p, outBytes));
    SGZ_HIP(ctx, hipMalloc(&dBandW.p, sizeof(float) * bandW.size()));
    SGZ_HIP(ctx, hipMemcpyAsync(dBandW.p, bandW.data(), sizeof(float) * bandW.size(), hipMemcpyHostToDevice, ctx->stream));
    // (a single pageable copy runs at 27 GB/s; long inputs go through the context's parallel pinned staging)
    if (native) {
        const int crc = sgz::copy_large(ctx, dPcm.p, pcm, pcmBytes, false);
        if (crc) return crc;
    } else {  // the samples as the file holds them, decoded on the device
        SGZ_HIP(ctx, hipMalloc(&dRaw.p, rawBytes + 16));
        const int crc = sgz::copy_large(ctx, dRaw.p, pcm, rawBytes, false);
        if (crc) return crc;
        if (!direct16) {
            const int drc = sgz::pcm_decode_launch(ctx, dRaw.p, (long long)num_frames * num_channels, sample_format,
                                                   (float *)dPcm.p, ctx->stream);
            if (drc) return drc;
        }
    }
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
    DevBuf dFast;
    FastTables ft{};
    if (fast) {
        SGZ_HIP(ctx, hipMalloc(&dFast.p, sizeof(float2) * fastTab.size()));
        SGZ_HIP(ctx, hipMemcpyAsync(dFast.p, fastTab.data(), sizeof(float2) * fastTab.size(), hipMemcpyHostToDevice,
                                    ctx->stream));
        ft.tw2 = (const float2 *)dFast.p;
        ft.tw3 = ft.tw2 + 64;
        ft.twU = ft.tw3 + 512;
        ft.hann2 = ft.twU + 512;
        ft.wgt = ft.hann2 + 512;
        ft.dct = (const float *)dDct.p;
    }
    SGZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    const size_t fastLds = sizeof(float) * (size_t)num_coeffs * kBands;
    const dim3 fastGrid((unsigned)((nOut + kFastWaves - 1) / kFastWaves)), fastBlock(64 * kFastWaves);
    const int chK = num_channels == 1 ? 1 : (num_channels == 2 && p.step % 2 == 0) ? 2 : 0;  // the kernel's kCh
    auto launchFast = [&](auto kernel, const void *src) {
        hipLaunchKernelGGL(kernel, fastGrid, fastBlock, fastLds, ctx->stream, (const float *)src, ft, st, p, (float *)dOut.p);
    };
    if (direct16 && chK == 1) launchFast(extract1024_kernel<1, true>, dRaw.p);
    else if (direct16 && chK == 2) launchFast(extract1024_kernel<2, true>, dRaw.p);
    else if (direct16) launchFast(extract1024_kernel<0, true>, dRaw.p);
    else if (fast && chK == 1) launchFast(extract1024_kernel<1, false>, dPcm.p);
    else if (fast && chK == 2) launchFast(extract1024_kernel<2, false>, dPcm.p);
    else if (fast) launchFast(extract1024_kernel<0, false>, dPcm.p);
    else
        hipLaunchKernelGGL(extract_kernel, dim3(grid), dim3(64 * kFramesPerBlock), lds, ctx->stream,
                           (const float *)dPcm.p, (const float2 *)dTw.p, (const float *)dHann.p, (const float *)dDct.p,
                           (const float *)dBandW.p, bt, p, (float *)dOut.p);
    SGZ_HIP(ctx, hipGetLastError());
    SGZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    ctx->ev_runs++;
    if (nOut > 0) {
        hipLaunchKernelGGL(fix_nans_kernel, dim3((unsigned)((nOut + 1 + 1023) / 1024)), dim3(256), 0, ctx->stream,
                           (float *)dOut.p, (long long)nOut, num_coeffs + 1);
        SGZ_HIP(ctx, hipGetLastError());
    }
    SGZ_HIP(ctx, hipMemcpyAsync(features_out, dOut.p, outBytes, hipMemcpyDeviceToHost, ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SGZ_OK;
}

}  // extern "C"
