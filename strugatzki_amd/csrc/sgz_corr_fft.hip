// Spectral screening of the sliding cross term
// (impl/FeatureCorrelationImpl.scala:193-232 evaluates, per window o and channel c,
//  sum_i a'[c][i] * b[c][o + i] through MathUtil.correlate, impl/MathUtil.scala:177-196).
//
// For a block of N frames of a file starting at frame s, the sums of the windows o = s + k, 0 <= k <= N - L, are
// the first N - L + 1 values of the circular correlation of the block with the zero-padded target,
//
//     y[k] = IFFT( conj(FFT(a')) * FFT(x[s .. s + N)) )[k],
//
// about 95 times fewer operations than the direct sum at L = 2584 and no longer bound by arithmetic at all: the
// kernel streams the block's spectra (kept in HBM per database / norms / target length, like the half-precision
// image of the matrix-core path), multiplies them with the target's spectra, adds the channels of a group in
// the frequency domain, and inverts ONE complex transform per block whose real part is the loudness cross term
// and whose imaginary part the cross term of the other channels.  It is bound by memory traffic: per block
// 2 N numCh bytes of spectra (16-bit integers with one scale per band of 64 bins; 4 N numCh as fp32, on request) and
// as much again of the target's out of the cache, per window 8 bytes of cached statistics in (20 where a window's
// boost comes near the gate) and, for a search, nothing out but the few windows that can hold their file's
// maximum (the largest lower bound of a file is kept per file).
//
// All arithmetic is fp32.  The result is a screening value with a certified error like the matrix-core kernel's.
// A transform of `passes` passes is taken to be accurate to eps = (1 + 40 u)^passes - 1 of its input's norm (per
// pass: a table-derived twiddle <= 10 ulp, its product, four butterfly levels with their inner factors; every output
// is an inner product evaluated through a butterfly tree, every path of which carries that factor).  Two things
// make the bound of a cross term more than eps |a'| |x|:
//   * transforms are PACKED -- two real channels per forward transform, the two channel groups' cross terms as real
//     and imaginary part of the inverse one -- and a complex transform's rounding errors are relative to the norm of
//     the whole complex sequence: the partner's energy leaks in.  Forward: the spectrum of channel c is off by
//     eps sqrt(N) |(x_c, x_partner)|; through sum_j |A_j dX_j| / N <= |a'| |dx| (Cauchy-Schwarz, Parseval) the
//     loudness term gets eps (|a_01| |x_T| + |a_T| |x_01|) and the other channels' term
//     eps (|a_P| |x_S| + |a_S| |x_P|), with 01 = channels 0 and 1 together and P^2 = T^2 + 2 S^2 (every other channel
//     counted with its partner).  Inverse: both outputs get eps (sum |Zt| + beta sum |Zs|) / N
//     <= eps (|a_T| |x_T| + beta |a_S| |x_S|), the imaginary one divided by beta again;
//   * so the kernel BALANCES the two parts of the inverse transform: Zs is scaled by beta = |a_T||x_T| / (|a_S||x_S|)
//     (per block, from the block's norms) before it is packed, and the imaginary output divided by it.  Each cross
//     term's inverse-transform error is then 2 eps of its own norm product, whatever the other group's energy.
// Plus the roundings outside the transforms (operands, products, channel sums, the scaling), and one that does not
// scale with the block's norm: a normalised value is rounded at the size of the value itself, (x - min) / range, before
// the file's centring constant is taken off, so a file whose level is far from zero carries 2^-24 of that level as
// noise on every centred value -- against a window's cross term at most 2^-23 |level| / sd (FftGeom::levelErr).  At N = 16384, L = 2584
// the bound comes to about 1e-4 of the similarity where half-precision products give 1e-3.
// Spectra stored as 16-bit integers add one more term, which is not estimated but measured: the kernel that stores
// a spectrum decodes it again with the reader's own expression and sums the squared distance to what it computed
// (per block and channel; per channel of the target); a spectrum off by d in the Euclidean norm of the half
// spectrum moves a cross term by at most |a'| sqrt(2 / N) |d| resp. sqrt(2 N) |d_A| |x_block|.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "sgz_corr_dev.hpp"
#include "sgz_fft_dev.hpp"
#include "sgz_internal.hpp"
#include "sgz_target_dev.hpp"

using namespace sgz;

namespace {

constexpr int kTgtDqParts = 4;  // parts the target's rounding sums are kept in, per channel (sgz_context::d_tgtdq)

// Deterministic sum of one float per thread over the workgroup (fixed order); result valid in thread 0.
template <int T>
__device__ __forceinline__ float block_sum(float v, float *scratch /* T / 64 floats */)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    if (threadIdx.x == 0)
        for (int k = 0; k < T / 64; k++) s += scratch[k];
    __syncthreads();
    return s;
}

// Two real sequences transformed as one complex one: spectra of the real and of the imaginary part at bin j
// from Z[j] and Z[N - j].
__device__ __forceinline__ void untangle(float2 zj, float2 zm, float2 &xa, float2 &xb)
{
    xa = make_float2(0.5f * (zj.x + zm.x), 0.5f * (zj.y - zm.y));
    xb = make_float2(0.5f * (zj.y + zm.y), 0.5f * (zm.x - zj.x));
}

// 16-bit spectra: a complex bin is two signed 16-bit integers (real part in the low half) times the scale of its
// band of 64 bins.  The product is one fp32 rounding; the builder measures what the stored value is off the fp32
// spectrum with the very same expression the reader decodes with.
__device__ __forceinline__ float2 q16_decode(unsigned int q, float scale)
{
    return make_float2((float)(short)(q & 0xffffu) * scale, (float)((int)q >> 16) * scale);
}

// Largest value over the 64 lanes of a wave, the same in every lane afterwards: four steps inside the rows of 16
// lanes and two across them on the cross-lane data path, the result taken from the last lane.  (Lanes without a
// source in a step keep their own value.)
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_max(float v)
{
    const int own = __float_as_int(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(own, own, kCtrl, kRowMask, 0xf, false)));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = dpp_max<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max<0x118, 0xf>(v);  // row_shr:8
    v = dpp_max<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_max<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Rounds one band (the 64 bins of a wave, one per lane) to 16 bits: the band's scale is its largest component / 32767
// (NaN when a component is not finite, so that the block's results come out as NaN like those of fp32 spectra; 0 when
// the band is all but empty).  Returns the packed bin, adds the squared distance to `dq`.
__device__ __forceinline__ unsigned int q16_encode_band(float2 v, float &scaleOut, float &dq)
{
    // (a component that is not finite counts as infinite: a maximum drops a NaN next to a number)
    const bool fin = fabsf(v.x) <= 3.0e38f && fabsf(v.y) <= 3.0e38f;
    const float mx = wave_max(fin ? fmaxf(fabsf(v.x), fabsf(v.y)) : INFINITY);
    const bool bad = !(mx <= 3.0e38f);
    float scale = 0.0f, inv = 0.0f;
    if (mx >= 1.0e-30f) {
        scale = mx * (1.0f / 32767.0f);
        inv = 32767.0f / mx;
    }
    if (bad) scale = __builtin_nanf("");
    const float qx = fminf(fmaxf(rintf(v.x * inv), -32767.0f), 32767.0f);
    const float qy = fminf(fmaxf(rintf(v.y * inv), -32767.0f), 32767.0f);
    const unsigned int q = ((unsigned int)(int)qx & 0xffffu) | ((unsigned int)(int)qy << 16);
    const float2 d = q16_decode(q, scale);
    const float ex = d.x - v.x, ey = d.y - v.y;
    if (!bad) dq = fmaf(ex, ex, fmaf(ey, ey, dq));
    scaleOut = scale;
    return q;
}

// fft_spectra_kernel: workgroup = (block, channel pair).  The block's frames of two channels, normalised and
// centred exactly as the half-precision preparation does it (the window statistics of the bounds are sums of
// these very values), one forward transform, the two spectra to HBM; per channel the sum of squares.
// Q16: the spectra go out as 16-bit integers (q16_encode_band), with the scales and, per channel, the squared
// distance of what was stored to what was computed.
template <int N, bool Q16>
__global__ __launch_bounds__(N / 16) void fft_spectra_kernel(const FileDesc *__restrict__ files,
                                                             const FftBlock *__restrict__ blocks, int numPairs,
                                                             int numUnits, NormParams nrm, int numCh,
                                                             const float2 *__restrict__ tab, float2 *__restrict__ spec,
                                                             float *__restrict__ blksq, float *__restrict__ specscale,
                                                             float *__restrict__ blkdq)
{
    constexpr int T = N / 16, NH = N / 2;
    extern __shared__ __attribute__((aligned(16))) float2 buf[];
    __shared__ float red[16];
    const int tid = threadIdx.x;
    // The channel pairs of a block read the same interleaved frames: workgroups are dealt to the chip's eight
    // partitions (each with its own L2) in turn, so the grid is renumbered to give consecutive pairs to one of them.
    const int perPart = gridDim.x >> 3;  // (the grid is a multiple of 8)
    const int unit = (blockIdx.x & 7) * perPart + (blockIdx.x >> 3);
    if (unit >= numUnits) return;
    const int blk = unit / numPairs, p = unit - blk * numPairs;
    const FftBlock bd = blocks[blk];
    const FileDesc fd = files[bd.file];
    const int c0 = 2 * p, c1 = 2 * p + 1;
    const bool two = c1 < numCh;
    const float mn0 = nrm.enabled ? nrm.mn[c0] : 0.0f, mn1 = nrm.enabled && two ? nrm.mn[c1] : 0.0f;
    // (the reference's fp32 quotient, impl/MathUtil.scala:132-152, as the exact path and the window records form it)
    const float d0 = nrm.enabled ? nrm.d[c0] : 1.0f, d1 = nrm.enabled && two ? nrm.d[c1] : 1.0f;
    const float k0 = c0 == 0 ? fd.cT : fd.cS, k1 = fd.cS;
    float sq0 = 0.0f, sq1 = 0.0f;
#pragma unroll 4
    for (int m = 0; m < 16; m++) {
        const int n = tid + m * T;
        const long long fr = (long long)bd.o0 + n;
        float a = 0.0f, b = 0.0f;
        if (fr < fd.frames) {
            const float *src = fd.raw + fr * numCh;
            a = (src[c0] - mn0) / d0 - k0;
            if (two) b = (src[c1] - mn1) / d1 - k1;
        }
        buf[fft_phys(n)] = make_float2(a, b);
        sq0 = fmaf(a, a, sq0);
        sq1 = fmaf(b, b, sq1);
    }
    const float t0 = block_sum<T>(sq0, red);
    const float t1 = block_sum<T>(sq1, red);
    if (tid == 0) {
        blksq[(size_t)blk * numCh + c0] = t0;
        if (two) blksq[(size_t)blk * numCh + c1] = t1;
    }
    __syncthreads();
    fft_lds<N, -1>(buf, tab, tid);
    float2 *out0 = spec + bd.spec + (long long)c0 * NH;
    float2 *out1 = spec + bd.spec + (long long)c1 * NH;
    unsigned int *q0 = reinterpret_cast<unsigned int *>(spec) + bd.spec + (long long)c0 * NH;
    unsigned int *q1 = reinterpret_cast<unsigned int *>(spec) + bd.spec + (long long)c1 * NH;
    float *sc0 = specscale + ((size_t)blk * numCh + c0) * (NH / 64), *sc1 = specscale + ((size_t)blk * numCh + c1) * (NH / 64);
    float dq0 = 0.0f, dq1 = 0.0f;
#pragma unroll 4
    for (int m = 0; m < 8; m++) {
        const int j = tid + m * T;
        const float2 zj = buf[fft_phys(j)], zm = buf[fft_phys((N - j) & (N - 1))];
        float2 xa, xb;
        untangle(zj, zm, xa, xb);
        if (j == 0) {  // both real: the Nyquist bin rides in the imaginary part
            const float2 zn = buf[fft_phys(NH)];
            xa = make_float2(zj.x, zn.x);
            xb = make_float2(zj.y, zn.y);
        }
        if constexpr (Q16) {  // (a wave's 64 lanes hold one band)
            float sa, sb;
            q0[j] = q16_encode_band(xa, sa, dq0);
            if ((tid & 63) == 0) sc0[j >> 6] = sa;
            if (two) {
                q1[j] = q16_encode_band(xb, sb, dq1);
                if ((tid & 63) == 0) sc1[j >> 6] = sb;
            }
        } else {
            out0[j] = xa;
            if (two) out1[j] = xb;
        }
    }
    if constexpr (Q16) {
        const float e0 = block_sum<T>(dq0, red);
        const float e1 = block_sum<T>(dq1, red);
        if (tid == 0) {
            blkdq[(size_t)blk * numCh + c0] = e0;
            if (two) blkdq[(size_t)blk * numCh + c1] = e1;
        }
    }
}

// fft_target_kernel: workgroup = channel pair of the target; the conjugated spectra divided by N (the inverse
// transform's factor), packed like the database spectra, from the centred fp32 target (the first search after the
// layout changed; a new target for a standing layout goes through fft_target_sub_kernel / fft_target_encode_kernel).
// Q16: as 16-bit integers like the database's (q16_encode_band), with the scales and per channel the squared
// distance of what was stored to what was computed.
template <int N, bool Q16>
__global__ __launch_bounds__(N / 16) void fft_target_kernel(const float *__restrict__ aprime, int Lpad, int L, int numCh,
                                                            const float2 *__restrict__ tab, float2 *__restrict__ aspec,
                                                            float *__restrict__ ascale, float *__restrict__ tgtdq)
{
    constexpr int T = N / 16, NH = N / 2;
    extern __shared__ __attribute__((aligned(16))) float2 buf[];
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const int c0 = 2 * blockIdx.x, c1 = c0 + 1;
    const bool two = c1 < numCh;
#pragma unroll 4
    for (int m = 0; m < 16; m++) {
        const int n = tid + m * T;
        float a = 0.0f, b = 0.0f;
        if (n < L) {
            a = aprime[(size_t)c0 * Lpad + n];
            if (two) b = aprime[(size_t)c1 * Lpad + n];
        }
        buf[fft_phys(n)] = make_float2(a, b);
    }
    __syncthreads();
    fft_lds<N, -1>(buf, tab, tid);
    constexpr float rN = 1.0f / (float)N;
    float dq0 = 0.0f, dq1 = 0.0f;
#pragma unroll 4
    for (int m = 0; m < 8; m++) {
        const int j = tid + m * T;
        const float2 zj = buf[fft_phys(j)], zm = buf[fft_phys((N - j) & (N - 1))];
        float2 xa, xb;
        untangle(zj, zm, xa, xb);
        xa = make_float2(xa.x * rN, -xa.y * rN);
        xb = make_float2(xb.x * rN, -xb.y * rN);
        if (j == 0) {
            const float2 zn = buf[fft_phys(NH)];
            xa = make_float2(zj.x * rN, zn.x * rN);
            xb = make_float2(zj.y * rN, zn.y * rN);
        }
        if constexpr (Q16) {
            unsigned int *aq = reinterpret_cast<unsigned int *>(aspec);
            float sa, sb;
            aq[(size_t)c0 * NH + j] = q16_encode_band(xa, sa, dq0);
            if ((tid & 63) == 0) ascale[(size_t)c0 * (NH / 64) + (j >> 6)] = sa;
            if (two) {
                aq[(size_t)c1 * NH + j] = q16_encode_band(xb, sb, dq1);
                if ((tid & 63) == 0) ascale[(size_t)c1 * (NH / 64) + (j >> 6)] = sb;
            }
        } else {
            aspec[(size_t)c0 * NH + j] = xa;
            if (two) aspec[(size_t)c1 * NH + j] = xb;
        }
    }
    if constexpr (Q16) {
        const float e0 = block_sum<T>(dq0, red);
        const float e1 = block_sum<T>(dq1, red);
        if (tid < kTgtDqParts) {  // (one workgroup: its sum is the first part, the others are empty)
            tgtdq[kTgtDqParts * c0 + tid] = tid == 0 ? e0 : 0.0f;
            if (two) tgtdq[kTgtDqParts * c1 + tid] = tid == 0 ? e1 : 0.0f;
        }
    }
}

// The target's spectra for a target that has just been normalised (the launch behind the first pass of the target
// preparation): what fft_target_kernel does, cut into many small workgroups and a short second launch,
// because a query waits for it.  The N-point transform of a channel pair is decimated in frequency by 16: with
// n = n1 + M n2, M = N / 16, and k = 16 k1 + k2,
//     Z[16 k1 + k2] = sum_n1 W_M^(n1 k1) { W_N^(n1 k2) sum_n2 z[n1 + M n2] W_16^(n2 k2) },
// sixteen independent M-point transforms (fft_wave: one wave each), and of the sum over n2 only the first
// ceil(L / M) terms are there -- the rest of the padded target is zero.  A pass like any other in the error account
// (fft_eps: a sum of at most 16 terms with a table factor each and one more factor, against four butterfly levels
// with their factors).
// fft_target_sub_kernel: workgroups [0, 4 numPairs): four waves = four values of k2 of one pair, centred values
// formed from the normalised target and the means, Z to memory as [k2][k1]; the other
// workgroups carry the centring pass (target_centre_body).  fft_target_encode_kernel: a workgroup per pair, what
// fft_target_kernel does after its transform.
template <int N>
__global__ __launch_bounds__(256) void fft_target_sub_kernel(int L, int numCh, const float2 *__restrict__ tab,
                                                             float2 *__restrict__ zbuf, TargetCentreArgs ca)
{
    constexpr int M = N / 16, B = M / 16;
    extern __shared__ __attribute__((aligned(16))) float2 dyn[];  // [4][M] exchange buffers, the pair's L values, 16 roots
    const int numPairs = (numCh + 1) / 2;
    if ((int)blockIdx.x >= 4 * numPairs) {
        target_centre_body<256>((int)blockIdx.x - 4 * numPairs, (int)gridDim.x - 4 * numPairs, ca.L, ca.numCh, ca.Lpad,
                                ca.tnorm, ca.aprime, ca.aprimeF, ca.state, ca.normWgs, ca.hflags, ca.numFiles, ca.report,
                                ca.seq);
        return;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.x >> 2, k2 = 4 * (blockIdx.x & 3) + wave;
    const int c0 = 2 * pair, c1 = c0 + 1;
    const bool two = c1 < numCh;
    __shared__ double rows1[256], sum1[2];
    // the pair's centred values, once per workgroup (its four waves sum over the same ones): a frame's two channels
    // are neighbours in the normalised target; twelve frames per thread in flight at a time (a 30 s target: all of
    // them -- what the first pass of the preparation wrote has to come from memory, a round trip of microseconds)
    const int terms = (L + M - 1) / M;  // of the sum over n2; the values are padded with zeros to terms * M
    float2 *wb = dyn + wave * M, *cin = dyn + 4 * M, *w16 = cin + terms * M;
    constexpr int U = 12;
    const bool packed = (numCh & 1) == 0;  // a frame's two channels as one 8-byte load
    // every table factor of the workgroup is asked for here, ahead of the first barrier: what is asked for behind a
    // barrier waits for a trip to memory of its own (and right after a launch nothing is in the cache)
    const float2 root = tab[(threadIdx.x & 15) * (kFftTab / 16)];
    FftWaveFactors<M> wf;
    fft_wave_factors<M>(wf, tab, lane);
    // the factors W_N^(n1 k2) of the sixteen points n1 = lane + r B: the lane's own (r = 0) times powers of W_N^(B k2),
    // which is the same for all lanes -- the powers 1, 2, 4, 8 from the table, the others by at most three
    // multiplications (a gather per point would be sixteen loads of 64 cache lines each)
    float2 wn[16];
    {
        const float2 base = tab[((lane & (B - 1)) * k2) * (kFftTab / N)];
        float2 p[16];
        p[1] = tab[((B * k2) & (N - 1)) * (kFftTab / N)];
        p[2] = tab[((2 * B * k2) & (N - 1)) * (kFftTab / N)];
        p[4] = tab[((4 * B * k2) & (N - 1)) * (kFftTab / N)];
        p[8] = tab[((8 * B * k2) & (N - 1)) * (kFftTab / N)];
        p[3] = c_mul(p[1], p[2]);
        p[5] = c_mul(p[1], p[4]);
        p[6] = c_mul(p[2], p[4]);
        p[9] = c_mul(p[1], p[8]);
        p[10] = c_mul(p[2], p[8]);
        p[12] = c_mul(p[4], p[8]);
        p[7] = c_mul(p[3], p[4]);
        p[11] = c_mul(p[3], p[8]);
        p[13] = c_mul(p[5], p[8]);
        p[14] = c_mul(p[6], p[8]);
        p[15] = c_mul(p[7], p[8]);
        wn[0] = base;
#pragma unroll
        for (int r = 1; r < 16; r++) wn[r] = c_mul(base, p[r]);
    }
    double mean0 = 0.0, mean1 = 0.0;
    for (int n0 = threadIdx.x; n0 < terms * M; n0 += U * 256) {
        float2 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int n = n0 + 256 * u;
            x[u] = make_float2(0.0f, 0.0f);
            if (n < L) {
                const float *src = ca.tnorm + (size_t)n * numCh + c0;
                if (packed) x[u] = *reinterpret_cast<const float2 *>(src);
                else x[u] = make_float2(src[0], two ? src[1] : 0.0f);
            }
        }
        if (n0 == (int)threadIdx.x) {
            // the means, from the first pass' partial sums (behind the loads above, so that both travel together;
            // MathUtil.stat: mean = sum / n)
            tp_first_pass_sums<256>(ca.state, ca.normWgs, rows1, sum1);
            const double mT = sum1[0] / (double)L, mS = sum1[1] / ((double)L * (double)(numCh - 1));
            mean0 = c0 == 0 ? mT : mS;
            mean1 = mS;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int n = n0 + 256 * u;
            if (n < terms * M)
                cin[n] = n < L ? make_float2((float)((double)x[u].x - mean0), two ? (float)((double)x[u].y - mean1) : 0.0f)
                               : make_float2(0.0f, 0.0f);
        }
    }
    if (threadIdx.x < 16) w16[threadIdx.x] = c_dir<-1>(root);  // the sixteenth roots
    __syncthreads();
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = make_float2(0.0f, 0.0f);
    if (lane < B) {
        for (int n2 = 0; n2 < terms; n2++) {
            const float2 w = w16[(n2 * k2) & 15];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                v[r] = c_add(v[r], c_mul(cin[lane + r * B + M * n2], w));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = c_mul(v[r], c_dir<-1>(wn[r]));
    }
    fft_wave<M, -1>(v, wb, wf, lane);
    float2 *z = zbuf + (size_t)pair * N + (size_t)k2 * M;  // Z[16 k1 + k2] at [k2][k1]
#pragma unroll
    for (int m = 0; m < M / 64; m++) {
        const int k1 = lane + 64 * m;
        z[k1] = wb[fft_phys(k1)];
    }
}

template <int N, bool Q16>
__global__ __launch_bounds__(N / 64) void fft_target_encode_kernel(const float2 *__restrict__ zbuf, int numCh,
                                                                   float2 *__restrict__ aspec, float *__restrict__ ascale,
                                                                   float *__restrict__ tgtdq)
{
    // kTgtDqParts workgroups per pair, each a quarter of the half spectrum: eight bins per thread
    constexpr int T = N / 64, NH = N / 2;
    static_assert(kTgtDqParts == 4 && T % 64 == 0, "a quarter of the bins per workgroup, whole bands per wave");
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const int pair = blockIdx.x / kTgtDqParts, part = blockIdx.x % kTgtDqParts;
    const int c0 = 2 * pair, c1 = c0 + 1;
    const bool two = c1 < numCh;
    const float2 *z = zbuf + (size_t)pair * N;
    constexpr float rN = 1.0f / (float)N;
    float2 zj[8], zm[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {  // all loads first
        const int j = part * (NH / kTgtDqParts) + tid + m * T, jm = (N - j) & (N - 1);
        zj[m] = z[(j & 15) * (N / 16) + (j >> 4)];  // (the sub-transforms' order: [k mod 16][k / 16])
        zm[m] = z[(jm & 15) * (N / 16) + (jm >> 4)];
    }
    const float2 zn = z[(NH & 15) * (N / 16) + (NH >> 4)];
    float dq0 = 0.0f, dq1 = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int j = part * (NH / kTgtDqParts) + tid + m * T;
        float2 xa, xb;
        untangle(zj[m], zm[m], xa, xb);
        xa = make_float2(xa.x * rN, -xa.y * rN);
        xb = make_float2(xb.x * rN, -xb.y * rN);
        if (j == 0) {
            xa = make_float2(zj[m].x * rN, zn.x * rN);
            xb = make_float2(zj[m].y * rN, zn.y * rN);
        }
        if constexpr (Q16) {
            unsigned int *aq = reinterpret_cast<unsigned int *>(aspec);
            float sa, sb;
            aq[(size_t)c0 * NH + j] = q16_encode_band(xa, sa, dq0);
            if ((tid & 63) == 0) ascale[(size_t)c0 * (NH / 64) + (j >> 6)] = sa;
            if (two) {
                aq[(size_t)c1 * NH + j] = q16_encode_band(xb, sb, dq1);
                if ((tid & 63) == 0) ascale[(size_t)c1 * (NH / 64) + (j >> 6)] = sb;
            }
        } else {
            aspec[(size_t)c0 * NH + j] = xa;
            if (two) aspec[(size_t)c1 * NH + j] = xb;
        }
    }
    if constexpr (Q16) {
        const float e0 = block_sum<T>(dq0, red);
        const float e1 = block_sum<T>(dq1, red);
        if (tid == 0) {
            tgtdq[kTgtDqParts * c0 + part] = e0;
            if (two) tgtdq[kTgtDqParts * c1 + part] = e1;
        }
    }
}

// Euclidean norms of a block, rounded up: loudness channel, the other channels, channels 0 and 1 together,
// sqrt(|x_T|^2 + 2 |x_S|^2).
// With 16-bit spectra (blkdq given) also what the rounding to 16 bits amounts to, as norms of time-domain values:
// a spectrum off by d (half spectrum, Euclidean norm) moves a cross term by at most |a'| sqrt(2 / N) |d| (Cauchy-Schwarz
// over the bins, Parseval for the target, every bin of the half spectrum standing for at most two of the whole one);
// qscale = sqrt(2 / N).  The block's norms are raised by the same amounts: what the kernel multiplies is the
// stored spectrum.
__global__ void fft_blknorm_kernel(const float *__restrict__ blksq, int numCh, int nblocks, float4 *__restrict__ blknorm,
                                   const float *__restrict__ blkdq, float qscale, float2 *__restrict__ blkq)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const float *sq = blksq + (size_t)b * numCh;
    float s = 0.0f;
    for (int c = 1; c < numCh; c++) s += sq[c];
    const float s1 = numCh > 1 ? sq[1] : 0.0f;
    float4 bn = make_float4(sqrtf(sq[0]) * 1.0001f, sqrtf(s) * 1.0001f, sqrtf(sq[0] + s1) * 1.0001f,
                            sqrtf(sq[0] + 2.0f * s) * 1.0001f);
    if (blkdq) {
        const float *dq = blkdq + (size_t)b * numCh;
        float ds = 0.0f;
        for (int c = 1; c < numCh; c++) ds += dq[c];
        const float dT = sqrtf(dq[0]) * qscale * 1.001f, dS = sqrtf(ds) * qscale * 1.001f;
        blkq[b] = make_float2(dT, dS);
        bn.x += dT;
        bn.y += dS;
        bn.z += dT + dS;
        bn.w += dT + 1.5f * dS;
    }
    blknorm[b] = bn;
}

// Smallest ln(average loudness) over a block's windows -- its largest boost is exp((lnAvgIn - that) / 0.6) -- or
// -inf when a window has none (average <= 0): where even that boost passes the gate, the kernel needs no window's
// own value.  One workgroup of 256 threads per block.
__global__ __launch_bounds__(256) void fft_blklnmin_kernel(const FftBlock *__restrict__ blocks, float *wstat,
                                                           long long totalWindows, float *__restrict__ blklnmin)
{
    __shared__ float part[4];
    const FftBlock bd = blocks[blockIdx.x];
    const SpectralStatics st = spectral_statics(wstat, totalWindows);
    float m = INFINITY;
    for (int k = threadIdx.x; k < bd.valid; k += 256) {
        const float v = st.lnAvg[bd.win + k];
        m = v == v ? fminf(m, v) : -INFINITY;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blklnmin[blockIdx.x] = fminf(fminf(part[0], part[1]), fminf(part[2], part[3]));
}

// What a search's launch of the spectral kernel reads per window is cut to 8 bytes: the two 1 / sd of the window.  The
// other four numbers of a window's record -- g (how far the rounding of the prefix sums can move the sd, relative) and
// u (mean offset over sd, plus what the sums can move the mean by) of either channel group, which only widen the
// error term -- are replaced by their largest values over the block's windows: every term they enter grows with them,
// so a bound formed with the block's maxima encloses the one formed with the window's own.  A window whose g or u
// is not finite or lies above kBlkGCap / kBlkUCap is left out of the maxima and gets 1 / sd = +inf, which makes its
// bounds -inf .. +inf (a certain candidate: what its own large g did before), so that a stretch without spread does not
// widen the bounds of the thousands of ordinary windows around it.  One workgroup of 256 threads per block.
// (The caps sit well above what windows with a usable spread show -- g about 1e-6 at a standard deviation of a tenth of
// the normalised range, 1e-3 at a four-hundredth of it -- and bound what one odd window can add to the error term of
// its block's other windows: 2e-3 of the weight through g, 2.5e-4 through u.)
constexpr float kBlkGCap = 2.0e-3f, kBlkUCap = 1024.0f;
__global__ __launch_bounds__(256) void fft_blkgu_kernel(const FftBlock *__restrict__ blocks, float *wstat,
                                                        long long totalWindows, float4 *__restrict__ blkgu,
                                                        float2 *__restrict__ rec2)
{
    __shared__ float4 part[4];
    const FftBlock bd = blocks[blockIdx.x];
    const SpectralStatics st = spectral_statics(wstat, totalWindows);
    float4 m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // gT, uT, gS, uS
    for (int k = threadIdx.x; k < bd.valid; k += 256) {
        const float4 r = st.rec[bd.win + k];
        const unsigned int guT = __float_as_uint(r.z), guS = __float_as_uint(r.w);
        const float gT = half_from_bits(guT & 0xffffu), uT = half_from_bits(guT >> 16);
        const float gS = half_from_bits(guS & 0xffffu), uS = half_from_bits(guS >> 16);
        const bool ok = gT <= kBlkGCap && gS <= kBlkGCap && uT <= kBlkUCap && uS <= kBlkUCap;  // (false for NaN)
        if (ok) m = make_float4(fmaxf(m.x, gT), fmaxf(m.y, uT), fmaxf(m.z, gS), fmaxf(m.w, uS));
        rec2[bd.win + k] = ok ? make_float2(r.x, r.y) : make_float2(INFINITY, INFINITY);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m.x = fmaxf(m.x, __shfl_xor(m.x, off, 64));
        m.y = fmaxf(m.y, __shfl_xor(m.y, off, 64));
        m.z = fmaxf(m.z, __shfl_xor(m.z, off, 64));
        m.w = fmaxf(m.w, __shfl_xor(m.w, off, 64));
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++)
            m = make_float4(fmaxf(m.x, part[i].x), fmaxf(m.y, part[i].y), fmaxf(m.z, part[i].z), fmaxf(m.w, part[i].w));
        blkgu[blockIdx.x] = m;
    }
}

// Boosts of a run of windows from their statics: what the spectral kernel leaves out of its output (see
// sgz_context::boost_lazy).
__global__ __launch_bounds__(256) void boost_fill_kernel(float *wstat, long long totalWindows, long long w0, long long n,
                                                         float lnAvgIn, float *__restrict__ boost)
{
    const SpectralStatics st = spectral_statics(wstat, totalWindows);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        boost[w0 + i] = screen_boost(lnAvgIn, st.lnAvg[w0 + i]);
}

// corr_fft_kernel: workgroup = one block of N frames, N / 16 threads.
//   1. per bin j < N / 2: Zt = A_0 X_0, Zs = sum_{c >= 1} A_c X_c (A: conjugated target spectra / N), and the two
//      Hermitian spectra combined into one, W[j] = Zt + i Zs, W[N - j] = conj(Zt) + i conj(Zs), written to LDS;
//   2. inverse transform in LDS: w[k] = crossT[k] + i crossS[k];
//   3. per window the bounds from its cached statics: a search reads the window's two 1 / sd (8 bytes, rec2) and the
//      block's largest g and u (blkgu); the general path the full record (sgz::SpectralStatics, 20 bytes per window).
// Q16: the spectra are 16-bit integers with band scales (four bins per 16-byte load); `blkq` holds the block's
// rounding terms.
template <int N, int NCH, bool Q16>
__global__ __launch_bounds__(N / 16, 4) void corr_fft_kernel(const FileDesc *__restrict__ files,
                                                          const FftBlock *__restrict__ blocks,
                                                          const float2 *__restrict__ spec,
                                                          const float2 *__restrict__ aspec,
                                                          const float2 *__restrict__ tab,
                                                          const float4 *__restrict__ blknorm,
                                                          const float *__restrict__ specscale,
                                                          const float2 *__restrict__ blkq,
                                                          const float *__restrict__ ascale,
                                                          const float *__restrict__ tgtdq,
                                                          const float *__restrict__ blklnmin,
                                                          const float4 *__restrict__ blkgu,
                                                          const float2 *__restrict__ rec2,
                                                          float *__restrict__ wstat, long long totalWindows,
                                                          CorrParams cp, FftGeom fg,
                                                          float *__restrict__ simOut, float *__restrict__ boostOut,
                                                          int *__restrict__ lokey,
                                                          int *__restrict__ lokeyNext, int *__restrict__ hflags,
                                                          PreList pre)
{
    constexpr int T = N / 16, NH = N / 2;
    extern __shared__ __attribute__((aligned(16))) float2 buf[];
    __shared__ float wbest[16];
    __shared__ int wbad[16];
    __shared__ float blockBest;
    const int tid = threadIdx.x;
    const FftBlock bd = blocks[blockIdx.x];
    const int numCh = NCH ? NCH : cp.numCh;
    const float2 *xs = spec + bd.spec;
    // the factor that balances the two parts of the packed inverse transform, and the block's error terms
    const float4 bn = blknorm[blockIdx.x];
    const float pT = fg.aT * bn.x, pS = fg.aS * bn.y;  // |a'| |x_block| of the two channel groups
    float beta = pT / pS;
    if (!(beta >= 2.4414062e-4f && beta <= 4096.0f)) beta = beta > 4096.0f ? 4096.0f : (beta < 2.4414062e-4f ? 2.4414062e-4f : 1.0f);
    const float rbeta = 1.0f / beta;
    const FileDesc fd = files[bd.file];
    float absT = (fg.epsT * (fg.a01 * bn.x + fg.aT * bn.z + pT + beta * pS) + fg.epsUT * pT) * 1.0001f * fg.kT +
                 fg.levelErr * fabsf(fd.cT);
    float absS = (fg.epsT * (fg.aP * bn.y + fg.aS * bn.w + pT * rbeta * 1.0001f + pS) + fg.epsUS * pS) * 1.0001f * fg.kS +
                 fg.levelErr * fabsf(fd.cS);
    if constexpr (Q16) {
        // the stored spectra against the computed ones: the block's (blkq, already as norms of time-domain values)
        // and the target's -- a target spectrum off by d (half spectrum) moves a cross term by at most
        // sqrt(2) |d| |X| = sqrt(2 N) |d| |x_block|
        const float2 bq = blkq[blockIdx.x];
        // (per channel in kTgtDqParts parts, one per workgroup that encoded a share of the spectrum)
        auto dqOf = [&](int c) {
            return (tgtdq[kTgtDqParts * c] + tgtdq[kTgtDqParts * c + 1]) + (tgtdq[kTgtDqParts * c + 2] + tgtdq[kTgtDqParts * c + 3]);
        };
        float tqS = 0.0f;
        for (int c = 1; c < numCh; c++) tqS += dqOf(c);
        const float tscale = __builtin_amdgcn_sqrtf(2.0f * (float)N) * 1.001f;
        const float tqT = __builtin_amdgcn_sqrtf(dqOf(0)) * tscale;
        tqS = __builtin_amdgcn_sqrtf(tqS) * tscale;
        absT += (fg.aT * bq.x + tqT * bn.x) * 1.0001f * fg.kT;
        absS += (fg.aS * bq.y + tqS * bn.y) * 1.0001f * fg.kS;
    }

    // SGZ_PROBE (timing builds of tools/spectral_phase_probe.sh only; results are wrong): 1 leaves out the spectra
    // phase, 2 the transform, 3 everything after the transform
#if defined(SGZ_PROBE) && SGZ_PROBE == 1
    for (int m = 0; m < 16; m++) buf[fft_phys(tid + m * T)] = make_float2(0.001f * tid, 0.002f * m);
#else
    if constexpr (Q16) {
        // four neighbouring bins per thread and step: one 16-byte load each of the block's spectrum and of the
        // target's, and the two scales of the band (a band is the bins of 16 neighbouring threads).  The scales are
        // multiplied first and applied to one side only: the target's integers go into the products as they are.
        const unsigned int *xq = reinterpret_cast<const unsigned int *>(spec) + bd.spec;
        const unsigned int *aq = reinterpret_cast<const unsigned int *>(aspec);
        const uint4 *xq4 = reinterpret_cast<const uint4 *>(xq), *aq4 = reinterpret_cast<const uint4 *>(aq);
        const float *sc = specscale + (size_t)blockIdx.x * numCh * (NH / 64);
        auto mac = [&](int c, const uint4 q, float scale, const uint4 qa, float2 (&zt)[4], float2 (&zs)[4]) {
            const float2 a[4] = {q16_decode(qa.x, 1.0f), q16_decode(qa.y, 1.0f), q16_decode(qa.z, 1.0f), q16_decode(qa.w, 1.0f)};
            const float2 x[4] = {q16_decode(q.x, scale), q16_decode(q.y, scale), q16_decode(q.z, scale),
                                 q16_decode(q.w, scale)};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (c == 0) {
                    zt[i] = c_mul(a[i], x[i]);
                    zs[i] = make_float2(0.0f, 0.0f);
                } else {
                    zs[i].x = fmaf(a[i].x, x[i].x, zs[i].x);
                    zs[i].x = fmaf(-a[i].y, x[i].y, zs[i].x);
                    zs[i].y = fmaf(a[i].x, x[i].y, zs[i].y);
                    zs[i].y = fmaf(a[i].y, x[i].x, zs[i].y);
                }
            }
        };
        auto scatter = [&](int j, const float2 (&zt)[4], const float2 (&zs)[4]) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float sx = zs[i].x * beta, sy = zs[i].y * beta;
                buf[fft_phys((N - j - i) & (N - 1))] = make_float2(zt[i].x + sy, sx - zt[i].y);
                buf[fft_phys(j + i)] = make_float2(zt[i].x - sy, zt[i].y + sx);
            }
        };
        // (the loop over the other channels is unrolled by seven, not completely: with all 28 pairs of loads of the
        // two steps in one block the compiler's schedule ends each step with loads that are waited for one by one;
        // measured 38.8 instead of 43.7 us at 240 blocks, 0.464 instead of 0.485 ms at 3000; by 3 .. 10: 38.8 - 39.5 us)
#pragma unroll 2
        for (int m = 0; m < 2; m++) {
            const int h = tid + m * T;  // bins 4 h .. 4 h + 3
            float2 zt[4], zs[4];
            mac(0, xq4[h], sc[h >> 4] * ascale[h >> 4], aq4[h], zt, zs);
#pragma unroll 7
            for (int c = 1; c < numCh; c++)
                mac(c, xq4[(size_t)c * (NH / 4) + h], sc[c * (NH / 64) + (h >> 4)] * ascale[c * (NH / 64) + (h >> 4)],
                    aq4[(size_t)c * (NH / 4) + h], zt, zs);
            scatter(4 * h, zt, zs);
        }
        if (tid == 0) {  // bin 0 holds two real bins, DC and Nyquist: component-wise products
            const float2 x0 = q16_decode(xq[0], sc[0] * ascale[0]), a0 = q16_decode(aq[0], 1.0f);
            float2 zt = make_float2(a0.x * x0.x, a0.y * x0.y);
            float2 zs = make_float2(0.0f, 0.0f);
            for (int c = 1; c < numCh; c++) {
                const float2 a = q16_decode(aq[(size_t)c * NH], 1.0f);
                const float2 x = q16_decode(xq[(size_t)c * NH], sc[c * (NH / 64)] * ascale[c * (NH / 64)]);
                zs.x = fmaf(a.x, x.x, zs.x);
                zs.y = fmaf(a.y, x.y, zs.y);
            }
            buf[fft_phys(0)] = make_float2(zt.x, zs.x * beta);
            buf[fft_phys(NH)] = make_float2(zt.y, zs.y * beta);
        }
    } else {
        // two neighbouring bins per thread and step: 16-byte loads (1 KB of a channel's spectrum per wave instruction)
        const float4 *xs4 = reinterpret_cast<const float4 *>(xs), *as4 = reinterpret_cast<const float4 *>(aspec);
#pragma unroll 4
        for (int m = 0; m < 4; m++) {
            const int h = tid + m * T, j = 2 * h;  // bins j and j + 1
            const float4 a0 = as4[h], x0 = xs4[h];
            float2 zt0 = c_mul(make_float2(a0.x, a0.y), make_float2(x0.x, x0.y));
            float2 zt1 = c_mul(make_float2(a0.z, a0.w), make_float2(x0.z, x0.w));
            float2 zs0 = make_float2(0.0f, 0.0f), zs1 = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int c = 1; c < numCh; c++) {
                const float4 a = as4[(size_t)c * (NH / 2) + h], x = xs4[(size_t)c * (NH / 2) + h];
                zs0.x = fmaf(a.x, x.x, zs0.x);
                zs0.x = fmaf(-a.y, x.y, zs0.x);
                zs0.y = fmaf(a.x, x.y, zs0.y);
                zs0.y = fmaf(a.y, x.x, zs0.y);
                zs1.x = fmaf(a.z, x.z, zs1.x);
                zs1.x = fmaf(-a.w, x.w, zs1.x);
                zs1.y = fmaf(a.z, x.w, zs1.y);
                zs1.y = fmaf(a.w, x.z, zs1.y);
            }
            zs0.x *= beta;
            zs0.y *= beta;
            zs1.x *= beta;
            zs1.y *= beta;
            buf[fft_phys((N - j) & (N - 1))] = make_float2(zt0.x + zs0.y, zs0.x - zt0.y);
            buf[fft_phys(j)] = make_float2(zt0.x - zs0.y, zt0.y + zs0.x);
            buf[fft_phys(N - j - 1)] = make_float2(zt1.x + zs1.y, zs1.x - zt1.y);
            buf[fft_phys(j + 1)] = make_float2(zt1.x - zs1.y, zt1.y + zs1.x);
        }
        if (tid == 0) {  // bin 0 holds two real bins, DC and Nyquist: component-wise products
            float2 zt = make_float2(aspec[0].x * xs[0].x, aspec[0].y * xs[0].y);
            float2 zs = make_float2(0.0f, 0.0f);
            for (int c = 1; c < numCh; c++) {
                const float2 a = aspec[(size_t)c * NH], x = xs[(size_t)c * NH];
                zs.x = fmaf(a.x, x.x, zs.x);
                zs.y = fmaf(a.y, x.y, zs.y);
            }
            buf[fft_phys(0)] = make_float2(zt.x, zs.x * beta);
            buf[fft_phys(NH)] = make_float2(zt.y, zs.y * beta);
        }
    }
#endif
    __syncthreads();
#if !defined(SGZ_PROBE) || SGZ_PROBE != 2
    fft_lds<N, 1>(buf, tab, tid);
#endif
#if defined(SGZ_PROBE) && SGZ_PROBE == 3
    if (tid == 0 && simOut) simOut[bd.win] = buf[5].x;
    return;
#endif

    CorrParams cpf = cp;  // the statistics of the fp32 target values
    cpf.sumAT = fg.sumAT;
    cpf.sumAS = fg.sumAS;
    float best = -INFINITY;
    int bad = 0;
    const SpectralStatics st = spectral_statics(wstat, totalWindows);
    // Boosts are written only on request (boostOut), and read only where the block's largest boost does not pass
    // the gate by itself: 8 of 28 bytes per window that most launches do not move.
    const bool allOpen = screen_gate(screen_boost((float)cp.lnAvgIn, blklnmin[blockIdx.x]) * 1.00002f, cp.maxBoost) == 1;
    const bool needLn = boostOut != nullptr || !allOpen;
    // (the loops are unrolled so that the sixteen upper bounds of a thread stay in registers for the candidate pass)
    float ups[16];
    if (!needLn && cp.w > 0.0f && cp.w < 1.0f) {
        // What a search runs: every window's gate open, both channel groups weighted, no boosts asked for.
        // The same bounds as window_coeffs_spectral / window_bounds give, with everything that is the same for all
        // windows taken out of the loop and no branch inside it (the kernel's transform and this loop are bound by
        // instruction issue, not by memory: about 40 instructions per window instead of 95).
        const float nT = (float)cp.L, nS = (float)cp.L * (float)(cp.numCh - 1);
        const float kA = cp.w / ((float)cp.aStdT * nT), kB = ((1.0f - cp.w) / ((float)cp.aStdS * nS)) * rbeta;
        const float mT = fabsf((float)cpf.sumAT) / ((float)cp.aStdT * nT) * 1.0001f;
        const float mS = fabsf((float)cpf.sumAS) / ((float)cp.aStdS * nS) * 1.0001f;
        const float wT = cp.w, wS = 1.0f - cp.w, relErr = fg.relErr;
        // g and u of the block (fft_blkgu_kernel): errX = mX uX + (1 + ehX) gX + ehX with ehX = absX / sd + relErr (1 + uX),
        // everything that does not depend on the window formed here
        const float4 gu = blkgu[blockIdx.x];
        const float ehT0 = relErr * (1.0f + gu.y), ehS0 = relErr * (1.0f + gu.w);
        const float erT0 = mT * gu.y + gu.x, erS0 = mS * gu.w + gu.z, erT1 = 1.0f + gu.x, erS1 = 1.0f + gu.z;
        // (the records of G windows per thread are on their way at a time; 8 instead of 4 changes nothing measurable)
        constexpr int G = 4;
#pragma unroll
        for (int m0 = 0; m0 < 16; m0 += G) {
#pragma unroll
            for (int u = 0; u < G; u++) ups[m0 + u] = -INFINITY;
            if (m0 * T < bd.valid) {  // (uniform over the workgroup)
                float2 rec[G];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int k = tid + (m0 + u) * T;
                    rec[u] = rec2[bd.win + (k < bd.valid ? k : 0)];
                }
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const int k = tid + (m0 + u) * T;
                    const bool live = k < bd.valid;
                    const float2 y = buf[fft_phys(live ? k : 0)];
                    const float ehT = fmaf(absT, rec[u].x, ehT0), ehS = fmaf(absS, rec[u].y, ehS0);
                    const float errT = fmaf(ehT, erT1, erT0), errS = fmaf(ehS, erS1, erS0);
                    const float e = fmaf(errT, wT, errS * wS) + 1e-5f;
                    const float sim = fmaf(kA * rec[u].x, y.x, (kB * rec[u].y) * y.y);
                    const bool eFin = e < INFINITY, fin = eFin && fabsf(sim) < INFINITY;
                    float up = fin ? sim + e : INFINITY, lo = fin ? sim - e : -INFINITY;
                    if (sim != sim && eFin) up = lo = sim;  // NaN data: stays NaN, as in the exact path
                    if (simOut && live) simOut[bd.win + k] = up;
                    ups[m0 + u] = live ? up : -INFINITY;
                    best = fmaxf(best, live ? lo : -INFINITY);
                    bad |= live && ((lo != lo) || (up != up));
                }
            }
        }
    } else {
#pragma unroll
        for (int m0 = 0; m0 < 16; m0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) ups[m0 + u] = -INFINITY;
            if (m0 * T < bd.valid) {  // (uniform over the workgroup)
                float rsdT[4], rsdS[4], lnAvg[4];
                uint2 gu[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = tid + (m0 + u) * T;
                    const long long w = bd.win + (k < bd.valid ? k : 0);
                    const float4 r = st.rec[w];
                    rsdT[u] = r.x;
                    rsdS[u] = r.y;
                    gu[u] = make_uint2(__float_as_uint(r.z), __float_as_uint(r.w));
                    lnAvg[u] = needLn ? st.lnAvg[w] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = tid + (m0 + u) * T;
                    if (k < bd.valid) {
                        const float2 y = buf[fft_phys(k)];
                        float sim, lo;
                        int gate = 1;
                        if (needLn) {
                            const float boost = screen_boost((float)cp.lnAvgIn, lnAvg[u]);
                            gate = screen_gate(boost, cp.maxBoost);
                            if (boostOut) boostOut[bd.win + k] = boost;
                        }
                        const WindowCoeffs wc = window_coeffs_spectral(cpf, rsdT[u], rsdS[u], gu[u], fg.relErr, absT, absS, gate);
                        window_bounds(wc, y.x, y.y * rbeta, sim, lo);
                        if (simOut) simOut[bd.win + k] = sim;
                        ups[m0 + u] = sim;
                        best = fmaxf(best, lo);
                        bad |= (lo != lo) || (sim != sim);
                    }
                }
            }
        }
    }
    // the file's largest lower bound (one atomic per workgroup); NaN similarities mark the file for the exact path
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        best = fmaxf(best, __shfl_xor(best, off, 64));
        bad |= __shfl_xor(bad, off, 64);
    }
    if ((tid & 63) == 0) {
        wbest[tid >> 6] = best;
        wbad[tid >> 6] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < T / 64; k++) {
            best = fmaxf(best, wbest[k]);
            bad |= wbad[k];
        }
        atomicMax(&lokey[bd.file], bound_key(best));
        if (bad) atomicOr(&hflags[bd.file], 2);
        // the run after this one folds its bounds into the other array: the file's first block resets its entry
        if (bd.o0 == 0) lokeyNext[bd.file] = bound_key(-INFINITY);
        blockBest = best;
    }
    if (!pre.list) return;
    // Candidate pass (lists instead of dense upper bounds): a window whose upper bound does not reach the largest
    // lower bound of its own block cannot hold its file's maximum; the few that do -- a handful per block -- are
    // listed, and the search filters that list by the per-file and the k-th bound instead of scanning every window.
    __syncthreads();
    const float bb = blockBest;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k = tid + i * T;
        if (k < bd.valid && ups[i] >= bb) {
            const unsigned long long slot = atomicAdd(pre.count, 1ull);
            if ((long long)slot < pre.capacity) {
                sgz_candidate c;
                c.file = bd.file;
                c.offset = bd.o0 + k;
                c.sim = ups[i];
                c.boost = screen_boost((float)cp.lnAvgIn, st.lnAvg[bd.win + k]);
                pre.list[slot] = c;
            }
        }
    }
}

// eps of the header comment for a transform of N points.
double fft_eps(int N)
{
    const double u = std::ldexp(1.0, -24);
    const int passes = N > 4096 ? 4 : 3;
    return std::expm1(passes * std::log1p(40.0 * u)) * 1.01;
}

// Raises a kernel's limit of dynamic LDS where it needs more than the default 64 KB -- once per context and kernel
// (the call costs microseconds, a search launches the same kernel every time; the size depends on the kernel's
// template arguments only).
template <typename K>
int set_lds(sgz_context *ctx, K kern, size_t bytes)
{
    if (bytes <= 64 * 1024) return SGZ_OK;
    const void *key = reinterpret_cast<const void *>(kern);
    for (const void *k : ctx->lds_attr_set)
        if (k == key) return SGZ_OK;
    SGZ_HIP(ctx, hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->lds_attr_set.push_back(key);
    return SGZ_OK;
}

int ensure_tab(sgz_context *ctx)
{
    if (ctx->d_fft_tab) return SGZ_OK;
    std::vector<float2> tab(kFftTabAll);
    fft_fill_table(tab.data());
    SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_fft_tab), sizeof(float2) * kFftTabAll));
    SGZ_HIP(ctx, hipMemcpy(ctx->d_fft_tab, tab.data(), sizeof(float2) * kFftTabAll, hipMemcpyHostToDevice));
    return SGZ_OK;
}

// The target's spectra (per target; N from the layout): one kernel on `stream`, reading the centred fp32 target --
// or, with `centre`, launched behind the first pass of the target preparation and carrying its second pass along in
// further workgroups (the centred target is not needed then).
int fft_target_launch(sgz_context *ctx, hipStream_t stream, const TargetCentreArgs *centre)
{
    FftGeom &fg = ctx->fg;
    const int N = fg.N, numCh = ctx->cp.numCh, L = ctx->cp.L;
    const int numPairs = (numCh + 1) / 2;
    int rc;
    if ((rc = ensure(ctx, ctx->d_aspec, ctx->aspec_cap, (size_t)numCh * (N / 2)))) return rc;
    const bool q16 = fg.q16 != 0;
    if (q16) {
        if ((rc = ensure(ctx, ctx->d_aspecscale, ctx->aspecscale_cap, (size_t)numCh * (N / 128)))) return rc;
        if ((rc = ensure(ctx, ctx->d_tgtdq, ctx->tgtdq_cap, (size_t)numCh * kTgtDqParts))) return rc;
    }
    const size_t lds = sizeof(float2) * fft_lds_elems(N);
    const int Lpad = ctx->cp.S * kR;
    if (centre) {
        // behind the first pass of the target preparation: sixteen sub-transforms per pair and the centring pass in
        // one launch, the encoding of the spectra in a second one
        if ((rc = ensure(ctx, ctx->d_tgt_z, ctx->tgt_z_cap, (size_t)numPairs * N))) return rc;
        const int extra = fft_target_centre_workgroups(ctx, numCh, Lpad);
        auto both = [&](auto subK, auto encK) -> int {
            const size_t subLds = sizeof(float2) * ((size_t)4 * (N / 16) + (size_t)((L + N / 16 - 1) / (N / 16)) * (N / 16) + 16);
            if ((rc = set_lds(ctx, subK, subLds))) return rc;
            hipLaunchKernelGGL(subK, dim3(4 * numPairs + extra), dim3(256), subLds, stream, L, numCh, ctx->d_fft_tab,
                               ctx->d_tgt_z, *centre);
            SGZ_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL(encK, dim3(numPairs * kTgtDqParts), dim3(N / 64), 0, stream, ctx->d_tgt_z, numCh, ctx->d_aspec,
                               q16 ? ctx->d_aspecscale : nullptr, q16 ? ctx->d_tgtdq : nullptr);
            SGZ_HIP(ctx, hipGetLastError());
            return SGZ_OK;
        };
        if (N == 4096)
            return q16 ? both(fft_target_sub_kernel<4096>, fft_target_encode_kernel<4096, true>)
                       : both(fft_target_sub_kernel<4096>, fft_target_encode_kernel<4096, false>);
        if (N == 8192)
            return q16 ? both(fft_target_sub_kernel<8192>, fft_target_encode_kernel<8192, true>)
                       : both(fft_target_sub_kernel<8192>, fft_target_encode_kernel<8192, false>);
        return q16 ? both(fft_target_sub_kernel<16384>, fft_target_encode_kernel<16384, true>)
                   : both(fft_target_sub_kernel<16384>, fft_target_encode_kernel<16384, false>);
    }
    auto launch = [&](auto kern) -> int {
        if ((rc = set_lds(ctx, kern, lds))) return rc;
        hipLaunchKernelGGL(kern, dim3(numPairs), dim3(N / 16), lds, stream, ctx->d_aprime_f, Lpad, L, numCh,
                           ctx->d_fft_tab, ctx->d_aspec, q16 ? ctx->d_aspecscale : nullptr, q16 ? ctx->d_tgtdq : nullptr);
        SGZ_HIP(ctx, hipGetLastError());
        return SGZ_OK;
    };
    if (N == 4096) return q16 ? launch(fft_target_kernel<4096, true>) : launch(fft_target_kernel<4096, false>);
    if (N == 8192) return q16 ? launch(fft_target_kernel<8192, true>) : launch(fft_target_kernel<8192, false>);
    return q16 ? launch(fft_target_kernel<16384, true>) : launch(fft_target_kernel<16384, false>);
}

// The target-dependent factors of the error bound, from the target's statistics (host arithmetic).
void fft_target_factors(sgz_context *ctx)
{
    FftGeom &fg = ctx->fg;
    const int N = fg.N, numCh = ctx->cp.numCh, L = ctx->cp.L;
    const bool q16 = fg.q16 != 0;
    const double nT = (double)L, nS = (double)L * (numCh - 1);
    fg.sumAT = ctx->tgt_sumT;
    fg.sumAS = ctx->tgt_sumS;
    // (a target without spread gives inf / NaN factors, like its similarities)
    // (16-bit spectra: the kernel multiplies the stored target spectrum, whose norm is within sqrt(128) / 65534 of the
    // computed one's -- every component is off by at most half a step of its band, 1 / 65534 of the band's largest,
    // and a band has 128 of them; four more roundings per product: the product of the two scales, its product with
    // the block's integer, and -- on either side -- the rounding of integer times scale in the value whose distance to
    // the computed spectrum was measured, where the kernel's operand is the unrounded product)
    const double u = std::ldexp(1.0, -24), up = q16 ? 1.0003 : 1.0001;
    const int extra = q16 ? 4 : 0;
    fg.epsT = (float)(fft_eps(N) * up);
    fg.epsUT = (float)((1 + 8 + extra) * 3.0 * u * up);            // operand roundings, products, the untangling
    fg.epsUS = (float)((numCh - 1 + 10 + extra) * 3.0 * u * up);   // the same, the channel sums and the scaling by beta
    fg.aT = (float)(ctx->tgt_nrmT * up);
    fg.aS = (float)(ctx->tgt_nrmS * up);
    fg.a01 = (float)(std::sqrt(ctx->tgt_nrmT * ctx->tgt_nrmT + ctx->tgt_nrm1 * ctx->tgt_nrm1) * up);
    fg.aP = (float)(std::sqrt(ctx->tgt_nrmT * ctx->tgt_nrmT + 2.0 * ctx->tgt_nrmS * ctx->tgt_nrmS) * up);
    fg.kT = (float)(up / (ctx->cp.aStdT * nT));
    fg.kS = numCh > 1 ? (float)(up / (ctx->cp.aStdS * nS)) : 0.0f;
    fg.levelErr = ctx->norm.enabled ? (float)std::ldexp(1.0, -23) : 0.0f;
    fg.relErr = (float)std::ldexp(1.0, -22);
}

int fft_target(sgz_context *ctx, hipStream_t stream)
{
    int rc;
    if ((rc = fft_target_launch(ctx, stream, nullptr))) return rc;
    fft_target_factors(ctx);
    ctx->aspec_dirty = false;
    return SGZ_OK;
}

}  // namespace

namespace sgz {

int fft_target_spectra(sgz_context *ctx, hipStream_t stream, const void *centre_args)
{
    return fft_target_launch(ctx, stream, static_cast<const TargetCentreArgs *>(centre_args));
}

int fft_target_centre_workgroups(const sgz_context *ctx, int numCh, int Lpad)
{
    // workgroups of 256 threads (fft_target_sub_kernel), four elements per thread
    const size_t per = (size_t)256 * kTpPer;
    return ctx->fg.N > 0 ? (int)(((size_t)numCh * Lpad + per - 1) / per) : 0;
}

void fft_target_ready(sgz_context *ctx)
{
    fft_target_factors(ctx);
    ctx->aspec_dirty = false;
}

void fft_screen_free(sgz_context *ctx)
{
    void *bufs[] = {ctx->d_fblocks, ctx->d_spec, ctx->d_aspec, ctx->d_fft_tab, ctx->d_blksq, ctx->d_blknorm, ctx->d_blklnmin, ctx->d_blkgu, ctx->d_wrec2, ctx->d_prelist, ctx->d_precount,
                    ctx->d_specscale, ctx->d_blkdq, ctx->d_blkq, ctx->d_aspecscale, ctx->d_tgtdq, ctx->d_tgt_z};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
}

// Transform length for the files and the target of the layout: the one that moves the fewest bytes,
// sum over files of blocks * N, among the lengths whose blocks hold at least N / 4 windows; 0 (not in use) when
// the caller asked for the matrix-core kernel, when no length fits the target, or -- left to itself -- when the
// target is short enough for the matrix cores to be faster.  Then the block table.
int fft_screen_layout(sgz_context *ctx)
{
    ctx->fg = FftGeom{};
    ctx->h_fblocks.clear();
    const int nf = (int)ctx->files.size();
    ctx->fblock_begin.assign((size_t)nf + 1, 0);
    ctx->fc_begin = ctx->fc_end = 0;
    ctx->aspec_dirty = true;
    const int L = ctx->cp.L, numCh = ctx->cp.numCh;
    int mode = ctx->screen_kernel;
    if (const char *e = std::getenv("SGZ_SCREEN_KERNEL")) {
        if (!std::strcmp(e, "matrix")) mode = 1;
        if (!std::strcmp(e, "spectral")) mode = 2;
    }
    if (mode == 1 || ctx->hg.S2 <= 0 || nf == 0) return SGZ_OK;
    if (mode == 0 && L < 1024) return SGZ_OK;
    int bestN = 0;
    double bestCost = 0.0;
    for (int N : {4096, 8192, 16384}) {
        const int hop = N - L + 1;
        if (hop < N / 4) continue;
        double cost = 0.0;
        for (int f = 0; f < nf; f++) {
            const long long w = ctx->h_files[(size_t)f].num_windows;
            cost += (double)((w + hop - 1) / hop) * N;
        }
        if (!bestN || cost < bestCost) {
            bestN = N;
            bestCost = cost;
        }
    }
    if (const char *e = std::getenv("SGZ_FFT_N")) {
        const int N = std::atoi(e);
        if ((N == 4096 || N == 8192 || N == 16384) && N - L + 1 >= 1) bestN = N;
    }
    if (!bestN) return SGZ_OK;
    FftGeom fg{};
    fg.N = bestN;
    fg.hop = bestN - L + 1;
    // 16-bit spectra unless the caller (or the environment, for tests) asks for fp32 ones
    int fmt = ctx->spectra_format;
    if (const char *e = std::getenv("SGZ_SPECTRA")) {
        if (!std::strcmp(e, "f32")) fmt = 1;
        if (!std::strcmp(e, "q16")) fmt = 2;
    }
    fg.q16 = fmt != 1;
    long long specOff = 0;
    for (int f = 0; f < nf; f++) {
        const FileDesc &fd = ctx->h_files[(size_t)f];
        ctx->fblock_begin[(size_t)f] = (int)ctx->h_fblocks.size();
        // The file's windows are dealt evenly over the blocks it needs (not hop, hop, ..., rest): a block costs its
        // spectra and its transform whatever it holds, but its epilogue by the window, and when there are no more
        // blocks than CUs the launch takes as long as its fullest block.
        const long long nblk = (fd.num_windows + fg.hop - 1) / fg.hop;
        const long long per = nblk > 0 ? (fd.num_windows + nblk - 1) / nblk : 0;
        for (long long o0 = 0; o0 < fd.num_windows; o0 += per) {
            FftBlock b;
            b.file = f;
            b.o0 = (int)o0;
            b.valid = (int)std::min<long long>(per, fd.num_windows - o0);
            b.pad = 0;
            b.spec = specOff;
            b.win = fd.win_off + o0;
            specOff += (long long)numCh * (bestN / 2);
            ctx->h_fblocks.push_back(b);
        }
    }
    ctx->fblock_begin[(size_t)nf] = (int)ctx->h_fblocks.size();
    const size_t nb = ctx->h_fblocks.size();
    if (nb == 0) return SGZ_OK;
    int rc;
    if ((rc = ensure_tab(ctx))) return rc;
    if ((rc = ensure(ctx, ctx->d_fblocks, ctx->d_fblocks_cap, nb))) return rc;
    if ((rc = ensure(ctx, ctx->d_spec, ctx->spec_cap, fg.q16 ? ((size_t)specOff + 1) / 2 : (size_t)specOff))) return rc;
    if (fg.q16) {
        if ((rc = ensure(ctx, ctx->d_specscale, ctx->specscale_cap, nb * (size_t)numCh * (bestN / 128)))) return rc;
        if ((rc = ensure(ctx, ctx->d_blkdq, ctx->blkdq_cap, nb * (size_t)numCh))) return rc;
        if ((rc = ensure(ctx, ctx->d_blkq, ctx->blkq_cap, nb))) return rc;
    }
    if ((rc = ensure(ctx, ctx->d_blksq, ctx->blksq_cap, nb * (size_t)numCh))) return rc;
    if ((rc = ensure(ctx, ctx->d_blknorm, ctx->blknorm_cap, nb))) return rc;
    if ((rc = ensure(ctx, ctx->d_blklnmin, ctx->blklnmin_cap, nb))) return rc;
    if ((rc = ensure(ctx, ctx->d_blkgu, ctx->blkgu_cap, nb))) return rc;
    if ((rc = ensure(ctx, ctx->d_wrec2, ctx->wrec2_cap, (size_t)std::max<long long>(ctx->total_windows, 1)))) return rc;
    SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_fblocks, ctx->h_fblocks.data(), sizeof(FftBlock) * nb, hipMemcpyHostToDevice,
                                ctx->stream));
    ctx->fg = fg;
    return SGZ_OK;
}

// Spectra and block norms of files [file_begin, file_end): after the half-precision preparation of the same
// files (it writes the centring constants this reads from the file table).
int fft_screen_prepare(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream)
{
    const int N = ctx->fg.N, numCh = ctx->cp.numCh;
    const bool q16 = ctx->fg.q16 != 0;
    const int b0 = ctx->fblock_begin[(size_t)file_begin], b1 = ctx->fblock_begin[(size_t)file_end];
    if (b1 == b0) return SGZ_OK;
    const int numPairs = (numCh + 1) / 2;
    const size_t lds = sizeof(float2) * fft_lds_elems(N);
    int rc = SGZ_OK;
    auto launch = [&](auto kern) -> int {
        if ((rc = set_lds(ctx, kern, lds))) return rc;
        const int units = (b1 - b0) * numPairs;
        hipLaunchKernelGGL(kern, dim3((unsigned)((units + 7) & ~7)), dim3(N / 16), lds, stream, ctx->d_files,
                           ctx->d_fblocks + b0, numPairs, units, ctx->norm, numCh, ctx->d_fft_tab, ctx->d_spec,
                           ctx->d_blksq + (size_t)b0 * numCh,
                           q16 ? ctx->d_specscale + (size_t)b0 * numCh * (N / 128) : nullptr,
                           q16 ? ctx->d_blkdq + (size_t)b0 * numCh : nullptr);
        SGZ_HIP(ctx, hipGetLastError());
        return SGZ_OK;
    };
    if (N == 4096) rc = q16 ? launch(fft_spectra_kernel<4096, true>) : launch(fft_spectra_kernel<4096, false>);
    else if (N == 8192) rc = q16 ? launch(fft_spectra_kernel<8192, true>) : launch(fft_spectra_kernel<8192, false>);
    else rc = q16 ? launch(fft_spectra_kernel<16384, true>) : launch(fft_spectra_kernel<16384, false>);
    if (rc) return rc;
    hipLaunchKernelGGL(fft_blknorm_kernel, dim3((b1 - b0 + 255) / 256), dim3(256), 0, stream,
                       ctx->d_blksq + (size_t)b0 * numCh, numCh, b1 - b0, ctx->d_blknorm + b0,
                       q16 ? ctx->d_blkdq + (size_t)b0 * numCh : nullptr, (float)(std::sqrt(2.0 / N) * 1.0001),
                       q16 ? ctx->d_blkq + b0 : nullptr);
    SGZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(fft_blklnmin_kernel, dim3(b1 - b0), dim3(256), 0, stream, ctx->d_fblocks + b0, ctx->d_wstat_h,
                       ctx->total_windows, ctx->d_blklnmin + b0);
    SGZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(fft_blkgu_kernel, dim3(b1 - b0), dim3(256), 0, stream, ctx->d_fblocks + b0, ctx->d_wstat_h,
                       ctx->total_windows, ctx->d_blkgu + b0, ctx->d_wrec2);
    SGZ_HIP(ctx, hipGetLastError());
    ctx->fc_begin = file_begin;
    ctx->fc_end = file_end;
    return SGZ_OK;
}

int boost_materialize(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream, bool *launched)
{
    if (launched) *launched = false;
    const int nf = (int)std::min(ctx->boost_lazy.size(), ctx->h_files.size());
    for (int f = std::max(file_begin, 0); f < std::min(file_end, nf);) {
        if (!ctx->boost_lazy[(size_t)f]) {
            f++;
            continue;
        }
        int e = f;
        while (e < std::min(file_end, nf) && ctx->boost_lazy[(size_t)e]) ctx->boost_lazy[(size_t)e++] = 0;
        const long long w0 = ctx->h_files[(size_t)f].win_off;
        const long long w1 = ctx->h_files[(size_t)e - 1].win_off + ctx->h_files[(size_t)e - 1].num_windows;
        if (w1 > w0) {
            const int grid = (int)std::min<long long>((w1 - w0 + 255) / 256, 2048);
            hipLaunchKernelGGL(boost_fill_kernel, dim3(grid), dim3(256), 0, stream, ctx->d_wstat_h, ctx->total_windows, w0,
                               w1 - w0, (float)ctx->boost_lazy_ln, ctx->d_boost);
            SGZ_HIP(ctx, hipGetLastError());
            if (launched) *launched = true;
        }
        f = e;
    }
    return SGZ_OK;
}

int require_dense(sgz_context *ctx, int file_begin, int file_end)
{
    const int nf = (int)ctx->no_dense.size();
    for (int f = std::max(file_begin, 0); f < std::min(file_end, nf); f++)
        if (ctx->no_dense[(size_t)f])
            return ctx->fail(SGZ_E_STATE, "the latest screening run over these files kept candidate lists only "
                                          "(sgz_corr_set_screen_output): no per-window results to read");
    return SGZ_OK;
}

int boost_source(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream, const float **lnAvg)
{
    *lnAvg = nullptr;
    const int nf = (int)ctx->boost_lazy.size();
    int lazy = 0, total = 0;
    for (int f = std::max(file_begin, 0); f < std::min(file_end, nf); f++) {
        if (ctx->h_files[(size_t)f].num_windows <= 0) continue;  // (no window reads anything)
        total++;
        lazy += ctx->boost_lazy[(size_t)f] ? 1 : 0;
    }
    if (lazy == 0) return SGZ_OK;
    if (lazy == total && ctx->fg.N > 0 && ctx->boost_lazy_ln == ctx->cp.lnAvgIn) {
        *lnAvg = ctx->d_wstat_h + spectral_statics_ln_offset(ctx->total_windows);
        return SGZ_OK;
    }
    return boost_materialize(ctx, file_begin, file_end, stream);
}

int fft_screen_launch(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream, hipEvent_t start,
                      hipEvent_t stop)
{
    const int N = ctx->fg.N;
    if (N <= 0) return ctx->fail(SGZ_E_STATE, "spectral screening is not set up for this layout");
    int rc;
    if (ctx->aspec_dirty && (rc = fft_target(ctx, stream))) return rc;
    // this run's boosts stay implicit; files that still owe theirs to a run with another target get them now
    if (ctx->boost_lazy.size() != ctx->files.size()) ctx->boost_lazy.assign(ctx->files.size(), 0);
    if (ctx->boost_lazy_ln != ctx->cp.lnAvgIn) {
        if ((rc = boost_materialize(ctx, 0, file_begin, stream))) return rc;
        if ((rc = boost_materialize(ctx, file_end, (int)ctx->files.size(), stream))) return rc;
        ctx->boost_lazy_ln = ctx->cp.lnAvgIn;
    }
    for (int f = file_begin; f < file_end; f++) ctx->boost_lazy[(size_t)f] = 1;
    // dense upper bounds, or the list of the windows that can hold their file's maximum
    if (ctx->no_dense.size() != ctx->files.size()) ctx->no_dense.assign(ctx->files.size(), 0);
    PreList pre{nullptr, nullptr, 0};
    const bool lists = ctx->screen_output == 1;
    if (lists) {
        const size_t cap = std::max<size_t>(8192, 16 * ctx->h_fblocks.size());
        if ((rc = ensure(ctx, ctx->d_prelist, ctx->prelist_cap, cap))) return rc;
        const unsigned long long *before = ctx->d_precount;
        if ((rc = ensure(ctx, ctx->d_precount, ctx->precount_cap, (size_t)2))) return rc;
        if (ctx->d_precount != before) ctx->pre_clean[0] = ctx->pre_clean[1] = false;
        const int pp = ctx->pre_parity;
        if (!ctx->pre_clean[pp])
            SGZ_HIP(ctx, hipMemsetAsync(ctx->d_precount + pp, 0, sizeof(unsigned long long), stream));
        ctx->pre_clean[pp] = false;
        ctx->pre_run_parity = pp;
        ctx->pre_capacity = (long long)ctx->prelist_cap;
        pre = PreList{ctx->d_prelist, ctx->d_precount + pp, ctx->pre_capacity};
        ctx->pre_valid = true;
        ctx->pre_fb = file_begin;
        ctx->pre_fe = file_end;
    } else if (ctx->pre_valid && file_begin < ctx->pre_fe && file_end > ctx->pre_fb) {
        ctx->pre_valid = false;
    }
    for (int f = file_begin; f < file_end; f++) ctx->no_dense[(size_t)f] = lists ? 1 : 0;
    float *simOut = lists ? nullptr : ctx->d_sim;
    const FftGeom &fg = ctx->fg;
    const int b0 = ctx->fblock_begin[(size_t)file_begin], b1 = ctx->fblock_begin[(size_t)file_end];
    // per-file bounds: two alternating arrays, as in the matrix-core launch (half_screen_launch)
    const int nf = (int)ctx->files.size();
    const int par = ctx->lokey_parity;
    int *lokey = ctx->d_lokey + (size_t)par * nf, *lokeyNext = ctx->d_lokey + (size_t)(1 - par) * nf;
    if (!ctx->lokey_clean[par])
        SGZ_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(lokey), (int)0x807fffff, (size_t)nf, stream));
    const bool whole = b0 == 0 && b1 == (int)ctx->h_fblocks.size();
    ctx->lokey_clean[par] = false;
    ctx->lokey_clean[1 - par] = whole || ctx->lokey_clean[1 - par];
    ctx->lokey_cur = lokey;
    ctx->lokey_parity = 1 - par;
    if (b1 == b0) return SGZ_OK;
    const size_t lds = sizeof(float2) * fft_lds_elems(N);
    const bool q16 = fg.q16 != 0;
    const float *scales = q16 ? ctx->d_specscale + (size_t)b0 * ctx->cp.numCh * (N / 128) : nullptr;
    const float2 *blkq = q16 ? ctx->d_blkq + b0 : nullptr;
    auto launch = [&](auto kern) -> int {
        if ((rc = set_lds(ctx, kern, lds))) return rc;
        if (start || stop)
            hipExtLaunchKernelGGL(kern, dim3(b1 - b0), dim3(N / 16), lds, stream, start, stop, 0, ctx->d_files, ctx->d_fblocks + b0,
                                  ctx->d_spec, ctx->d_aspec, ctx->d_fft_tab, ctx->d_blknorm + b0, scales, blkq, ctx->d_aspecscale, ctx->d_tgtdq, ctx->d_blklnmin + b0, ctx->d_blkgu + b0, ctx->d_wrec2,
                                  ctx->d_wstat_h, ctx->total_windows, ctx->cp, fg, simOut, (float *)nullptr,
                                  lokey, lokeyNext, ctx->d_hflags, pre);
        else
            hipLaunchKernelGGL(kern, dim3(b1 - b0), dim3(N / 16), lds, stream, ctx->d_files, ctx->d_fblocks + b0, ctx->d_spec,
                               ctx->d_aspec, ctx->d_fft_tab, ctx->d_blknorm + b0, scales, blkq, ctx->d_aspecscale, ctx->d_tgtdq, ctx->d_blklnmin + b0, ctx->d_blkgu + b0, ctx->d_wrec2,
                               ctx->d_wstat_h, ctx->total_windows, ctx->cp, fg, simOut, (float *)nullptr,
                               lokey, lokeyNext, ctx->d_hflags, pre);
        SGZ_HIP(ctx, hipGetLastError());
        return SGZ_OK;
    };
    const bool std14 = ctx->cp.numCh == 14;
    if (q16) {
        if (N == 4096) return std14 ? launch(corr_fft_kernel<4096, 14, true>) : launch(corr_fft_kernel<4096, 0, true>);
        if (N == 8192) return std14 ? launch(corr_fft_kernel<8192, 14, true>) : launch(corr_fft_kernel<8192, 0, true>);
        return std14 ? launch(corr_fft_kernel<16384, 14, true>) : launch(corr_fft_kernel<16384, 0, true>);
    }
    if (N == 4096) return std14 ? launch(corr_fft_kernel<4096, 14, false>) : launch(corr_fft_kernel<4096, 0, false>);
    if (N == 8192) return std14 ? launch(corr_fft_kernel<8192, 14, false>) : launch(corr_fft_kernel<8192, 0, false>);
    return std14 ? launch(corr_fft_kernel<16384, 14, false>) : launch(corr_fft_kernel<16384, 0, false>);
}

}  // namespace sgz

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int sgz_corr_set_screen_kernel(sgz_context *ctx, int kernel)
{
    if (!ctx) return SGZ_E_ARG;
    if (kernel < 0 || kernel > 2) return ctx->fail(SGZ_E_ARG, "screen kernel: 0 (auto), 1 (matrix cores) or 2 (spectral)");
    if (kernel != ctx->screen_kernel) ctx->layout_dirty = true;
    ctx->screen_kernel = kernel;
    return SGZ_OK;
}

int sgz_corr_set_spectra_format(sgz_context *ctx, int format)
{
    if (!ctx) return SGZ_E_ARG;
    if (format < 0 || format > 2) return ctx->fail(SGZ_E_ARG, "spectra format: 0 (auto), 1 (fp32) or 2 (16-bit integers)");
    if (format != ctx->spectra_format) ctx->layout_dirty = true;
    ctx->spectra_format = format;
    return SGZ_OK;
}

int sgz_corr_spectra_format_in_use(const sgz_context *ctx)
{
    if (!ctx || !ctx->have_target || ctx->layout_dirty) return -1;
    if (ctx->fg.N <= 0) return 0;
    return ctx->fg.q16 ? 2 : 1;
}

int sgz_corr_set_screen_output(sgz_context *ctx, int lists_only)
{
    if (!ctx) return SGZ_E_ARG;
    if (lists_only != 0 && lists_only != 1) return ctx->fail(SGZ_E_ARG, "screen output: 0 (per-window results) or 1 (candidate lists)");
    ctx->screen_output = lists_only;
    return SGZ_OK;
}

int sgz_corr_screen_kernel_in_use(const sgz_context *ctx)
{
    if (!ctx || !ctx->have_target || ctx->layout_dirty) return -1;
    if (ctx->fg.N > 0) return 2;
    return ctx->hg.S2 > 0 ? 1 : 0;
}

int sgz_corr_screen_geometry(const sgz_context *ctx, int *n_out, int *hop_out, long long *blocks_out)
{
    if (!ctx) return SGZ_E_ARG;
    const bool on = ctx->have_target && !ctx->layout_dirty && ctx->fg.N > 0;
    if (n_out) *n_out = on ? ctx->fg.N : 0;
    if (hop_out) *hop_out = on ? ctx->fg.hop : 0;
    if (blocks_out) *blocks_out = on ? (long long)ctx->h_fblocks.size() : 0;
    return SGZ_OK;
}

// The transform the kernels run, thread by thread on the host (same index arithmetic, same fp32 operations).
int sgz_fft_reference_host(int n, int inverse, const float *in, float *out)
{
    if (!in || !out || (n != 4096 && n != 8192 && n != 16384)) return SGZ_E_ARG;
    std::vector<float2> tab(kFftTabAll);
    fft_fill_table(tab.data());
    std::vector<float2> buf((size_t)fft_lds_elems(n));
    for (int i = 0; i < n; i++) buf[(size_t)fft_phys(i)] = make_float2(in[2 * i], in[2 * i + 1]);
    auto run = [&](auto N_, auto DIR_) {
        constexpr int N = decltype(N_)::value, DIR = decltype(DIR_)::value, T = N / 16;
        std::vector<float2> regs((size_t)T * 16);
        auto pass = [&](auto Ns_) {
            constexpr int Ns = decltype(Ns_)::value;
            for (int t = 0; t < T; t++) {
                float2(&v)[16] = *reinterpret_cast<float2(*)[16]>(&regs[(size_t)t * 16]);
                fft16_load<N>(buf.data(), t, v);
                fft16_math<DIR, Ns>(v, t, tab.data());
            }
            for (int t = 0; t < T; t++) {
                float2(&v)[16] = *reinterpret_cast<float2(*)[16]>(&regs[(size_t)t * 16]);
                fft16_store<Ns>(buf.data(), t, v);
            }
        };
        pass(std::integral_constant<int, 1>{});
        pass(std::integral_constant<int, 16>{});
        pass(std::integral_constant<int, 256>{});
        if constexpr (N > 4096)
            for (int j = 0; j < 4096; j++) fft_last<N, DIR>(buf.data(), j, tab.data());
    };
    using I = std::integral_constant<int, 1>;
    using F = std::integral_constant<int, -1>;
    if (n == 4096) inverse ? run(std::integral_constant<int, 4096>{}, I{}) : run(std::integral_constant<int, 4096>{}, F{});
    else if (n == 8192) inverse ? run(std::integral_constant<int, 8192>{}, I{}) : run(std::integral_constant<int, 8192>{}, F{});
    else inverse ? run(std::integral_constant<int, 16384>{}, I{}) : run(std::integral_constant<int, 16384>{}, F{});
    for (int i = 0; i < n; i++) {
        out[2 * i] = buf[(size_t)fft_phys(i)].x;
        out[2 * i + 1] = buf[(size_t)fft_phys(i)].y;
    }
    return SGZ_OK;
}

}  // extern "C"
