// FeatureCorrelation on gfx950: database preparation, the sliding correlation
// kernel, candidate compaction and the synthetic-data generator.
//
// Arithmetic follows the reference's impl/MathUtil.scala (stat :29-62,
// correlate :177-196, avg :109-118, normalize :132-152) and
// impl/FeatureCorrelationImpl.scala (calcBoost :73-78, window loop :190-210):
// fp32 normalisation, fp64 sums, fp32 results, fp32 weight mix.
//
// Data flow per sgz_corr_run():
//   prep_kernel   raw interleaved fp32 frames -> normalised, centred fp64 in a
//                 "chunk-major" planar layout [ch][j][m] (frame = 16*m + j) plus
//                 per-frame first / second moments of the loudness channel and of
//                 the MFCC channels
//                 and their prefix sums inside each 1024-frame tile
//   tile_scan_kernel  exclusive scan of the tile totals per file (window mean /
//                 std-dev become two subtractions of prefix values)
//   corr_kernel   cross terms sum_i a'[c][i] * b[c][o+i] for 1024 windows per
//                 workgroup, fp64 FMA, then boost / similarity per window
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include <type_traits>

#include "sgz_corr_dev.hpp"
#include "sgz_internal.hpp"
#include "sgz_target_dev.hpp"

using namespace sgz;

namespace {

// ---------------------------------------------------------------------------
// Synthetic feature frames: must produce the same bits as the C and numpy
// generators used by the tests (counter-based hash, 8-tap moving average summed
// in index order, fp32 without contraction).
// ---------------------------------------------------------------------------
__host__ __device__ inline unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline float synth_u(unsigned long long seed, long long file, long long frame, int ch,
                                         unsigned long long stream)
{
    const unsigned long long k =
        mix64(seed + (unsigned long long)file * 0xD1342543DE82EF95ull + stream * 0xA0761D6478BD642Full);
    const unsigned long long h = mix64(k ^ ((unsigned long long)frame * 16ull + (unsigned long long)ch));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}
__host__ __device__ inline float synth_base(unsigned long long seed, long long file, long long frame, int ch)
{
    float s = 0.0f;
    for (int j = 0; j < 8; j++) s = s + synth_u(seed, file, frame + j, ch, 0);
    s = s * 0.125f;
    if (ch == 0) return 0.05f + 1.45f * s;
    return s;
}

__global__ void synth_kernel(float *out, unsigned long long seed, long long file, long long frames, int numCh,
                             const sgz_plant *plants, int numPlants)
{
    const long long total = frames * numCh;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long fr = e / numCh;
        const int ch = (int)(e - fr * numCh);
        int hit = -1;
        for (int k = 0; k < numPlants; k++) {
            if (plants[k].file == file && fr >= plants[k].offset && fr < plants[k].offset + plants[k].length) {
                hit = k;
                break;
            }
        }
        float v;
        if (hit >= 0) {
            const sgz_plant p = plants[hit];
            const float b = synth_base(seed, p.src_file, p.src_offset + (fr - p.offset), ch);
            const float nz = synth_u(seed, file, fr, ch, 1) - 0.5f;
            v = b + p.amp * nz;
            if (ch == 0) v = v * p.gain;
        } else {
            v = synth_base(seed, file, fr, ch);
        }
        out[e] = v;
    }
}

// ---------------------------------------------------------------------------
// prep_kernel: one workgroup per `cb` chunks (cb*16 frames) of one file.
// LDS holds the normalised fp32 values as xs[ch][j][mm] with row strides chosen
// so that both the frame-major fill and the chunk-major drain are (nearly)
// conflict free.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prep_kernel(const FileDesc *__restrict__ files,
                                                   const PrepTile *__restrict__ ptiles,
                                                   T *__restrict__ planar, double *__restrict__ prefix,
                                                   double *__restrict__ tbase, NormParams nrm, int numCh, int cb,
                                                   double kT, double kS, int *__restrict__ hflags)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ double wtot[4][4];
    const PrepTile pt = ptiles[blockIdx.x];
    const FileDesc fd = files[pt.file];
    if constexpr (std::is_same<T, _Float16>::value) {  // the half-precision image is centred per file
        kT = (double)fd.cT;
        kS = (double)fd.cS;
    }
    const int jstride = cb + 1;
    const int cstride = kR * jstride + 1;
    const long long f0 = (long long)pt.chunk0 * kR;
    const int tileFrames = cb * kR;
    const int tid = threadIdx.x;

    // phase 1: coalesced read of the interleaved frames, normalise in fp32.
    // (fr, c) of element e = tid + 256*k advance by a fixed (dfr, dc) per k.
    const int total = tileFrames * numCh;
    const float *src = fd.raw + f0 * numCh;
    const long long validElems = (fd.frames - f0) * numCh;  // may be <= 0 for padding tiles
    {
        // 16-byte loads, four per thread in flight; (fr, c) of a vector's first element by one division
        const int nvec = total >> 2;  // total is a multiple of 64
        for (int v0 = tid; v0 < nvec; v0 += 4 * 256) {
            float4 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int v = v0 + u * 256;
                const long long e0 = 4ll * v;
                x[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (v < nvec) {
                    if (e0 + 3 < validElems) {
                        x[u] = *reinterpret_cast<const float4 *>(src + e0);
                    } else {  // the file ends inside this vector
                        if (e0 < validElems) x[u].x = src[e0];
                        if (e0 + 1 < validElems) x[u].y = src[e0 + 1];
                        if (e0 + 2 < validElems) x[u].z = src[e0 + 2];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int v = v0 + u * 256;
                if (v < nvec) {
                    const int e0 = 4 * v;
                    int fr = e0 / numCh, c = e0 - fr * numCh;
                    const float xv[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float val = xv[j];
                        if (nrm.enabled && e0 + j < validElems) {
                            // (the reference's fp32 quotient, impl/MathUtil.scala:132-152, for every element type: one
                            // arithmetic for the normalised value wherever it is formed)
                            val = (val - nrm.mn[c]) / nrm.d[c];
                        }
                        xs[c * cstride + (fr & (kR - 1)) * jstride + (fr >> 4)] = val;
                        if (++c == numCh) {
                            c = 0;
                            fr++;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // phase 2: planar image, centred; frames past the end are 0
    T *dst = planar + fd.planar_off;
    const int rows = numCh * kR;
    if constexpr (std::is_same<T, _Float16>::value) {
        // half precision: natural frame order [numCh][16 M], 8 frames (16 bytes) per thread
        using half8 = __attribute__((ext_vector_type(8))) _Float16;
        const long long rowLen = fd.M * kR;
        const int vecPerRow = tileFrames / 8;
        bool outOfRange = false;
        for (int idx = tid; idx < numCh * vecPerRow; idx += 256) {
            const int c = idx / vecPerRow, v = idx - c * vecPerRow;
            const long long fr0 = f0 + 8 * v;
            if (fr0 < rowLen) {
                const float k = (float)(c == 0 ? kT : kS);
                half8 out;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int fr = 8 * v + e;
                    float val = 0.0f;
                    if (f0 + fr < fd.frames) val = xs[c * cstride + (fr & (kR - 1)) * jstride + (fr >> 4)] - k;
                    outOfRange |= !(fabsf(val) <= 60000.0f) && val == val;  // NaN stays NaN, as in the exact path
                    out[e] = (_Float16)val;
                }
                *reinterpret_cast<half8 *>(dst + (long long)c * rowLen + fr0) = out;
            }
        }
        if (outOfRange) atomicOr(&hflags[pt.file], 1);
    } else {
        // idx = tid + 256*k over [rows][cb]; cb is a power of two
        const int cbShift = 31 - __clz(cb);
        for (int idx = tid; idx < rows * cb; idx += 256) {
            const int mm = idx & (cb - 1);
            const int row = idx >> cbShift;
            const int c = row >> 4;
            const int j = row & (kR - 1);
            const long long m = (long long)pt.chunk0 + mm;
            if (m < fd.M) {
                const long long fr = m * kR + j;
                double v = 0.0;
                if (fr < fd.frames) v = (double)xs[c * cstride + j * jstride + mm] - (c == 0 ? kT : kS);
                dst[(long long)row * fd.M + m] = (T)v;
            }
        }
    }

    // phase 3: per-frame moments (loudness: x, x^2; MFCC rows: sum x, sum x^2, all centred) and their
    // inclusive prefix sums inside this tile; the tile totals go to tbase for the cross-tile scan.
    if (f0 < fd.frames) {
        const int per = tileFrames >= 256 ? tileFrames / 256 : 1;  // consecutive frames per thread
        double run[4] = {0.0, 0.0, 0.0, 0.0};
        double m[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int fr = tid * per + k;
            const long long gf = f0 + fr;
            m[k][0] = m[k][1] = m[k][2] = m[k][3] = 0.0;
            if (k < per && fr < tileFrames && gf < fd.frames) {
                const int j = fr & (kR - 1), mm = fr >> 4;
                double t, s1, s2;
                if constexpr (std::is_same<T, _Float16>::value) {
                    // screening statistics: the 13-term sums of one frame in fp32, the running sums in fp64
                    const float kSf = (float)kS;
                    float f1 = 0.0f, f2 = 0.0f;
                    for (int c = 1; c < numCh; c++) {
                        const float d = xs[c * cstride + j * jstride + mm] - kSf;
                        f1 += d;
                        f2 = fmaf(d, d, f2);
                    }
                    t = (double)(xs[j * jstride + mm] - (float)kT);
                    s1 = (double)f1;
                    s2 = (double)f2;
                } else {
                    t = (double)xs[j * jstride + mm] - kT;
                    s1 = 0.0;
                    s2 = 0.0;
                    for (int c = 1; c < numCh; c++) {
                        const double d = (double)xs[c * cstride + j * jstride + mm] - kS;
                        s1 += d;
                        s2 = fma(d, d, s2);
                    }
                }
                m[k][0] = t;
                m[k][1] = t * t;
                m[k][2] = s1;
                m[k][3] = s2;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) run[q] += m[k][q];
        }
        // exclusive scan of run[] over the 256 threads
        double inc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) inc[q] = run[q];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double up = __shfl_up(inc[q], off, 64);
                if ((tid & 63) >= off) inc[q] += up;
            }
        }
        if ((tid & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 4; q++) wtot[tid >> 6][q] = inc[q];
        }
        __syncthreads();
        double base[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double bsum = 0.0;
            for (int wv = 0; wv < (tid >> 6); wv++) bsum += wtot[wv][q];
            base[q] = bsum + inc[q] - run[q];
        }
        double *pre = prefix + fd.prefix_off;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int fr = tid * per + k;
            const long long gf = f0 + fr;
            if (k < per && fr < tileFrames && gf < fd.frames) {
#pragma unroll
                for (int q = 0; q < 4; q++) base[q] += m[k][q];
                if constexpr (std::is_same<T, _Float16>::value) {
                    // screening: one fp32 record per frame (its bounds account for that rounding); `prefix` is
                    // the record buffer of the half-precision path here
                    (reinterpret_cast<float4 *>(prefix) + fd.prefix_off / 4)[gf] =
                        make_float4((float)base[0], (float)base[1], (float)base[2], (float)base[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) pre[q * fd.frames + gf] = base[q];
                }
            }
        }
        if (tid == 255) {
            const int tileIdx = pt.chunk0 / cb;
#pragma unroll
            for (int q = 0; q < 4; q++)
                tbase[fd.tbase_off + (long long)q * fd.nptiles + tileIdx] = wtot[0][q] + wtot[1][q] + wtot[2][q] + inc[q];
        }
    }
}

// ---------------------------------------------------------------------------
// prep_half_kernel: the half-precision preparation (same outputs as prep_kernel<_Float16>) for a channel
// count known at compile time.  A workgroup takes 512 frames: coalesced 16-byte loads into a linear LDS
// copy, then every thread owns two consecutive frames in registers -- normalisation, centring, the
// frame moments and the half conversion need no further LDS traffic or address arithmetic, the norms are
// scalar operands, and the 256-thread scan of the moments runs on the cross-lane data path.
// ---------------------------------------------------------------------------
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_add(double v)
{
    // lanes without a source (and rows switched off by the mask) add 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_inclusive_scan(double v)
{
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

constexpr int kPrepHalfFrames = 512;  // frames per workgroup of prep_half_kernel: two per thread

template <int NCH>
__global__ __launch_bounds__(256) void prep_half_kernel(const FileDesc *__restrict__ files,
                                                        const PrepTile *__restrict__ ptiles,
                                                        _Float16 *__restrict__ planar, float4 *__restrict__ prefix,
                                                        double *__restrict__ tbase, NormParams nrm,
                                                        int *__restrict__ hflags)
{
    static_assert((2 * NCH) % 4 == 0, "two frames must be whole 16-byte vectors");
    constexpr int kVecPerThread = 2 * NCH / 4;
    __shared__ __attribute__((aligned(16))) float raw[kPrepHalfFrames * NCH];
    __shared__ double wtot[4][4];
    const PrepTile pt = ptiles[blockIdx.x];
    const FileDesc fd = files[pt.file];
    const long long f0 = (long long)pt.chunk0 * kR;
    const int tid = threadIdx.x;
    const float *src = fd.raw + f0 * NCH;
    const long long validElems = (fd.frames - f0) * NCH;  // may be <= 0 for padding tiles

    // phase 1: the tile's frames as they lie in memory
    {
        float4 x[kVecPerThread];
        if (validElems >= (long long)kPrepHalfFrames * NCH) {
#pragma unroll
            for (int u = 0; u < kVecPerThread; u++)
                x[u] = *reinterpret_cast<const float4 *>(src + 4 * (tid + 256 * u));
        } else {  // the file ends inside this tile
#pragma unroll
            for (int u = 0; u < kVecPerThread; u++) {
                const long long e0 = 4ll * (tid + 256 * u);
                x[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (e0 + 3 < validElems) {
                    x[u] = *reinterpret_cast<const float4 *>(src + e0);
                } else {
                    if (e0 < validElems) x[u].x = src[e0];
                    if (e0 + 1 < validElems) x[u].y = src[e0 + 1];
                    if (e0 + 2 < validElems) x[u].z = src[e0 + 2];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kVecPerThread; u++) *reinterpret_cast<float4 *>(raw + 4 * (tid + 256 * u)) = x[u];
    }
    __syncthreads();

    // phase 2: frames 2 tid and 2 tid + 1 of the tile
    float v[2 * NCH];
#pragma unroll
    for (int u = 0; u < kVecPerThread; u++) {
        const float4 q = *reinterpret_cast<const float4 *>(raw + 2 * NCH * tid + 4 * u);
        v[4 * u] = q.x;
        v[4 * u + 1] = q.y;
        v[4 * u + 2] = q.z;
        v[4 * u + 3] = q.w;
    }
    const long long gf0 = f0 + 2 * tid;
    const bool ok[2] = {gf0 < fd.frames, gf0 + 1 < fd.frames};
    const float kTf = fd.cT, kSf = fd.cS;  // the file's own mean levels (file_const_kernel)
    float peak = 0.0f;
    float mom[2][4];
    // the reference's fp32 quotient (impl/MathUtil.scala:132-152), as the exact path and the spectra form it: this
    // kernel runs once per layout; (x - 0) / 1 leaves x as it is where normalisation is off
    float mn[NCH], dv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        mn[c] = nrm.enabled ? nrm.mn[c] : 0.0f;
        dv[c] = nrm.enabled ? nrm.d[c] : 1.0f;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float f1 = 0.0f, f2 = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            float val = (v[k * NCH + c] - mn[c]) / dv[c];
            val = ok[k] ? val - (c == 0 ? kTf : kSf) : 0.0f;  // frames past the end are 0
            peak = fmaxf(peak, fabsf(val));                   // NaN stays NaN, as in the exact path
            if (c > 0) {
                f1 += val;
                f2 = fmaf(val, val, f2);
            }
            v[k * NCH + c] = val;
        }
        mom[k][0] = v[k * NCH];
        mom[k][2] = f1;
        mom[k][3] = f2;
    }
    const long long rowLen = fd.M * kR;
    if (gf0 < rowLen) {
        using half2v = __attribute__((ext_vector_type(2))) _Float16;
        _Float16 *dst = planar + fd.planar_off + gf0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            half2v h;
            h[0] = (_Float16)v[c];
            h[1] = (_Float16)v[NCH + c];
            *reinterpret_cast<half2v *>(dst + (long long)c * rowLen) = h;
        }
    }
    if (peak > 60000.0f) atomicOr(&hflags[pt.file], 1);

    // phase 3: frame moments (loudness: x, x^2; MFCC rows: sum x, sum x^2, all centred; the 13-term sums of
    // one frame in fp32, the running sums in fp64), inclusive prefix sums inside this tile; the tile totals go
    // to tbase for the cross-tile scan.
    if (f0 < fd.frames) {
        double m[2][4], run[4], inc[4];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const double t = (double)mom[k][0];
            m[k][0] = t;
            m[k][1] = t * t;
            m[k][2] = (double)mom[k][2];
            m[k][3] = (double)mom[k][3];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            run[q] = m[0][q] + m[1][q];
            inc[q] = wave_inclusive_scan(run[q]);
        }
        if ((tid & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 4; q++) wtot[tid >> 6][q] = inc[q];
        }
        __syncthreads();
        // one fp32 record of the four running sums per frame (the screening bounds account for that rounding)
        float4 *pre = prefix + fd.prefix_off / 4;  // (prefix_off counts four values per frame)
        double base[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            base[q] = 0.0;
            for (int wv = 0; wv < (tid >> 6); wv++) base[q] += wtot[wv][q];
            base[q] += inc[q] - run[q];
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
#pragma unroll
            for (int q = 0; q < 4; q++) base[q] += m[k][q];
            if (ok[k]) pre[gf0 + k] = make_float4((float)base[0], (float)base[1], (float)base[2], (float)base[3]);
        }
        if (tid == 255) {
            const int tileIdx = pt.chunk0 / (kPrepHalfFrames / kR);
#pragma unroll
            for (int q = 0; q < 4; q++)
                tbase[fd.tbase_off + (long long)q * fd.nptiles + tileIdx] = wtot[0][q] + wtot[1][q] + wtot[2][q] + inc[q];
        }
    }
}

// ---------------------------------------------------------------------------
// file_const_kernel: the centring constants of a file's half-precision image -- the mean level of the
// normalised loudness channel and of all other normalised channels -- written into the file table on the
// device.  Any constant gives correct bounds (the target image is centred on its own mean, so the constant
// drops out of the cross term); one near the file's mean keeps the half-precision values, and with them the
// error bound, small.  So a sample is enough: the first `rows` (up to 64) frames of every block of 256, read as
// one contiguous run by a workgroup per file, thread t on channel t % numCh, eight loads in flight; sums in a
// fixed order.  A file whose mean is not finite gets 0: its bounds are NaN then and the exact path takes over.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void file_const_kernel(FileDesc *__restrict__ files, int fileBegin, NormParams nrm,
                                                          int numCh)
{
    __shared__ double part[1024];
    __shared__ double chan[SGZ_MAX_CHANNELS];
    FileDesc &fd = files[fileBegin + blockIdx.x];
    const int tid = threadIdx.x;
    const int rows = min(64, 1024 / numCh), used = rows * numCh;
    const long long frames = fd.frames, blockVals = 256ll * numCh;
    const long long nblocks = (frames + 255) / 256;
    double s = 0.0;
    if (tid < used) {
        const int c = tid % numCh, r = tid / numCh;
        const float mn = nrm.enabled ? nrm.mn[c] : 0.0f, rcp = nrm.enabled ? 1.0f / nrm.d[c] : 1.0f;
        const float *src = fd.raw + tid;
        for (long long b0 = 0; b0 < nblocks; b0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const long long b = b0 + u;
                v[u] = (b < nblocks && b * 256 + r < frames) ? src[b * blockVals] : mn;  // (mn adds 0)
            }
#pragma unroll
            for (int u = 0; u < 8; u++) s += (double)((v[u] - mn) * rcp);
        }
    }
    part[tid] = s;
    __syncthreads();
    if (tid < numCh) {
        double t = 0.0;
        for (int k = tid; k < used; k += numCh) t += part[k];
        chan[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double sp = 0.0;
        for (int c = 1; c < numCh; c++) sp += chan[c];
        // frames in the sample: `rows` of every whole block, and what the last block holds of its first `rows`
        const long long whole = frames / 256, rest = frames - whole * 256;
        const double n = (double)(whole * rows + min((long long)rows, rest));
        float cT = n > 0 ? (float)(chan[0] / n) : 0.0f, cS = n > 0 ? (float)(sp / (n * (numCh - 1))) : 0.0f;
        if (!(fabsf(cT) < 3.0e38f)) cT = 0.0f;
        if (!(fabsf(cS) < 3.0e38f)) cS = 0.0f;
        fd.cT = cT;
        fd.cS = cS;
    }
}

// window_static_kernel: for every window of the files of a half-precision preparation, the part of its bounds
// that does not depend on the target's values (sgz::WindowStatic, 32 bytes), from the fp32 prefix records and the
// fp64 tile bases.  Runs once per database / norms / target length, after tile_scan_kernel; the matrix-core
// kernel then reads two 16-byte words per window instead of doing these statistics in its epilogue.  For a layout
// whose bounds come from the spectral kernel the same buffer takes that kernel's compact form instead
// (compactWindows = all windows of the layout; sgz::SpectralStatics).
__global__ __launch_bounds__(256) void window_static_kernel(const FileDesc *__restrict__ files,
                                                            const TileDesc *__restrict__ tiles,
                                                            const float4 *__restrict__ prefix,
                                                            const double *__restrict__ tbase, int L, int numCh,
                                                            int tileShift, float4 *__restrict__ wstat,
                                                            long long compactWindows)
{
    const TileDesc td = tiles[blockIdx.x];
    const FileDesc fd = files[td.file];
    HalfPrefix hp;
    hp.pre = prefix + fd.prefix_off / 4;  // (prefix_off counts four values per frame)
    hp.tbase = tbase + fd.tbase_off;
    hp.nptiles = fd.nptiles;
    hp.tileShift = tileShift;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long o = (long long)td.o0 + k * 256 + threadIdx.x;
        if (o < fd.num_windows) {
            const WindowStatic ws = window_static(L, numCh, hp, (unsigned)o, window_records(hp, (unsigned)o, L), (double)fd.cT);
            if (compactWindows > 0) {  // the layout of the spectral kernel: a 16-byte record and a value of its own per window
                spectral_statics_store(reinterpret_cast<float *>(wstat), compactWindows, fd.win_off + o, ws);
            } else {
                wstat[2 * (fd.win_off + o)] = make_float4(ws.rsdT, ws.dmT, ws.gT, ws.lnAvg);
                wstat[2 * (fd.win_off + o) + 1] = make_float4(ws.rsdS, ws.dmS, ws.gS, __uint_as_float(ws.hTS));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// tile_scan_kernel: per file and moment, turns the tile totals into exclusive
// tile bases (one wave per moment).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_scan_kernel(const FileDesc *__restrict__ files, int fileBegin,
                                                        double *__restrict__ tbase)
{
    const FileDesc fd = files[fileBegin + blockIdx.x];
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *p = tbase + fd.tbase_off + (long long)q * fd.nptiles;
    double carry = 0.0;
    for (long long i0 = 0; i0 < fd.nptiles; i0 += 64) {
        const long long i = i0 + lane;
        const double v = i < fd.nptiles ? p[i] : 0.0;
        double inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double up = __shfl_up(inc, off, 64);
            if (lane >= off) inc += up;
        }
        if (i < fd.nptiles) p[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
}

// ---------------------------------------------------------------------------
// corr_kernel
//
// Workgroup = 8 waves, one tile of 1024 consecutive windows of one file.  Lane t
// owns windows o0 + 16t .. o0 + 16t + 15 (16 fp64 accumulators).  A "step" is 16
// consecutive lags: with x[0..31) the 31 frames starting at the lane's first
// window + lag, acc[r] += a'[lag + q] * x[q + r] for q, r in 0..15, i.e. 256 FMAs
// for 16 newly read frames and 16 target values.  The target values are
// wave-uniform (scalar loads feeding the FMA's SGPR operand); the frames come
// from an LDS image of the planar channel block in chunk-major order, so that
// the 64 lanes read 64 consecutive doubles per ds_read_b64.  The 8 waves split
// the lag steps of each (channel, block) phase; their partial sums are added in
// wave order into a small LDS array (fixed order: results are reproducible).
// Staging of the next phase's block (global -> registers -> LDS) is issued
// before and retired after the FMA work of the current phase.
// Two workgroups per CU = 4 waves per SIMD: 2 x (64 KiB staging + 16 KiB sums).
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_chunk(T (&x)[kR], const T *p)
{
#pragma unroll
    for (int j = 0; j < kR; j++) x[j] = p[j * kMT];
}

template <typename T>
__device__ __forceinline__ void fma_step(T (&acc)[kR], const T *__restrict__ a, const T (&x0)[kR], const T (&x1)[kR])
{
#pragma unroll
    for (int q = 0; q < kR; q++) {
        const T aq = a[q];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const int k = q + r;
            acc[r] = fma(aq, k < kR ? x0[k] : x1[k - kR], acc[r]);
        }
    }
}

constexpr int kThreads = kWaves * 64;
constexpr int kStageCols = 256;  // columns one staging pass covers (>= kMT)
static_assert(kStageCols >= kMT && kMT % 4 == 0, "staging pass must cover a whole LDS row");
constexpr int kRedDoubles = kR * 64;  // one sum array (16 windows x 64 lanes)

template <typename T>
constexpr size_t corr_lds_bytes()
{
    return (size_t)2 * kR * kMT * sizeof(T) + 2 * kRedDoubles * sizeof(double);
}

// Adds every wave's 16 partial sums into red[r][lane], wave 0 first.
__device__ __forceinline__ void reduce_waves(double *red, const double (&acc)[kR], int wave, int lane)
{
    for (int wv = 0; wv < kWaves; wv++) {
        if (wave == wv) {
#pragma unroll
            for (int r = 0; r < kR; r++) {
                const double prev = wv == 0 ? 0.0 : red[r * 64 + lane];
                red[r * 64 + lane] = prev + acc[r];
            }
        }
        __syncthreads();
    }
}

// Windows whose prefix-sum statistics cannot be trusted (window_result): listed by the exact kernel for a
// two-pass evaluation, or just given `fill` (+inf in a screening run: a certain candidate, re-scored anyway).
struct SuspectList {
    sgz_candidate *list;        // null: no list
    unsigned long long *count;  // zero when the kernel starts
    long long capacity;
    float fill;                 // the similarity written for such a window (exact run: NaN until re-evaluated)
    const int *wholeFile;       // per file (or null): != 0 = every window of the file is re-evaluated from its
                                // frames anyway (it holds a value that is not a number): not listed here
};

// T = double: the exact path.  T = float: the screening path -- fp32 products and
// fp32 sums over one phase (<= 176 * 16 lags of one channel per wave), folded into
// fp64 sums after every phase.
template <typename T>
__global__ __launch_bounds__(kThreads, 4) void corr_kernel(const FileDesc *__restrict__ files,
                                                           const TileDesc *__restrict__ tiles,
                                                           const T *__restrict__ planar,
                                                           const double *__restrict__ prefix,
                                                           const double *__restrict__ tbase,
                                                           const T *__restrict__ aprime, CorrParams cp,
                                                           float *__restrict__ simOut,
                                                           float *__restrict__ boostOut, SuspectList sus)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    T *lds = reinterpret_cast<T *>(ldsRaw);  // 2 x [kR][kMT] of T, then 2 x [kR][64] doubles
    constexpr bool kExact = sizeof(T) == 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileDesc td = tiles[blockIdx.x];
    const FileDesc fd = files[td.file];
    const long long m0 = td.o0 / kR;
    const T *fplanar = planar + fd.planar_off;
    const int Lpad = cp.S * kR;
    const int nPh = cp.numCh * cp.nb;
    double *redS = reinterpret_cast<double *>(ldsRaw + (size_t)2 * kR * kMT * sizeof(T));
    double *redT = redS + kRedDoubles;

    T acc[kR];
    double accD[kR];  // only used by the screening path
#pragma unroll
    for (int r = 0; r < kR; r++) {
        acc[r] = (T)0;
        accD[r] = 0.0;
    }

    // staging geometry: 16-byte vectors; thread copies row it*rowsPerIt + tid/vecPerRow, columns scol..scol+EV-1
    constexpr int EV = 16 / sizeof(T);
    constexpr int vecPerRow = kStageCols / EV;
    constexpr int rowsPerIt = kThreads / vecPerRow;
    constexpr int stageIters = kR / rowsPerIt;
    const int srow = tid / vecPerRow;
    const int scol = EV * (tid % vecPerRow);
    float4 st[stageIters];

    auto stage_load = [&](int ph) {
        const int c = ph / cp.nb;
        const int kb = ph - c * cp.nb;
        const int s0 = kb * cp.SB;
        const int scount = min(cp.SB, cp.S - s0);
        const int used = 64 + scount;  // chunks of this block that are read
        const T *src = fplanar + ((long long)c * kR) * fd.M + m0 + s0 + scol;
#pragma unroll
        for (int it = 0; it < stageIters; it++) {
            const int row = it * rowsPerIt + srow;
            if (scol < used) st[it] = *reinterpret_cast<const float4 *>(src + (long long)row * fd.M);
        }
    };
    auto stage_store = [&](int buf) {
        T *dst = lds + buf * (kR * kMT) + scol;
#pragma unroll
        for (int it = 0; it < stageIters; it++) {
            const int row = it * rowsPerIt + srow;
            if (scol < kMT) *reinterpret_cast<float4 *>(dst + row * kMT) = st[it];
        }
    };

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int ph = 0; ph < nPh; ph++) {
        const int c = ph / cp.nb;
        const int kb = ph - c * cp.nb;
        const int s0 = kb * cp.SB;
        const int scount = min(cp.SB, cp.S - s0);
        if (ph + 1 < nPh) stage_load(ph + 1);

        const int sb = (scount * wave) / kWaves;
        const int se = (scount * (wave + 1)) / kWaves;
        const int n = se - sb;
        const T *a = aprime + (long long)c * Lpad + (long long)(s0 + sb) * kR;
        const T *xp = lds + (ph & 1) * (kR * kMT) + lane + sb;
        // A second, opaque copy of the pointer keeps the compiler from pairing the
        // loads of adjacent chunks into ds_read2 forms (half the LDS rate, and a
        // narrow offset field that costs an address add per row).
        // (The offset, not the pointer, is made opaque: the pointer keeps its LDS address space.)
        int one = 1;
        asm volatile("" : "+v"(one));
        const T *xq = xp + one;

        T xa[kR], xb[kR];
        load_chunk(xa, xp);
        int s = 0;
        for (; s + 2 <= n; s += 2) {
            load_chunk(xb, xq + s);
            fma_step(acc, a + s * kR, xa, xb);
            load_chunk(xa, xp + s + 2);
            fma_step(acc, a + (s + 1) * kR, xb, xa);
        }
        if (s < n) {
            load_chunk(xb, xq + s);
            fma_step(acc, a + s * kR, xa, xb);
        }
        if constexpr (!kExact) {
#pragma unroll
            for (int r = 0; r < kR; r++) {
                accD[r] += (double)acc[r];
                acc[r] = (T)0;
            }
        }

        if (ph + 1 < nPh) stage_store((ph + 1) & 1);
        __syncthreads();
        if (c == 0 && kb == cp.nb - 1) {  // loudness channel done: its sums go to redT
            if constexpr (kExact) {
                double tmp[kR];
#pragma unroll
                for (int r = 0; r < kR; r++) tmp[r] = (double)acc[r];
                reduce_waves(redT, tmp, wave, lane);
            } else {
                reduce_waves(redT, accD, wave, lane);
            }
#pragma unroll
            for (int r = 0; r < kR; r++) {
                acc[r] = (T)0;
                accD[r] = 0.0;
            }
        }
    }
    if constexpr (kExact) {
        double tmp[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) tmp[r] = (double)acc[r];
        reduce_waves(redS, tmp, wave, lane);
    } else {
        reduce_waves(redS, accD, wave, lane);
    }

    Epilogue ep;
    ep.pre = prefix + fd.prefix_off;
    ep.tbase = tbase + fd.tbase_off;
    ep.frames = fd.frames;
    ep.nptiles = fd.nptiles;
    ep.tileShift = cp.prepTileShift;

    // thread handles lane l = tid % 64 and r = 2*(tid/64), +1  -> 2 consecutive windows
    constexpr int perThread = kR / kWaves;
    const int g = tid >> 6;
    const long long oBase = (long long)td.o0 + (long long)lane * kR + perThread * g;
#pragma unroll
    for (int k = 0; k < perThread; k++) {
        const int r = perThread * g + k;
        const long long o = oBase + k;
        if (o < fd.num_windows) {
            float sim, boost;
            bool suspect;
            window_result(cp, ep, o, redT[r * 64 + lane], redS[r * 64 + lane], sim, boost, suspect);
            if (suspect) {
                // statistics too close to the rounding of the prefix sums: the window is listed for a two-pass
                // evaluation from its frames (exact run) or made a certain candidate (screening run)
                sim = sus.fill;
                if (sus.list && !(sus.wholeFile && sus.wholeFile[td.file])) {
                    const unsigned long long slot = atomicAdd(sus.count, 1ull);
                    if ((long long)slot < sus.capacity) {
                        sgz_candidate c;
                        c.file = td.file;
                        c.offset = (int)o;
                        c.sim = sim;
                        c.boost = boost;
                        sus.list[slot] = c;
                    }
                }
            }
            simOut[fd.win_off + o] = sim;
            boostOut[fd.win_off + o] = boost;
        }
    }
}

// Per file: does it hold a value that is not a finite number once normalised?  MathUtil.stat and correlate
// (impl/MathUtil.scala:29-62, 177-196) run over the frames of one window, so such a value spoils exactly the
// windows that contain it; window statistics from prefix sums would spoil every later window of the file as well.
// The exact run therefore evaluates every window of such a file from its frames (list_windows_kernel and the
// re-scoring kernels).  kScanParts workgroups per file (flags cleared beforehand).
constexpr int kScanParts = 32;
__global__ __launch_bounds__(256) void nonfinite_scan_kernel(const FileDesc *__restrict__ files,
                                                             const float *__restrict__ normTab, int normEnabled,
                                                             int numCh, int *__restrict__ flags)
{
    const int file = blockIdx.x / kScanParts, part = blockIdx.x % kScanParts;
    const FileDesc fd = files[file];
    const long long n = fd.frames * numCh;
    const long long per = ((n + kScanParts - 1) / kScanParts + 3) & ~3ll;
    const long long e0 = part * per, e1 = min(n, e0 + per);
    int bad = 0;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
        float v = fd.raw[e];
        if (normEnabled) {
            const int c = (int)(e % numCh);
            v = (v - normTab[c]) / normTab[SGZ_MAX_CHANNELS + c];
        }
        bad |= !(fabsf(v) <= 3.4028234e38f);
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0 && bad) atomicOr(&flags[file], 1);
}

// Windows [o0, o0 + n) of one file as a candidate list (for the re-scoring kernels).
__global__ __launch_bounds__(256) void list_windows_kernel(int file, long long o0, long long n,
                                                           sgz_candidate *__restrict__ list,
                                                           unsigned long long *__restrict__ count)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *count = (unsigned long long)n;
    if (i < n) {
        sgz_candidate c;
        c.file = file;
        c.offset = (int)(o0 + i);
        c.sim = NAN;
        c.boost = NAN;
        list[i] = c;
    }
}

// Writes re-evaluated windows back into the dense arrays.
__global__ __launch_bounds__(256) void scatter_kernel(const FileDesc *__restrict__ files,
                                                      const sgz_candidate *__restrict__ list,
                                                      const unsigned long long *__restrict__ count,
                                                      long long capacity, float *__restrict__ simOut,
                                                      float *__restrict__ boostOut)
{
    const long long n = min((long long)*count, capacity);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const sgz_candidate c = list[i];
        const long long at = files[c.file].win_off + c.offset;
        simOut[at] = c.sim;
        boostOut[at] = c.boost;
    }
}

// ---------------------------------------------------------------------------
// compact_kernel: one workgroup per correlation tile; appends windows with
// sim > threshold.  NaN similarities never pass (as in the reference's `>`).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_kernel(const FileDesc *__restrict__ files,
                                                      const TileDesc *__restrict__ tiles,
                                                      const float *__restrict__ sim,
                                                      const float *__restrict__ boost, float threshold,
                                                      sgz_candidate *__restrict__ out, long long capacity,
                                                      unsigned long long *__restrict__ count,
                                                      const int *__restrict__ hflags, int flagBegin, int flagEnd,
                                                      unsigned long long *__restrict__ zeroOther,
                                                      const float *__restrict__ lnAvgLazy, float lnAvgIn)
{
    const TileDesc td = tiles[blockIdx.x];
    const FileDesc fd = files[td.file];
    // the counter of the next call (callers that alternate between two counters need no separate clear)
    if (zeroOther && blockIdx.x == 0 && threadIdx.x == 0) *zeroOther = 0ull;
    // A file whose half-precision image overflowed has no usable screening values: report it as a
    // candidate overflow, which sends the caller to the exact dense path.  (The marks belong to the image: they
    // count for the files [flagBegin, flagEnd) that have one, whose stored values may be bounds of a half-precision
    // run; a marked file that has been through an exact run since is reported all the same, which is safe.)
    if (td.file >= flagBegin && td.file < flagEnd && hflags[td.file] && td.o0 == 0 && threadIdx.x == 0)
        atomicAdd(count, (unsigned long long)capacity + 1ull);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long o = (long long)td.o0 + k * 256 + threadIdx.x;
        if (o < fd.num_windows) {
            const float s = sim[fd.win_off + o];
            if (s > threshold) {
                const unsigned long long slot = atomicAdd(count, 1ull);
                if ((long long)slot < capacity) {
                    sgz_candidate c;
                    c.file = td.file;
                    c.offset = (int)o;
                    c.sim = s;
                    // (boosts the spectral kernel left implicit are evaluated here, as that kernel would have)
                    c.boost = lnAvgLazy ? screen_boost(lnAvgIn, lnAvgLazy[fd.win_off + o]) : boost[fd.win_off + o];
                    out[slot] = c;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Search over bounds (half-precision screening run): with numPerFile = 1 a head
// file contributes its largest similarity, first occurrence (the sliding loop of
// impl/FeatureCorrelationImpl.scala:234-239 keeps a running strict maximum).
// The matrix-core kernel leaves, per file, the largest lower bound (an atomic
// maximum over order-preserving integer keys): every window whose upper bound
// reaches it may hold the maximum.  bound_min_kernel takes the minimum of those
// over the head files: no match set that the head files fill
// can end with a smaller worst entry, so tail windows whose upper bound does not
// exceed it cannot be offered.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bound_min_kernel(const int *__restrict__ lokey, int n, float *__restrict__ out)
{
    float v = INFINITY;
    for (int k = threadIdx.x; k < n; k += 64) v = fminf(v, bound_from_key(lokey[k]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    if (threadIdx.x == 0) out[0] = v;
}

// The k-th largest of the per-file bounds (lokey[0 .. n)), -inf when there are fewer than k files: with
// numPerFile = 1 and k = numMatches, a file whose similarities all lie below it cannot be in the result when at
// least k files reach it with distinct values (the caller checks that on the finished selection).  Work of one
// workgroup of 256 threads (every thread returns the value); radix selection over the order-preserving keys, a
// byte per round.
__device__ float bound_kth_select(const int *__restrict__ lokey, int n, int k)
{
    __shared__ int hist[256];
    __shared__ int waveTotal[4];
    __shared__ unsigned int chosenPrefix;
    __shared__ int chosenLeft;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n < k || k < 1) return -INFINITY;
    // (the keys of up to four files per thread stay in registers; more are read again every round)
    unsigned int mine[4];
#pragma unroll
    for (int j = 0; j < 4; j++) mine[j] = tid + 256 * j < n ? (unsigned int)lokey[tid + 256 * j] ^ 0x80000000u : 0u;
    unsigned int prefix = 0, mask = 0;
    int left = k;  // the wanted key is the left-th largest of those that match the prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (tid + 256 * j < n && (mine[j] & mask) == prefix) atomicAdd(&hist[(mine[j] >> shift) & 255u], 1);
        for (int i = tid + 1024; i < n; i += 256) {
            const unsigned int u = (unsigned int)lokey[i] ^ 0x80000000u;
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1);
        }
        __syncthreads();
        // thread b: how many matching keys have a byte >= b (suffix sums: within the wave by shuffles, then the
        // totals of the waves above); the wanted byte is the one where that count first reaches `left`
        const int h = hist[tid];
        int atLeast = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_down(atLeast, off, 64);
            if (lane + off < 64) atLeast += t;
        }
        if (lane == 0) waveTotal[wave] = atLeast;
        __syncthreads();
        for (int w = wave + 1; w < 4; w++) atLeast += waveTotal[w];
        if (atLeast >= left && atLeast - h < left) {  // exactly one thread
            chosenPrefix = prefix | ((unsigned int)tid << shift);
            chosenLeft = left - (atLeast - h);
        }
        __syncthreads();
        prefix = chosenPrefix;
        left = chosenLeft;
        mask |= 255u << shift;
    }
    return bound_from_key((int)(prefix ^ 0x80000000u));
}

// Windows of head files (file < headEnd) whose upper bound reaches the file's own bound (and, with topK > 0, the
// topK-th largest of the head files' bounds), windows of the other files whose upper bound exceeds the minimum of
// the head bounds (tailThr[0]).  A workgroup works through tiles blockIdx.x, blockIdx.x + gridDim.x, ...; with
// topK > 0 every workgroup first selects that bound for itself from the per-file keys -- a few microseconds of
// redundant work that cost less than a kernel of one workgroup ahead of this one -- and workgroup 0 leaves it in
// topThrOut[0] for the host.
__global__ __launch_bounds__(256) void compact_bounded_kernel(const FileDesc *__restrict__ files,
                                                              const TileDesc *__restrict__ tiles, int numTiles,
                                                              const float *__restrict__ up,
                                                              const float *__restrict__ boost,
                                                              const int *__restrict__ lokey,
                                                              const float *__restrict__ tailThr, int headEnd,
                                                              sgz_candidate *__restrict__ out, long long capacity,
                                                              unsigned long long *__restrict__ count,
                                                              const int *__restrict__ hflags,
                                                              unsigned long long *__restrict__ zeroOther, int topK,
                                                              float *__restrict__ topThrOut,
                                                              const float *__restrict__ lnAvgLazy, float lnAvgIn)
{
    if (zeroOther && blockIdx.x == 0 && threadIdx.x == 0) *zeroOther = 0ull;
    float top = -INFINITY;
    if (topK > 0) {
        top = bound_kth_select(lokey, headEnd, topK);
        if (blockIdx.x == 0 && threadIdx.x == 0) topThrOut[0] = top;
    }
    // The descriptors a tile needs hang on one another (tile -> file -> its bound and mark): the tile record is
    // fetched two rounds ahead and the file's data one round ahead, so that a round waits for its windows only.
    struct Ahead {
        long long numWindows, winOff;
        int file, o0, key, flag;
    };
    auto describe = [&](const TileDesc &t) {
        const FileDesc f = files[t.file];
        return Ahead{f.num_windows, f.win_off, t.file, t.o0, lokey[t.file], hflags[t.file]};
    };
    const int stride = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= numTiles) return;
    Ahead cur = describe(tiles[tile]);
    TileDesc td2 = tiles[min(tile + stride, numTiles - 1)];
    const float tail = tailThr[0];
    for (; tile < numTiles; tile += stride) {
        const TileDesc td3 = tiles[min(tile + 2 * stride, numTiles - 1)];
        float s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long o = (long long)cur.o0 + k * 256 + threadIdx.x;
            s[k] = o < cur.numWindows ? up[cur.winOff + o] : -INFINITY;
        }
        const Ahead nxt = describe(td2);
        // (every file this kernel sees holds bounds of a half-precision run: the host checks)
        if (cur.flag && cur.o0 == 0 && threadIdx.x == 0) atomicAdd(count, (unsigned long long)capacity + 1ull);
        const bool headFile = cur.file < headEnd;
        float t = headFile ? bound_from_key(cur.key) : tail;
        if (headFile) t = fmaxf(t, top);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long o = (long long)cur.o0 + k * 256 + threadIdx.x;
            if (o < cur.numWindows && (headFile ? s[k] >= t : s[k] > t)) {
                const unsigned long long slot = atomicAdd(count, 1ull);
                if ((long long)slot < capacity) {
                    sgz_candidate c;
                    c.file = cur.file;
                    c.offset = (int)o;
                    c.sim = s[k];
                    c.boost = lnAvgLazy ? screen_boost(lnAvgIn, lnAvgLazy[cur.winOff + o]) : boost[cur.winOff + o];
                    out[slot] = c;
                }
            }
        }
        cur = nxt;
        td2 = td3;
    }
}

__global__ __launch_bounds__(256) void bound_kth_kernel(const int *__restrict__ lokey, int n, int k, float *__restrict__ out)
{
    const float v = bound_kth_select(lokey, n, k);
    if (threadIdx.x == 0) out[0] = v;
}

// The search over bounds from the candidate list of a spectral run (sgz::PreList: the windows whose upper bound
// reaches their own block's largest lower bound) instead of from dense upper bounds: a record stays when its
// upper bound reaches its file's bound and, with topK > 0, the topK-th largest of the files' bounds -- what
// compact_bounded_kernel does for head files, over a few records per block instead of every window.  Files the
// bounds cannot decide and a list that did not hold every record are reported through the count, as there.  Clears
// the other counter of the list (the next run's) and of the output.
__global__ __launch_bounds__(256) void prelist_filter_kernel(const sgz_candidate *__restrict__ pre,
                                                             const unsigned long long *__restrict__ preCount,
                                                             long long preCapacity,
                                                             unsigned long long *__restrict__ preOther,
                                                             const int *__restrict__ lokey, int headEnd,
                                                             const int *__restrict__ hflags, int fileBegin, int fileEnd,
                                                             sgz_candidate *__restrict__ out, long long capacity,
                                                             unsigned long long *__restrict__ count,
                                                             unsigned long long *__restrict__ zeroOther, int topK,
                                                             float *__restrict__ topThrOut)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (zeroOther) *zeroOther = 0ull;
        *preOther = 0ull;
    }
    float top = -INFINITY;
    if (topK > 0) {
        top = bound_kth_select(lokey, headEnd, topK);
        if (blockIdx.x == 0 && threadIdx.x == 0) topThrOut[0] = top;
    }
    const unsigned long long listed = *preCount;
    if (blockIdx.x == 0) {
        for (int f = fileBegin + threadIdx.x; f < fileEnd; f += 256)
            if (hflags[f]) atomicAdd(count, (unsigned long long)capacity + 1ull);
        if (threadIdx.x == 0 && listed > (unsigned long long)preCapacity) atomicAdd(count, (unsigned long long)capacity + 1ull);
    }
    const long long n = listed > (unsigned long long)preCapacity ? preCapacity : (long long)listed;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const sgz_candidate c = pre[i];
        const float t = fmaxf(bound_from_key(lokey[c.file]), top);
        if (c.sim >= t) {
            const unsigned long long slot = atomicAdd(count, 1ull);
            if ((long long)slot < capacity) out[slot] = c;
        }
    }
}

// ---------------------------------------------------------------------------
// Exact re-evaluation of listed windows, straight from the raw frames: fp32 normalise with the IEEE divide
// (MathUtil.normalize, impl/MathUtil.scala:132-152), every sum in fp64.  What MathUtil.stat and
// MathUtil.correlate (impl/MathUtil.scala:29-62, 177-196) compute in two passes over the window --
// mean, then sum (b - mean)^2 and sum (a - aMean)(b - mean) -- comes from ONE pass over values shifted by a
// sample of the window itself, m0 = the window's first value of the channel group:
//     S = sum (b - m0),  Q = sum (b - m0)^2,  X = sum (a - aMean)(b - m0)
//     mean = m0 + S / n,  sum (b - mean)^2 = Q - S^2 / n,  sum (a - aMean)(b - mean) = X - (S / n) sum (a - aMean)
// b - m0 is exact in fp64 (a difference of two fp32 values), and because m0 lies inside the window's range the
// subtraction Q - S^2 / n loses no more than a two-pass evaluation does: a constant window gives exactly 0
// (the reference's NaN similarity), a nearly constant one its tiny variance to 1e-15 relative.  The window is
// read once instead of twice.
//
// A window is cut into `slices` runs of frames; one 256-thread workgroup takes
// one (window, slice) pair, so that a short list still fills the GPU:
//   rescore_pass_kernel   partial sums S, Q, X of both channel groups   -> part[.][0..5]
//   rescore_final_kernel  one thread per window: totals in slice order, sim, boost
// Thread t of the first (256 / numCh) * numCh threads owns channel t % numCh and
// every (256 / numCh)-th frame of the slice: consecutive threads read consecutive
// values, and a thread's channel constants stay in registers.
// ---------------------------------------------------------------------------
constexpr int kRsThreads = 256;
constexpr int kRsParts = 7;  // sumT, sumS, sqT, sqS, crossT, crossS; the window's loudness shift m0T (the same in every slice)
constexpr int kRsSums = 6;

// Records of a candidate list that may be read.  A count beyond the capacity comes from two sources: more
// candidates than slots (then every slot is written), or a compaction kernel that reports a file it cannot
// screen by adding capacity + 1 without writing anything (then slots may hold stale records of an earlier,
// possibly larger database, whose file indices and frame pointers are no longer valid).  Only a caller that
// knows its list is of the first kind (`truncatedOk`: the suspect list of an exact run) has the first
// `capacity` records read; everybody else gets none -- such a list is reported as an overflow and discarded.
__device__ __forceinline__ long long listed_records(unsigned long long count, long long capacity, int truncatedOk)
{
    if (count > (unsigned long long)capacity) return truncatedOk ? capacity : 0;
    return (long long)count;
}

struct RescoreGeom {
    int slices;       // per window
    int sliceFrames;  // frames per slice (the last one may be shorter)
};

// The shift of a window's channel group: its first normalised value (loudness: channel 0, others: channel 1).
__device__ __forceinline__ void window_shift(const float *__restrict__ first, const float *__restrict__ normTab,
                                             int normEnabled, double &m0T, double &m0S)
{
    float vT = first[0], vS = first[1];
    if (normEnabled) {
        vT = (vT - normTab[0]) / normTab[SGZ_MAX_CHANNELS];
        vS = (vS - normTab[1]) / normTab[SGZ_MAX_CHANNELS + 1];
    }
    // (a first value that is not a finite number is no shift to take: with 0 the sums come out as those of the
    // reference's own passes do -- infinite or NaN -- and with them boost, gate and similarity)
    m0T = fabsf(vT) <= 3.4028234e38f ? (double)vT : 0.0;
    m0S = fabsf(vS) <= 3.4028234e38f ? (double)vS : 0.0;
}

__global__ __launch_bounds__(kRsThreads) void rescore_pass_kernel(
    const FileDesc *__restrict__ files, const float *__restrict__ tframes /* normalised target [L][numCh] */,
    const float *__restrict__ normTab /* min[64], max - min[64] */, int normEnabled, CorrParams cp, RescoreGeom rg,
    double aMeanT, double aMeanS,
    const sgz_candidate *__restrict__ cands, const unsigned long long *__restrict__ count, long long capacity,
    int truncatedOk, double *__restrict__ part /* [window][slice][kRsParts] */)
{
    constexpr int U = 16;  // loads in flight per thread
    __shared__ double scratch6[kRsSums][4];
    const long long n = listed_records(*count, capacity, truncatedOk);
    const long long work = n * rg.slices;
    if ((long long)blockIdx.x >= work) return;
    const int numCh = cp.numCh, L = cp.L;
    const int tid = threadIdx.x;
    const int rows = kRsThreads / numCh;  // frames per sweep
    const int stride = rows * numCh;      // values per sweep
    const int i0 = tid / numCh, c = tid - i0 * numCh;
    const bool active = tid < stride;
    const bool isT = c == 0;
    const float mn = active ? normTab[c] : 0.0f, dv = active ? normTab[SGZ_MAX_CHANNELS + c] : 1.0f;
    const double aMean = isT ? aMeanT : aMeanS;
    for (long long w = blockIdx.x; w < work; w += gridDim.x) {
        const long long ci = w / rg.slices;
        const int g = (int)(w - ci * rg.slices);
        const int f0 = g * rg.sliceFrames;
        const int fl = min(rg.sliceFrames, L - f0);                            // frames of this slice
        const int K = active && i0 < fl ? (fl - i0 + rows - 1) / rows : 0;     // values of this thread
        const sgz_candidate cd = cands[ci];
        const FileDesc fd = files[cd.file];
        double m0T, m0S;
        window_shift(fd.raw + (long long)cd.offset * numCh, normTab, normEnabled, m0T, m0S);
        const double m0 = isT ? m0T : m0S;
        const float *raw = fd.raw + ((long long)cd.offset + f0) * numCh + tid;
        const float *tn = tframes + (long long)f0 * numCh + tid;
        double s = 0.0, q = 0.0, x = 0.0;
        // batches of U loads in flight; the last batch repeats the thread's final value and skips it in the
        // sums (no one-at-a-time tail: each of its loads would cost a full memory round trip)
        for (int k = 0; k < K; k += U) {
            float v[U], af[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int kk = min(k + u, K - 1);
                v[u] = raw[kk * stride];
                af[u] = tn[kk * stride];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (normEnabled) v[u] = (v[u] - mn) / dv;
                const bool live = k + u < K;
                const double d = live ? (double)v[u] - m0 : 0.0;
                s += d;
                q += d * d;
                x += ((double)af[u] - aMean) * d;
            }
        }
        double *out = part + w * kRsParts;
        // the six sums over the workgroup at once: wave sums by shuffles, the four waves' results through LDS
        double v6[kRsSums] = {isT ? s : 0.0, isT ? 0.0 : s, isT ? q : 0.0, isT ? 0.0 : q, isT ? x : 0.0, isT ? 0.0 : x};
#pragma unroll
        for (int j = 0; j < kRsSums; j++)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v6[j] += __shfl_xor(v6[j], off, 64);
        __syncthreads();  // scratch may still be read from the window before
        if ((tid & 63) == 0)
#pragma unroll
            for (int j = 0; j < kRsSums; j++) scratch6[j][tid >> 6] = v6[j];
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < kRsSums; j++) out[j] = (scratch6[j][0] + scratch6[j][1]) + (scratch6[j][2] + scratch6[j][3]);
            out[6] = m0T;  // (the final kernel then needs neither the file table nor the frames)
        }
    }
}

// Where the last workgroup of rescore_final_kernel leaves a copy of the result for the host: pinned memory
// the host polls, which saves a copy command and the wake-up of a blocking stream wait per search.
// sgz_corr_prefetch_dense: similarities and boosts of a run of windows to pinned host memory, then a sequence
// word the host polls (the last workgroup publishes it).  A copy command on another stream would be carried
// out by a copy kernel on the compute queue, behind whatever kernel the caller has queued next -- for the head
// of a search that is the screening run over all other files.
__global__ __launch_bounds__(256) void dense_to_host_kernel(const float *__restrict__ sim, const float *__restrict__ boost,
                                                            long long n, float *__restrict__ simHost,
                                                            float *__restrict__ boostHost, unsigned int *done,
                                                            unsigned long long *seqHost, unsigned long long seq)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        simHost[i] = sim[i];
        boostHost[i] = boost[i];
    }
    __shared__ bool isLast;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        isLast = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (isLast && threadIdx.x == 0) {
        __threadfence_system();
        *done = 0u;
        __hip_atomic_store(seqHost, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct HostReport {
    unsigned long long *host;       // pinned: headers and records as in the device buffer; word maxWords = sequence
    const unsigned long long *dev;  // device buffer: [two count headers][records]
    unsigned int *done;             // device: workgroups that have finished (reset by the last one)
    int headWords;                  // words of the headers
    int maxWords;                   // capacity of the pinned buffer in words (headers included)
    unsigned long long seq;         // value the host waits for
};

__global__ __launch_bounds__(256) void rescore_final_kernel(const FileDesc *__restrict__ files,
                                                            const float *__restrict__ normTab, int normEnabled,
                                                            CorrParams cp, RescoreGeom rg,
                                                            sgz_candidate *__restrict__ cands,
                                                            const unsigned long long *__restrict__ count,
                                                            long long capacity, int truncatedOk,
                                                            const double *__restrict__ part, HostReport hr)
{
    const long long n = listed_records(*count, capacity, truncatedOk);
    // The workgroups that have records (the first one in any case) take part; a search's few dozen records are one
    // workgroup's, which then reports without the fence, the counter and the wait that finding the last of several
    // costs (about 3 us of a launch of 7).
    const unsigned int busy = (unsigned int)max(1ll, min((long long)gridDim.x, (n + blockDim.x - 1) / blockDim.x));
    if (blockIdx.x >= busy) return;
    // (the headers the report starts with, asked for now: they were written before this launch)
    unsigned long long head = 0ull;
    if (hr.host && (int)threadIdx.x < hr.headWords) head = hr.dev[threadIdx.x];
    const double nT = (double)cp.L, nS = (double)cp.L * (double)(cp.numCh - 1);
    for (long long ci = (long long)blockIdx.x * blockDim.x + threadIdx.x; ci < n; ci += (long long)busy * blockDim.x) {
        double t[kRsSums] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < rg.slices; k++)
#pragma unroll
            for (int j = 0; j < kRsSums; j++) t[j] += part[(ci * rg.slices + k) * kRsParts + j];
        sgz_candidate c = cands[ci];
        const double m0T = part[ci * rg.slices * kRsParts + 6];
        const double dT = t[0] / nT, dS = t[1] / nS;  // mean - m0
        const float avgF = (float)(m0T + dT);
        const double lnAvgB = log((double)avgF);
        const float boost = (float)exp((cp.lnAvgIn - lnAvgB) / 0.6);
        float sim = 0.0f;
        if (boost <= cp.maxBoost) {
            float temporal = 0.0f, spectral = 0.0f;
            if (cp.w > 0.0f) {
                double var = t[2] / nT - dT * dT;
                if (var < 0.0) var = 0.0;  // rounding of an (almost) constant window; NaN stays NaN
                temporal = (float)((t[4] - dT * cp.sumAT) / (cp.aStdT * sqrt(var) * nT));
            }
            if (cp.w < 1.0f) {
                double var = t[3] / nS - dS * dS;
                if (var < 0.0) var = 0.0;
                spectral = (float)((t[5] - dS * cp.sumAS) / (cp.aStdS * sqrt(var) * nS));
            }
            sim = temporal * cp.w + spectral * (1.0f - cp.w);
        }
        c.sim = sim;
        c.boost = boost;
        cands[ci] = c;
        if (hr.host && hr.headWords + 2 * ci + 1 < hr.maxWords)  // every workgroup reports its own records
            reinterpret_cast<sgz_candidate *>(hr.host + hr.headWords)[ci] = c;
    }
    if (!hr.host) return;
    __shared__ int isLast;
    __syncthreads();
    if (busy > 1) {
        if (threadIdx.x == 0) {
            __threadfence_system();  // one release for the workgroup's records (a fence per thread costs a cache
                                     // write-back each)
            isLast = atomicAdd(hr.done, 1u) == busy - 1;
        }
        __syncthreads();
        if (!isLast) return;
    }
    // the last workgroup adds the headers and then the sequence number the host polls for
    if ((int)threadIdx.x < hr.headWords) hr.host[threadIdx.x] = head;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (busy > 1) *hr.done = 0u;
        __threadfence_system();  // records and headers before the sequence word
        __hip_atomic_store(hr.host + hr.maxWords, hr.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------
// Target preparation (readInBuffer, impl/FeatureCorrelationImpl.scala:83-98): normalise the raw frames
// (MathUtil.normalize, impl/MathUtil.scala:132-152: fp32 subtraction and IEEE division), the two-pass
// statistics of the loudness row and of the other rows (MathUtil.stat, :29-62: fp64 sums), the centred
// target in fp64 and rounded to fp32, and what the screening bounds need of the fp32 values.
//
// Two kernels, one per pass of MathUtil.stat, each over all elements with four elements per thread (all loads in
// flight at once; the first kernel reads the raw frames from where the host left them, pinned host memory: 145 KB
// for a 30 s target, cheaper to read in place than to copy first).  Every sum is taken in a fixed order that
// depends on nothing but the target's shape -- a thread adds its four elements in index order, the 64 lanes of a
// wave are folded by a butterfly, the 4 waves of a workgroup added in wave order, and the workgroups' sums are added
// in workgroup order (by every workgroup of the second pass for itself; the second pass' by the workgroup that
// finishes last) -- so a target always gets the same statistics; they differ
// from a sequential summation in the reference's order by a few units of the last place of the fp64 sums.
// The scalars go to pinned host memory, followed by a sequence word the host polls: the host needs them as
// kernel arguments of everything that follows.
// ---------------------------------------------------------------------------
constexpr int kTpThreads = 256;
constexpr int kTpTile = kTpThreads * kTpPer;     // elements per workgroup

__global__ __launch_bounds__(kTpThreads) void target_norm_kernel(const float *__restrict__ frames, int L, int numCh,
                                                                 const float *__restrict__ normTab, int normEnabled,
                                                                 float *__restrict__ tnorm, double *__restrict__ state)
{
    __shared__ double red[(kTpThreads / 64) * 2];
    // elements in the order of the interleaved frames (e = frame * numCh + channel): reads and writes of a wave are
    // contiguous
    const unsigned int n = (unsigned int)L * (unsigned int)numCh;
    const unsigned int e0 = blockIdx.x * kTpTile + threadIdx.x;
    float raw[kTpPer];
#pragma unroll
    for (int k = 0; k < kTpPer; k++) {
        const unsigned int e = e0 + k * kTpThreads;
        raw[k] = e < n ? frames[e] : 0.0f;
    }
    double s1[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kTpPer; k++) {
        const unsigned int e = e0 + k * kTpThreads;
        if (e < n) {
            const unsigned int c = e % (unsigned int)numCh;
            float v = raw[k];
            if (normEnabled) v = (v - normTab[c]) / normTab[SGZ_MAX_CHANNELS + c];
            tnorm[e] = v;
            s1[c > 0] += (double)v;
        }
    }
    // (the launch ends here, with the workgroups' partial sums: the second pass adds them up itself)
    tp_wg_sums<kTpThreads, 2>(s1, red, state + 4 + (size_t)blockIdx.x * kTpSums);
}

__global__ __launch_bounds__(kTpThreads) void target_centre_kernel(TargetCentreArgs a)
{
    target_centre_body<kTpThreads>((int)blockIdx.x, (int)gridDim.x, a.L, a.numCh, a.Lpad, a.tnorm, a.aprime, a.aprimeF,
                                   a.state, a.normWgs, a.hflags, a.numFiles, a.report, a.seq);
}

// ---------------------------------------------------------------------------
// Scratch of sgz_corr_rescore_device_noalloc: the partial sums of the exact re-scoring
// ([window][slice][kRsParts] doubles), which sgz_corr_rescore_device keeps in a buffer it allocates on
// demand.  A file-scope device array: every GPU the library runs on holds its own copy, at its own address,
// from the moment the code object is loaded there until the process ends.  Host code must never pass the
// array's name to a kernel (on the host it is only a handle for hipGetSymbolAddress); the address is looked
// up once per device and kept in g_scratch_ptr.
// ---------------------------------------------------------------------------
constexpr size_t kNoAllocScratchBytes = (size_t)4 << 20;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
constexpr int kScratchMaxDevices = 64;
static void *g_scratch_ptr[kScratchMaxDevices];  // per device; null until that device's first sgz_create
static std::mutex g_scratch_mutex;

// Picks the chunks-per-tile of prep_kernel so that its LDS image stays <= 32 KiB
// (the kernel is bandwidth bound: several workgroups per CU matter more than tile size).
int pick_prep_cb(int numCh)
{
    for (int cb = 64; cb >= 2; cb >>= 1) {
        const size_t bytes = (size_t)numCh * (kR * (cb + 1) + 1) * sizeof(float);
        if (bytes <= 32 * 1024) return cb;
    }
    return 1;
}

int build_layout(sgz_context *ctx)
{
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->screen_stream));
    const int L = ctx->cp.L;
    const int S = ctx->cp.S;
    const int nf = (int)ctx->files.size();
    ctx->h_files.assign(nf, FileDesc{});
    ctx->h_tiles.clear();
    ctx->h_ptiles.clear();
    ctx->h_tiles_h.clear();
    ctx->tile_h_begin.assign(nf + 1, 0);
    const bool halfOk = ctx->hg.S2 > 0;
    if (halfOk) {  // tile width of the matrix-core kernel for these files and this target
        std::vector<long long> nw((size_t)nf);
        for (int f = 0; f < nf; f++) {
            const long long fr = std::max<long long>(ctx->files[f].frames - ctx->tail_trim, 0);
            nw[(size_t)f] = fr >= L ? fr - L + 1 : (fr > 0 ? 1 : 0);
        }
        int rcp;
        if ((rcp = half_screen_pick_tile(ctx, nw.data(), nf))) return rcp;
    }
    const long long kHTileW = 256ll * (halfOk ? ctx->hg.NB : kHNB);
    ctx->tile_begin.assign(nf + 1, 0);
    ctx->ptile_begin.assign(nf + 1, 0);
    ctx->prep_cb = pick_prep_cb(ctx->numCh);
    long long planar = 0, prefix = 0, wins = 0, tb = 0;
    const long long prepTileFrames = (long long)ctx->prep_cb * kR;
    ctx->cp.prepTileShift = 0;
    while ((1ll << ctx->cp.prepTileShift) < prepTileFrames) ctx->cp.prepTileShift++;
    // The reference keeps one ring of L frames across all files (impl/FeatureCorrelationImpl.scala:154): a file
    // with fewer than L frames overwrites the first positions only and is scored once, over its own frames
    // followed by whatever earlier files left in the rest of the ring (:190-195).  Such a file is given a private
    // L-frame image: its own frames, then either what the caller supplied (sgz_corr_set_ring_image: the
    // database is spread over several contexts) or what this context's earlier files leave behind
    // (sgz_corr_ring_runs over its own sequence of files).
    for (float *p : ctx->stale_images) SGZ_HIP(ctx, hipFree(p));
    ctx->stale_images.clear();
    std::vector<long long> passFrames((size_t)nf);
    // a pass that stops tail_trim frames before the end of every file (the punch-in pass of a search with
    // punch-out, :182-185) neither scores nor leaves in the ring what lies behind that point
    for (int f = 0; f < nf; f++) passFrames[(size_t)f] = std::max<long long>(ctx->files[f].frames - ctx->tail_trim, 0);
    const std::vector<sgz_ring_run> runs = ring_runs(passFrames.data(), nf, L);
    size_t runPos = 0;
    const size_t rowBytes = sizeof(float) * (size_t)ctx->numCh;
    for (int f = 0; f < nf; f++) {
        FileDesc &fd = ctx->h_files[f];
        fd.raw = ctx->files[f].d_raw;
        fd.frames = passFrames[(size_t)f];
        const long long own = fd.frames;
        if (own > 0 && own < L) {
            float *img = nullptr;
            SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&img), rowBytes * (size_t)L));
            ctx->stale_images.push_back(img);
            SGZ_HIP(ctx, hipMemcpyAsync(img, fd.raw, rowBytes * (size_t)own, hipMemcpyDeviceToDevice, ctx->stream));
            const auto given = ctx->ring_images.find(f);
            if (given != ctx->ring_images.end()) {
                if ((long long)given->second.size() != (L - own) * ctx->numCh)
                    return ctx->fail(SGZ_E_STATE, "the ring image set for file " + std::to_string(f) +
                                                      " does not fit the target length");
                SGZ_HIP(ctx, hipMemcpy(img + own * ctx->numCh, given->second.data(), rowBytes * (size_t)(L - own),
                                       hipMemcpyHostToDevice));
            }
            for (; runPos < runs.size() && runs[runPos].file == f; runPos++) {
                if (given != ctx->ring_images.end()) continue;
                const sgz_ring_run &r = runs[runPos];
                if (r.src_file < 0) {  // never written: raw values that normalise to exactly 0
                    std::vector<float> fill((size_t)r.count * ctx->numCh);
                    for (size_t k = 0; k < fill.size(); k++)
                        fill[k] = ctx->norm.enabled ? ctx->norm.mn[k % (size_t)ctx->numCh] : 0.0f;
                    SGZ_HIP(ctx, hipMemcpy(img + (long long)r.dest * ctx->numCh, fill.data(), sizeof(float) * fill.size(),
                                           hipMemcpyHostToDevice));
                } else {
                    SGZ_HIP(ctx, hipMemcpyAsync(img + (long long)r.dest * ctx->numCh,
                                                ctx->files[(size_t)r.src_file].d_raw + r.src_frame * ctx->numCh,
                                                rowBytes * (size_t)r.count, hipMemcpyDeviceToDevice, ctx->stream));
                }
            }
            fd.raw = img;
            fd.frames = L;
        }
        fd.num_windows = fd.frames >= L ? fd.frames - L + 1 : 0;
        const long long ntiles = (fd.num_windows + kTileW - 1) / kTileW;
        long long M = ntiles * 64 + S + 8;  // slack: staging copies whole 16-byte vectors
        const long long need = (fd.frames + kR - 1) / kR;
        if (M < need) M = need;
        const long long ntilesH = (fd.num_windows + kHTileW - 1) / kHTileW;
        if (halfOk) {  // the half-precision kernel stages 8 * XU frames from the first window of a tile
            const long long needH = (ntilesH * kHTileW + 8ll * ctx->hg.XU + kR - 1) / kR;
            if (M < needH) M = needH;
        }
        M = (M + 3) & ~3ll;
        fd.M = M;
        fd.planar_off = planar;
        fd.prefix_off = prefix;
        fd.win_off = wins;
        planar += (long long)ctx->numCh * kR * M;
        prefix += 4 * fd.frames;
        fd.nptiles = (fd.frames + prepTileFrames - 1) / prepTileFrames;
        fd.tbase_off = tb;
        tb += 4 * fd.nptiles;
        wins += fd.num_windows;
        ctx->tile_begin[f] = (int)ctx->h_tiles.size();
        for (long long t = 0; t < ntiles; t++) ctx->h_tiles.push_back(TileDesc{f, (int)(t * kTileW)});
        ctx->tile_h_begin[f] = (int)ctx->h_tiles_h.size();
        if (halfOk)
            for (long long t = 0; t < ntilesH; t++) {
                HalfTile ht;
                ht.file = f;
                ht.o0 = (int)(t * kHTileW);
                ht.valid = (int)std::min<long long>(kHTileW, fd.num_windows - t * kHTileW);
                ht.pad = 0;
                ht.planar = fd.planar_off + t * kHTileW;
                ht.rowLen = fd.M * kR;
                ht.win = fd.win_off + t * kHTileW;
                ctx->h_tiles_h.push_back(ht);
            }
        ctx->ptile_begin[f] = (int)ctx->h_ptiles.size();
        if (fd.num_windows > 0) {
            for (long long c0 = 0; c0 < M; c0 += ctx->prep_cb) ctx->h_ptiles.push_back(PrepTile{f, (int)c0});
        }
    }
    ctx->tile_begin[nf] = (int)ctx->h_tiles.size();
    ctx->ptile_begin[nf] = (int)ctx->h_ptiles.size();
    ctx->tile_h_begin[nf] = (int)ctx->h_tiles_h.size();
    ctx->total_windows = wins;

    int rc;
    if ((rc = fft_screen_layout(ctx))) return rc;  // (its table is uploaded on ctx->stream, synchronised below)
    if ((rc = ensure(ctx, ctx->d_files, ctx->d_files_cap, (size_t)nf))) return rc;
    if ((rc = ensure(ctx, ctx->d_tiles, ctx->d_tiles_cap, ctx->h_tiles.size()))) return rc;
    if ((rc = ensure(ctx, ctx->d_ptiles, ctx->d_ptiles_cap, ctx->h_ptiles.size()))) return rc;
    if ((rc = ensure(ctx, ctx->d_planar, ctx->planar_cap, (size_t)planar))) return rc;
    if ((rc = ensure(ctx, ctx->d_planar_f, ctx->planar_f_cap, (size_t)planar))) return rc;
    if (halfOk) {
        if ((rc = ensure(ctx, ctx->d_planar_h, ctx->planar_h_cap, (size_t)planar))) return rc;
        if ((rc = ensure(ctx, ctx->d_tiles_h, ctx->d_tiles_h_cap, ctx->h_tiles_h.size()))) return rc;
    }
    if ((rc = ensure(ctx, ctx->d_hflags, ctx->hflags_cap, (size_t)nf))) return rc;
    if (nf) SGZ_HIP(ctx, hipMemsetAsync(ctx->d_hflags, 0, sizeof(int) * nf, ctx->stream));
    ctx->hc_begin = ctx->hc_end = 0;  // no half-precision preparation of this layout yet
    ctx->boost_lazy.assign((size_t)nf, 0);
    ctx->no_dense.assign((size_t)nf, 0);
    ctx->pre_valid = false;
    ctx->prefetch_valid = false;
    if (halfOk) {
        if ((rc = ensure(ctx, ctx->d_prefix_h, ctx->prefix_h_cap, (size_t)prefix))) return rc;
        if ((rc = ensure(ctx, ctx->d_tbase_h, ctx->tbase_h_cap, (size_t)tb))) return rc;
    }
    if ((rc = ensure(ctx, ctx->d_prefix, ctx->prefix_cap, (size_t)prefix))) return rc;
    if ((rc = ensure(ctx, ctx->d_tbase, ctx->tbase_cap, (size_t)tb))) return rc;
    if ((rc = ensure(ctx, ctx->d_sim, ctx->dense_cap, (size_t)wins))) return rc;
    if ((rc = ensure(ctx, ctx->d_boost, ctx->boost_cap, (size_t)wins))) return rc;
    if (halfOk && (rc = ensure(ctx, ctx->d_wstat_h, ctx->wstat_h_cap,
                                std::max((size_t)8 * (size_t)wins, spectral_statics_floats(wins))))) return rc;
    if (halfOk && (rc = ensure(ctx, ctx->d_lokey, ctx->lokey_cap, (size_t)2 * nf))) return rc;
    if (halfOk && nf)  // key of -inf in both arrays; files without windows keep it (no tile ever writes them)
        SGZ_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_lokey), (int)0x807fffff, (size_t)2 * nf,
                                       ctx->stream));
    ctx->lokey_clean[0] = ctx->lokey_clean[1] = true;
    ctx->lokey_parity = 0;
    ctx->lokey_cur = ctx->d_lokey;
    ctx->lo_begin = ctx->lo_end = 0;
    // the three tables are consumed by kernels queued later on ctx->stream
    if (nf) SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_files, ctx->h_files.data(), sizeof(FileDesc) * nf,
                                        hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->h_tiles.empty())
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_tiles, ctx->h_tiles.data(), sizeof(TileDesc) * ctx->h_tiles.size(),
                                    hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->h_tiles_h.empty())
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_tiles_h, ctx->h_tiles_h.data(), sizeof(HalfTile) * ctx->h_tiles_h.size(),
                                    hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->h_ptiles.empty())
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_ptiles, ctx->h_ptiles.data(), sizeof(PrepTile) * ctx->h_ptiles.size(),
                                    hipMemcpyHostToDevice, ctx->stream));
    // which files hold a value that is not a finite number (under the norms in force)
    ctx->h_nonfinite.assign((size_t)nf, 0);
    if (nf) {
        if ((rc = ensure(ctx, ctx->d_nonfinite, ctx->nonfinite_cap, (size_t)nf))) return rc;
        SGZ_HIP(ctx, hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(int) * (size_t)nf, ctx->stream));
        hipLaunchKernelGGL(nonfinite_scan_kernel, dim3((unsigned)nf * kScanParts), dim3(256), 0, ctx->stream, ctx->d_files,
                           ctx->d_normtab, ctx->norm.enabled, ctx->numCh, ctx->d_nonfinite);
        SGZ_HIP(ctx, hipGetLastError());
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->h_nonfinite.data(), ctx->d_nonfinite, sizeof(int) * (size_t)nf,
                                    hipMemcpyDeviceToHost, ctx->stream));
    }
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors may be rebuilt afterwards
    for (auto &r : ctx->runs) ctx->event_pool.push_back(r.done);
    ctx->runs.clear();
    ctx->layout_dirty = false;
    return SGZ_OK;
}

}  // namespace

// Called by sgz_create with the context's device current: looks up, once per device, where that device keeps
// its copy of `scratch`.
int sgz::noalloc_scratch_init(sgz_context *ctx)
{
    if (ctx->device < 0 || ctx->device >= kScratchMaxDevices) {
        ctx->noalloc_scratch = nullptr;  // the entry point then reports that it is not available there
        return SGZ_OK;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[ctx->device]) {
        void *ptr = nullptr;
        SGZ_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return ctx->fail(SGZ_E_HIP, "hipGetSymbolAddress(scratch) returned no address");
        g_scratch_ptr[ctx->device] = ptr;
    }
    ctx->noalloc_scratch = g_scratch_ptr[ctx->device];
    return SGZ_OK;
}

// Device memory for the frames of a database file: a piece of the current slab, or an allocation of its own for a
// large file (f.owned).  256-byte aligned either way.
static int db_file_alloc(sgz_context *ctx, size_t bytes, DbFile &f)
{
    constexpr size_t kSlab = (size_t)32 << 20, kLargest = (size_t)8 << 20;
    const size_t need = (std::max<size_t>(bytes, 4) + 255) & ~(size_t)255;
    if (need > kLargest) {
        SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&f.d_raw), need));
        f.owned = true;
        return SGZ_OK;
    }
    if (need > ctx->db_slab_left) {
        void *slab = nullptr;
        SGZ_HIP(ctx, hipMalloc(&slab, kSlab));
        ctx->db_slabs.push_back(slab);
        ctx->db_slab_cur = static_cast<char *>(slab);
        ctx->db_slab_left = kSlab;
    }
    f.d_raw = reinterpret_cast<float *>(ctx->db_slab_cur);
    f.owned = false;
    ctx->db_slab_cur += need;
    ctx->db_slab_left -= need;
    return SGZ_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int sgz_db_reset(sgz_context *ctx, int num_channels)
{
    if (!ctx) return SGZ_E_ARG;
    if (num_channels < 2 || num_channels > SGZ_MAX_CHANNELS)
        return ctx->fail(SGZ_E_ARG, "num_channels must be in [2, 64]");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->screen_stream));
    for (auto &f : ctx->files)
        if (f.owned && f.d_raw) SGZ_HIP(ctx, hipFree(f.d_raw));
    ctx->files.clear();
    for (void *slab : ctx->db_slabs) (void)hipFree(slab);
    ctx->db_slabs.clear();
    ctx->db_slab_cur = nullptr;
    ctx->db_slab_left = 0;
    ctx->db_tag.clear();
    ctx->db_serial++;
    ctx->ring_images.clear();
    ctx->numCh = num_channels;
    ctx->layout_dirty = true;
    ctx->have_target = false;
    ctx->tail_trim = 0;
    return SGZ_OK;
}

int sgz_corr_set_tail_trim(sgz_context *ctx, int64_t frames)
{
    if (!ctx) return SGZ_E_ARG;
    if (frames < 0) return ctx->fail(SGZ_E_ARG, "tail trim must be >= 0");
    if (frames != ctx->tail_trim) {
        ctx->tail_trim = frames;
        ctx->layout_dirty = true;
    }
    return SGZ_OK;
}

int sgz_db_add_file(sgz_context *ctx, const float *frames, int64_t num_frames)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (num_frames < 0 || (num_frames > 0 && !frames)) return ctx->fail(SGZ_E_ARG, "bad frames / num_frames");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    DbFile f;
    f.frames = num_frames;
    const size_t bytes = sizeof(float) * (size_t)num_frames * ctx->numCh;
    if (const int arc = db_file_alloc(ctx, bytes, f)) return arc;
    if (bytes && hipMemcpy(f.d_raw, frames, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        if (f.owned) (void)hipFree(f.d_raw);
        return ctx->fail(SGZ_E_HIP, "upload of the frames failed");
    }
    ctx->files.push_back(f);
    ctx->db_tag.clear();
    ctx->db_serial++;
    ctx->layout_dirty = true;
    return SGZ_OK;
}

int sgz_db_add_file_pcm(sgz_context *ctx, const void *samples, int sample_format, int64_t num_frames)
{
    if (!ctx) return SGZ_E_ARG;
    if (sample_format == SGZ_PCM_F32) return sgz_db_add_file(ctx, static_cast<const float *>(samples), num_frames);
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (num_frames < 0 || (num_frames > 0 && !samples)) return ctx->fail(SGZ_E_ARG, "bad samples / num_frames");
    int sampleBytes, bigEndian, isFloat;
    if (sgz::pcm_format(sample_format, &sampleBytes, &bigEndian, &isFloat)) return ctx->fail(SGZ_E_ARG, "unknown sample format");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    DbFile f;
    f.frames = num_frames;
    const size_t count = (size_t)num_frames * ctx->numCh, bytes = sizeof(float) * count, rawBytes = (size_t)sampleBytes * count;
    if (const int arc = db_file_alloc(ctx, bytes, f)) return arc;
    if (count && sampleBytes == 4) {
        // a layout as wide as the decoded frames (what feature files hold: big-endian floats): the file's bytes go
        // straight to the file's place and are decoded there, on the stream every later reader is ordered after --
        // no staging buffer, nothing to wait for
        int rc = SGZ_OK;
        if (hipMemcpy(f.d_raw, samples, rawBytes, hipMemcpyHostToDevice) != hipSuccess) rc = ctx->fail(SGZ_E_HIP, "upload of the samples failed");
        if (!rc) rc = sgz::pcm_decode_launch(ctx, f.d_raw, (long long)count, sample_format, f.d_raw, ctx->stream);
        if (rc) {
            if (f.owned) (void)hipFree(f.d_raw);
            return rc;
        }
    } else if (count) {
        // the file's own bytes go up (into a buffer kept on the context) and are decoded there: no pass over the
        // samples on the host
        if (rawBytes + 16 > ctx->pcm_stage_cap) {
            if (ctx->d_pcm_stage) (void)hipFree(ctx->d_pcm_stage);
            ctx->d_pcm_stage = nullptr;
            ctx->pcm_stage_cap = 0;
            const size_t want = rawBytes + rawBytes / 4 + 16;
            if (hipMalloc(&ctx->d_pcm_stage, want) != hipSuccess) {
                if (f.owned) (void)hipFree(f.d_raw);
                return ctx->fail(SGZ_E_HIP, "hipMalloc of the sample staging buffer failed");
            }
            ctx->pcm_stage_cap = want;
        }
        int rc = SGZ_OK;
        if (hipMemcpy(ctx->d_pcm_stage, samples, rawBytes, hipMemcpyHostToDevice) != hipSuccess) rc = ctx->fail(SGZ_E_HIP, "upload of the samples failed");
        if (!rc) rc = sgz::pcm_decode_launch(ctx, ctx->d_pcm_stage, (long long)count, sample_format, f.d_raw, ctx->stream);
        // (the staging buffer is reused by the next file: its decode must have read it)
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ctx->fail(SGZ_E_HIP, "sample decode failed");
        if (rc) {
            if (f.owned) (void)hipFree(f.d_raw);
            return rc;
        }
    }
    ctx->files.push_back(f);
    ctx->db_tag.clear();
    ctx->db_serial++;
    ctx->layout_dirty = true;
    return SGZ_OK;
}

int sgz_db_add_file_device(sgz_context *ctx, const float *device_frames, int64_t num_frames)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (num_frames < 0 || (num_frames > 0 && !device_frames))
        return ctx->fail(SGZ_E_ARG, "bad device_frames / num_frames");
    if (reinterpret_cast<uintptr_t>(device_frames) & 15)
        return ctx->fail(SGZ_E_ARG, "device_frames must be 16-byte aligned");
    DbFile f;
    f.frames = num_frames;
    f.owned = false;
    f.d_raw = const_cast<float *>(device_frames);
    ctx->files.push_back(f);
    ctx->db_tag.clear();
    ctx->db_serial++;
    ctx->layout_dirty = true;
    return SGZ_OK;
}

int sgz_db_add_synthetic(sgz_context *ctx, uint64_t seed, int64_t file_id, int64_t num_frames,
                         const sgz_plant *plants, int num_plants)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (num_frames < 0 || num_plants < 0 || (num_plants > 0 && !plants))
        return ctx->fail(SGZ_E_ARG, "bad num_frames / plants");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    // only the plants of this file travel to the GPU
    std::vector<sgz_plant> mine;
    for (int k = 0; k < num_plants; k++)
        if (plants[k].file == file_id) mine.push_back(plants[k]);
    DbFile f;
    f.frames = num_frames;
    const size_t elems = (size_t)num_frames * ctx->numCh;
    if (const int arc = db_file_alloc(ctx, elems * sizeof(float), f)) return arc;
    sgz_plant *d_pl = nullptr;
    if (!mine.empty()) {
        SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d_pl), sizeof(sgz_plant) * mine.size()));
        SGZ_HIP(ctx, hipMemcpy(d_pl, mine.data(), sizeof(sgz_plant) * mine.size(), hipMemcpyHostToDevice));
    }
    if (elems) {
        const int grid = (int)std::min<size_t>((elems + 255) / 256, 8192);
        hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(256), 0, ctx->stream, f.d_raw,
                           (unsigned long long)seed, (long long)file_id, (long long)num_frames, ctx->numCh, d_pl,
                           (int)mine.size());
        SGZ_HIP(ctx, hipGetLastError());
    }
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d_pl) SGZ_HIP(ctx, hipFree(d_pl));
    ctx->files.push_back(f);
    ctx->db_tag.clear();
    ctx->db_serial++;
    ctx->layout_dirty = true;
    return SGZ_OK;
}

int sgz_db_num_files(const sgz_context *ctx) { return ctx ? (int)ctx->files.size() : 0; }

int64_t sgz_db_file_frames(const sgz_context *ctx, int file)
{
    if (!ctx || file < 0 || file >= (int)ctx->files.size()) return -1;
    return ctx->files[file].frames;
}

int sgz_db_read_file(sgz_context *ctx, int file, float *frames_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (file < 0 || file >= (int)ctx->files.size() || !frames_out) return ctx->fail(SGZ_E_ARG, "bad file / buffer");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    const DbFile &f = ctx->files[file];
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (f.frames)
        SGZ_HIP(ctx, hipMemcpy(frames_out, f.d_raw, sizeof(float) * (size_t)f.frames * ctx->numCh,
                               hipMemcpyDeviceToHost));
    return SGZ_OK;
}

int sgz_db_set_tag(sgz_context *ctx, const char *tag)
{
    if (!ctx) return SGZ_E_ARG;
    ctx->db_tag = tag ? tag : "";
    return SGZ_OK;
}

const char *sgz_db_tag(const sgz_context *ctx) { return ctx ? ctx->db_tag.c_str() : ""; }

int sgz_db_read_frames(sgz_context *ctx, int file, int64_t first, int64_t count, float *frames_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (file < 0 || file >= (int)ctx->files.size()) return ctx->fail(SGZ_E_ARG, "bad file index");
    const DbFile &f = ctx->files[file];
    if (first < 0 || count < 0 || first + count > f.frames || (count > 0 && !frames_out))
        return ctx->fail(SGZ_E_ARG, "frames outside the file / no buffer");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (count)
        SGZ_HIP(ctx, hipMemcpy(frames_out, f.d_raw + first * ctx->numCh, sizeof(float) * (size_t)count * ctx->numCh,
                               hipMemcpyDeviceToHost));
    return SGZ_OK;
}

int sgz_corr_set_ring_image(sgz_context *ctx, int file, const float *frames, int num_frames)
{
    if (!ctx) return SGZ_E_ARG;
    if (file < 0 || file >= (int)ctx->files.size()) return ctx->fail(SGZ_E_ARG, "bad file index");
    if (num_frames < 0 || (num_frames > 0 && !frames)) return ctx->fail(SGZ_E_ARG, "bad ring image");
    // (the same image again -- a caller that hands it over with every search -- leaves the layout as it is)
    const auto it = ctx->ring_images.find(file);
    if (!frames) {
        if (it == ctx->ring_images.end()) return SGZ_OK;
        ctx->ring_images.erase(it);
    } else {
        const size_t n = (size_t)num_frames * ctx->numCh;
        if (it != ctx->ring_images.end() && it->second.size() == n &&
            std::memcmp(it->second.data(), frames, sizeof(float) * n) == 0)
            return SGZ_OK;
        ctx->ring_images[file].assign(frames, frames + n);
    }
    ctx->layout_dirty = true;
    return SGZ_OK;
}

int sgz_norm_zero_frame(const sgz_context *ctx, float *frame_out)
{
    if (!ctx || !frame_out || ctx->numCh == 0) return SGZ_E_ARG;
    for (int c = 0; c < ctx->numCh; c++) frame_out[c] = ctx->norm.enabled ? ctx->norm.mn[c] : 0.0f;
    return SGZ_OK;
}

int sgz_set_norms(sgz_context *ctx, const float *norms)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    NormParams nrm;
    std::memset(&nrm, 0, sizeof nrm);
    if (norms) {
        nrm.enabled = 1;
        for (int c = 0; c < ctx->numCh; c++) {
            nrm.mn[c] = norms[2 * c];
            nrm.d[c] = norms[2 * c + 1] - norms[2 * c];
        }
    }
    // the same norms again (a caller that sets them with every query): everything derived from them stays valid
    if (ctx->norms_set && std::memcmp(&nrm, &ctx->norm, sizeof nrm) == 0) return SGZ_OK;
    ctx->norms_set = true;
    ctx->norm = nrm;
    ctx->have_target = false;  // target statistics depend on the norms
    ctx->layout_dirty = true;  // so do the images of files shorter than the target
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rc;
    if ((rc = ensure(ctx, ctx->d_normtab, ctx->normtab_cap, (size_t)2 * SGZ_MAX_CHANNELS))) return rc;
    float tab[2 * SGZ_MAX_CHANNELS];
    for (int c = 0; c < SGZ_MAX_CHANNELS; c++) {
        tab[c] = ctx->norm.mn[c];
        tab[SGZ_MAX_CHANNELS + c] = ctx->norm.enabled ? ctx->norm.d[c] : 1.0f;
    }
    SGZ_HIP(ctx, hipMemcpy(ctx->d_normtab, tab, sizeof tab, hipMemcpyHostToDevice));
    return SGZ_OK;
}

// frames_on_device: the raw frames are in memory of the context's GPU already (what a query's feature extraction
// leaves there); the first kernel then reads them where they are.
static int set_target_impl(sgz_context *ctx, const float *frames, int num_frames, float temporal_weight, float max_boost,
                           bool frames_on_device)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (!frames || num_frames < 1) return ctx->fail(SGZ_E_ARG, "target needs at least one frame");
    if (((long long)num_frames + 2 * kR) * ctx->numCh >= (1ll << 31)) return ctx->fail(SGZ_E_ARG, "target too long");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    const int numCh = ctx->numCh, L = num_frames;
    // The host hands over the raw frames and nothing else: they are written into a pinned buffer that
    // target_prep_kernel reads in place; normalisation, statistics and the centred copies are device work
    // (a query service sets a new target per search).  Pinned layout: raw frames, the kernel's report, and the
    // staging of the matrix-core kernel's target image (fp64 centred target + the image, filled when that kernel
    // first runs with this target).
    int S = (L + kR - 1) / kR;
    S = (S + 1) & ~1;
    const size_t Lpad = (size_t)S * kR;
    const size_t nTb = (size_t)numCh * L, nAp = (size_t)numCh * Lpad;
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // (the report: the first pass' two sums, the second pass' partial sums per unit of 1024 elements, a sequence word
    // per workgroup)
    const size_t tpUnits = (nAp + kTpTile - 1) / kTpTile;
    const size_t repBytes = sizeof(double) * (2 + tpUnits * kTpSums) + sizeof(unsigned long long) * tpUnits;
    const size_t offRaw = 0, offRep = up256(offRaw + sizeof(float) * nTb), offAp = up256(offRep + repBytes);
    const size_t offImg = up256(offAp + sizeof(double) * nAp);
    const size_t imgHalfwords = sgz::half_image_halfwords(L, numCh);
    const size_t stageBytes = up256(offImg + sizeof(uint16_t) * imgHalfwords);
    // Kernels of the search before read the old target on `stream`, where the new one is written in order behind
    // them; the fp32 screening reads it on the other stream.  The previous target's kernel has finished with the
    // pinned frames (its report was waited for).
    if (ctx->screen_stream_used) SGZ_HIP(ctx, hipStreamSynchronize(ctx->screen_stream));
    if (stageBytes > ctx->tgt_stage_cap) {
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (a copy out of the old buffer may be on its way)
        if (ctx->h_tgt_stage) (void)hipHostFree(ctx->h_tgt_stage);
        ctx->h_tgt_stage = nullptr;
        ctx->tgt_stage_cap = 0;
        SGZ_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_tgt_stage), stageBytes, hipHostMallocDefault));
        ctx->tgt_stage_cap = stageBytes;
    }
    if (!frames_on_device) std::memcpy(ctx->h_tgt_stage + offRaw, frames, sizeof(float) * nTb);
    TargetReportLayout repl;
    repl.sum1 = reinterpret_cast<double *>(ctx->h_tgt_stage + offRep);
    repl.rows = repl.sum1 + 2;
    repl.flags = reinterpret_cast<unsigned long long *>(repl.rows + tpUnits * kTpSums);
    int rc;
    if ((rc = ensure(ctx, ctx->d_aprime, ctx->aprime_cap, nAp))) return rc;
    if ((rc = ensure(ctx, ctx->d_tnorm, ctx->tnorm_cap, nTb))) return rc;
    if ((rc = ensure(ctx, ctx->d_aprime_f, ctx->aprime_f_cap, nAp))) return rc;
    const unsigned long long seq = ++ctx->tgt_seq;
    // (the report may sit at another offset than last time: the words the host polls must not hold this number yet)
    for (size_t w = 0; w < tpUnits; w++) __atomic_store_n(repl.flags + w, 0ull, __ATOMIC_RELEASE);
    // (the second pass runs over the padded copies, in workgroups of 256 threads or -- carried along by the launch
    // that transforms the target -- of more: the larger grid sizes the buffer of partial sums)
    const int tpGridA = (int)((nTb + kTpTile - 1) / kTpTile), tpGrid = (int)((nAp + kTpTile - 1) / kTpTile);
    if ((rc = ensure(ctx, ctx->d_tgt_state, ctx->tgt_state_cap, (size_t)4 + (size_t)tpGridA * kTpSums))) return rc;
    // the marks of the previous target's screening runs go with it (a layout that is rebuilt clears all marks itself)
    const int nfMarks = ctx->d_hflags && ctx->hflags_cap >= ctx->files.size() && !ctx->layout_dirty ? (int)ctx->files.size() : 0;
    hipLaunchKernelGGL(target_norm_kernel, dim3(tpGridA), dim3(kTpThreads), 0, ctx->stream,
                       frames_on_device ? frames : reinterpret_cast<const float *>(ctx->h_tgt_stage + offRaw), L, numCh,
                       ctx->d_normtab,
                       ctx->norm.enabled, ctx->d_tnorm, ctx->d_tgt_state);
    SGZ_HIP(ctx, hipGetLastError());
    CorrParams &cp = ctx->cp;
    const int oldL = ctx->have_target ? cp.L : -1;
    cp.numCh = numCh;
    cp.L = L;
    cp.S = S;
    cp.nb = (S + kSBMax - 1) / kSBMax;
    int SB = (S + cp.nb - 1) / cp.nb;
    SB = (SB + 3) & ~3;  // staged blocks start on 16-byte boundaries for both element types
    cp.SB = SB;
    cp.w = temporal_weight;
    cp.maxBoost = max_boost;
    ctx->aspec_dirty = true;
    TargetCentreArgs ca;
    ca.L = L;
    ca.numCh = numCh;
    ca.Lpad = (int)Lpad;
    ca.tnorm = ctx->d_tnorm;
    ca.aprime = ctx->d_aprime;
    ca.aprimeF = ctx->d_aprime_f;
    ca.state = ctx->d_tgt_state;
    ca.normWgs = tpGridA;
    ca.hflags = ctx->d_hflags;
    ca.numFiles = nfMarks;
    ca.report = repl;
    ca.seq = seq;
    // Where the spectral kernel screens and the layout stays, the launch that transforms the target needs nothing of
    // the report: it is queued now and carries the second pass along, so that the host's wait for the scalars, the
    // centring and the transform overlap.
    const bool early = oldL == L && !ctx->layout_dirty && ctx->fg.N > 0;
    if (early) {
        if ((rc = fft_target_spectra(ctx, ctx->stream, &ca))) return rc;
    } else {
        hipLaunchKernelGGL(target_centre_kernel, dim3(tpGrid), dim3(kTpThreads), 0, ctx->stream, ca);
        SGZ_HIP(ctx, hipGetLastError());
    }
    {
        // every workgroup of the second pass reports by itself (those of the fused launch are 256 threads as well: a
        // workgroup per unit)
        const auto t0 = std::chrono::steady_clock::now();
        auto allThere = [&]() {
            for (size_t w = 0; w < tpUnits; w++)
                if (__atomic_load_n(repl.flags + w, __ATOMIC_ACQUIRE) != seq) return false;
            return true;
        };
        bool arrived = false;
        for (unsigned long long spins = 0;; spins++) {
            if (allThere()) {
                arrived = true;
                break;
            }
            if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) break;
        }
        if (!arrived) {  // something is wrong on the device: the stream says what
            SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!allThere()) return ctx->fail(SGZ_E_HIP, "the target preparation kernel did not report");
        }
    }
    // the units' sums in unit order
    TargetReport repSums{};
    TargetReport *rep = &repSums;
    {
        double fin[kTpSums] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (size_t u = 0; u < tpUnits; u++)
            for (int j = 0; j < kTpSums; j++) fin[j] += repl.rows[u * kTpSums + j];
        rep->sum1[0] = repl.sum1[0];
        rep->sum1[1] = repl.sum1[1];
        rep->sum2[0] = fin[0];
        rep->sum2[1] = fin[1];
        rep->csum[0] = fin[2];
        rep->csum[1] = fin[3];
        rep->fsum[0] = fin[4];
        rep->fsum[1] = fin[5];
        rep->fsq[0] = fin[6];
        rep->fsq[1] = fin[7];
        rep->fsq1 = fin[8];
        rep->inRange = fin[9] == 0.0 ? 1ull : 0ull;
    }
    // MathUtil.stat: mean = sum / n, sd = sqrt(sum of squared deviations / n)
    const double nT = (double)L, nS = (double)L * (double)(numCh - 1);
    const double tMean = rep->sum1[0] / nT, sMean = rep->sum1[1] / nS;
    const double tStd = std::sqrt(rep->sum2[0] / nT), sStd = std::sqrt(rep->sum2[1] / nS);
    const float avgF = (float)(rep->sum1[0] / L);  // MathUtil.avg over the loudness row
    cp.kT = tMean;
    cp.kS = sMean;
    cp.aStdT = tStd;
    cp.aStdS = sStd;
    cp.lnAvgIn = std::log((double)avgF);
    cp.sumAT = rep->csum[0];
    cp.sumAS = rep->csum[1];
    ctx->tgt_nrm1 = std::sqrt(rep->fsq1);
    ctx->tgt_sumT = rep->fsum[0];
    ctx->tgt_sumS = rep->fsum[1];
    ctx->tgt_nrmT = std::sqrt(rep->fsq[0]);
    ctx->tgt_nrmS = std::sqrt(rep->fsq[1]);
    ctx->aMeanT = tMean;
    ctx->aMeanS = sMean;
    const bool finite = rep->inRange != 0;
    ctx->have_target = true;
    ctx->tgt_serial++;
    // half-precision target image of the matrix-core screening kernel (none when the geometry does not fit)
    const bool hadHalf = ctx->hg.S2 > 0;
    // (the image itself is built when that kernel first runs with this target: it takes the fp64 target back from
    // the device and converts 36 000 values to half precision on the host, which takes longer than the rest of this
    // call, and searches screened by the spectral kernel never read it)
    ctx->half_stage_ap = reinterpret_cast<double *>(ctx->h_tgt_stage + offAp);
    ctx->half_stage_lpad = Lpad;
    ctx->half_stage_img = reinterpret_cast<uint16_t *>(ctx->h_tgt_stage + offImg);
    ctx->half_stage_halfwords = imgHalfwords;
    if ((rc = half_screen_set_target(ctx, finite))) return rc;
    // (the fp32 screening runs on the other stream: it waits for the preparation.  A context that has never run it
    // -- every search over bounds -- is spared the two calls; the first such run waits for the stream itself.)
    if (ctx->screen_stream_used) {
        if (!ctx->tgt_event) SGZ_HIP(ctx, hipEventCreateWithFlags(&ctx->tgt_event, hipEventDisableTiming));
        SGZ_HIP(ctx, hipEventRecord(ctx->tgt_event, ctx->stream));
        SGZ_HIP(ctx, hipStreamWaitEvent(ctx->screen_stream, ctx->tgt_event, 0));
    }
    if (oldL != L || hadHalf != (ctx->hg.S2 > 0)) ctx->layout_dirty = true;
    if (early && !ctx->layout_dirty) fft_target_ready(ctx);
    return SGZ_OK;
}

int sgz_corr_set_target(sgz_context *ctx, const float *frames, int num_frames, float temporal_weight,
                        float max_boost)
{
    return set_target_impl(ctx, frames, num_frames, temporal_weight, max_boost, false);
}

int sgz_corr_set_target_device(sgz_context *ctx, const float *device_frames, int num_frames, float temporal_weight,
                               float max_boost)
{
    return set_target_impl(ctx, device_frames, num_frames, temporal_weight, max_boost, true);
}

int sgz_corr_target_stats(const sgz_context *ctx, double *stats_out)
{
    if (!ctx || !stats_out) return SGZ_E_ARG;
    if (!ctx->have_target) return ctx->fail(SGZ_E_STATE, "sgz_corr_set_target has not been called");
    stats_out[0] = ctx->cp.kT;
    stats_out[1] = ctx->cp.aStdT;
    stats_out[2] = ctx->cp.kS;
    stats_out[3] = ctx->cp.aStdS;
    stats_out[4] = ctx->cp.lnAvgIn;
    return SGZ_OK;
}

int64_t sgz_corr_num_windows(const sgz_context *ctx, int file)
{
    if (!ctx || !ctx->have_target || file < 0 || file >= (int)ctx->files.size()) return -1;
    const long long n = std::max<long long>(ctx->files[file].frames - ctx->tail_trim, 0);
    if (n >= ctx->cp.L) return n - ctx->cp.L + 1;
    return n > 0 ? 1 : 0;  // the single window over the ring's stale contents
}

enum RunMode { kRunExact = 0, kRunScreenF32 = 1, kRunScreenF16 = 2 };

static int corr_run_impl(sgz_context *ctx, int file_begin, int file_end, int mode);
static int rescore_device_impl(sgz_context *ctx, sgz_candidate *device_cands, int64_t capacity,
                               const unsigned long long *device_count, const HostReport &hr, bool truncated_ok = false,
                               double *part_noalloc = nullptr);
constexpr int64_t kSuspectCap = 1 << 16;  // windows an exact run can re-evaluate from their frames

int sgz_corr_run(sgz_context *ctx, int file_begin, int file_end)
{
    return corr_run_impl(ctx, file_begin, file_end, kRunExact);
}

int sgz_corr_run_screen(sgz_context *ctx, int file_begin, int file_end)
{
    return corr_run_impl(ctx, file_begin, file_end, kRunScreenF32);
}

int sgz_corr_screen_f16_available(const sgz_context *ctx) { return ctx && half_screen_fits(ctx) ? 1 : 0; }

int sgz_corr_run_screen_f16(sgz_context *ctx, int file_begin, int file_end)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->have_target && !half_screen_fits(ctx))
        return ctx->fail(SGZ_E_STATE, "half-precision screening needs normalised features and a target that fits LDS");
    return corr_run_impl(ctx, file_begin, file_end, kRunScreenF16);
}

static int corr_run_impl(sgz_context *ctx, int file_begin, int file_end, int mode)
{
    const bool exact = mode == kRunExact;
    if (!ctx) return SGZ_E_ARG;
    if (!ctx->have_target) return ctx->fail(SGZ_E_STATE, "sgz_corr_set_target has not been called");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (ctx->layout_dirty && (rc = build_layout(ctx))) return rc;
    ctx->ev_valid = false;
    ctx->last_windows = 0;
    ctx->last_flops = 0.0;
    if (file_begin == file_end) return SGZ_OK;
    if (ctx->prefetch_valid && file_begin < ctx->prefetch_fe && file_end > ctx->prefetch_fb)
        ctx->prefetch_valid = false;  // these files get new values

    const int pt0 = ctx->ptile_begin[file_begin], pt1 = ctx->ptile_begin[file_end];
    const bool half = mode == kRunScreenF16;
    const bool spectral = half && ctx->fg.N > 0;  // the bounds come from the spectral kernel instead
    const int t0 = half ? ctx->tile_h_begin[file_begin] : ctx->tile_begin[file_begin];
    const int t1 = half ? ctx->tile_h_begin[file_end] : ctx->tile_begin[file_end];
    long long wins = 0;
    for (int f = file_begin; f < file_end; f++) wins += ctx->h_files[f].num_windows;
    ctx->last_windows = wins;
    ctx->last_flops = (double)wins * 2.0 * ctx->cp.numCh * ctx->cp.L;
    if (t1 == t0) return SGZ_OK;

    // kernel timing: every run, or one run in timing_every (sgz_set_kernel_timing)
    const bool timed = ctx->timing_every > 0 && (ctx->timing_count++ % ctx->timing_every) == 0;
    if (timed && (rc = ctx->next_event_pair())) return rc;
    const int cb = ctx->prep_cb;
    const size_t prepLds = (size_t)ctx->numCh * (kR * (cb + 1) + 1) * sizeof(float);
    // The screening path has its own planar image (fp32) and its own stream, so that a screening run
    // over one file range overlaps with an exact run over another.
    float *planarF = ctx->d_planar_f;
    // (the matrix-core screening is compute bound like the exact kernel: nothing to gain from a second
    // stream, and its consumers on ctx->stream then need no cross-stream wait)
    hipStream_t stream = mode == kRunScreenF32 ? ctx->screen_stream : ctx->stream;
    if (mode == kRunScreenF32 && !ctx->screen_stream_used) {
        // the first run on the other stream: whatever the main stream holds of the target's preparation comes first
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->screen_stream_used = true;
    }
    // an earlier run over some of these files (possibly on the other stream) writes the same outputs
    for (auto &run : ctx->runs)
        if (run.fb < file_end && run.fe > file_begin && (rc = ctx->wait_for_run(run, stream))) return rc;
    // Half precision: image, prefix records, tile bases, range marks and centring constants depend on database,
    // norms and target length only; they are built when a run first covers these files and kept until the layout
    // is rebuilt.  A search with a new target of the same length then starts with the matrix-core kernel.
    const bool halfCached = half && ctx->hc_begin <= file_begin && file_end <= ctx->hc_end;
    if (half && !halfCached) {
        SGZ_HIP(ctx, hipMemsetAsync(ctx->d_hflags + file_begin, 0, sizeof(int) * (size_t)(file_end - file_begin), stream));
        hipLaunchKernelGGL(file_const_kernel, dim3(file_end - file_begin), dim3(1024), 0, stream, ctx->d_files, file_begin,
                           ctx->norm, ctx->numCh);
        SGZ_HIP(ctx, hipGetLastError());
    }
    if (exact)
        hipLaunchKernelGGL(prep_kernel<double>, dim3(pt1 - pt0), dim3(256), prepLds, stream, ctx->d_files,
                           ctx->d_ptiles + pt0, ctx->d_planar, ctx->d_prefix, ctx->d_tbase, ctx->norm, ctx->numCh, cb,
                           ctx->cp.kT, ctx->cp.kS, ctx->d_hflags);
    else if (halfCached)
        ;
    else if (half && ctx->numCh == 14 && cb * kR == kPrepHalfFrames)  // the reference's default: 13 MFCC + loudness
        hipLaunchKernelGGL(prep_half_kernel<14>, dim3(pt1 - pt0), dim3(256), 0, stream, ctx->d_files,
                           ctx->d_ptiles + pt0, reinterpret_cast<_Float16 *>(ctx->d_planar_h),
                           reinterpret_cast<float4 *>(ctx->d_prefix_h), ctx->d_tbase_h, ctx->norm, ctx->d_hflags);
    else if (half)
        hipLaunchKernelGGL(prep_kernel<_Float16>, dim3(pt1 - pt0), dim3(256), prepLds, stream, ctx->d_files,
                           ctx->d_ptiles + pt0, reinterpret_cast<_Float16 *>(ctx->d_planar_h),
                           reinterpret_cast<double *>(ctx->d_prefix_h), ctx->d_tbase_h, ctx->norm, ctx->numCh, cb, 0.0, 0.0,
                           ctx->d_hflags);
    else
        hipLaunchKernelGGL(prep_kernel<float>, dim3(pt1 - pt0), dim3(256), prepLds, stream, ctx->d_files,
                           ctx->d_ptiles + pt0, planarF, ctx->d_prefix, ctx->d_tbase, ctx->norm, ctx->numCh, cb,
                           ctx->cp.kT, ctx->cp.kS, ctx->d_hflags);
    SGZ_HIP(ctx, hipGetLastError());
    if (!halfCached) {
        hipLaunchKernelGGL(tile_scan_kernel, dim3(file_end - file_begin), dim3(256), 0, stream, ctx->d_files,
                           file_begin, half ? ctx->d_tbase_h : ctx->d_tbase);
        SGZ_HIP(ctx, hipGetLastError());
    }
    if (half && !halfCached) {
        const int s0 = ctx->tile_begin[file_begin], s1 = ctx->tile_begin[file_end];
        if (s1 > s0) {
            hipLaunchKernelGGL(window_static_kernel, dim3(s1 - s0), dim3(256), 0, stream, ctx->d_files, ctx->d_tiles + s0,
                               reinterpret_cast<const float4 *>(ctx->d_prefix_h), ctx->d_tbase_h, ctx->cp.L, ctx->numCh,
                               ctx->cp.prepTileShift, reinterpret_cast<float4 *>(ctx->d_wstat_h),
                               ctx->fg.N > 0 ? ctx->total_windows : 0ll);
            SGZ_HIP(ctx, hipGetLastError());
        }
        ctx->hc_begin = file_begin;
        ctx->hc_end = file_end;
    }
    if (spectral && !(ctx->fc_begin <= file_begin && file_end <= ctx->fc_end) &&
        (rc = fft_screen_prepare(ctx, file_begin, file_end, stream)))
        return rc;
    const size_t corrLds = exact ? corr_lds_bytes<double>() : corr_lds_bytes<float>();
    // windows whose prefix-sum statistics cannot be trusted (constant stretches): see SuspectList
    SuspectList suspects{nullptr, nullptr, 0, INFINITY, nullptr};
    if (exact) {
        if ((rc = ensure(ctx, ctx->d_suspect, ctx->suspect_cap, (size_t)kSuspectCap + 1))) return rc;
        suspects.list = ctx->d_suspect + 1;
        suspects.count = reinterpret_cast<unsigned long long *>(ctx->d_suspect);
        suspects.capacity = kSuspectCap;
        suspects.fill = NAN;
        suspects.wholeFile = ctx->d_nonfinite;
        SGZ_HIP(ctx, hipMemsetAsync(ctx->d_suspect, 0, sizeof(sgz_candidate), stream));
    }
    // kernel time for sgz_corr_last_kernel_ms: the matrix-core kernel carries the two events in its own
    // dispatch (no separate marker packets, which would each cost a few idle microseconds on the stream)
    if (!half && timed) SGZ_HIP(ctx, hipEventRecord(ctx->ev0, stream));
    if (spectral) {
        if ((rc = fft_screen_launch(ctx, file_begin, file_end, stream, timed ? ctx->ev0 : nullptr,
                                    timed ? ctx->ev1 : nullptr)))
            return rc;
    } else if (half) {
        if ((rc = half_screen_launch(ctx, t0, t1, stream, timed ? ctx->ev0 : nullptr, timed ? ctx->ev1 : nullptr)))
            return rc;
    } else if (exact)
        hipLaunchKernelGGL(corr_kernel<double>, dim3(t1 - t0), dim3(kThreads), corrLds, stream, ctx->d_files,
                           ctx->d_tiles + t0, ctx->d_planar, ctx->d_prefix, ctx->d_tbase, ctx->d_aprime, ctx->cp,
                           ctx->d_sim, ctx->d_boost, suspects);
    else
        hipLaunchKernelGGL(corr_kernel<float>, dim3(t1 - t0), dim3(kThreads), corrLds, stream, ctx->d_files,
                           ctx->d_tiles + t0, planarF, ctx->d_prefix, ctx->d_tbase, ctx->d_aprime_f, ctx->cp,
                           ctx->d_sim, ctx->d_boost, suspects);
    SGZ_HIP(ctx, hipGetLastError());
    if (!half && timed) SGZ_HIP(ctx, hipEventRecord(ctx->ev1, stream));
    if (exact) {
        // two-pass evaluation of the listed windows, written back into the dense arrays (three small kernels
        // that return at once when the list is empty); windows beyond the list's capacity stay NaN
        if ((rc = rescore_device_impl(ctx, suspects.list, kSuspectCap, suspects.count, HostReport{}, true))) return rc;
        hipLaunchKernelGGL(scatter_kernel, dim3(64), dim3(256), 0, stream, ctx->d_files, suspects.list, suspects.count,
                           (long long)kSuspectCap, ctx->d_sim, ctx->d_boost);
        SGZ_HIP(ctx, hipGetLastError());
        // files that hold a value that is not a number: every window from its frames, a list's worth at a time
        for (int f = file_begin; f < file_end; f++) {
            if (f >= (int)ctx->h_nonfinite.size() || !ctx->h_nonfinite[(size_t)f]) continue;
            const long long nw = ctx->h_files[(size_t)f].num_windows;
            for (long long o0 = 0; o0 < nw; o0 += kSuspectCap) {
                const long long n = std::min<long long>(kSuspectCap, nw - o0);
                hipLaunchKernelGGL(list_windows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, f, o0, n,
                                   suspects.list, suspects.count);
                SGZ_HIP(ctx, hipGetLastError());
                if ((rc = rescore_device_impl(ctx, suspects.list, kSuspectCap, suspects.count, HostReport{}, true))) return rc;
                hipLaunchKernelGGL(scatter_kernel, dim3(64), dim3(256), 0, stream, ctx->d_files, suspects.list,
                                   suspects.count, (long long)kSuspectCap, ctx->d_sim, ctx->d_boost);
                SGZ_HIP(ctx, hipGetLastError());
            }
        }
    }
    if (timed) {
        ctx->ev_valid = true;
        ctx->ev_runs++;
    }
    if (!spectral) {  // every other kernel writes the boosts and the per-window results of its windows
        for (int f = file_begin; f < file_end && f < (int)ctx->boost_lazy.size(); f++) ctx->boost_lazy[(size_t)f] = 0;
        for (int f = file_begin; f < file_end && f < (int)ctx->no_dense.size(); f++) ctx->no_dense[(size_t)f] = 0;
        if (ctx->pre_valid && file_begin < ctx->pre_fe && file_end > ctx->pre_fb) ctx->pre_valid = false;
    }
    if (half) {
        ctx->lo_begin = file_begin;
        ctx->lo_end = file_end;
    } else if (file_begin < ctx->lo_end && file_end > ctx->lo_begin) {
        ctx->lo_begin = ctx->lo_end = 0;  // these files no longer hold bounds
    }
    // completion marker of this run, so that reading results of an earlier run
    // does not have to wait for later ones
    {
        sgz_context::Run run;
        run.fb = file_begin;
        run.fe = file_end;
        if (!ctx->event_pool.empty()) {
            run.done = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            SGZ_HIP(ctx, hipEventCreateWithFlags(&run.done, hipEventDisableTiming));
        }
        run.stream = stream;
        run.recorded = false;
        // a newer run over the same files supersedes older markers
        for (size_t i = 0; i < ctx->runs.size();) {
            if (ctx->runs[i].fb >= file_begin && ctx->runs[i].fe <= file_end) {
                ctx->event_pool.push_back(ctx->runs[i].done);
                ctx->runs.erase(ctx->runs.begin() + (std::ptrdiff_t)i);
            } else {
                i++;
            }
        }
        ctx->runs.push_back(run);
    }
    return SGZ_OK;
}

// Pinned staging of dense read-backs: room for `count` similarities and as many boosts (up to 16 MB each; a copy
// straight into pageable memory is staged by the runtime in small pieces, each with its own wait) and the
// sequence word of a prefetch.  False when the range is larger or no pinned memory is to be had.
static bool dense_staging(sgz_context *ctx, size_t count)
{
    constexpr size_t kPinFloats = (size_t)4 << 20;
    if (count == 0 || count > kPinFloats) return false;
    if (ctx->dense_pin_floats < count) {
        // (a prefetch kernel may still be writing into the buffer that is about to go)
        if (ctx->h_dense_pin && ctx->prefetch_seq > 0) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->h_dense_pin) (void)hipHostFree(ctx->h_dense_pin);
        ctx->h_dense_pin = nullptr;
        ctx->dense_pin_floats = 0;
        ctx->prefetch_valid = false;
        const size_t want = std::min(kPinFloats, std::max(count, (size_t)1 << 18));
        if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_dense_pin), 2 * want * sizeof(float) + 64, hipHostMallocDefault) !=
            hipSuccess) {
            ctx->h_dense_pin = nullptr;
            return false;
        }
        ctx->dense_pin_floats = want;
        std::memset(ctx->h_dense_pin + 2 * want, 0, 64);
    }
    return true;
}

int sgz_corr_prefetch_dense(sgz_context *ctx, int file_begin, int file_end)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    ctx->prefetch_valid = false;
    if (file_begin == file_end) return SGZ_OK;
    {
        const int drc = require_dense(ctx, file_begin, file_end);
        if (drc) return drc;
    }
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    const long long w0 = ctx->h_files[file_begin].win_off;
    const long long w1 = ctx->h_files[file_end - 1].win_off + ctx->h_files[file_end - 1].num_windows;
    if (!dense_staging(ctx, (size_t)(w1 - w0))) return SGZ_OK;  // too large: the read will copy the plain way
    {
        const int brc = boost_materialize(ctx, file_begin, file_end, ctx->stream);
        if (brc) return brc;
    }
    bool covered = false;
    for (auto &run : ctx->runs)
        if (run.fb < file_end && run.fe > file_begin) {
            const int wrc = ctx->wait_for_run(run, ctx->stream);
            if (wrc) return wrc;
            covered = true;
        }
    if (!covered) return ctx->fail(SGZ_E_STATE, "sgz_corr_run has not covered these files");
    int rc;
    if ((rc = ensure(ctx, ctx->d_prefetch_done, ctx->prefetch_done_cap, (size_t)1))) return rc;
    if (ctx->prefetch_seq == 0) SGZ_HIP(ctx, hipMemsetAsync(ctx->d_prefetch_done, 0, sizeof(unsigned int), ctx->stream));
    const unsigned long long seq = ++ctx->prefetch_seq;
    const int grid = (int)std::min<long long>((w1 - w0 + 255) / 256, 512);
    hipLaunchKernelGGL(dense_to_host_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->d_sim + w0, ctx->d_boost + w0,
                       w1 - w0, ctx->h_dense_pin, ctx->h_dense_pin + ctx->dense_pin_floats, ctx->d_prefetch_done,
                       reinterpret_cast<unsigned long long *>(ctx->h_dense_pin + 2 * ctx->dense_pin_floats), seq);
    SGZ_HIP(ctx, hipGetLastError());
    ctx->prefetch_valid = true;
    ctx->prefetch_fb = file_begin;
    ctx->prefetch_fe = file_end;
    return SGZ_OK;
}

int sgz_corr_read_dense_range(sgz_context *ctx, int file_begin, int file_end, float *sim_out, float *boost_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    if (file_begin == file_end) return SGZ_OK;
    {
        const int drc = require_dense(ctx, file_begin, file_end);
        if (drc) return drc;
    }
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    // wait (on the copy stream only) for the runs that produced these files
    bool covered = false;
    for (auto &run : ctx->runs) {
        if (run.fb < file_end && run.fe > file_begin) {
            const int wrc = ctx->wait_for_run(run, ctx->copy_stream);
            if (wrc) return wrc;
            covered = true;
        }
    }
    if (!covered) return ctx->fail(SGZ_E_STATE, "sgz_corr_run has not covered these files");
    if (boost_out && !(ctx->prefetch_valid && ctx->prefetch_fb == file_begin && ctx->prefetch_fe == file_end)) {
        // boosts the spectral kernel left implicit (the copies below run on another stream)
        bool filled = false;
        const int brc = boost_materialize(ctx, file_begin, file_end, ctx->stream, &filled);
        if (brc) return brc;
        if (filled) SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const long long w0 = ctx->h_files[file_begin].win_off;
    const long long w1 = ctx->h_files[file_end - 1].win_off + ctx->h_files[file_end - 1].num_windows;
    const size_t count = (size_t)(w1 - w0), bytes = sizeof(float) * count;
    if (ctx->prefetch_valid && ctx->prefetch_fb == file_begin && ctx->prefetch_fe == file_end) {
        // a copy by kernel is on its way (or done): wait for its sequence word
        ctx->prefetch_valid = false;
        const unsigned long long *flag =
            reinterpret_cast<const unsigned long long *>(ctx->h_dense_pin + 2 * ctx->dense_pin_floats);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived = false;
        for (unsigned long long spins = 0;; spins++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ctx->prefetch_seq) {
                arrived = true;
                break;
            }
            if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) break;
        }
        if (arrived) {
            if (sim_out) std::memcpy(sim_out, ctx->h_dense_pin, bytes);
            if (boost_out) std::memcpy(boost_out, ctx->h_dense_pin + ctx->dense_pin_floats, bytes);
            return SGZ_OK;
        }
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // something is wrong on the device: this reports it
    }
    // (while a prefetch of other files is pending the staging buffer is its: copy the plain way; an abandoned
    // prefetch -- its files were run again -- may still be writing: let it finish first)
    if (!ctx->prefetch_valid && ctx->prefetch_seq > 0 && ctx->h_dense_pin &&
        __atomic_load_n(reinterpret_cast<const unsigned long long *>(ctx->h_dense_pin + 2 * ctx->dense_pin_floats),
                        __ATOMIC_ACQUIRE) != ctx->prefetch_seq)
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const bool staged = !ctx->prefetch_valid && dense_staging(ctx, count);
    float *simTo = staged ? ctx->h_dense_pin : sim_out, *boostTo = staged ? ctx->h_dense_pin + ctx->dense_pin_floats : boost_out;
    if (bytes && sim_out)
        SGZ_HIP(ctx, hipMemcpyAsync(simTo, ctx->d_sim + w0, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    if (bytes && boost_out)
        SGZ_HIP(ctx, hipMemcpyAsync(boostTo, ctx->d_boost + w0, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (staged) {
        if (sim_out) std::memcpy(sim_out, simTo, bytes);
        if (boost_out) std::memcpy(boost_out, boostTo, bytes);
    }
    return SGZ_OK;
}

int sgz_corr_read_dense(sgz_context *ctx, int file, float *sim_out, float *boost_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (file < 0 || file >= (int)ctx->files.size()) return ctx->fail(SGZ_E_ARG, "bad file index");
    return sgz_corr_read_dense_range(ctx, file, file + 1, sim_out, boost_out);
}

// `zero_other`: a second counter the kernel clears for the caller's next call; with `count_is_zero` the
// caller vouches that device_count was cleared that way.  *launched: whether a kernel ran (and cleared it).
static int compact_device_impl(sgz_context *ctx, int file_begin, int file_end, float threshold,
                               sgz_candidate *device_out, int64_t capacity, unsigned long long *device_count,
                               bool count_is_zero, unsigned long long *zero_other, bool *launched)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    if (!device_count || capacity < 0 || (capacity > 0 && !device_out))
        return ctx->fail(SGZ_E_ARG, "bad output buffers");
    {
        const int drc = require_dense(ctx, file_begin, file_end);
        if (drc) return drc;
    }
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    for (auto &run : ctx->runs)
        if (run.fb < file_end && run.fe > file_begin) {
            const int wrc = ctx->wait_for_run(run, ctx->stream);
            if (wrc) return wrc;
        }
    if (!count_is_zero) SGZ_HIP(ctx, hipMemsetAsync(device_count, 0, sizeof(unsigned long long), ctx->stream));
    const int t0 = ctx->tile_begin[file_begin], t1 = ctx->tile_begin[file_end];
    if (launched) *launched = t1 > t0;
    if (t1 > t0) {
        const float *lnAvgLazy = nullptr;
        int brc;
        if ((brc = boost_source(ctx, file_begin, file_end, ctx->stream, &lnAvgLazy))) return brc;
        hipLaunchKernelGGL(compact_kernel, dim3(t1 - t0), dim3(256), 0, ctx->stream, ctx->d_files,
                           ctx->d_tiles + t0, ctx->d_sim, ctx->d_boost, threshold, device_out,
                           (long long)capacity, device_count, ctx->d_hflags, ctx->hc_begin, ctx->hc_end, zero_other,
                           lnAvgLazy, (float)ctx->cp.lnAvgIn);
        SGZ_HIP(ctx, hipGetLastError());
    }
    return SGZ_OK;
}

int sgz_corr_compact_device(sgz_context *ctx, int file_begin, int file_end, float threshold,
                            sgz_candidate *device_out, int64_t capacity, unsigned long long *device_count)
{
    return compact_device_impl(ctx, file_begin, file_end, threshold, device_out, capacity, device_count, false,
                               nullptr, nullptr);
}

// top_k > 0: the header that starts with the count also takes, as a float in its bytes 8..11, the top_k-th
// largest bound of the files [0, head_end), and head files are filtered by it as well.
static int compact_bounded_device_impl(sgz_context *ctx, int file_begin, int file_end, int head_end,
                                       sgz_candidate *device_out, int64_t capacity,
                                       unsigned long long *device_count, bool count_is_zero,
                                       unsigned long long *zero_other, bool *launched, int top_k = 0)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    if (head_end < 1 || head_end > nf) return ctx->fail(SGZ_E_ARG, "bad head_end");
    if (!device_count || capacity < 0 || (capacity > 0 && !device_out))
        return ctx->fail(SGZ_E_ARG, "bad output buffers");
    const int need_end = std::max(file_end, head_end);
    if (ctx->lo_begin != 0 || ctx->lo_end < need_end)
        return ctx->fail(SGZ_E_STATE, "files are not covered by one sgz_corr_run_screen_f16 run starting at file 0");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->d_thr, ctx->thr_cap, (size_t)1))) return rc;
    if (!count_is_zero) SGZ_HIP(ctx, hipMemsetAsync(device_count, 0, sizeof(unsigned long long), ctx->stream));
    for (auto &run : ctx->runs)
        if (run.fb < need_end && (rc = ctx->wait_for_run(run, ctx->stream))) return rc;
    if (head_end < file_end) {  // tail files present: they are filtered by the smallest head bound
        hipLaunchKernelGGL(bound_min_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->lokey_cur, head_end, ctx->d_thr);
        SGZ_HIP(ctx, hipGetLastError());
    }
    // (the header that starts with the count takes the top_k-th largest bound as a float in its bytes 8..11)
    float *topThr = top_k > 0 ? reinterpret_cast<float *>(device_count) + 2 : nullptr;
    const int t0 = ctx->tile_begin[file_begin], t1 = ctx->tile_begin[file_end];
    if (launched) *launched = t1 > t0;
    if (t1 > t0 && ctx->pre_valid && ctx->pre_fb == file_begin && ctx->pre_fe == file_end && head_end >= file_end) {
        // the run left the list of the windows that can hold their file's maximum instead of dense upper bounds
        const int pp = ctx->pre_run_parity;
        hipLaunchKernelGGL(prelist_filter_kernel, dim3(32), dim3(256), 0, ctx->stream, ctx->d_prelist, ctx->d_precount + pp,
                           ctx->pre_capacity, ctx->d_precount + (1 - pp), ctx->lokey_cur, head_end, ctx->d_hflags,
                           file_begin, file_end, device_out, (long long)capacity, device_count, zero_other, top_k, topThr);
        SGZ_HIP(ctx, hipGetLastError());
        ctx->pre_clean[1 - pp] = true;
        ctx->pre_parity = 1 - pp;
        return SGZ_OK;
    }
    if ((rc = require_dense(ctx, file_begin, file_end))) return rc;
    if (t1 > t0) {
        // four tiles per workgroup and more: enough workgroups to fill the chip, few enough that the bound
        // selection each of them repeats stays cheap
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const int grid = std::max(1, std::min((t1 - t0 + 3) / 4, 8 * cus));
        const float *lnAvgLazy = nullptr;
        if ((rc = boost_source(ctx, file_begin, file_end, ctx->stream, &lnAvgLazy))) return rc;
        hipLaunchKernelGGL(compact_bounded_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->d_files,
                           ctx->d_tiles + t0, t1 - t0, ctx->d_sim, ctx->d_boost, ctx->lokey_cur, ctx->d_thr, head_end,
                           device_out, (long long)capacity, device_count, ctx->d_hflags, zero_other, top_k, topThr,
                           lnAvgLazy, (float)ctx->cp.lnAvgIn);
        SGZ_HIP(ctx, hipGetLastError());
    } else if (top_k > 0) {
        hipLaunchKernelGGL(bound_kth_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->lokey_cur, head_end, top_k, topThr);
        SGZ_HIP(ctx, hipGetLastError());
    }
    return SGZ_OK;
}

int sgz_corr_compact_bounded_device(sgz_context *ctx, int file_begin, int file_end, int head_end,
                                    sgz_candidate *device_out, int64_t capacity, unsigned long long *device_count)
{
    return compact_bounded_device_impl(ctx, file_begin, file_end, head_end, device_out, capacity, device_count, false,
                                       nullptr, nullptr);
}

static int rescore_device_impl(sgz_context *ctx, sgz_candidate *device_cands, int64_t capacity,
                               const unsigned long long *device_count, const HostReport &hr, bool truncated_ok,
                               double *part_noalloc)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    if (!device_count || capacity < 0 || (capacity > 0 && !device_cands)) return ctx->fail(SGZ_E_ARG, "bad buffers");
    if (capacity == 0) return SGZ_OK;
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    // A slice holds about 4 K values (with the few dozen windows a search over bounds leaves, 4 K beat 12 K by a
    // third; with hundreds of windows the two are within 2 %); the number of candidates is only known on the device, so a fixed grid
    // of small workgroups strides over the (window, slice) pairs.
    const long long perWindow = (long long)ctx->cp.L * ctx->cp.numCh;
    const long long sliceVals = 4096;
    RescoreGeom rg;
    rg.slices = (int)std::min<long long>(std::max<long long>((perWindow + sliceVals - 1) / sliceVals, 1), 64);
    rg.sliceFrames = (ctx->cp.L + rg.slices - 1) / rg.slices;
    rg.slices = (ctx->cp.L + rg.sliceFrames - 1) / rg.sliceFrames;  // no empty slice
    int rc;
    double *part = part_noalloc;
    const float *normtab = ctx->d_normtab;  // (exists from sgz_create on)
    if (part_noalloc) {
        // the variant that allocates nothing: partial sums in the device's scratch array.  Every value read is
        // written earlier in the same call, so nothing has to be cleared.
        const size_t perRecord = (size_t)rg.slices * kRsParts * sizeof(double);
        if ((size_t)capacity > kNoAllocScratchBytes / perRecord)  // (by division: the product may not fit)
            return ctx->fail(SGZ_E_ARG, "sgz_corr_rescore_device_noalloc: capacity " + std::to_string(capacity) + " x " +
                                            std::to_string(rg.slices) + " slices of this target needs more than the " +
                                            std::to_string(kNoAllocScratchBytes) + " bytes of scratch (" +
                                            std::to_string(kNoAllocScratchBytes / perRecord) + " records at most)");
    } else {
        if ((rc = ensure(ctx, ctx->d_rs_part, ctx->rs_part_cap, (size_t)capacity * rg.slices * kRsParts))) return rc;
        part = ctx->d_rs_part;
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int grid = (int)std::min<long long>((long long)capacity * rg.slices, 8ll * cus);
    const float *tframes = ctx->d_tnorm;
    hipLaunchKernelGGL(rescore_pass_kernel, dim3(grid), dim3(kRsThreads), 0, ctx->stream, ctx->d_files, tframes,
                       normtab, ctx->norm.enabled, ctx->cp, rg, ctx->aMeanT, ctx->aMeanS, device_cands, device_count,
                       (long long)capacity, (int)truncated_ok, part);
    SGZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(rescore_final_kernel, dim3((int)std::min<int64_t>((capacity + 255) / 256, 64)), dim3(256), 0,
                       ctx->stream, ctx->d_files, normtab, ctx->norm.enabled, ctx->cp, rg, device_cands, device_count,
                       (long long)capacity, (int)truncated_ok, part, hr);
    SGZ_HIP(ctx, hipGetLastError());
    return SGZ_OK;
}

int sgz_corr_compact_top_device(sgz_context *ctx, int file_begin, int file_end, int num_top,
                                sgz_candidate *device_out, int64_t capacity, unsigned long long *device_header)
{
    if (!ctx) return SGZ_E_ARG;
    if (num_top < 1) return ctx->fail(SGZ_E_ARG, "bad num_top");
    return compact_bounded_device_impl(ctx, file_begin, file_end, (int)ctx->files.size(), device_out, capacity,
                                       device_header, false, nullptr, nullptr, num_top);
}

int sgz_corr_rescore_device(sgz_context *ctx, sgz_candidate *device_cands, int64_t capacity,
                            const unsigned long long *device_count)
{
    if (!ctx) return SGZ_E_ARG;
    return rescore_device_impl(ctx, device_cands, capacity, device_count, HostReport{});
}

int sgz_corr_rescore_device_noalloc(sgz_context *ctx, sgz_candidate *device_cands, int64_t capacity,
                                    const unsigned long long *device_count)
{
    if (!ctx) return SGZ_E_ARG;
    if (!ctx->noalloc_scratch)
        return ctx->fail(SGZ_E_STATE, "sgz_corr_rescore_device_noalloc: no scratch array on this device");
    return rescore_device_impl(ctx, device_cands, capacity, device_count, HostReport{}, false,
                               reinterpret_cast<double *>(ctx->noalloc_scratch));
}

static int compact_host(sgz_context *ctx, int file_begin, int file_end, float threshold, bool rescore,
                        sgz_candidate *out, int64_t capacity, int64_t *count, int head_end = 0, int top_k = 0,
                        float *top_out = nullptr);

int sgz_corr_compact_top(sgz_context *ctx, int file_begin, int file_end, int num_top, sgz_candidate *out,
                         int64_t capacity, int64_t *count, float *threshold_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (num_top < 1 || !threshold_out) return ctx->fail(SGZ_E_ARG, "bad num_top / threshold_out");
    if (ctx->files.empty()) return ctx->fail(SGZ_E_STATE, "no database");
    return compact_host(ctx, file_begin, file_end, 0.0f, true, out, capacity, count, (int)ctx->files.size(), num_top,
                        threshold_out);
}

int sgz_corr_compact_bounded(sgz_context *ctx, int file_begin, int file_end, int head_end, sgz_candidate *out,
                             int64_t capacity, int64_t *count)
{
    if (!ctx) return SGZ_E_ARG;
    if (head_end < 1) return ctx->fail(SGZ_E_ARG, "bad head_end");
    return compact_host(ctx, file_begin, file_end, 0.0f, true, out, capacity, count, head_end);
}

int sgz_corr_compact_rescored(sgz_context *ctx, int file_begin, int file_end, float threshold, sgz_candidate *out,
                              int64_t capacity, int64_t *count)
{
    return compact_host(ctx, file_begin, file_end, threshold, true, out, capacity, count);
}

int sgz_corr_compact(sgz_context *ctx, int file_begin, int file_end, float threshold, sgz_candidate *out,
                     int64_t capacity, int64_t *count)
{
    return compact_host(ctx, file_begin, file_end, threshold, false, out, capacity, count);
}

static int compact_host(sgz_context *ctx, int file_begin, int file_end, float threshold, bool rescore,
                        sgz_candidate *out, int64_t capacity, int64_t *count, int head_end, int top_k, float *top_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (!count || capacity < 0 || (capacity > 0 && !out)) return ctx->fail(SGZ_E_ARG, "bad output buffers");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    // device buffer = [two record-sized headers, each holding a count][capacity records].  Calls alternate
    // between the two counters; the compaction kernel of one call clears the counter of the next, which
    // saves a clear (and its idle time on the stream) per call.
    const sgz_candidate *before = ctx->d_cand;
    if ((rc = ensure(ctx, ctx->d_cand, ctx->cand_cap, (size_t)capacity + 2))) return rc;
    if (ctx->d_cand != before) {  // fresh memory: no stale record may ever be taken for a candidate
        SGZ_HIP(ctx, hipMemsetAsync(ctx->d_cand, 0, sizeof(sgz_candidate) * ctx->cand_cap, ctx->stream));
        ctx->cand_clean = true;
    }
    unsigned long long *d_count = reinterpret_cast<unsigned long long *>(ctx->d_cand + ctx->cand_parity);
    unsigned long long *d_other = reinterpret_cast<unsigned long long *>(ctx->d_cand + (1 - ctx->cand_parity));
    sgz_candidate *d_rec = ctx->d_cand + 2;
    const bool clean = ctx->cand_clean;
    ctx->cand_clean = false;  // until this call has got as far as its kernel
    bool launched = false;
    if (head_end > 0)
        rc = compact_bounded_device_impl(ctx, file_begin, file_end, head_end, d_rec, capacity, d_count, clean, d_other,
                                         &launched, top_k);
    else
        rc = compact_device_impl(ctx, file_begin, file_end, threshold, d_rec, capacity, d_count, clean, d_other,
                                 &launched);
    if (rc) return rc;
    const int parity = ctx->cand_parity;
    ctx->cand_parity = 1 - parity;
    ctx->cand_clean = launched;
    // first transfer: headers + as many records as usually suffice
    constexpr size_t kPinned = 16384;  // records in the pinned buffer, the two headers included
    const size_t first = (size_t)std::min<int64_t>(capacity, (int64_t)kPinned - 2);
    if (!ctx->h_pinned) {  // + the sequence word of a host report
        SGZ_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_pinned), sizeof(sgz_candidate) * (kPinned + 1)));
        std::memset(ctx->h_pinned, 0, sizeof(sgz_candidate) * (kPinned + 1));
        // every record slot starts as "not written" (file index -1), see the safety net below
        std::memset(ctx->h_pinned + 2, 0xff, sizeof(sgz_candidate) * (kPinned - 2));
    }
    bool reported = false;
    if (rescore && capacity > 0) {
        // the last workgroup of the re-scoring writes headers and records to the pinned buffer and then the
        // sequence number the host polls for
        if ((rc = ensure(ctx, ctx->d_done, ctx->done_cap, (size_t)1))) return rc;
        if (!ctx->done_init) {
            SGZ_HIP(ctx, hipMemsetAsync(ctx->d_done, 0, sizeof(unsigned int), ctx->stream));
            ctx->done_init = true;
        }
        HostReport hr;
        hr.host = reinterpret_cast<unsigned long long *>(ctx->h_pinned);
        hr.dev = reinterpret_cast<const unsigned long long *>(ctx->d_cand);
        hr.done = ctx->d_done;
        hr.headWords = 4;
        hr.maxWords = 2 * (int)kPinned;
        hr.seq = ++ctx->report_seq;
        // Safety net: the slots the previous report filled are set to an impossible file index (-1) before the
        // kernels start; a record that has not arrived when the sequence word has is recognised below.
        if (ctx->report_prev_n)
            std::memset(ctx->h_pinned + 2, 0xff, sizeof(sgz_candidate) * ctx->report_prev_n);
        ctx->report_prev_n = 0;
        if ((rc = rescore_device_impl(ctx, d_rec, capacity, d_count, hr))) return rc;
        const unsigned long long *flag = hr.host + hr.maxWords;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned long long spins = 0;; spins++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == hr.seq) {
                reported = true;
                break;
            }
            if ((spins & 0xffff) == 0xffff &&
                std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
                break;  // something is wrong on the device: let the stream wait below report it
        }
    }
    if (reported) {
        unsigned long long nr = 0;
        std::memcpy(&nr, ctx->h_pinned + parity, sizeof nr);
        const size_t inRep = (size_t)std::min<unsigned long long>({nr, (unsigned long long)capacity, (unsigned long long)first});
        for (size_t i = 0; i < inRep; i++)
            if (ctx->h_pinned[2 + i].file < 0) reported = false;  // not all there: fetch the list the plain way
        ctx->report_prev_n = inRep;
    }
    if (!reported) {
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_cand, sizeof(sgz_candidate) * (first + 2),
                                    hipMemcpyDeviceToHost, ctx->stream));
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->report_prev_n = first;  // these slots now hold valid-looking records: invalidated before the next report
    }
    unsigned long long n = 0;
    std::memcpy(&n, ctx->h_pinned + parity, sizeof n);
    *count = (int64_t)n;
    if (top_out) std::memcpy(top_out, reinterpret_cast<const char *>(ctx->h_pinned + parity) + 8, sizeof(float));
    const size_t got = (size_t)std::min<unsigned long long>(n, (unsigned long long)capacity);
    const size_t inFirst = std::min(got, first);
    if (inFirst) std::memcpy(out, ctx->h_pinned + 2, sizeof(sgz_candidate) * inFirst);
    if (got > first)
        SGZ_HIP(ctx, hipMemcpy(out + first, d_rec + first, sizeof(sgz_candidate) * (got - first),
                               hipMemcpyDeviceToHost));
    std::sort(out, out + got, [](const sgz_candidate &a, const sgz_candidate &b) {
        return a.file != b.file ? a.file < b.file : a.offset < b.offset;
    });
    if (n > (unsigned long long)capacity) return ctx->fail(SGZ_E_OVERFLOW, "candidate buffer too small");
    return SGZ_OK;
}

float sgz_corr_last_kernel_ms(sgz_context *ctx)
{
    if (!ctx || !ctx->ev_valid) return -1.0f;
    if (hipSetDevice(ctx->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}

float sgz_last_kernel_ms(sgz_context *ctx) { return sgz_corr_last_kernel_ms(ctx); }

int sgz_set_kernel_timing(sgz_context *ctx, int every)
{
    if (!ctx || every < 0) return SGZ_E_ARG;
    ctx->timing_every = every;
    ctx->timing_count = 0;
    return SGZ_OK;
}

int sgz_kernel_ms_history(sgz_context *ctx, float *out, int n)
{
    if (!ctx || n < 0 || (n > 0 && !out)) return SGZ_E_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return SGZ_E_HIP;
    const size_t ring = ctx->ev_ring.size();
    const size_t have = std::min<size_t>({(size_t)n, ctx->ev_runs, ring});
    for (size_t k = 0; k < have; k++) {  // oldest first
        const auto &pr = ctx->ev_ring[(ctx->ev_pos + ring - (have - 1 - k)) % ring];
        float ms = -1.0f;
        if (hipEventSynchronize(pr.second) != hipSuccess || hipEventElapsedTime(&ms, pr.first, pr.second) != hipSuccess)
            ms = -1.0f;
        out[k] = ms;
    }
    return (int)have;
}

int sgz_corr_last_work(const sgz_context *ctx, int64_t *windows, double *flops)
{
    if (!ctx) return SGZ_E_ARG;
    if (windows) *windows = ctx->last_windows;
    if (flops) *flops = ctx->last_flops;
    return SGZ_OK;
}

}  // extern "C"
