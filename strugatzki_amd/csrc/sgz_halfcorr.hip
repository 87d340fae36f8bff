// FeatureSegmentation and SelfSimilarity on gfx950: both are built on
// MathUtil.correlateHalf (impl/MathUtil.scala:80-99): joint mean / std-dev over a
// window of 2H frames, then the normalised product sum of its two halves.
//
// correlate_half_dev (segmentation, and the pixels of the self-similarity matrix
// that need it) keeps the reference's summation order (channel-major, ring slot
// order for the statistics, frame order for the cross term), plain
// multiply-then-add in fp64, so its results agree with a sequential CPU
// evaluation to the last bit.  The self-similarity matrix as a whole is a matrix
// product on the fp32 matrix cores with fp64 window statistics (further down):
// within 1e-5 of that evaluation, typically 1e-7.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
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
// (a call, not inline code: the power function is some hundred instructions, and the tile kernels evaluate the colour
// map in loops that have to be unrolled)
__device__ __noinline__ float gray_warp(float m0, double warp) { return (float)pow((double)m0, warp); }

__device__ __forceinline__ int gray_color(float sim, double warp, float scale, int inv)
{
    const float m0 = (sim != sim) ? sim : (sim > 0.0f ? sim : 0.0f);
    const float v = (warp == 1.0 ? m0 : gray_warp(m0, warp)) * scale;  // pow(x, 1) == x
    const float s = inv ? (1.0f - v) : v;
    const double d = (double)(s * 255) + 0.5;
    int i;
    if (d != d) i = 0;
    else if (d >= 255.0) i = 255;
    else if (d <= 0.0) i = 0;
    else i = (int)d;
    return (i << 16) | (i << 8) | i;
}

// ---------------------------------------------------------------------------
// Tiled self-similarity.  correlateHalf over a window made of the half windows a (file 1 at x) and b (file 2 at
// y), n values each, is a function of five sums over values shifted by any common constant:
//     S = sum(a) + sum(b),  Q = sum(a^2) + sum(b^2),  P = sum(a * b)
//     joint mean m = S / 2n,  sum (v - m)^2 = Q - S^2 / 2n,  sum (a - m)(b - m) = P - S^2 / 4n
//     sim = 2 (P - S^2 / 4n) / (Q - S^2 / 2n)
// selfsim_gather_kernel stores every window as a column of n values minus the window's own mean level mu (so the
// stored values are small and mostly exact differences) together with their fp64 sums.  With d = mu_a - mu_b
// the sums in the frame of b are  S = Sa + n d + Sb,  Q = Qa + 2 d Sa + n d^2 + Qb,  P = P'' + d Sb,  where only
// P'' = sum(a'' b'') couples x and y: the matrix product of the two column sets.  A workgroup of
// selfsim_tile_kernel evaluates it for 128 x 128 pixels on the matrix cores in fp32 (v_mfma_f32_32x32x2_f32) over
// runs of 64 products that are folded into fp64; everything else is fp64.
// The error of P'' is at most 67 * 2^-24 * sqrt(Qa Qb); a pixel for which that could move the similarity by more
// than 8e-6 is evaluated by correlate_half_dev instead.
// ---------------------------------------------------------------------------
constexpr int kSsTile = 128;   // pixels per workgroup tile edge
constexpr int kSsRun = 32;     // products staged at a time (two of these per fp32 run)

struct SelfSimStat {
    double S, Q;    // sum and sum of squares of the stored (shifted) values
    double mu;      // the shift
    double SR, QR;  // sum and sum of squares of the values themselves
    double pad;
};

// Window w (frame afStart + w*decim, H frames) of planar matrix x -> column w of cols[numCh * H][stride]: rows
// [0, H) = channel 0 (temporal group), rows [H, numCh * H) = channels 1.. (spectral group), each minus the mean
// level of its group in this window; stats[2 w + group].  Columns in [imgExt, stride) are zero.
__global__ __launch_bounds__(256) void selfsim_gather_kernel(const float *__restrict__ x, long long frames, int numCh,
                                                             int afStart, int H, int decim, int imgExt,
                                                             long long stride, float *__restrict__ cols,
                                                             SelfSimStat *__restrict__ stats)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= stride) return;
    if (w >= imgExt) {
        for (long long e = 0; e < (long long)numCh * H; e++) cols[e * stride + w] = 0.0f;
        return;
    }
    const long long f0 = (long long)afStart + w * decim;
    for (int group = 0; group < 2; group++) {
        const int c0 = group == 0 ? 0 : 1, c1 = group == 0 ? 1 : numCh;
        double sum = 0.0;
        for (int c = c0; c < c1; c++)
            for (int k = 0; k < H; k++) sum += (double)x[(long long)c * frames + f0 + k];
        float mu = (float)(sum / ((double)(c1 - c0) * H));
        if (!(fabsf(mu) < 3.0e38f)) mu = 0.0f;
        double S = 0.0, Q = 0.0;
        for (int c = c0; c < c1; c++)
            for (int k = 0; k < H; k++) {
                const float v = x[(long long)c * frames + f0 + k] - mu;
                cols[((long long)c * H + k) * stride + w] = v;
                S += (double)v;
                Q = fma((double)v, (double)v, Q);
            }
        SelfSimStat st;
        st.S = S;
        st.Q = Q;
        st.mu = (double)mu;
        const double cnt = (double)(c1 - c0) * H;
        st.SR = S + cnt * (double)mu;
        st.QR = Q + 2.0 * (double)mu * S + cnt * (double)mu * (double)mu;
        st.pad = 0.0;
        stats[2 * w + group] = st;
    }
}

struct SelfSimArgs {
    const float *x1, *x2;            // planar normalised [numCh][frames]
    long long frames1, frames2;
    const float *cols1, *cols2;      // gathered windows [numCh * H][stride]
    const SelfSimStat *st1, *st2;
    long long stride;
    int numCh, afStart, H, decim, imgExt;
    int tilesEdge, superEdge;        // tiles and blocks of 8 x 8 tiles per edge of the matrix
    int c0, c1;                      // channel group of this pass
    int first, last;                 // first pass: no earlier part to read; last pass: writes the final values
    float wThis;                     // weight of this group's similarity
    double warp;
    float scale;
    int inv;
    float *simOut;                   // [imgExt][imgExt]; between two passes: the weighted first part, upper triangle
    int wantSim;                     // the last pass writes the matrix
    int keepUpper;                   // the last pass writes at least the half with y >= x (wantSim, or for the list)
    int *imgOut;
    float wT;                        // temporal weight of the whole similarity
    // pixels whose gray level the certified error of the similarity could change: listed by selfsim_list_kernel,
    // evaluated as the reference does by selfsim_fix_kernel
    unsigned long long *fixCount;
    int2 *fixList;
    long long fixCap;
    // of those, the few whose quotient needs the reference's own order of additions (selfsim_seq_kernel)
    unsigned int *seqCount;
    struct SelfSimSeq *seqList;
    unsigned int seqCap;
    int forceSeq;  // test knob: every listed pixel goes that way
    float slack;   // test knob: added to every pixel's error bound by selfsim_tile8_kernel (more uncertain pixels)
    // selfsim_tile8_kernel: the gathered windows as three 8-bit digits per value, [digit][unit of 16 values][column]
    // in 16-byte units (the temporal group's kgT units first, then the other channels' kgS), and what the digits
    // leave out per window
    const uint4 *limbs1, *limbs2;
    const struct SelfSimLimbs *sp1, *sp2;
    int kgT, kgS;
    int doT, doS;  // channel groups with a weight (the reference skips a group whose weight is 0)
};

struct SelfSimSeq {
    int px, py;
    float temporal, spectral;  // the group that is certain already
    int redo;                  // bit 0: temporal, bit 1: spectral to be evaluated in sequence
    int pad;
};

// Largest error of a similarity that left selfsim_pixel as "sure": 8e-6 from the fp32 product sums (both groups,
// weights adding to 1) and about 1e-6 from the fp32 quotient and mix.
constexpr float kSsSimErr = 1.0e-5f;

__device__ __noinline__ float correlate_half_exact(const float *xl, long long strideL, long long left, const float *xr,
                                                   long long strideR, long long right, int H, int c0, int c1)
{
    return correlate_half_dev(xl, strideL, left, xr, strideR, right, H, 0, c0, c1);
}

// Similarity of one pixel's channel group from the product sum of its shifted columns and the statistics of its
// two windows, or the mark for the exact evaluation.
// Window a: shifted sum Sa, shift mua, raw sums SaR / QaR, and in fp32 its shifted sum of squares qa and
// 1e-7 * QaR (ra); likewise b.  In fp64 the raw sums can be combined directly (their cancellation costs a relative
// 1e-16 * QR / varsum, kept below 1e-8 by the second test); only the product sum needs the shifted columns:
// sum(a b) = P'' + mua SbR + mub Sa.
__device__ __forceinline__ float selfsim_pixel(double Sa, double mua, double SaR, double QaR, float qa, float ra,
                                               double mub, double SbR, double QbR, float qb, float rb, double pSum,
                                               double inv2n, float unsure)
{
    const double S = SaR + SbR;
    const double s2 = S * S * inv2n;
    const double varsum = (QaR + QbR) - s2;
    const double cross2 = 2.0 * (fma(mua, SbR, fma(mub, Sa, pSum))) - s2;  // twice the centred product sum
    const float v = (float)varsum;
    // 2 * 67 * 2^-24 * sqrt(Qa Qb) <= 8e-6 * varsum, squared; both tests fail for NaN as well
    const bool sure = 0.997f * qa * qb <= v * v && ra + rb <= v;
    return sure ? (float)cross2 / v : unsure;
}

typedef __attribute__((address_space(3))) void ss_lds_void;

__global__ __launch_bounds__(256, 2) void selfsim_tile_kernel(SelfSimArgs g)
{
    // Workgroups go round robin over the 8 XCDs, each with its own L2: the 64 workgroups that follow each other
    // on one XCD form a block of 8 x 8 tiles, which share 8 + 8 column sets.
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int super = (slot >> 6) * 8 + xcd, within = slot & 63;
    const int tileX = (super % g.superEdge) * 8 + (within & 7), tileY = (super / g.superEdge) * 8 + (within >> 3);
    if (tileY < tileX || tileY >= g.tilesEdge) return;  // pixels with y >= x only (:127-163); the rest is mirrored
    // two stages of one run for both sides (rows of 128 columns: side 1, then side 2); afterwards the
    // similarities of half a tile (64 x 128, rows padded)
    constexpr int kStage = 2 * kSsRun * kSsTile;  // floats
    __shared__ __align__(16) float lds[2 * kStage];
    static_assert(2 * kStage >= 64 * (kSsTile + 1), "the similarities of half a tile must fit");
    const int tid = threadIdx.x;
    const int lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int X0 = tileX * kSsTile, Y0 = tileY * kSsTile;
    const int H = g.H, nCh = g.c1 - g.c0;
    const int n = H * nCh;
    const int group = g.c0 == 0 ? 0 : 1;
    const long long eBeg = (long long)g.c0 * H;

    // Staging with buffer loads whose destination is the LDS: a wave instruction moves two rows (2 x 512 bytes);
    // the descriptor starts at the run's first row and ends with its last, so the rows of a short last run that
    // do not exist read as zeros.
    auto stage = [&](int e0, int buf) {
        const int len = min(kSsRun, n - e0);
        const unsigned bytes = (unsigned)(((long long)(len - 1) * g.stride + kSsTile) * 4);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.cols1 + (eBeg + e0) * g.stride + X0), 0, bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(g.cols2 + (eBeg + e0) * g.stride + Y0), 0, bytes, 0x00020000);
        float *dst = lds + buf * kStage;
        for (int piece = q; piece < kSsRun; piece += 4) {  // pieces 0..15: side 1, 16..31: side 2
            const int row = 2 * (piece & 15) + (lane >> 5);
            const unsigned off = (unsigned)(((long long)row * g.stride + (lane & 31) * 4) * 4);
            ss_lds_void *to = (ss_lds_void *)(dst + piece * 256);
            if (piece < 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, to, 16, off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, to, 16, off, 0, 0, 0);
        }
    };

    // Wave q owns the 64 x 64 pixels at (64 wx, 64 wy) of the tile as 2 x 2 matrix-core tiles of 32 x 32
    // (v_mfma_f32_32x32x2_f32: lane l supplies row l % 32 of each operand for run element l / 32 and receives,
    // in register v, the product at row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32).
    typedef float v16f __attribute__((ext_vector_type(16)));
    const int wx = q & 1, wy = q >> 1, l32 = lane & 31, lhi = lane >> 5;
    v16f acc[2][2];
    double sum[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                acc[i][j][v] = 0.0f;
                sum[i][j][v] = 0.0;
            }

    stage(0, 0);
    int buf = 0;
    for (int e0 = 0; e0 < n; e0 += kSsRun, buf ^= 1) {
        // every direct-to-LDS load of this wave has to have landed before the barrier lets other waves read the
        // stage (the compiler only waits for what this wave's own reads may touch): vmcnt(0)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();  // everyone is done with the other stage
        if (e0 + kSsRun < n) stage(e0 + kSsRun, buf ^ 1);
        const float *Aw = lds + buf * kStage + lhi * kSsTile + 64 * wx + l32;
        const float *Bw = lds + buf * kStage + kSsRun * kSsTile + lhi * kSsTile + 64 * wy + l32;
        {
            // Always the whole stage: rows past the end of a short last run were staged as zeros.  Straight-line;
            // the operands of the next step are fetched while the matrix cores work on this one (the scheduling
            // barriers keep that order).
            float a[2][2], b[2][2];
            a[0][0] = Aw[0], a[0][1] = Aw[32], b[0][0] = Bw[0], b[0][1] = Bw[32];
#pragma unroll
            for (int r = 0; r < kSsRun; r += 2) {
                const int cur = (r >> 1) & 1, nxt = cur ^ 1;
                if (r + 2 < kSsRun) {
                    a[nxt][0] = Aw[(r + 2) * kSsTile], a[nxt][1] = Aw[(r + 2) * kSsTile + 32];
                    b[nxt][0] = Bw[(r + 2) * kSsTile], b[nxt][1] = Bw[(r + 2) * kSsTile + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (buf == 0 && e0 + kSsRun < n) continue;  // fold after every second run and after the last
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    sum[i][j][v] += (double)acc[i][j][v];
                    acc[i][j][v] = 0.0f;
                }
    }

    // Similarities, half a tile at a time through LDS (row rho of the staging area = pixel row
    // 64 (rho / 32) + 32 half + rho % 32 of the tile) so that the rows of the matrix and of its mirror image are
    // written by neighbouring lanes.
    const double inv2n = 1.0 / (2.0 * (double)n);
    constexpr int kRow = kSsTile + 1;
    const float kUnsure = __int_as_float(0x7fc0beef);  // marks a pixel for the exact evaluation
    // statistics of the tile's 128 + 128 windows (indices past the end clamped: those pixels are never written)
    double *stS = reinterpret_cast<double *>(lds + 64 * kRow), *stMu = stS + 2 * kSsTile, *stSR = stMu + 2 * kSsTile,
           *stQR = stSR + 2 * kSsTile;
    float *stQ32 = reinterpret_cast<float *>(stQR + 2 * kSsTile), *stR32 = stQ32 + 2 * kSsTile;
    static_assert((64 * kRow) % 2 == 0 && 64 * kRow + 20 * kSsTile <= 2 * kStage, "statistics must fit behind the rows");
    __syncthreads();
    {
        const int wdw = min((tid < kSsTile ? X0 + tid : Y0 + tid - kSsTile), g.imgExt - 1);
        const SelfSimStat st = (tid < kSsTile ? g.st1 : g.st2)[2 * (long long)wdw + group];
        stS[tid] = st.S;
        stMu[tid] = st.mu;
        stSR[tid] = st.SR;
        stQR[tid] = st.QR;
        stQ32[tid] = (float)st.Q;
        stR32[tid] = (float)(1.0e-7 * st.QR);
    }
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int pyl = 64 * wy + 32 * j + l32, yb = kSsTile + pyl;
            const double mub = stMu[yb], SbR = stSR[yb], QbR = stQR[yb];
            const float qb = stQ32[yb], rb = stR32[yb];
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int rowInTile = 8 * (v >> 2) + 4 * lhi + (v & 3);
                const int pxl = 64 * wx + 32 * half + rowInTile;
                const double pSum = half == 0 ? sum[0][j][v] : sum[1][j][v];
                lds[(32 * wx + rowInTile) * kRow + pyl] =
                    selfsim_pixel(stS[pxl], stMu[pxl], stSR[pxl], stQR[pxl], stQ32[pxl], stR32[pxl], mub, SbR, QbR, qb,
                                  rb, pSum, inv2n, kUnsure);
            }
        }
        __syncthreads();
        const int XH = X0 + 32 * half;
        // rows of the matrix: neighbouring lanes = neighbouring y
        for (int idx = tid; idx < 64 * kSsTile; idx += 256) {
            const int pxl = idx >> 7, pyl = idx & (kSsTile - 1);
            const int px = XH + 64 * (pxl >> 5) + (pxl & 31), py = Y0 + pyl;
            if (px >= g.imgExt || py >= g.imgExt || py < px) continue;
            float part = lds[pxl * kRow + pyl];
            if (__float_as_int(part) == __float_as_int(kUnsure))
                part = correlate_half_exact(g.x1, g.frames1, (long long)g.afStart + (long long)px * g.decim, g.x2,
                                            g.frames2, (long long)g.afStart + (long long)py * g.decim, H, g.c0, g.c1);
            // sim = temporal * w + spectral * (1 - w), impl/SelfSimilarityImpl.scala:146
            float sim = part * g.wThis;
            const long long up = (long long)px * g.imgExt + py;
            if (!g.first) sim = g.simOut[up] + sim;
            if (!g.last || g.keepUpper) g.simOut[up] = sim;
            if (g.last) {
                lds[pxl * kRow + pyl] = sim;
                // (the raster's row is imgExt - 1 - y: of the two pixels that show this value, the one whose
                // neighbours along y lie in the same row)
                if (g.imgOut)
                    g.imgOut[(long long)(g.imgExt - 1 - px) * g.imgExt + py] = gray_color(sim, g.warp, g.scale, g.inv);
            }
        }
        if (!g.last) continue;
        __syncthreads();
        // rows of the mirror image: neighbouring lanes = neighbouring x
        for (int idx = tid; idx < 64 * kSsTile; idx += 256) {
            const int pxl = idx & 63, pyl = idx >> 6;
            const int px = XH + 64 * (pxl >> 5) + (pxl & 31), py = Y0 + pyl;
            if (px >= g.imgExt || py >= g.imgExt || py <= px) continue;
            const float sim = lds[pxl * kRow + pyl];
            if (g.wantSim) g.simOut[(long long)py * g.imgExt + px] = sim;
            if (g.imgOut)
                g.imgOut[(long long)(g.imgExt - 1 - py) * g.imgExt + px] = gray_color(sim, g.warp, g.scale, g.inv);
        }
    }
}

// ---------------------------------------------------------------------------
// The same matrix on the integer matrix cores (v_mfma_i32_32x32x32_i8: 32 products per instruction and row, exact
// 32-bit sums).  A stored value v (fp32, shifted by its window's mean level) is multiplied by a power of two 2^e of
// its window -- the one that puts the window's largest magnitude just below 2^22 -- rounded to an integer q, and q
// is written with three signed digits of 8 bits, q = h 2^16 + m 2^8 + l (each within [-128, 127]).  Then
//     sum qa qb  =  2^32 sum ha hb  +  2^24 sum (ha mb + ma hb)  +  2^16 sum (ha lb + ma mb + la hb)  +  rest,
//     |rest| <= 2^8 (|ma| |lb| + |la| |mb|) + |la| |lb|                  (Euclidean norms over the window)
// and the three sums that are kept are exact: no rounding happens before the fp64 epilogue, so nothing is folded
// along the way and the error of a pixel is what the digits leave out -- about 2^-22 of the product of the windows'
// norms, a twentieth of what fp32 or half-precision accumulation certifies -- plus what the integers leave of the
// values themselves, rho = |v - q 2^-e|.  None of it is estimated: selfsim_gather_limbs_kernel measures |m|, |l| and
// rho per window (SelfSimLimbs).  A pixel for which all that could move the similarity by more than 8e-6 is
// evaluated by correlate_half_dev instead, as in the fp32 kernel (windows with values that are not finite, windows
// without spread).
// Both channel groups in one launch: the temporal group first, its similarities wait in registers while the other
// channels' products are formed, one epilogue, one trip of the result to memory.
// Workgroup = 128 x 128 pixels, 8 waves of 64 (x) x 32 (y) pixels each = 2 matrix-core tiles; a stage holds 32
// products of both sides and the three digits: [side][digit][group of 16 products][128 columns][16 bytes] = 24 KB,
// two stages, filled by buffer loads that write the LDS directly.
// ---------------------------------------------------------------------------
constexpr int kSfThreads = 512;
constexpr int kSfStageBytes = 2 * 3 * 2 * kSsTile * 16;  // 24 KB
constexpr int kSfFrontBytes = 64 * 1024;                 // the two stages; later similarities, statistics, uncertain pixels
constexpr int kSfMaxProducts = 40000;                    // per window and group: 3 * 2^14 * this stays below 2^31

// Per window and channel group: how its shifted values v went into integers, v 2^e ~ q = h 2^16 + m 2^8 + l.
struct SelfSimLimbs {
    float rho;    // Euclidean norm of v - q 2^-e over the window, rounded up
    float lamM;   // 2^8 times the Euclidean norm of m 2^-e, rounded up
    float lamL;   // Euclidean norm of l 2^-e, rounded up
    float sq;     // Euclidean norm of v, rounded up (NaN: the window holds values that are not finite)
    int exp2;     // e
    int pad[3];
};

// Window w -> its column of the three digits' images (plane d of `limbs` holds digit d: 0 = h, 1 = m, 2 = l; a
// 16-byte unit = 16 consecutive values of the window), its statistics (as selfsim_gather_kernel) and its SelfSimLimbs.
__global__ __launch_bounds__(256) void selfsim_gather_limbs_kernel(const float *__restrict__ x, long long frames,
                                                                   int numCh, int afStart, int H, int decim, int imgExt,
                                                                   long long stride, int kgT, int kgS,
                                                                   uint4 *__restrict__ limbs,
                                                                   SelfSimStat *__restrict__ stats,
                                                                   SelfSimLimbs *__restrict__ split)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= stride) return;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    const long long plane = ((long long)kgT + kgS) * stride;
    if (w >= imgExt) {
        for (long long kg = 0; kg < (long long)kgT + kgS; kg++)
            for (int d = 0; d < 3; d++) limbs[d * plane + kg * stride + w] = zero;
        return;
    }
    const long long f0 = (long long)afStart + w * decim;
    for (int group = 0; group < 2; group++) {
        const int c0 = group == 0 ? 0 : 1, c1 = group == 0 ? 1 : numCh;
        const long long kg0 = group == 0 ? 0 : kgT;
        const int kgN = group == 0 ? kgT : kgS;
        double sum = 0.0;
        for (int c = c0; c < c1; c++)
            for (int k = 0; k < H; k++) sum += (double)x[(long long)c * frames + f0 + k];
        const double cnt = (double)(c1 - c0) * H;
        float mu = (float)(sum / cnt);
        if (!(fabsf(mu) < 3.0e38f)) mu = 0.0f;
        float mx = 0.0f;
        bool bad = false;
        for (int c = c0; c < c1; c++)
            for (int k = 0; k < H; k++) {
                const float a = fabsf(x[(long long)c * frames + f0 + k] - mu);
                bad |= !(a <= 3.0e38f);
                mx = fmaxf(mx, a);
            }
        int e = 0;
        if (!bad && mx > 0.0f) {
            int ex;
            (void)frexpf(mx, &ex);  // mx = f 2^ex, 0.5 <= f < 1
            e = 22 - ex;            // mx 2^e in [2^21, 2^22)
        }
        double S = 0.0, Q = 0.0, R = 0.0, Mm = 0.0, Lm = 0.0;
        unsigned int pd[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        int idx = 0;
        auto flush = [&](long long kg) {
            for (int d = 0; d < 3; d++) {
                limbs[d * plane + (kg0 + kg) * stride + w] = make_uint4(pd[d][0], pd[d][1], pd[d][2], pd[d][3]);
                pd[d][0] = pd[d][1] = pd[d][2] = pd[d][3] = 0u;
            }
        };
        for (int c = c0; c < c1; c++)
            for (int k = 0; k < H; k++, idx++) {
                const float v = x[(long long)c * frames + f0 + k] - mu;
                S += (double)v;
                Q = fma((double)v, (double)v, Q);
                // (a window with values that are not finite is marked below; its digits are not looked at)
                const int q = bad ? 0 : (int)rintf(ldexpf(v, e));
                const int l = ((q + 128) & 255) - 128, q1 = (q - l) >> 8;
                const int m = ((q1 + 128) & 255) - 128, h = (q1 - m) >> 8;
                const double r = (double)v - ldexp((double)q, -e);
                R = fma(r, r, R);
                Mm += (double)(m * m);
                Lm += (double)(l * l);
                const int j = idx & 15, sh = 8 * (j & 3);
                pd[0][j >> 2] |= ((unsigned int)h & 255u) << sh;
                pd[1][j >> 2] |= ((unsigned int)m & 255u) << sh;
                pd[2][j >> 2] |= ((unsigned int)l & 255u) << sh;
                if (j == 15) flush(idx >> 4);
            }
        int kg = idx >> 4;
        if (idx & 15) flush(kg++);  // the last, partly filled unit
        for (; kg < kgN; kg++) flush(kg);
        SelfSimStat st;
        st.S = S;
        st.Q = Q;
        st.mu = (double)mu;
        st.SR = S + cnt * (double)mu;
        st.QR = Q + 2.0 * (double)mu * S + cnt * (double)mu * (double)mu;
        st.pad = 0.0;
        stats[2 * w + group] = st;
        SelfSimLimbs sp;
        sp.rho = (float)(sqrt(R) * 1.0001);
        sp.lamM = (float)(ldexp(sqrt(Mm), 8 - e) * 1.0001);
        sp.lamL = (float)(ldexp(sqrt(Lm), -e) * 1.0001);
        sp.sq = bad ? __builtin_nanf("") : (float)(sqrt(Q) * 1.0001);
        sp.exp2 = e;
        sp.pad[0] = sp.pad[1] = sp.pad[2] = 0;
        split[2 * w + group] = sp;
    }
}

// Similarity of one pixel's channel group from the product sum P of the integers (already divided by the two windows'
// powers of two) and the windows' statistics, or the mark for the exact evaluation: selfsim_pixel with the error
// terms of the header comment.  With SR, QR the windows' raw sums, mu their shifts, S their shifted sums:
//     sum (v - m)^2        = QaR + QbR - (SaR + SbR)^2 / 2n                  = Aa + Ab - Ua SbR
//     2 sum (a - m)(b - m) = 2 (mua SbR + mub Sa + P) - (SaR + SbR)^2 / 2n   = 2 P + Ca SbR + Db Sa - (Ba + Bb)
// with what depends on one window alone formed once per window and tile (selfsim_window_terms): A = QR - SR^2 / 2n,
// B = SR^2 / 2n, U = SR / n, C = 2 mu - U, D = 2 mu.  (ra, rb: 1e-7 of the windows' raw sums of squares, the guard of
// the fp64 cancellation as in selfsim_pixel.)  The quotient is a product with the reciprocal (v_rcp_f32, one unit of
// the last place): kSsMixErr has room for it.  `bound`: how far the returned similarity can be from the exact quotient
// because of the product sum's error -- at most 8e-6, for ordinary windows about 2e-6.
__device__ __forceinline__ float selfsim_pixel8(double Sa, double Aa, double Ba, double Ca, double Ua, float sqa,
                                                float rhoa, float lamMa, float lamLa, float ra, double SbR, double Ab,
                                                double Bb, double Db, float sqb, float rhob, float lamMb, float lamLb,
                                                float rb, double P, float unsure, float &bound)
{
    const double varsum = fma(-Ua, SbR, Aa + Ab);
    const double cross2 = fma(P, 2.0, fma(Ca, SbR, fma(Db, Sa, -(Ba + Bb))));  // twice the centred product sum
    const float v = (float)varsum;
    // digits left out, what the integers leave of the values, and the fp64 roundings of the sum's three terms
    const float err = fmaf(lamMa, lamLb, fmaf(lamLa, lamMb, fmaf(lamLa, lamLb, fmaf(rhoa, sqb, fmaf(rhob, sqa,
                      fmaf(3.0f * rhoa, rhob, 1.0e-14f * sqa * sqb))))));
    // 2 err <= 8e-6 varsum (both tests fail for NaN as well)
    const bool sure = err * 1.0001f <= 4.0e-6f * v && ra + rb <= v;
    const float rv = __builtin_amdgcn_rcpf(v);
    bound = 2.0003f * err * rv;
    return sure ? (float)cross2 * rv : unsure;
}

// What the fp32 steps add to the distance between a pixel's similarity and the reference's: the reference rounds each
// group's fp64 quotient to fp32 once, multiplies by the weight and adds (impl/SelfSimilarityImpl.scala:143-146); here a
// group's part is fl(cross) * rcp(fl(var)), four roundings and the reciprocal's unit of the last place against one,
// then the same two products and one sum with arguments that differ by the bounds: with u = 2^-24 and similarities
// of magnitude 1 at most that is 7 u for the parts, 2 u for the products and 2 u for the sum, 11 u = 6.6e-7.
constexpr float kSsMixErr = 7.0e-7f;
constexpr int kSfErrBytes = 32 * kSfThreads * 4;  // the first group's error bounds: a float per pixel
constexpr int kSfHits = 2048;                     // uncertain pixels a tile collects before it appends them one by one

__global__ __launch_bounds__(kSfThreads, 2) void selfsim_tile8_kernel(SelfSimArgs g)
{
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int super = (slot >> 6) * 8 + xcd, within = slot & 63;
    const int tileX = (super % g.superEdge) * 8 + (within & 7), tileY = (super / g.superEdge) * 8 + (within >> 3);
    if (tileY < tileX || tileY >= g.tilesEdge) return;  // pixels with y >= x only (:127-163); the rest is mirrored
    extern __shared__ __align__(16) unsigned char ldsRaw[];  // kSfFrontBytes, then kSfErrBytes
    __shared__ int hitCount;
    __shared__ unsigned long long hitBase;
    float *lds = reinterpret_cast<float *>(ldsRaw);
    float *errLds = reinterpret_cast<float *>(ldsRaw + kSfFrontBytes);  // [pixel of the thread][thread]
    const int tid = threadIdx.x;
    if (tid == 0) hitCount = 0;  // (the loop below has barriers before anybody counts)
    const int lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = q & 1, wy = q >> 1, l32 = lane & 31, lhi = lane >> 5;
    const int X0 = tileX * kSsTile, Y0 = tileY * kSsTile;
    const int H = g.H;
    typedef int v16i __attribute__((ext_vector_type(16)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr int kRow = kSsTile + 1;
    const float kUnsure = __int_as_float(0x7fc0beef);  // marks a pixel for the exact evaluation
    // behind the similarities of half a tile (64 rows): the statistics of the tile's 128 + 128 windows
    // (S: shifted sum, SR: raw sum, A, B, C, D, U: see selfsim_pixel8, F: the window's power of two)
    double *stS = reinterpret_cast<double *>(lds + 64 * kRow), *stSR = stS + 2 * kSsTile, *stA = stSR + 2 * kSsTile,
           *stB = stA + 2 * kSsTile, *stC = stB + 2 * kSsTile, *stD = stC + 2 * kSsTile, *stU = stD + 2 * kSsTile,
           *stF = stU + 2 * kSsTile;
    float *stSq = reinterpret_cast<float *>(stF + 2 * kSsTile), *stRho = stSq + 2 * kSsTile, *stLamM = stRho + 2 * kSsTile,
          *stLamL = stLamM + 2 * kSsTile, *stR7 = stLamL + 2 * kSsTile;
    // behind the statistics: the tile's uncertain pixels (x | y << 8 within the tile)
    int *hits = reinterpret_cast<int *>(stR7 + 2 * kSsTile);
    static_assert((64 * kRow) % 2 == 0 && (64 * kRow + (16 + 5) * 2 * kSsTile + kSfHits) * 4 <= kSfFrontBytes &&
                      2 * kSfStageBytes <= kSfFrontBytes,
                  "stages resp. similarities, statistics and the uncertain pixels must fit the front memory");
    // the temporal group's part, later the pixel's similarity -- or, where only the raster is asked for, its colour
    float simv[2][16];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int v = 0; v < 16; v++) simv[i][v] = 0.0f;

    for (int group = g.doT ? 0 : 1; group < (g.doS ? 2 : 1); group++) {
        const int nkg = group == 0 ? g.kgT : g.kgS;  // units of 16 products: a multiple of 2
        const long long kg0 = group == 0 ? 0 : g.kgT;
        const long long plane = ((long long)g.kgT + g.kgS) * g.stride;
        const unsigned bytes = (unsigned)(((long long)(nkg - 1) * g.stride + kSsTile) * 16);
        // [side][digit]
        const __amdgpu_buffer_rsrc_t rs[6] = {
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(g.limbs1 + kg0 * g.stride + X0), 0, bytes, 0x00020000),
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(g.limbs1 + plane + kg0 * g.stride + X0), 0, bytes, 0x00020000),
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(g.limbs1 + 2 * plane + kg0 * g.stride + X0), 0, bytes, 0x00020000),
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(g.limbs2 + kg0 * g.stride + Y0), 0, bytes, 0x00020000),
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(g.limbs2 + plane + kg0 * g.stride + Y0), 0, bytes, 0x00020000),
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(g.limbs2 + 2 * plane + kg0 * g.stride + Y0), 0, bytes, 0x00020000)};
        // a stage = 12 rows of 2 KB ([side][digit][unit]); a wave instruction moves half a row, a wave three of them
        auto stage = [&](int s0, int buf) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int piece = q * 3 + t, row = piece >> 1, hc = piece & 1;
                const unsigned off = (unsigned)((((long long)(s0 + (row & 1))) * g.stride + 64 * hc + lane) * 16);
                ss_lds_void *to = (ss_lds_void *)(ldsRaw + buf * kSfStageBytes + row * 2048 + hc * 1024);
                switch (row >> 1) {
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[0], to, 16, off, 0, 0, 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[1], to, 16, off, 0, 0, 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[2], to, 16, off, 0, 0, 0); break;
                case 3: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[3], to, 16, off, 0, 0, 0); break;
                case 4: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[4], to, 16, off, 0, 0, 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[5], to, 16, off, 0, 0, 0); break;
                }
            }
        };
        v16i acc[3][2];  // [order: h h; h m + m h; h l + m m + l h][matrix-core tile]
#pragma unroll
        for (int o = 0; o < 3; o++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[o][i][v] = 0;
        __syncthreads();  // (whatever read the LDS before -- the first group's statistics -- is done)
        stage(0, 0);
        int buf = 0;
        for (int s0 = 0; s0 < nkg; s0 += 2, buf ^= 1) {
            // every direct-to-LDS load of this wave has to have landed before the barrier lets other waves read the
            // stage (the compiler only waits for what this wave's own reads may touch): vmcnt(0)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();  // everyone is done with the other stage
            if (s0 + 2 < nkg) stage(s0 + 2, buf ^ 1);
            const unsigned char *st = ldsRaw + buf * kSfStageBytes;
            // (the lanes' upper half takes the stage's second unit: the instruction's 32 products)
            const int colB = (32 * wy + l32) * 16;
            const v4i bH = *reinterpret_cast<const v4i *>(st + (6 + lhi) * 2048 + colB);
            const v4i bM = *reinterpret_cast<const v4i *>(st + (8 + lhi) * 2048 + colB);
            const v4i bL = *reinterpret_cast<const v4i *>(st + (10 + lhi) * 2048 + colB);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int colA = (64 * wx + 32 * i + l32) * 16;
                const v4i aH = *reinterpret_cast<const v4i *>(st + (0 + lhi) * 2048 + colA);
                const v4i aM = *reinterpret_cast<const v4i *>(st + (2 + lhi) * 2048 + colA);
                const v4i aL = *reinterpret_cast<const v4i *>(st + (4 + lhi) * 2048 + colA);
                acc[0][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aH, bH, acc[0][i], 0, 0, 0);
                acc[1][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aH, bM, acc[1][i], 0, 0, 0);
                acc[1][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aM, bH, acc[1][i], 0, 0, 0);
                acc[2][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aH, bL, acc[2][i], 0, 0, 0);
                acc[2][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aM, bM, acc[2][i], 0, 0, 0);
                acc[2][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aL, bH, acc[2][i], 0, 0, 0);
            }
        }

        // the group's similarities: statistics of the tile's windows (indices past the end clamped: those pixels
        // are never written) into LDS, then every thread its 32 pixels
        __syncthreads();
        if (tid < 2 * kSsTile) {
            const int wdw = min((tid < kSsTile ? X0 + tid : Y0 + tid - kSsTile), g.imgExt - 1);
            const SelfSimStat st = (tid < kSsTile ? g.st1 : g.st2)[2 * (long long)wdw + group];
            const SelfSimLimbs sp = (tid < kSsTile ? g.sp1 : g.sp2)[2 * (long long)wdw + group];
            const double nn = (double)(H * (group == 0 ? 1 : g.numCh - 1));
            const double b = st.SR * st.SR / (2.0 * nn), u = st.SR / nn;
            stS[tid] = st.S;
            stSR[tid] = st.SR;
            stA[tid] = st.QR - b;
            stB[tid] = b;
            stC[tid] = 2.0 * st.mu - u;
            stD[tid] = 2.0 * st.mu;
            stU[tid] = u;
            stF[tid] = ldexp(1.0, 8 - sp.exp2);  // (2^16 of the sum's common factor, half on either side)
            stSq[tid] = sp.sq;
            stRho[tid] = sp.rho;
            stLamM[tid] = sp.lamM;
            stLamL[tid] = sp.lamL;
            stR7[tid] = (float)(1.0e-7 * st.QR);
        }
        __syncthreads();
        const int c0 = group == 0 ? 0 : 1, c1 = group == 0 ? 1 : g.numCh;
        const float wThis = group == 0 ? g.wT : 1.0f - g.wT;
        const int pyl = 32 * wy + l32, yb = kSsTile + pyl, py = Y0 + pyl;
        const double SbR = stSR[yb], Ab = stA[yb], Bb = stB[yb], Db = stD[yb], fb = stF[yb];
        const float sqb = stSq[yb], rhob = stRho[yb], lamMb = stLamM[yb], lamLb = stLamL[yb], rb = stR7[yb];
        const bool both = g.doT && g.doS, lastGroup = group == (g.doS ? 1 : 0);
        // (the two matrix-core tiles of a thread one after the other in a loop, not side by side in the code: the
        // accumulators and similarities of tile i are picked by selects.  With both tiles' 32 pixels unrolled the
        // kernel was 57 KB of code and spilled 124 registers per thread; so: 37 KB, none, 14.5 -> 13.1 ms for the
        // matrix of a 30-minute file.  Four pixels per pass of the loop: 13.4 ms.)
#pragma unroll 1
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int pxl = 64 * wx + 32 * i + 8 * (v >> 2) + 4 * lhi + (v & 3), px = X0 + pxl;
                // sum qa qb / 2^16 (every term an integer below 2^53: exact)
                const int e0 = i ? acc[0][1][v] : acc[0][0][v], e1 = i ? acc[1][1][v] : acc[1][0][v],
                          e2 = i ? acc[2][1][v] : acc[2][0][v];
                const double t = fma((double)e0, 65536.0, fma((double)e1, 256.0, (double)e2));
                float bound;
                float part = selfsim_pixel8(stS[pxl], stA[pxl], stB[pxl], stC[pxl], stU[pxl], stSq[pxl], stRho[pxl],
                                            stLamM[pxl], stLamL[pxl], stR7[pxl], SbR, Ab, Bb, Db, sqb, rhob, lamMb, lamLb,
                                            rb, t * stF[pxl] * fb, kUnsure, bound);
                const bool inside = px < g.imgExt && py < g.imgExt && py >= px;
                if (__float_as_int(part) == __float_as_int(kUnsure)) {
                    part = 0.0f;
                    bound = 0.0f;  // (the reference's own value)
                    if (inside)
                        part = correlate_half_exact(g.x1, g.frames1, (long long)g.afStart + (long long)px * g.decim, g.x2,
                                                    g.frames2, (long long)g.afStart + (long long)py * g.decim, H, c0, c1);
                }
                // sim = temporal * w + spectral * (1 - w), impl/SelfSimilarityImpl.scala:146
                const float term = part * wThis;
                const float sim = (group == 0 || !g.doT) ? term : (i ? simv[1][v] : simv[0][v]) + term;
                float keep = sim;  // (what simv[i][v] becomes)
                // The gray level is integer work: it is certain when both ends of the similarity's error interval map
                // to the same level (the colour map is monotonic); the other pixels are listed for
                // selfsim_fix_kernel, which evaluates them as the reference does.  (Only where a raster is asked for.)
                if (g.fixList && !lastGroup) errLds[(16 * i + v) * kSfThreads + tid] = bound * wThis;
                if (g.fixList && lastGroup) {
                    float e = fmaf(bound, wThis, kSsMixErr + g.slack);
                    if (both) e += errLds[(16 * i + v) * kSfThreads + tid];
                    const int cLo = gray_color(sim - e, g.warp, g.scale, g.inv), cHi = gray_color(sim + e, g.warp, g.scale, g.inv);
                    if (!g.wantSim) keep = __int_as_float(cLo);
                    if (cLo != cHi && inside) {
                        const int mine = atomicAdd(&hitCount, 1);
                        if (mine < kSfHits) {
                            hits[mine] = pxl | (pyl << 8);
                        } else {
                            const unsigned long long slot = atomicAdd(g.fixCount, 1ull);
                            if (slot < (unsigned long long)g.fixCap) g.fixList[slot] = make_int2(px, py);
                        }
                    }
                }
                simv[0][v] = i ? simv[0][v] : keep;
                simv[1][v] = i ? keep : simv[1][v];
            }
    }
    if (g.fixList) {  // the tile's uncertain pixels: one place in the list for all of them
        __syncthreads();
        const int n = min(hitCount, kSfHits);
        if (n > 0) {
            if (tid == 0) hitBase = atomicAdd(g.fixCount, (unsigned long long)n);
            __syncthreads();
            for (int i = tid; i < n; i += kSfThreads)
                if (hitBase + (unsigned long long)i < (unsigned long long)g.fixCap)
                    g.fixList[hitBase + (unsigned long long)i] = make_int2(X0 + (hits[i] & 255), Y0 + (hits[i] >> 8));
        }
    }

    // Out, half a tile at a time through LDS (row rho of the staging area = pixel row 64 (rho / 32) + 32 half +
    // rho % 32 of the tile) so that the rows of the matrix and of its mirror image are written by neighbouring lanes.
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int rowInTile = 8 * (v >> 2) + 4 * lhi + (v & 3);
            lds[(32 * wx + rowInTile) * kRow + 32 * wy + l32] = half == 0 ? simv[0][v] : simv[1][v];
        }
        __syncthreads();
        const int XH = X0 + 32 * half;
        // (kColour: only the raster is asked for and the staged values are colours already)
        auto store = [&](auto colourOnly) {
            constexpr bool kColour = decltype(colourOnly)::value;
            // rows of the matrix: neighbouring lanes = neighbouring y
            {
                const int pyl = tid & (kSsTile - 1), py = Y0 + pyl;
#pragma unroll
                for (int it = 0; it < 64 * kSsTile / kSfThreads; it++) {
                    const int pxl = (tid >> 7) + it * (kSfThreads / kSsTile);
                    const int px = XH + 64 * (pxl >> 5) + (pxl & 31);
                    if (px < g.imgExt && py < g.imgExt && py >= px) {
                        const float sim = lds[pxl * kRow + pyl];
                        if (!kColour) g.simOut[(long long)px * g.imgExt + py] = sim;
                        // (the raster's row is imgExt - 1 - y: of the two pixels that show this value, the one whose
                        // neighbours along y lie in the same row)
                        if (g.imgOut)
                            g.imgOut[(long long)(g.imgExt - 1 - px) * g.imgExt + py] =
                                kColour ? __float_as_int(sim) : gray_color(sim, g.warp, g.scale, g.inv);
                    }
                }
            }
            // rows of the mirror image: neighbouring lanes = neighbouring x
            {
                const int pxl = tid & 63, px = XH + 64 * (pxl >> 5) + (pxl & 31);
#pragma unroll
                for (int it = 0; it < 64 * kSsTile / kSfThreads; it++) {
                    const int pyl = (tid >> 6) + it * (kSfThreads / 64), py = Y0 + pyl;
                    if (px < g.imgExt && py < g.imgExt && py > px) {
                        const float sim = lds[pxl * kRow + pyl];
                        if (!kColour) g.simOut[(long long)py * g.imgExt + px] = sim;
                        if (g.imgOut)
                            g.imgOut[(long long)(g.imgExt - 1 - py) * g.imgExt + px] =
                                kColour ? __float_as_int(sim) : gray_color(sim, g.warp, g.scale, g.inv);
                    }
                }
            }
        };
        if (g.wantSim) store(std::false_type{});
        else store(std::true_type{});
    }
}

// Lists the pixels (x, y >= x) whose gray level is not certain: the colour map is monotonic, so the level is
// certain when both ends of the similarity's error interval map to the same one.  A workgroup per tile of
// 128 x 128 pixels, so that neighbours in the list are neighbours in the matrix (selfsim_fix_kernel then finds
// the frames of both windows in cache); hits are collected in LDS and appended with one atomic on the global
// counter per tile.
__global__ __launch_bounds__(256) void selfsim_list_kernel(SelfSimArgs g)
{
    constexpr int kLocal = 4096;
    __shared__ int hits[kLocal];
    __shared__ int count;
    __shared__ unsigned long long base;
    const int tileX = blockIdx.x % g.tilesEdge, tileY = blockIdx.x / g.tilesEdge;
    if (tileY < tileX) return;
    const int tid = threadIdx.x, X0 = tileX * kSsTile, Y0 = tileY * kSsTile;
    if (tid == 0) count = 0;
    __syncthreads();
    const int col = tid & (kSsTile - 1), py = Y0 + col;
    for (int r0 = tid >> 7; r0 < kSsTile; r0 += 16) {  // eight rows in flight
        float sim[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int px = X0 + r0 + 2 * u;
            sim[u] = (py < g.imgExt && py >= px) ? g.simOut[(long long)px * g.imgExt + py] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int px = X0 + r0 + 2 * u;
            if (py < g.imgExt && py >= px &&
                gray_color(sim[u] - kSsSimErr, g.warp, g.scale, g.inv) !=
                    gray_color(sim[u] + kSsSimErr, g.warp, g.scale, g.inv)) {
                const int mine = atomicAdd(&count, 1);
                if (mine < kLocal) {
                    hits[mine] = (r0 + 2 * u) | (col << 8);
                } else {
                    const unsigned long long slot = atomicAdd(g.fixCount, 1ull);
                    if (slot < (unsigned long long)g.fixCap) g.fixList[slot] = make_int2(px, py);
                }
            }
        }
    }
    __syncthreads();
    const int n = min(count, kLocal);
    if (n == 0) return;
    if (tid == 0) base = atomicAdd(g.fixCount, (unsigned long long)n);
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        if (base + (unsigned long long)i < (unsigned long long)g.fixCap)
            g.fixList[base + (unsigned long long)i] = make_int2(X0 + (hits[i] & 255), Y0 + (hits[i] >> 8));
}

// Quotient of correlateHalf (impl/MathUtil.scala:80-99) from one-pass sums over values shifted by one of the
// window's own (d = v - m0): S = sum d over both halves, Q = sum d^2, X = sum da db give
//     sum (v - mean)^2 = Q - S^2 / 2n,   sum (a - mean)(b - mean) = X - S^2 / 4n.
// Returns it before its rounding to float and, in `tol`, a bound on how far the sequential evaluation in the
// reference's order can be from it: the sums carry at most g = (3 t + 8) 2^-53 relative to Q each (t = the terms
// one lane added up), which the subtraction amplifies by Q / var (at most 1 + 2n, about 2 for ordinary windows):
// 4 g (Q / var) (1 + |r|); the sequential two-pass sums over 2n terms: 4 n 2^-53 (1 + |r|); doubled.  tol is NaN
// when the bound does not apply (no variance, non-finite values).
__device__ __forceinline__ double shifted_quotient(double S, double Q, double X, int n, int laneTerms, double &tol)
{
    const double n2 = 2.0 * n;
    const double var = Q - S * S / n2, cross = X - S * S / (2.0 * n2);
    const double stdDev = sqrt(var / n2);
    const double r = cross / (stdDev * stdDev * n);
    const double u53 = 1.1102230246251565e-16;
    const double g = (3.0 * laneTerms + 8.0) * u53;
    tol = var > 0.0 ? 2.0 * (4.0 * g * (Q / var) + 4.0 * n * u53) * (1.0 + fabs(r)) : NAN;
    return r;
}

// v of another lane of the same row of 16: kCtrl 0xB1 / 0x4E = the lane 1 / 2 apart (quad permutations 1,0,3,2 and
// 2,3,0,1), 0x141 / 0x140 = the mirror image within 8 / 16 lanes.
template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// float(v) when every fp64 value within tol of v rounds to the same float (then the sequential evaluation, which
// lies in that interval, rounds to it as well); false otherwise.
__device__ __forceinline__ bool certain_float(double v, double tol, float &out)
{
    out = (float)v;
    if (!(fabs(v) < 1.0e300 && tol < 1.0e300)) return false;  // also NaN
    return __float_as_int((float)(v - tol)) == __float_as_int((float)(v + tol));
}

// The listed pixels (or, without a list, every pixel with y >= x) as the reference computes them
// (impl/SelfSimilarityImpl.scala:143-150, MathUtil.correlateHalf): similarity, gray level, both mirror positions.
// One wave per pixel.  The frames of a channel are dealt over the lanes, eight channels (32 loads a lane) in
// flight, both channel groups in the same pass; a quotient so close to where fp32 rounding flips that the changed
// order of operations could matter is left to selfsim_seq_kernel (a few in 10^4), which evaluates it in the
// reference's own order.
__global__ __launch_bounds__(256) void selfsim_fix_kernel(SelfSimArgs g, const int2 *__restrict__ list,
                                                          const unsigned long long *__restrict__ count, long long total)
{
    const int lane = threadIdx.x & 63;
    const long long n = list ? min((long long)*count, g.fixCap) : total;
    const long long step = (long long)gridDim.x * 4;
    const bool doT = g.wT > 0.0f, doS = g.wT < 1.0f;
    const int chEnd = doS ? g.numCh : 1, H = g.H;
    long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int2 next = make_int2(0, 0);
    if (list && i < n) next = list[i];
    for (; i < n; i += step) {
        int px, py;
        if (list) {
            px = next.x;
            py = next.y;
            if (i + step < n) next = list[i + step];  // on its way while this pixel is worked on
        } else {
            px = (int)(i / g.imgExt);
            py = (int)(i - (long long)px * g.imgExt);
            if (py < px) continue;
        }
        // the same for every lane: said so, the addresses below are scalar work
        px = __builtin_amdgcn_readfirstlane(px);
        py = __builtin_amdgcn_readfirstlane(py);
        const long long left = (long long)g.afStart + (long long)px * g.decim;
        const long long right = (long long)g.afStart + (long long)py * g.decim;
        // shifts: the first value of each group's window
        const float m0Tf = g.x1[left], m0Sf = g.x1[g.frames1 + left];
        const double m0T = (double)m0Tf, m0S = (double)m0Sf;
        double ST = 0.0, QT = 0.0, XT = 0.0, SS = 0.0, QS = 0.0, XS = 0.0;
        for (int kb = lane; kb < H; kb += 128) {
            const bool far = kb + 64 < H;
            for (int cb = 0; cb < chEnd; cb += 8) {
                float a[2][8], b[2][8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int c = cb + u;
                    const bool have = c < chEnd;
                    const float fill = c == 0 ? m0Tf : m0Sf;  // a value equal to the shift adds nothing to any sum
                    const float *pa = g.x1 + (long long)c * g.frames1 + left + kb;
                    const float *pb = g.x2 + (long long)c * g.frames2 + right + kb;
                    a[0][u] = have ? pa[0] : fill;
                    b[0][u] = have ? pb[0] : fill;
                    a[1][u] = have && far ? pa[64] : fill;
                    b[1][u] = have && far ? pb[64] : fill;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (cb + u == 0) {
                            const double da = (double)a[j][u] - m0T, db = (double)b[j][u] - m0T;
                            ST += da + db;
                            QT = fma(da, da, fma(db, db, QT));
                            XT = fma(da, db, XT);
                        } else {
                            const double da = (double)a[j][u] - m0S, db = (double)b[j][u] - m0S;
                            SS += da + db;
                            QS = fma(da, da, fma(db, db, QS));
                            XS = fma(da, db, XS);
                        }
                    }
                }
            }
        }
        // butterflies of all six sums side by side: within a row of 16 lanes by lane permutations in the
        // registers (partners 1 and 2 apart, then the mirror image within 8 and within 16), across rows by shuffles
        auto fold = [&](auto partner) {
            const double s0 = partner(ST), s1 = partner(QT), s2 = partner(XT);
            const double s3 = partner(SS), s4 = partner(QS), s5 = partner(XS);
            ST += s0, QT += s1, XT += s2, SS += s3, QS += s4, XS += s5;
        };
        fold([](double v) { return dpp_f64<0xB1>(v); });
        fold([](double v) { return dpp_f64<0x4E>(v); });
        fold([](double v) { return dpp_f64<0x141>(v); });
        fold([](double v) { return dpp_f64<0x140>(v); });
        fold([](double v) { return __shfl_xor(v, 16); });
        fold([](double v) { return __shfl_xor(v, 32); });
        const int laneSteps = 2 * ((H + 127) / 128);
        float temporal = 0.0f, spectral = 0.0f;
        double tol;
        int redo = 0;
        if (doT && !certain_float(shifted_quotient(ST, QT, XT, H, laneSteps, tol), tol, temporal)) redo |= 1;
        if (doS && !certain_float(shifted_quotient(SS, QS, XS, (g.numCh - 1) * H, laneSteps * (g.numCh - 1), tol), tol,
                                  spectral))
            redo |= 2;
        if (g.forceSeq) redo = (doT ? 1 : 0) | (doS ? 2 : 0);
        if (redo) {
            unsigned int slot = 0;
            if (lane == 0) slot = atomicAdd(g.seqCount, 1u);
            slot = __shfl(slot, 0);
            if (slot < g.seqCap) {
                if (lane == 0) g.seqList[slot] = SelfSimSeq{px, py, temporal, spectral, redo, 0};
                continue;
            }
            // the list is full: here and now, however long the walk through global memory takes
            if (redo & 1) temporal = correlate_half_exact(g.x1, g.frames1, left, g.x2, g.frames2, right, H, 0, 1);
            if (redo & 2) spectral = correlate_half_exact(g.x1, g.frames1, left, g.x2, g.frames2, right, H, 1, g.numCh);
        }
        if (lane != 0) continue;
        const float sim = temporal * g.wT + spectral * (1.0f - g.wT);
        const int colr = gray_color(sim, g.warp, g.scale, g.inv);
        g.imgOut[(long long)(g.imgExt - 1 - px) * g.imgExt + py] = colr;
        g.imgOut[(long long)(g.imgExt - 1 - py) * g.imgExt + px] = colr;
        if (g.wantSim) {
            g.simOut[(long long)px * g.imgExt + py] = sim;
            g.simOut[(long long)py * g.imgExt + px] = sim;
        }
    }
}

// The pixels selfsim_fix_kernel left over: correlateHalf in the reference's order, one wave per pixel.  The
// additions of that order depend on each other, so what counts is the latency of a value: the window of the
// group is staged in LDS first (ldsFloats = 2 * channels * H of the larger group, 0 when that does not fit; the
// walk then reads global memory).
__global__ __launch_bounds__(64) void selfsim_seq_kernel(SelfSimArgs g, int ldsFloats)
{
    extern __shared__ float win[];
    const unsigned int n = min(*g.seqCount, g.seqCap);
    const int lane = threadIdx.x, H = g.H;
    for (unsigned int i = blockIdx.x; i < n; i += gridDim.x) {
        const SelfSimSeq e = g.seqList[i];
        const long long left = (long long)g.afStart + (long long)e.px * g.decim;
        const long long right = (long long)g.afStart + (long long)e.py * g.decim;
        float part[2] = {e.temporal, e.spectral};
        for (int group = 0; group < 2; group++) {
            if (!(e.redo & (1 << group))) continue;
            const int c0 = group, c1 = group == 0 ? 1 : g.numCh;
            if (2 * (c1 - c0) * H <= ldsFloats) {
                __syncthreads();
                for (int c = c0; c < c1; c++)
                    for (int k = lane; k < H; k += 64) {
                        win[(c - c0) * 2 * H + k] = g.x1[(long long)c * g.frames1 + left + k];
                        win[(c - c0) * 2 * H + H + k] = g.x2[(long long)c * g.frames2 + right + k];
                    }
                __syncthreads();
                part[group] = correlate_half_dev(win, 2 * H, 0, win, 2 * H, H, H, 0, 0, c1 - c0);
            } else {
                part[group] = correlate_half_exact(g.x1, g.frames1, left, g.x2, g.frames2, right, H, c0, c1);
            }
        }
        if (lane != 0) continue;
        const float sim = part[0] * g.wT + part[1] * (1.0f - g.wT);
        const int colr = gray_color(sim, g.warp, g.scale, g.inv);
        g.imgOut[(long long)(g.imgExt - 1 - e.px) * g.imgExt + e.py] = colr;
        g.imgOut[(long long)(g.imgExt - 1 - e.py) * g.imgExt + e.px] = colr;
        if (g.wantSim) {
            g.simOut[(long long)e.px * g.imgExt + e.py] = sim;
            g.simOut[(long long)e.py * g.imgExt + e.px] = sim;
        }
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

// sgz_selfsim_run; keep_sim != nullptr: the matrix is computed and stays on the device (the caller owns *keep_sim).
static int selfsim_impl(sgz_context *ctx, const float *frames1, int64_t num_frames1, const float *frames2,
                        int64_t num_frames2, int num_channels, const float *norms, int af_start, int half_win,
                        int decim, int img_ext,
                        float temporal_weight, int colors, float color_warp, float color_ceil, int color_inv,
                        float *sim_out, int32_t *img_out, float **keep_sim)
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
    // the reference skips a group whose weight is 0 (:143-145)
    const bool doT = temporal_weight > 0.0f, doS = temporal_weight < 1.0f || !doT;
    DevBuf st1, st2, cols1, cols2, limbs1, limbs2, sp1, sp2;
    const unsigned gt = (unsigned)((img_ext + kSsTile - 1) / kSsTile);
    const long long stride = (long long)gt * kSsTile;
    // units of 16 products per channel group, padded to whole stages of 32 products
    const long long kgT = ((long long)half_win + 31) / 32 * 2, kgS = ((long long)(num_channels - 1) * half_win + 31) / 32 * 2;
    // The integer matrix cores take it unless a plane of the windows' digits would pass the 4 GB a buffer
    // descriptor's 32-bit offsets reach, or a window has more products than 32-bit sums hold (windows of minutes):
    // then the fp32 kernel, one pass per group.
    bool limbs = std::max(kgT, kgS) * stride * 16 < (1ll << 32) && std::max(kgT, kgS) * 16 <= kSfMaxProducts;
    if (const char *e = std::getenv("SGZ_SELFSIM_KERNEL")) {  // test knob: "f32" forces the fp32 kernel
        if (std::strcmp(e, "f32") == 0) limbs = false;
    }
    // (the fp32 kernel keeps its first group's part and, for the list of uncertain pixels, the finished half with
    // y >= x in the matrix; the integer one needs it only when the caller wants the matrix)
    if (sim_out || keep_sim || (!limbs && (img_out || (doT && doS)))) SGZ_HIP(ctx, hipMalloc(&simd.p, pix * sizeof(float)));
    if (img_out) SGZ_HIP(ctx, hipMalloc(&imgd.p, pix * sizeof(int32_t)));
    SGZ_HIP(ctx, hipMalloc(&st1.p, 2 * (size_t)stride * sizeof(SelfSimStat)));
    const unsigned gs = (unsigned)((stride + 255) / 256);
    const size_t colFloats = (size_t)num_channels * half_win * (size_t)stride;
    const size_t planeBytes = (size_t)(kgT + kgS) * (size_t)stride * 16;
    auto gather = [&](const void *planar, long long nframes, DevBuf &st, DevBuf &cols, DevBuf &lb, DevBuf &sp) -> int {
        if (limbs) {
            SGZ_HIP(ctx, hipMalloc(&lb.p, 3 * planeBytes));
            SGZ_HIP(ctx, hipMalloc(&sp.p, 2 * (size_t)stride * sizeof(SelfSimLimbs)));
            hipLaunchKernelGGL(selfsim_gather_limbs_kernel, dim3(gs), dim3(256), 0, ctx->stream, (const float *)planar,
                               nframes, num_channels, af_start, half_win, decim, img_ext, stride, (int)kgT, (int)kgS,
                               (uint4 *)lb.p, (SelfSimStat *)st.p, (SelfSimLimbs *)sp.p);
        } else {
            SGZ_HIP(ctx, hipMalloc(&cols.p, colFloats * sizeof(float)));
            hipLaunchKernelGGL(selfsim_gather_kernel, dim3(gs), dim3(256), 0, ctx->stream, (const float *)planar, nframes,
                               num_channels, af_start, half_win, decim, img_ext, stride, (float *)cols.p,
                               (SelfSimStat *)st.p);
        }
        SGZ_HIP(ctx, hipGetLastError());
        return SGZ_OK;
    };
    int grc;
    if ((grc = gather(p1.p, (long long)num_frames1, st1, cols1, limbs1, sp1))) return grc;
    if (!same) {
        SGZ_HIP(ctx, hipMalloc(&st2.p, 2 * (size_t)stride * sizeof(SelfSimStat)));
        if ((grc = gather(p2.p, (long long)num_frames2, st2, cols2, limbs2, sp2))) return grc;
    }
    SelfSimArgs a;
    a.x1 = (const float *)p1.p;
    a.x2 = (const float *)(same ? p1.p : p2.p);
    a.frames1 = num_frames1;
    a.frames2 = num_frames2;
    a.cols1 = (const float *)cols1.p;
    a.cols2 = (const float *)(same ? cols1.p : cols2.p);
    a.st1 = (const SelfSimStat *)st1.p;
    a.st2 = (const SelfSimStat *)(same ? st1.p : st2.p);
    a.limbs1 = (const uint4 *)limbs1.p;
    a.limbs2 = (const uint4 *)(same ? limbs1.p : limbs2.p);
    a.sp1 = (const SelfSimLimbs *)sp1.p;
    a.sp2 = (const SelfSimLimbs *)(same ? sp1.p : sp2.p);
    a.kgT = (int)kgT;
    a.kgS = (int)kgS;
    a.doT = doT ? 1 : 0;
    a.doS = doS ? 1 : 0;
    a.stride = stride;
    a.numCh = num_channels;
    a.afStart = af_start;
    a.H = half_win;
    a.decim = decim;
    a.imgExt = img_ext;
    a.tilesEdge = (int)gt;
    a.superEdge = (int)((gt + 7) / 8);
    const unsigned supers = (unsigned)a.superEdge * a.superEdge;
    const unsigned gridWGs = ((supers + 7) / 8) * 8 * 64;
    a.warp = (double)color_warp;
    a.scale = 1.0f / color_ceil;
    a.inv = color_inv;
    a.simOut = (float *)simd.p;
    a.wantSim = (sim_out || keep_sim) ? 1 : 0;
    a.keepUpper = (sim_out || img_out || keep_sim) ? 1 : 0;
    a.imgOut = (int *)imgd.p;
    a.wT = temporal_weight;
    // Gray levels are integer work: a pixel whose level the similarity's error bound could change is listed and
    // evaluated again in the reference's order.  The list holds 1/16 of the pixels (with 255 level boundaries and
    // an error of 1e-5 about 1/200 of them are expected); data that overflows it gets every pixel evaluated that way.
    DevBuf fixd;
    long long fixCap = std::max<long long>(1 << 16, (long long)(pix / 16));
    if (const char *e = std::getenv("SGZ_SELFSIM_FIX_CAP")) {  // tuning / test knob: entries of that list
        const long long v = std::atoll(e);
        if (v >= 0) fixCap = v;
    }
    a.fixCount = nullptr;
    a.fixList = nullptr;
    a.fixCap = fixCap;
    DevBuf seqd;
    a.seqCount = nullptr;
    a.seqList = nullptr;
    a.seqCap = 1u << 20;
    a.forceSeq = 0;
    a.slack = 0.0f;
    if (const char *e = std::getenv("SGZ_SELFSIM_SLACK")) a.slack = std::max(0.0f, (float)std::atof(e));
    if (const char *e = std::getenv("SGZ_SELFSIM_SEQ")) {  // test knob: "force[,capacity]"
        a.forceSeq = std::strncmp(e, "force", 5) == 0;
        if (const char *c = std::strchr(e, ',')) a.seqCap = (unsigned)std::max(0ll, std::atoll(c + 1));
    }
    // LDS of selfsim_seq_kernel: both halves of the larger channel group, when 64 KB hold them
    const size_t seqWin = 2 * (size_t)std::max(1, num_channels - 1) * half_win;
    const int seqLds = seqWin * sizeof(float) <= 65536 ? (int)seqWin : 0;
    if (img_out) {
        SGZ_HIP(ctx, hipMalloc(&fixd.p, sizeof(int2) * (size_t)(fixCap + 1)));
        SGZ_HIP(ctx, hipMemsetAsync(fixd.p, 0, sizeof(int2), ctx->stream));
        a.fixCount = reinterpret_cast<unsigned long long *>(fixd.p);
        a.fixList = reinterpret_cast<int2 *>(fixd.p) + 1;
        SGZ_HIP(ctx, hipMalloc(&seqd.p, sizeof(SelfSimSeq) * ((size_t)a.seqCap + 1)));
        SGZ_HIP(ctx, hipMemsetAsync(seqd.p, 0, sizeof(SelfSimSeq), ctx->stream));
        a.seqCount = reinterpret_cast<unsigned int *>(seqd.p);
        a.seqList = reinterpret_cast<SelfSimSeq *>(seqd.p) + 1;
    }
    if (limbs) {
        a.c0 = a.c1 = a.first = 0;
        a.last = 1;
        a.wThis = 0.0f;
        a.keepUpper = a.wantSim;
        constexpr int ldsBytes = kSfFrontBytes + kSfErrBytes;
        SGZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(selfsim_tile8_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
        hipLaunchKernelGGL(selfsim_tile8_kernel, dim3(gridWGs), dim3(kSfThreads), ldsBytes, ctx->stream, a);
        SGZ_HIP(ctx, hipGetLastError());
    }
    if (!limbs && doT) {
        a.c0 = 0;
        a.c1 = 1;
        a.first = 1;
        a.last = doS ? 0 : 1;
        a.wThis = temporal_weight;
        hipLaunchKernelGGL(selfsim_tile_kernel, dim3(gridWGs), dim3(256), 0, ctx->stream, a);
        SGZ_HIP(ctx, hipGetLastError());
    }
    if (!limbs && doS) {
        a.c0 = 1;
        a.c1 = num_channels;
        a.first = doT ? 0 : 1;
        a.last = 1;
        a.wThis = 1.0f - temporal_weight;
        hipLaunchKernelGGL(selfsim_tile_kernel, dim3(gridWGs), dim3(256), 0, ctx->stream, a);
        SGZ_HIP(ctx, hipGetLastError());
    }
    if (img_out) {
        if (!limbs) {  // (the integer kernel lists its uncertain pixels itself)
            hipLaunchKernelGGL(selfsim_list_kernel, dim3(gt * gt), dim3(256), 0, ctx->stream, a);
            SGZ_HIP(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(selfsim_fix_kernel, dim3(2048), dim3(256), 0, ctx->stream, a, (const int2 *)a.fixList,
                           (const unsigned long long *)a.fixCount, 0ll);
        SGZ_HIP(ctx, hipGetLastError());
        unsigned long long listed = 0;
        SGZ_HIP(ctx, hipMemcpyAsync(&listed, a.fixCount, sizeof listed, hipMemcpyDeviceToHost, ctx->stream));
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (std::getenv("SGZ_SELFSIM_VERBOSE"))
            std::fprintf(stderr, "selfsim: %llu of %zu pixels listed for the exact evaluation (list of %lld)\n", listed,
                         pix, fixCap);
        if (listed > (unsigned long long)fixCap) {  // more than the list holds: every pixel that way
            SGZ_HIP(ctx, hipMemsetAsync(a.seqCount, 0, sizeof(unsigned int), ctx->stream));
            hipLaunchKernelGGL(selfsim_fix_kernel, dim3(8192), dim3(256), 0, ctx->stream, a, (const int2 *)nullptr,
                               (const unsigned long long *)nullptr, (long long)pix);
            SGZ_HIP(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(selfsim_seq_kernel, dim3(4096), dim3(64), (size_t)seqLds * sizeof(float), ctx->stream, a,
                           seqLds);
        SGZ_HIP(ctx, hipGetLastError());
    }
    int rc;
    if (sim_out && (rc = copy_large(ctx, sim_out, simd.p, pix * sizeof(float), true))) return rc;
    if (img_out && (rc = copy_large(ctx, img_out, imgd.p, pix * sizeof(int32_t), true))) return rc;
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (keep_sim) {
        *keep_sim = (float *)simd.p;
        simd.p = nullptr;
    }
    return SGZ_OK;
}

int sgz_selfsim_run(sgz_context *ctx, const float *frames1, int64_t num_frames1, const float *frames2,
                    int64_t num_frames2, int num_channels, const float *norms, int af_start, int half_win,
                    int decim, int img_ext,
                    float temporal_weight, int colors, float color_warp, float color_ceil, int color_inv,
                    float *sim_out, int32_t *img_out)
{
    return selfsim_impl(ctx, frames1, num_frames1, frames2, num_frames2, num_channels, norms, af_start, half_win, decim,
                        img_ext, temporal_weight, colors, color_warp, color_ceil, color_inv, sim_out, img_out, nullptr);
}

// A digest of everything the matrix depends on (FNV-1a over 64-bit words of the frames, then the parameters).
static unsigned long long selfsim_digest(const float *frames1, int64_t n1, const float *frames2, int64_t n2, int numCh,
                                         const float *norms, int af_start, int half_win, int decim, int img_ext,
                                         float temporal_weight)
{
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&h](const void *p, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        size_t i = 0;
        for (; i + 8 <= bytes; i += 8) {
            unsigned long long w;
            std::memcpy(&w, b + i, 8);
            h = (h ^ w) * 1099511628211ull;
        }
        for (; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ull;
    };
    mix(frames1, sizeof(float) * (size_t)n1 * numCh);
    const bool same = frames1 == frames2 && n1 == n2;
    if (!same) mix(frames2, sizeof(float) * (size_t)n2 * numCh);
    const long long par[8] = {n1, n2, numCh, af_start, half_win, decim, img_ext, same ? 1 : 0};
    mix(par, sizeof par);
    mix(&temporal_weight, sizeof temporal_weight);
    if (norms) mix(norms, sizeof(float) * 2 * (size_t)numCh);
    return h ? h : 1ull;
}

int sgz_selfsim_rows(sgz_context *ctx, const float *frames1, int64_t num_frames1, const float *frames2,
                     int64_t num_frames2, int num_channels, const float *norms, int af_start, int half_win, int decim,
                     int img_ext, float temporal_weight, int x_begin, int x_end, float *sim_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (!frames1 || !frames2 || num_frames1 < 0 || num_frames2 < 0 || num_channels < 2 ||
        num_channels > SGZ_MAX_CHANNELS)
        return ctx->fail(SGZ_E_ARG, "bad frames / channels");
    if (x_begin < 0 || x_end < x_begin || x_end > img_ext || (x_end > x_begin && !sim_out))
        return ctx->fail(SGZ_E_ARG, "bad band of rows");
    const unsigned long long key = selfsim_digest(frames1, num_frames1, frames2, num_frames2, num_channels, norms, af_start,
                                                  half_win, decim, img_ext, temporal_weight);
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    if (key != ctx->ss_key || !ctx->d_ss_sim) {
        if (ctx->d_ss_sim) (void)hipFree(ctx->d_ss_sim);
        ctx->d_ss_sim = nullptr;
        ctx->ss_key = 0;
        if (img_ext > 0) {
            const int rc = selfsim_impl(ctx, frames1, num_frames1, frames2, num_frames2, num_channels, norms, af_start,
                                        half_win, decim, img_ext, temporal_weight, 0, 1.0f, 1.0f, 0, nullptr, nullptr,
                                        &ctx->d_ss_sim);
            if (rc) return rc;
            ctx->ss_key = key;
        }
    }
    if (x_end > x_begin) {
        const int rc = copy_large(ctx, sim_out, ctx->d_ss_sim + (size_t)x_begin * img_ext,
                                  sizeof(float) * (size_t)(x_end - x_begin) * img_ext, true);
        if (rc) return rc;
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SGZ_OK;
}

int sgz_selfsim_release(sgz_context *ctx)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->d_ss_sim) {
        SGZ_HIP(ctx, hipSetDevice(ctx->device));
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SGZ_HIP(ctx, hipFree(ctx->d_ss_sim));
    }
    ctx->d_ss_sim = nullptr;
    ctx->ss_key = 0;
    return SGZ_OK;
}

}  // extern "C"
