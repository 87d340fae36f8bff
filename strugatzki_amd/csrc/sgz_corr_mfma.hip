// Half-precision matrix-core screening of the sliding cross term
// (impl/FeatureCorrelationImpl.scala:193-232 evaluates, per window o,
//  sum_i a'[c][i] * b[c][o + i] through MathUtil.correlate, impl/MathUtil.scala:177-196).
//
// With o = 16 m + r and kk = i + r the sum over lags becomes a dense matrix product
//
//     out[r][m] = sum_kk  A[r][kk] * X[kk][m],   A[r][kk] = a'[kk - r],   X[kk][m] = b[16 m + kk]
//
// (A is the target in 16 shifted copies, zero outside [0, L); X is the file read
// at chunk stride 16).  v_mfma_f32_16x16x32_f16 multiplies a 16 x 32 slice of A
// (one "step" s: kk = 32 s .. 32 s + 31) with the 32 x 16 slice of X of one
// column tile (16 chunks).  The operand of tile t at step s is the operand of
// tile t + 1 at step s - 8, so wave q of a workgroup takes the steps
// s = q, q + 8, q + 16, ... and keeps a sliding window of NB operands in
// registers: every step costs one 16-byte LDS read of database frames and one
// target operand for NB matrix instructions.  NB (the width of a workgroup's tile,
// 256 NB windows) is a template parameter: which width wastes the fewest matrix
// instructions on the empty end of a file's last tile, and fills the last round of
// workgroups best, depends on the file lengths and the target (half_screen_pick_tile).
//
// The products are half precision (11 significant bits), the sums fp32; window
// means and std-devs still come from the fp64 prefix sums.  The result is a
// screening value: callers re-score every window that comes within a margin of
// their bound with the exact kernel (rescore_kernel) before using it.  What the
// kernel stores is an upper bound of the similarity (window_bounds): the
// half-precision value plus the largest error the roundings can cause for that
// window, so a threshold test on it never loses a window.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sgz_corr_dev.hpp"
#include "sgz_internal.hpp"

using namespace sgz;

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kHThreads = kHWaves * 64;
// one partial sum per window of a tile, skewed (see reduce)
__host__ __device__ constexpr int half_slab_floats(int NB) { return NB * 256 + NB * 32; }

// Relative error of a cross term of `channels` channels, against sum |a'| |b|.  Two half-precision roundings per
// product (target and database value, 2^-11 each: (1 + 2^-11)^2 - 1 = 2^-10 + 2^-22; the product of two
// half-precision numbers is exact in fp32).  The sums are fp32: an output element goes through
// channels * ceil(S2 / 8) matrix instructions in sequence, each of which may round its 32-term sum and the
// accumulation several times (counted as 6 roundings of 2^-24), then through the 8 additions of the cross-wave
// sum.  The term grows with the target length: 5e-5 at 2 584 frames, 2.3e-4 at 12 500.
float half_rel_err(int channels, int S2)
{
    const double steps = (double)channels * (double)((S2 + 7) / 8);
    return (float)(std::ldexp(1.0, -10) + std::ldexp(1.0, -22) + (6.0 * steps + 8.0) * std::ldexp(1.0, -24)) * 1.0001f;
}

// One staged channel: XU 16-byte units of database frames, then the two target copies (LAd dwords each),
// each part rounded up to whole 1 KiB pieces, the unit one wave-wide direct-to-LDS load fills
// (+1 KiB: the operand requests of the step past the last one stay inside the buffer).
__host__ __device__ constexpr int half_x_pieces(int XU) { return (XU + 63) >> 6; }
__host__ __device__ constexpr int half_a_pieces(int LAd) { return (LAd / 2 + 63) >> 6; }
__host__ __device__ constexpr size_t half_buf_bytes(int XU, int LAd, int NB)
{
    const size_t bytes = (size_t)1024 * (half_x_pieces(XU) + half_a_pieces(LAd) + 1);
    // once a buffer's channel is consumed it holds two slabs of partial sums and a few words
    const size_t sums = 2 * half_slab_floats(NB) * sizeof(float) + 1024;
    return bytes > sums ? bytes : sums;
}

typedef __attribute__((address_space(3))) void lds_void;

size_t half_lds_bytes(const HalfGeom &hg) { return 2 * half_buf_bytes(hg.XU, hg.LAd, hg.NB); }

// Workgroups per CU a variant is built for: up to 5 column tiles per wave the loop fits 80 registers and three
// workgroups share a CU (when their LDS does, too), above that two.
constexpr int half_wgs_per_cu(int NB) { return NB <= 5 ? 3 : 2; }

// Workgroup = 8 waves, one tile of 256 NB consecutive windows of one file.
// Per channel the workgroup stages (double buffered) the frames
// [o0, o0 + 8 XU) of the file in natural order and the two dword-aligned
// copies of the target channel (even shifts read copy 0, odd shifts copy 1,
// which is stored one element further left).  Staging uses direct-to-LDS loads
// (buffer_load_dwordx4 ... lds: per-lane source offset, destination = wave-uniform
// base + 16 * lane), issued a phase ahead and retired by the barrier that ends
// the phase, so no registers hold data in flight.
template <int NB>
__global__ __launch_bounds__(kHThreads, 2 * half_wgs_per_cu(NB)) void corr_mfma_kernel(
    const HalfTile *__restrict__ tiles, const _Float16 *__restrict__ planarH, const float4 *__restrict__ wstat,
    const uint32_t *__restrict__ aimg, CorrParams cp, HalfGeom hg, float *__restrict__ simOut,
    float *__restrict__ boostOut, int *__restrict__ lokey, int *__restrict__ lokeyNext,
    int *__restrict__ hflags)
{
    constexpr int kHTileW = NB * 256, kHSlabFloats = half_slab_floats(NB);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const HalfTile td = tiles[blockIdx.x];
    const size_t bufBytes = half_buf_bytes(hg.XU, hg.LAd, NB);
    const long long rowLen = td.rowLen;
    const _Float16 *fplanar = planarH + td.planar;

    // Staging through buffer loads with the LDS as destination: the descriptor carries the wave-uniform row
    // base and the valid length (lanes past it fill padding with zeros), the lane its 16-byte offset.  (The
    // flat-addressed form of these loads makes the compiler drain every LDS read completely while one is
    // in flight, which the loop below cannot afford.)
    const int nXP = half_x_pieces(hg.XU), nAP = half_a_pieces(hg.LAd);
    auto stage = [&](int c, int buf) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(fplanar + (long long)c * rowLen), 0, hg.XU * 16, 0x00020000);
        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(aimg + (size_t)c * 2 * hg.LAd), 0, hg.LAd * 8, 0x00020000);
        unsigned char *dst = lds + buf * bufBytes;
        for (int piece = q; piece < nXP + nAP; piece += kHWaves) {
            lds_void *to = (lds_void *)(dst + (size_t)piece * 1024);
            if (piece < nXP)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, to, 16, (piece * 64 + lane) * 16, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, to, 16, ((piece - nXP) * 64 + lane) * 16, 0, 0, 0);
        }
    };

    const int m = lane & 15, g = lane >> 4;
    const int hBase = 2 * m + g + 4 * q;                                              // 16-byte units
    const int aBase = (m & 1) * hg.LAd + 4 * g + 8 - ((m + 1) >> 1) + 16 * q;          // dwords
    const int P = (hg.S2 - q + 7) >> 3;  // steps of this wave: s = q, q + 8, ... < S2

    // The cross-wave sums take the place of staged channels that have been consumed: the MFCC sum goes where
    // the last MFCC channel was (the loudness channel sits in the other buffer by then), the loudness sum
    // where the loudness channel was.
    float4 *redS4 = reinterpret_cast<float4 *>(lds + ((cp.numCh - 2) & 1) * bufBytes);
    float4 *redT4 = reinterpret_cast<float4 *>(lds + ((cp.numCh - 1) & 1) * bufBytes);
    // Cross-wave sum in a fixed order (results are reproducible): even waves add into one slab, odd waves into
    // a second one, a pair per barrier round; readers add the two slabs.  Units of four windows are skewed by
    // one unit in eight, which keeps the 16-byte accesses of eight neighbouring lanes on distinct banks.
    auto red_unit = [](int u) { return u + (u >> 3); };
    auto reduce = [&](float4 *red, const f32x4 (&acc)[NB]) {
        float4 *slab = red + (q & 1) * (kHSlabFloats / 4);
        for (int round = 0; round < kHWaves / 2; round++) {
            if ((q >> 1) == round) {
#pragma unroll
                for (int t = 0; t < NB; t++) {
                    const int idx = red_unit(64 * t + 4 * m + g);  // windows 256 t + 16 m + 4 g + (0..3)
                    float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                    if (round) {
                        const float4 pv = slab[idx];
                        v.x += pv.x;
                        v.y += pv.y;
                        v.z += pv.z;
                        v.w += pv.w;
                    }
                    slab[idx] = v;
                }
            }
            __syncthreads();
        }
    };
    auto red_read = [&](const float4 *red, int wdw) {
        const float *f = reinterpret_cast<const float *>(red);
        const int pos = 4 * red_unit(wdw >> 2) + (wdw & 3);
        return f[pos] + f[kHSlabFloats + pos];
    };

    // phase order: the MFCC channels 1 .. numCh-1 first (one sum), the loudness channel last
    const int numCh = cp.numCh;
    auto channel_of = [&](int ph) { return ph + 1 < numCh ? ph + 1 : 0; };

    stage(channel_of(0), 0);

    constexpr int kPerThread = (kHTileW + kHThreads - 1) / kHThreads;  // (the last one of an odd NB: half the threads)
    __syncthreads();

    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ph = 0; ph < numCh; ph++) {
        if (ph + 1 < numCh) stage(channel_of(ph + 1), (ph + 1) & 1);
        const uint4 *Xu = reinterpret_cast<const uint4 *>(lds + (ph & 1) * bufBytes) + hBase;
        const uint32_t *Ad = reinterpret_cast<const uint32_t *>(lds + (ph & 1) * bufBytes + (size_t)nXP * 1024) + aBase;

        half8 h[NB];
#pragma unroll
        for (int t = 0; t < NB; t++) h[t] = __builtin_bit_cast(half8, Xu[32 * t]);
        auto load_a = [&](int p) {
            const uint32_t *ap = Ad + 128 * p;
            uint4 av;
            av.x = ap[0];
            av.y = ap[1];
            av.z = ap[2];
            av.w = ap[3];
            return __builtin_bit_cast(half8, av);
        };
        // Wave q runs its P steps (the image has room for whole blocks: a request past the wave's last step
        // reads zeros or padding and is not used).  Straight-line blocks of NB steps:
        // the target operand of the next step is requested before, the freed slot of the operand window
        // right after this step's first matrix instruction, so each wait covers exactly the reads it needs.
        half8 a[2];
        a[0] = load_a(0);
        int pb = 0;
        for (; pb + NB <= P; pb += NB) {
#pragma unroll
            for (int u = 0; u < NB; u++) {
                a[(u + 1) & 1] = load_a(pb + u + 1);  // the image has room for one step past the last
                __builtin_amdgcn_sched_barrier(0);    // (keeps the requests ahead of the matrix instructions)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1], h[u], acc[0], 0, 0, 0);
                h[u] = __builtin_bit_cast(half8, Xu[32 * (pb + u + NB)]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 1; t < NB; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1], h[(u + t) % NB], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NB % 2 != 0) a[0] = a[1];  // the next block starts with slot 0 again
        }
#pragma unroll
        for (int u = 0; u < NB - 1; u++) {
            if (pb + u >= P) break;
            a[(u + 1) & 1] = load_a(pb + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1], h[u], acc[0], 0, 0, 0);
            h[u] = __builtin_bit_cast(half8, Xu[32 * (pb + u + NB)]);  // for the steps that follow, if any
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 1; t < NB; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1], h[(u + t) % NB], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        __syncthreads();  // also retires the loads staged above
        if (ph == numCh - 2) {  // all MFCC channels done
            reduce(redS4, acc);
#pragma unroll
            for (int t = 0; t < NB; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // What the bounds need of every window besides its cross terms (WindowStatic: two 16-byte loads a window,
    // kept from the first search over this database on) is requested before the last reduction and arrives
    // during it.
    float4 ws0[kPerThread], ws1[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
        const int wdw = tid + k * kHThreads;
        const long long w = td.win + (wdw < td.valid ? wdw : 0);
        ws0[k] = wstat[2 * w];
        ws1[k] = wstat[2 * w + 1];
    }
    reduce(redT4, acc);
    CorrParams cph = cp;  // the statistics of the half-precision target image
    cph.sumAT = hg.sumAT;
    cph.sumAS = hg.sumAS;

    float best = -INFINITY;
    int bad = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
        const int wdw = tid + k * kHThreads;
        if (wdw < td.valid) {
            float sim, lo;
            float boost;
            WindowStatic ws;
            ws.rsdT = ws0[k].x, ws.dmT = ws0[k].y, ws.gT = ws0[k].z, ws.lnAvg = ws0[k].w;
            ws.rsdS = ws1[k].x, ws.dmS = ws1[k].y, ws.gS = ws1[k].z, ws.hTS = __float_as_uint(ws1[k].w);
            const WindowCoeffs wc = window_coeffs(cph, ws, hg.relErrT, hg.relErrS, boost);
            window_bounds(wc, red_read(redT4, wdw), red_read(redS4, wdw), sim, lo);
            boostOut[td.win + wdw] = boost;
            simOut[td.win + wdw] = sim;
            best = fmaxf(best, lo);
            bad |= (lo != lo) || (sim != sim);
        }
    }
    // the file's largest lower bound (one atomic per workgroup); a NaN similarity changes what the reference's
    // loop keeps, so such files are flagged and left to the exact path
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        best = fmaxf(best, __shfl_xor(best, off, 64));
        bad |= __shfl_xor(bad, off, 64);
    }
    float *wbest = reinterpret_cast<float *>(lds) + 2 * kHSlabFloats;  // behind the sums of buffer 0
    int *wbad = reinterpret_cast<int *>(wbest) + kHWaves;
    if (lane == 0) {
        wbest[q] = best;
        wbad[q] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < kHWaves; k++) {
            best = fmaxf(best, wbest[k]);
            bad |= wbad[k];
        }
        atomicMax(&lokey[td.file], bound_key(best));
        if (bad) atomicOr(&hflags[td.file], 2);
        // the run after this one folds its bounds into the other array: the file's first tile resets its entry
        // there (nothing reads that array before then: kernels of one stream run in order)
        if (td.o0 == 0) lokeyNext[td.file] = bound_key(-INFINITY);
    }
}

}  // namespace

namespace sgz {

// tile widths (column tiles per wave) a kernel exists for; the first is the default
constexpr int kHalfWidths[] = {kHNB, 4, 5, 6, 7, 9, 10};

// Geometry for a target of L frames; false when one staged channel does not fit
// the kernel's registers / LDS (the caller then screens with the fp32 kernel).
static bool half_geom(int L, int NB, HalfGeom &hg)
{
    const int S2 = (L + 15 + 31) / 32;
    const int Pmax = (S2 + 7) / 8;
    const int LAd = 128 * Pmax + 16;  // all 8 Pmax steps of 16 dwords + the shift room; = 16 mod 32
    const int XU = 32 * (Pmax + NB);
    hg.NB = NB;
    hg.S2 = S2;
    hg.LAd = LAd;
    hg.XU = XU;
    hg.relErrT = hg.relErrS = 0.0f;  // set with the target (they depend on the channel count)
    if (half_lds_bytes(hg) > 160 * 1024) return false;
    return true;
}

bool half_screen_fits(const sgz_context *ctx)
{
    // (half precision needs a bounded value range: a target that leaves it has no image, half_screen_set_target;
    // database files that leave it are marked by the preparation and sent to the exact path)
    return ctx->have_target && ctx->hg.S2 > 0;
}

// The tile width for the files and the target of the context's layout: the variant with the smallest estimate of
// the kernel's time.  A file of C column tiles costs ceil(C / NB) tiles, whatever part of the last one's columns
// is empty, and a tile NB + 1.2 column times (start, reductions and epilogue are worth 1.2 columns).  The chip
// holds CUs * k tiles at once (k workgroups per CU: 3 up to NB = 5 if their LDS fits, else 2) and works them off
// in rounds; the workgroups of a CU share its matrix pipe, so a round of k per CU takes t(k) = 0.55, 1, 1.5 tile
// times for k = 1, 2, 3, and the last round only as long as its fullest CU needs:
//     (full rounds * t(k) + t(ceil(tiles left / CUs))) * (NB + 1.2).
// Within 2 % of the measured order of the variants for 120 files of 91 columns (30 s target) and 1500 files of
// 81 columns (60 s target): 8 resp. 9 win there, 9 by 8 % over 8.  SGZ_HALF_NB=<n> overrides the choice.
int half_screen_pick_tile(sgz_context *ctx, const long long *numWindows, int numFiles)
{
    if (ctx->hg.S2 <= 0) return SGZ_OK;
    int numCUs = 256;
    if (hipDeviceGetAttribute(&numCUs, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || numCUs < 1)
        numCUs = 256;
    int forced = 0;
    if (const char *e = std::getenv("SGZ_HALF_NB")) forced = std::atoi(e);
    double bestCost = 0.0;
    HalfGeom best{};
    bool have = false;
    for (int nb : kHalfWidths) {
        HalfGeom hg = ctx->hg;
        if (!half_geom(ctx->cp.L, nb, hg)) continue;
        hg.relErrT = ctx->hg.relErrT;
        hg.relErrS = ctx->hg.relErrS;
        const int wgs = (int)std::min<size_t>((size_t)half_wgs_per_cu(nb), (160 * 1024) / half_lds_bytes(hg));
        if (wgs < 1) continue;
        long long tiles = 0;
        for (int f = 0; f < numFiles; f++) tiles += (numWindows[f] + 256ll * nb - 1) / (256ll * nb);
        const long long slots = (long long)numCUs * wgs;
        const long long left = tiles % slots;
        auto roundTime = [](long long perCU) { return perCU <= 0 ? 0.0 : perCU == 1 ? 0.55 : 0.5 * (double)perCU; };
        const double cost = ((double)(tiles / slots) * roundTime(wgs) + roundTime((left + numCUs - 1) / numCUs)) * (nb + 1.2);
        if (forced ? nb == forced : (!have || cost < bestCost * 0.995)) {  // (near ties: the earlier, wider default)
            best = hg;
            bestCost = cost;
            have = true;
        }
    }
    if (have) ctx->hg = best;
    return SGZ_OK;
}

// Halfwords of the target image (two shifted half-precision copies of every channel: copy k = 0, 1 holds a'[p - 16 + k]
// at halfword position p) at the widest geometry that fits: what sgz_corr_set_target reserves in its staging buffer.
size_t half_image_halfwords(int L, int numCh)
{
    size_t need = 0;
    for (size_t k = 0; k < sizeof kHalfWidths / sizeof kHalfWidths[0]; k++) {
        HalfGeom hg{};
        if (half_geom(L, kHalfWidths[k], hg)) need = std::max(need, (size_t)numCh * 2 * 2 * (size_t)hg.LAd);
    }
    return need;
}

// Geometry of the matrix-core kernel for the new target (none when it does not fit the LDS or the target leaves the
// half-precision range); the image is left to half_image_build.
int half_screen_set_target(sgz_context *ctx, bool in_half_range)
{
    HalfGeom hg{};
    const int L = ctx->cp.L, numCh = ctx->cp.numCh;
    // the width in use stays (the layout is built for it; a new target length rebuilds the layout and picks again)
    bool fits = ctx->hg.S2 > 0 && half_geom(L, ctx->hg.NB, hg);
    for (size_t k = 0; !fits && k < sizeof kHalfWidths / sizeof kHalfWidths[0]; k++) fits = half_geom(L, kHalfWidths[k], hg);
    const size_t total = (size_t)numCh * 2 * 2 * (size_t)hg.LAd;
    if (!fits || !in_half_range || total > ctx->half_stage_halfwords) {
        ctx->hg = HalfGeom{};
        ctx->half_img_dirty = false;
        return SGZ_OK;
    }
    hg.relErrT = half_rel_err(1, hg.S2);
    hg.relErrS = half_rel_err(numCh - 1, hg.S2);
    ctx->hg = hg;
    ctx->half_img_dirty = true;
    return SGZ_OK;
}

// The target's half-precision image (every channel at shifts 0 and 1 halfword) and the sums of its values, from the
// fp64 target of sgz_corr_set_target; the copy to the device is queued on `stream`.
static int half_image_build(sgz_context *ctx, hipStream_t stream)
{
    HalfGeom &hg = ctx->hg;
    const int L = ctx->cp.L, numCh = ctx->cp.numCh;
    const size_t perCopy = (size_t)2 * hg.LAd;  // halfwords
    const size_t total = (size_t)numCh * 2 * perCopy, Lpad = ctx->half_stage_lpad;
    uint16_t *stage = ctx->half_stage_img;
    double *aprime = ctx->half_stage_ap;
    if (!stage || !aprime || total > ctx->half_stage_halfwords) return ctx->fail(SGZ_E_STATE, "no staged target");
    // the centred fp64 target, which the preparation kernel left on the device
    SGZ_HIP(ctx, hipMemcpyAsync(aprime, ctx->d_aprime, sizeof(double) * (size_t)numCh * Lpad, hipMemcpyDeviceToHost, stream));
    SGZ_HIP(ctx, hipStreamSynchronize(stream));
    std::memset(stage, 0, total * sizeof(uint16_t));
    double sumT = 0.0, sumS = 0.0;
    for (int c = 0; c < numCh; c++) {
        uint16_t *c0 = stage + ((size_t)c * 2) * perCopy + 16, *c1 = stage + ((size_t)c * 2 + 1) * perCopy + 15;
        for (int i = 0; i < L; i++) {
            const _Float16 hv = (_Float16)aprime[(size_t)c * Lpad + i];
            uint16_t bits;
            std::memcpy(&bits, &hv, 2);
            if (c == 0) sumT += (double)hv; else sumS += (double)hv;
            c0[i] = bits;  // the image at shift 0 and, one halfword earlier, at shift 1 (L + 16 <= 2 LAd)
            c1[i] = bits;
        }
    }
    hg.sumAT = sumT;
    hg.sumAS = sumS;
    int rc;
    if ((rc = ensure(ctx, ctx->d_aimg_h, ctx->aimg_h_cap, total))) return rc;
    SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_aimg_h, stage, total * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
    ctx->half_img_dirty = false;
    return SGZ_OK;
}

int half_screen_launch(sgz_context *ctx, int tile_begin, int tile_end, hipStream_t stream, hipEvent_t start,
                       hipEvent_t stop)
{
    if (ctx->hg.S2 <= 0) return ctx->fail(SGZ_E_STATE, "half-precision screening is not available for this target");
    if (ctx->half_img_dirty) {
        const int irc = half_image_build(ctx, stream);
        if (irc) return irc;
    }
    const HalfGeom &hg = ctx->hg;
    const size_t ldsBytes = half_lds_bytes(hg);
    const _Float16 *planarH = reinterpret_cast<const _Float16 *>(ctx->d_planar_h);
    const uint32_t *aimg = reinterpret_cast<const uint32_t *>(ctx->d_aimg_h);
    // the array of per-file bounds this run folds into must hold -inf everywhere: the previous run reset it,
    // unless it covered other files than this one does
    const int nf = (int)ctx->files.size();
    const int par = ctx->lokey_parity;
    int *lokey = ctx->d_lokey + (size_t)par * nf, *lokeyNext = ctx->d_lokey + (size_t)(1 - par) * nf;
    if (!ctx->lokey_clean[par])
        SGZ_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(lokey), (int)0x807fffff, (size_t)nf, stream));
    const bool whole = tile_begin == 0 && tile_end == (int)ctx->h_tiles_h.size();
    ctx->lokey_clean[par] = false;
    ctx->lokey_clean[1 - par] = whole || ctx->lokey_clean[1 - par];
    ctx->lokey_cur = lokey;
    ctx->lokey_parity = 1 - par;
    auto launch = [&](auto kern) -> int {
        if (ldsBytes > 64 * 1024)
            SGZ_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
        if (start || stop)
            hipExtLaunchKernelGGL(kern, dim3(tile_end - tile_begin), dim3(kHThreads), ldsBytes, stream, start, stop, 0,
                                  ctx->d_tiles_h + tile_begin, planarH, reinterpret_cast<const float4 *>(ctx->d_wstat_h),
                                  aimg, ctx->cp, hg, ctx->d_sim, ctx->d_boost, lokey, lokeyNext,
                                  ctx->d_hflags);
        else
            hipLaunchKernelGGL(kern, dim3(tile_end - tile_begin), dim3(kHThreads), ldsBytes, stream,
                               ctx->d_tiles_h + tile_begin, planarH, reinterpret_cast<const float4 *>(ctx->d_wstat_h),
                               aimg, ctx->cp, hg, ctx->d_sim, ctx->d_boost, lokey, lokeyNext, ctx->d_hflags);
        SGZ_HIP(ctx, hipGetLastError());
        return SGZ_OK;
    };
    switch (hg.NB) {
    case 4: return launch(corr_mfma_kernel<4>);
    case 5: return launch(corr_mfma_kernel<5>);
    case 6: return launch(corr_mfma_kernel<6>);
    case 7: return launch(corr_mfma_kernel<7>);
    case 8: return launch(corr_mfma_kernel<8>);
    case 9: return launch(corr_mfma_kernel<9>);
    case 10: return launch(corr_mfma_kernel<10>);
    default: return ctx->fail(SGZ_E_STATE, "no matrix-core kernel of that tile width");
    }
}

}  // namespace sgz
