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
//   scan_kernel   per-file prefix sums of those moments (window mean / std-dev
//                 become two subtractions)
//   corr_kernel   cross terms sum_i a'[c][i] * b[c][o+i] for 1024 windows per
//                 workgroup, fp64 FMA, then boost / similarity per window
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "sgz_internal.hpp"

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
__global__ __launch_bounds__(256) void prep_kernel(const FileDesc *__restrict__ files,
                                                   const PrepTile *__restrict__ ptiles,
                                                   double *__restrict__ planar, double *__restrict__ prefix,
                                                   NormParams nrm, int numCh, int cb, double kT, double kS)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const PrepTile pt = ptiles[blockIdx.x];
    const FileDesc fd = files[pt.file];
    const int jstride = cb + 1;
    const int cstride = kR * jstride + 1;
    const long long f0 = (long long)pt.chunk0 * kR;
    const int tileFrames = cb * kR;
    const int tid = threadIdx.x;

    // phase 1: coalesced read of the interleaved frames, normalise in fp32
    const int total = tileFrames * numCh;
    const float *src = fd.raw + f0 * numCh;
    const long long validElems = (fd.frames - f0) * numCh;  // may be <= 0 for padding tiles
    for (int e = tid; e < total; e += 256) {
        const int fr = e / numCh;
        const int c = e - fr * numCh;
        float v = 0.0f;
        if (e < validElems) {
            v = src[e];
            if (nrm.enabled) v = (v - nrm.mn[c]) / nrm.d[c];
        }
        xs[c * cstride + (fr & (kR - 1)) * jstride + (fr >> 4)] = v;
    }
    __syncthreads();

    // phase 2: chunk-major planar fp64, centred; frames past the end are 0
    double *dst = planar + fd.planar_off;
    const int rows = numCh * kR;
    for (int idx = tid; idx < rows * cb; idx += 256) {
        const int mm = idx % cb;
        const int row = idx / cb;
        const int c = row >> 4;
        const int j = row & (kR - 1);
        const long long m = (long long)pt.chunk0 + mm;
        if (m < fd.M) {
            const long long fr = m * kR + j;
            double v = 0.0;
            if (fr < fd.frames) v = (double)xs[c * cstride + j * jstride + mm] - (c == 0 ? kT : kS);
            dst[(long long)row * fd.M + m] = v;
        }
    }

    // phase 3: per-frame moments, stored at index frame+1 of the prefix arrays
    double *pre = prefix + fd.prefix_off;
    const long long n1 = fd.frames + 1;
    for (int fr = tid; fr < tileFrames; fr += 256) {
        const long long gf = f0 + fr;
        if (gf < fd.frames) {
            const int j = fr & (kR - 1), mm = fr >> 4;
            const double t = (double)xs[j * jstride + mm] - kT;
            double s1 = 0.0, s2 = 0.0;
            for (int c = 1; c < numCh; c++) {
                const double d = (double)xs[c * cstride + j * jstride + mm] - kS;
                s1 += d;
                s2 = fma(d, d, s2);
            }
            pre[gf + 1] = t;
            pre[n1 + gf + 1] = t * t;
            pre[2 * n1 + gf + 1] = s1;
            pre[3 * n1 + gf + 1] = s2;
        }
    }
}

// ---------------------------------------------------------------------------
// scan_kernel: in-place inclusive scan of P[1..n] per (array, file); P[0] = 0.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_kernel(const FileDesc *__restrict__ files, int fileBegin,
                                                    double *__restrict__ prefix)
{
    __shared__ double wsum[16];
    const FileDesc fd = files[fileBegin + blockIdx.y];
    const long long n = fd.frames;
    double *p = prefix + fd.prefix_off + (long long)blockIdx.x * (n + 1);
    const int tid = threadIdx.x;
    const long long per = (n + 1023) / 1024;
    const long long b = 1 + (long long)tid * per;
    const long long e = std::min(n + 1, b + per);
    double local = 0.0;
    for (long long i = b; i < e; i++) local += p[i];
    // inclusive scan of `local` across the wave, then across the 16 waves
    double inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(inc, off, 64);
        if ((tid & 63) >= off) inc += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    double base = 0.0;
    for (int wv = 0; wv < (tid >> 6); wv++) base += wsum[wv];
    double run = base + inc - local;  // exclusive prefix of this thread
    for (long long i = b; i < e; i++) {
        run += p[i];
        p[i] = run;
    }
    if (tid == 0) p[0] = 0.0;
}

// ---------------------------------------------------------------------------
// corr_kernel
//
// Workgroup = 4 waves, one tile of 1024 consecutive windows of one file.  Lane t
// owns windows o0 + 16t .. o0 + 16t + 15 (16 fp64 accumulators).  A "step" is 16
// consecutive lags: with x[0..31) the 31 frames starting at the lane's first
// window + lag, acc[r] += a'[lag + q] * x[q + r] for q, r in 0..15, i.e. 256 FMAs
// for 16 newly read frames and 16 target values.  The target values are
// wave-uniform (scalar loads feeding the FMA's SGPR operand); the frames come
// from an LDS image of the planar channel block in chunk-major order, so that
// the 64 lanes read 64 consecutive doubles per ds_read_b64.  The 4 waves split
// the lag steps of each (channel, block) phase; partial sums meet in LDS at the
// end.  Staging of the next phase's block (global -> registers -> LDS) is
// issued before and retired after the FMA work of the current phase.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_chunk(double (&x)[kR], const double *p)
{
#pragma unroll
    for (int j = 0; j < kR; j++) x[j] = p[j * kMT];
}

__device__ __forceinline__ void fma_step(double (&acc)[kR], const double *__restrict__ a, const double (&x0)[kR],
                                         const double (&x1)[kR])
{
#pragma unroll
    for (int q = 0; q < kR; q++) {
        const double aq = a[q];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const int k = q + r;
            acc[r] = fma(aq, k < kR ? x0[k] : x1[k - kR], acc[r]);
        }
    }
}

struct Epilogue {
    const double *pt1, *pt2, *ps1, *ps2;
};

__device__ __forceinline__ void window_result(const CorrParams &cp, const Epilogue &ep, long long o,
                                              double crossT, double crossS, float &simOut, float &boostOut)
{
    const double nT = (double)cp.L;
    const double nS = (double)cp.L * (double)(cp.numCh - 1);
    const double dmT = (ep.pt1[o + cp.L] - ep.pt1[o]) / nT;  // window mean - kT
    const double avgD = cp.kT + dmT;
    const float avgF = (float)avgD;  // MathUtil.avg returns Float
    const double lnAvgB = log((double)avgF);
    const float boost = (float)exp((cp.lnAvgIn - lnAvgB) / 0.6);
    float sim = 0.0f;
    if (boost <= cp.maxBoost) {
        float temporal = 0.0f, spectral = 0.0f;
        if (cp.w > 0.0f) {
            double var = (ep.pt2[o + cp.L] - ep.pt2[o]) / nT - dmT * dmT;
            if (var < 0.0) var = 0.0;
            const double bStd = sqrt(var);
            const double cross = crossT - dmT * cp.sumAT;
            temporal = (float)(cross / (cp.aStdT * bStd * nT));
        }
        if (cp.w < 1.0f) {
            const double dmS = (ep.ps1[o + cp.L] - ep.ps1[o]) / nS;
            double var = (ep.ps2[o + cp.L] - ep.ps2[o]) / nS - dmS * dmS;
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

constexpr int kStageIters = (kR * kMT) / (kWaves * 64 * 2);  // double2 copies per thread per block

__global__ __launch_bounds__(kWaves * 64, 2) void corr_kernel(const FileDesc *__restrict__ files,
                                                              const TileDesc *__restrict__ tiles,
                                                              const double *__restrict__ planar,
                                                              const double *__restrict__ prefix,
                                                              const double *__restrict__ aprime, CorrParams cp,
                                                              float *__restrict__ simOut,
                                                              float *__restrict__ boostOut)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];  // 2 x [kR][kMT]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileDesc td = tiles[blockIdx.x];
    const FileDesc fd = files[td.file];
    const long long m0 = td.o0 / kR;
    const double *fplanar = planar + fd.planar_off;
    const int Lpad = cp.S * kR;
    const int nPh = cp.numCh * cp.nb;

    double acc[kR], accT[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) {
        acc[r] = 0.0;
        accT[r] = 0.0;
    }

    // staging geometry: thread copies (row = it*rowsPerIt + tid/128, cols 2*(tid%128), +1)
    constexpr int rowsPerIt = (kWaves * 64) / (kMT / 2);
    const int srow = tid / (kMT / 2);
    const int scol = 2 * (tid % (kMT / 2));
    double2 st[kStageIters];

    auto stage_load = [&](int ph) {
        const int c = ph / cp.nb;
        const int kb = ph - c * cp.nb;
        const int s0 = kb * cp.SB;
        const int scount = min(cp.SB, cp.S - s0);
        const int used = 64 + scount;  // chunks of this block that are read
        const double *src = fplanar + ((long long)c * kR) * fd.M + m0 + s0 + scol;
#pragma unroll
        for (int it = 0; it < kStageIters; it++) {
            const int row = it * rowsPerIt + srow;
            if (scol < used) st[it] = *reinterpret_cast<const double2 *>(src + (long long)row * fd.M);
        }
    };
    auto stage_store = [&](int buf) {
        double *dst = lds + buf * (kR * kMT) + scol;
#pragma unroll
        for (int it = 0; it < kStageIters; it++) {
            const int row = it * rowsPerIt + srow;
            *reinterpret_cast<double2 *>(dst + row * kMT) = st[it];
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
        const double *a = aprime + (long long)c * Lpad + (long long)(s0 + sb) * kR;
        const double *xp = lds + (ph & 1) * (kR * kMT) + lane + sb;

        double xa[kR], xb[kR];
        load_chunk(xa, xp);
        int s = 0;
        for (; s + 2 <= n; s += 2) {
            load_chunk(xb, xp + s + 1);
            fma_step(acc, a + s * kR, xa, xb);
            load_chunk(xa, xp + s + 2);
            fma_step(acc, a + (s + 1) * kR, xb, xa);
        }
        if (s < n) {
            load_chunk(xb, xp + s + 1);
            fma_step(acc, a + s * kR, xa, xb);
        }

        if (c == 0 && kb == cp.nb - 1) {
#pragma unroll
            for (int r = 0; r < kR; r++) {
                accT[r] = acc[r];
                acc[r] = 0.0;
            }
        }
        if (ph + 1 < nPh) stage_store((ph + 1) & 1);
        __syncthreads();
    }

    // cross-wave reduction through LDS: red[part][wave][r][lane]
    double *redS = lds;
    double *redT = lds + kWaves * kR * 64;
#pragma unroll
    for (int r = 0; r < kR; r++) {
        redS[(wave * kR + r) * 64 + lane] = acc[r];
        redT[(wave * kR + r) * 64 + lane] = accT[r];
    }
    __syncthreads();

    Epilogue ep;
    const long long n1 = fd.frames + 1;
    ep.pt1 = prefix + fd.prefix_off;
    ep.pt2 = ep.pt1 + n1;
    ep.ps1 = ep.pt2 + n1;
    ep.ps2 = ep.ps1 + n1;

    // thread handles lane l = tid % 64, r = 4*(tid/64) .. +3  -> 4 consecutive windows
    const int g = tid >> 6;
    float simv[4], boostv[4];
    const long long oBase = (long long)td.o0 + (long long)lane * kR + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = 4 * g + k;
        double cs = 0.0, ct = 0.0;
#pragma unroll
        for (int wv = 0; wv < kWaves; wv++) {
            cs += redS[(wv * kR + r) * 64 + lane];
            ct += redT[(wv * kR + r) * 64 + lane];
        }
        const long long o = oBase + k;
        simv[k] = 0.0f;
        boostv[k] = 0.0f;
        if (o < fd.num_windows) window_result(cp, ep, o, ct, cs, simv[k], boostv[k]);
    }
    float *so = simOut + fd.win_off + oBase;
    float *bo = boostOut + fd.win_off + oBase;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (oBase + k < fd.num_windows) {
            so[k] = simv[k];
            bo[k] = boostv[k];
        }
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
                                                      unsigned long long *__restrict__ count)
{
    const TileDesc td = tiles[blockIdx.x];
    const FileDesc fd = files[td.file];
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
                    c.boost = boost[fd.win_off + o];
                    out[slot] = c;
                }
            }
        }
    }
}

// Picks the chunks-per-tile of prep_kernel so that its LDS image stays <= 64 KiB.
int pick_prep_cb(int numCh)
{
    for (int cb = 64; cb >= 2; cb >>= 1) {
        const size_t bytes = (size_t)numCh * (kR * (cb + 1) + 1) * sizeof(float);
        if (bytes <= 64 * 1024) return cb;
    }
    return 1;
}

int build_layout(sgz_context *ctx)
{
    const int L = ctx->cp.L;
    const int S = ctx->cp.S;
    const int nf = (int)ctx->files.size();
    ctx->h_files.assign(nf, FileDesc{});
    ctx->h_tiles.clear();
    ctx->h_ptiles.clear();
    ctx->tile_begin.assign(nf + 1, 0);
    ctx->ptile_begin.assign(nf + 1, 0);
    ctx->prep_cb = pick_prep_cb(ctx->numCh);
    long long planar = 0, prefix = 0, wins = 0;
    for (int f = 0; f < nf; f++) {
        FileDesc &fd = ctx->h_files[f];
        fd.raw = ctx->files[f].d_raw;
        fd.frames = ctx->files[f].frames;
        fd.num_windows = fd.frames >= L ? fd.frames - L + 1 : 0;
        const long long ntiles = (fd.num_windows + kTileW - 1) / kTileW;
        long long M = ntiles * 64 + S + 2;
        const long long need = (fd.frames + kR - 1) / kR;
        if (M < need) M = need;
        M = (M + 1) & ~1ll;
        fd.M = M;
        fd.planar_off = planar;
        fd.prefix_off = prefix;
        fd.win_off = wins;
        planar += (long long)ctx->numCh * kR * M;
        prefix += 4 * (fd.frames + 1);
        wins += fd.num_windows;
        // keep every file's window block 16-byte aligned for the float4-wide tail of a tile
        wins = (wins + 3) & ~3ll;
        ctx->tile_begin[f] = (int)ctx->h_tiles.size();
        for (long long t = 0; t < ntiles; t++) ctx->h_tiles.push_back(TileDesc{f, (int)(t * kTileW)});
        ctx->ptile_begin[f] = (int)ctx->h_ptiles.size();
        if (fd.num_windows > 0) {
            for (long long c0 = 0; c0 < M; c0 += ctx->prep_cb) ctx->h_ptiles.push_back(PrepTile{f, (int)c0});
        }
    }
    ctx->tile_begin[nf] = (int)ctx->h_tiles.size();
    ctx->ptile_begin[nf] = (int)ctx->h_ptiles.size();
    ctx->total_windows = wins;

    int rc;
    if ((rc = ensure(ctx, ctx->d_files, ctx->d_files_cap, (size_t)nf))) return rc;
    if ((rc = ensure(ctx, ctx->d_tiles, ctx->d_tiles_cap, ctx->h_tiles.size()))) return rc;
    if ((rc = ensure(ctx, ctx->d_ptiles, ctx->d_ptiles_cap, ctx->h_ptiles.size()))) return rc;
    if ((rc = ensure(ctx, ctx->d_planar, ctx->planar_cap, (size_t)planar))) return rc;
    if ((rc = ensure(ctx, ctx->d_prefix, ctx->prefix_cap, (size_t)prefix))) return rc;
    if ((rc = ensure(ctx, ctx->d_sim, ctx->dense_cap, (size_t)wins))) return rc;
    if ((rc = ensure(ctx, ctx->d_boost, ctx->boost_cap, (size_t)wins))) return rc;
    // the three tables are consumed by kernels queued later on ctx->stream
    if (nf) SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_files, ctx->h_files.data(), sizeof(FileDesc) * nf,
                                        hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->h_tiles.empty())
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_tiles, ctx->h_tiles.data(), sizeof(TileDesc) * ctx->h_tiles.size(),
                                    hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->h_ptiles.empty())
        SGZ_HIP(ctx, hipMemcpyAsync(ctx->d_ptiles, ctx->h_ptiles.data(), sizeof(PrepTile) * ctx->h_ptiles.size(),
                                    hipMemcpyHostToDevice, ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors may be rebuilt afterwards
    ctx->layout_dirty = false;
    return SGZ_OK;
}

}  // namespace

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
    for (auto &f : ctx->files)
        if (f.owned && f.d_raw) SGZ_HIP(ctx, hipFree(f.d_raw));
    ctx->files.clear();
    ctx->numCh = num_channels;
    ctx->layout_dirty = true;
    ctx->have_target = false;
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
    f.owned = true;
    const size_t bytes = sizeof(float) * (size_t)num_frames * ctx->numCh;
    SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&f.d_raw), bytes ? bytes : 4));
    if (bytes) SGZ_HIP(ctx, hipMemcpy(f.d_raw, frames, bytes, hipMemcpyHostToDevice));
    ctx->files.push_back(f);
    ctx->layout_dirty = true;
    return SGZ_OK;
}

int sgz_db_add_file_device(sgz_context *ctx, const float *device_frames, int64_t num_frames)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (num_frames < 0 || (num_frames > 0 && !device_frames))
        return ctx->fail(SGZ_E_ARG, "bad device_frames / num_frames");
    DbFile f;
    f.frames = num_frames;
    f.owned = false;
    f.d_raw = const_cast<float *>(device_frames);
    ctx->files.push_back(f);
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
    f.owned = true;
    const size_t elems = (size_t)num_frames * ctx->numCh;
    SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&f.d_raw), elems ? elems * sizeof(float) : 4));
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

int sgz_set_norms(sgz_context *ctx, const float *norms)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    std::memset(&ctx->norm, 0, sizeof ctx->norm);
    if (norms) {
        ctx->norm.enabled = 1;
        for (int c = 0; c < ctx->numCh; c++) {
            ctx->norm.mn[c] = norms[2 * c];
            ctx->norm.d[c] = norms[2 * c + 1] - norms[2 * c];
        }
    }
    ctx->have_target = false;  // target statistics depend on the norms
    return SGZ_OK;
}

int sgz_corr_set_target(sgz_context *ctx, const float *frames, int num_frames, float temporal_weight,
                        float max_boost)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->numCh == 0) return ctx->fail(SGZ_E_STATE, "sgz_db_reset has not been called");
    if (!frames || num_frames < 1) return ctx->fail(SGZ_E_ARG, "target needs at least one frame");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    const int numCh = ctx->numCh, L = num_frames;
    // readInBuffer (impl/FeatureCorrelationImpl.scala:83-98): de-interleave, normalise (fp32)
    std::vector<float> tb((size_t)numCh * L);
    for (int c = 0; c < numCh; c++) {
        for (int i = 0; i < L; i++) {
            float v = frames[(size_t)i * numCh + c];
            if (ctx->norm.enabled) v = (v - ctx->norm.mn[c]) / ctx->norm.d[c];
            tb[(size_t)c * L + i] = v;
        }
    }
    // MathUtil.stat over the loudness row and over the MFCC rows (two passes, fp64)
    auto stat = [&](int c0, int c1, double &mean, double &sd) {
        double sum = 0.0;
        for (int c = c0; c < c1; c++)
            for (int i = 0; i < L; i++) sum += tb[(size_t)c * L + i];
        const int sz = L * (c1 - c0);
        mean = sum / sz;
        sum = 0.0;
        for (int c = c0; c < c1; c++)
            for (int i = 0; i < L; i++) {
                const double d = tb[(size_t)c * L + i] - mean;
                sum += d * d;
            }
        sd = std::sqrt(sum / sz);
    };
    double tMean, tStd, sMean, sStd;
    stat(0, 1, tMean, tStd);
    stat(1, numCh, sMean, sStd);
    double lsum = 0.0;
    for (int i = 0; i < L; i++) lsum += tb[i];
    const float avgF = (float)(lsum / L);

    CorrParams &cp = ctx->cp;
    const int oldL = ctx->have_target ? cp.L : -1;
    cp.numCh = numCh;
    cp.L = L;
    int S = (L + kR - 1) / kR;
    S = (S + 1) & ~1;
    cp.S = S;
    cp.nb = (S + kSBMax - 1) / kSBMax;
    int SB = (S + cp.nb - 1) / cp.nb;
    SB = (SB + 1) & ~1;
    cp.SB = SB;
    cp.kT = tMean;
    cp.kS = sMean;
    cp.aStdT = tStd;
    cp.aStdS = sStd;
    cp.lnAvgIn = std::log((double)avgF);
    cp.w = temporal_weight;
    cp.maxBoost = max_boost;

    const size_t Lpad = (size_t)S * kR;
    std::vector<double> ap((size_t)numCh * Lpad, 0.0);
    double sumT = 0.0, sumS = 0.0;
    for (int c = 0; c < numCh; c++) {
        const double mean = c == 0 ? tMean : sMean;
        for (int i = 0; i < L; i++) {
            const double v = (double)tb[(size_t)c * L + i] - mean;  // ca(i) + aAdd, MathUtil.scala:190
            ap[(size_t)c * Lpad + i] = v;
            if (c == 0) sumT += v; else sumS += v;
        }
    }
    cp.sumAT = sumT;
    cp.sumAS = sumS;
    int rc;
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = ensure(ctx, ctx->d_aprime, ctx->aprime_cap, ap.size()))) return rc;
    SGZ_HIP(ctx, hipMemcpy(ctx->d_aprime, ap.data(), sizeof(double) * ap.size(), hipMemcpyHostToDevice));
    ctx->have_target = true;
    if (oldL != L) ctx->layout_dirty = true;
    return SGZ_OK;
}

int64_t sgz_corr_num_windows(const sgz_context *ctx, int file)
{
    if (!ctx || !ctx->have_target || file < 0 || file >= (int)ctx->files.size()) return -1;
    const long long n = ctx->files[file].frames;
    return n >= ctx->cp.L ? n - ctx->cp.L + 1 : 0;
}

int sgz_corr_run(sgz_context *ctx, int file_begin, int file_end)
{
    if (!ctx) return SGZ_E_ARG;
    if (!ctx->have_target) return ctx->fail(SGZ_E_STATE, "sgz_corr_set_target has not been called");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (ctx->layout_dirty && (rc = build_layout(ctx))) return rc;
    if (!ctx->ev0) {
        SGZ_HIP(ctx, hipEventCreate(&ctx->ev0));
        SGZ_HIP(ctx, hipEventCreate(&ctx->ev1));
    }
    ctx->ev_valid = false;
    ctx->last_windows = 0;
    ctx->last_flops = 0.0;
    if (file_begin == file_end) return SGZ_OK;

    const int pt0 = ctx->ptile_begin[file_begin], pt1 = ctx->ptile_begin[file_end];
    const int t0 = ctx->tile_begin[file_begin], t1 = ctx->tile_begin[file_end];
    long long wins = 0;
    for (int f = file_begin; f < file_end; f++) wins += ctx->h_files[f].num_windows;
    ctx->last_windows = wins;
    ctx->last_flops = (double)wins * 2.0 * ctx->cp.numCh * ctx->cp.L;
    if (t1 == t0) return SGZ_OK;

    const int cb = ctx->prep_cb;
    const size_t prepLds = (size_t)ctx->numCh * (kR * (cb + 1) + 1) * sizeof(float);
    hipLaunchKernelGGL(prep_kernel, dim3(pt1 - pt0), dim3(256), prepLds, ctx->stream, ctx->d_files,
                       ctx->d_ptiles + pt0, ctx->d_planar, ctx->d_prefix, ctx->norm, ctx->numCh, cb, ctx->cp.kT,
                       ctx->cp.kS);
    SGZ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(scan_kernel, dim3(4, file_end - file_begin), dim3(1024), 0, ctx->stream, ctx->d_files,
                       file_begin, ctx->d_prefix);
    SGZ_HIP(ctx, hipGetLastError());
    const size_t corrLds = 2 * (size_t)kR * kMT * sizeof(double);
    static_assert(2 * kR * kMT >= 2 * kWaves * kR * 64, "reduction scratch must fit the staging buffers");
    SGZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(corr_kernel, dim3(t1 - t0), dim3(kWaves * 64), corrLds, ctx->stream, ctx->d_files,
                       ctx->d_tiles + t0, ctx->d_planar, ctx->d_prefix, ctx->d_aprime, ctx->cp, ctx->d_sim,
                       ctx->d_boost);
    SGZ_HIP(ctx, hipGetLastError());
    SGZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return SGZ_OK;
}

int sgz_corr_read_dense(sgz_context *ctx, int file, float *sim_out, float *boost_out)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    if (file < 0 || file >= (int)ctx->files.size()) return ctx->fail(SGZ_E_ARG, "bad file index");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    const FileDesc &fd = ctx->h_files[file];
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t bytes = sizeof(float) * (size_t)fd.num_windows;
    if (bytes && sim_out) SGZ_HIP(ctx, hipMemcpy(sim_out, ctx->d_sim + fd.win_off, bytes, hipMemcpyDeviceToHost));
    if (bytes && boost_out)
        SGZ_HIP(ctx, hipMemcpy(boost_out, ctx->d_boost + fd.win_off, bytes, hipMemcpyDeviceToHost));
    return SGZ_OK;
}

int sgz_corr_compact_device(sgz_context *ctx, int file_begin, int file_end, float threshold,
                            sgz_candidate *device_out, int64_t capacity, unsigned long long *device_count)
{
    if (!ctx) return SGZ_E_ARG;
    if (ctx->layout_dirty || !ctx->have_target) return ctx->fail(SGZ_E_STATE, "no correlation results");
    const int nf = (int)ctx->files.size();
    if (file_begin < 0 || file_end > nf || file_begin > file_end) return ctx->fail(SGZ_E_ARG, "bad file range");
    if (!device_count || capacity < 0 || (capacity > 0 && !device_out))
        return ctx->fail(SGZ_E_ARG, "bad output buffers");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    SGZ_HIP(ctx, hipMemsetAsync(device_count, 0, sizeof(unsigned long long), ctx->stream));
    const int t0 = ctx->tile_begin[file_begin], t1 = ctx->tile_begin[file_end];
    if (t1 > t0) {
        hipLaunchKernelGGL(compact_kernel, dim3(t1 - t0), dim3(256), 0, ctx->stream, ctx->d_files,
                           ctx->d_tiles + t0, ctx->d_sim, ctx->d_boost, threshold, device_out,
                           (long long)capacity, device_count);
        SGZ_HIP(ctx, hipGetLastError());
    }
    return SGZ_OK;
}

int sgz_corr_compact(sgz_context *ctx, int file_begin, int file_end, float threshold, sgz_candidate *out,
                     int64_t capacity, int64_t *count)
{
    if (!ctx) return SGZ_E_ARG;
    if (!count || capacity < 0 || (capacity > 0 && !out)) return ctx->fail(SGZ_E_ARG, "bad output buffers");
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (!ctx->d_count) SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_count), sizeof(unsigned long long)));
    if ((rc = ensure(ctx, ctx->d_cand, ctx->cand_cap, (size_t)capacity))) return rc;
    if ((rc = sgz_corr_compact_device(ctx, file_begin, file_end, threshold, ctx->d_cand, capacity, ctx->d_count)))
        return rc;
    unsigned long long n = 0;
    SGZ_HIP(ctx, hipMemcpyAsync(&n, ctx->d_count, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = (int64_t)n;
    const size_t got = (size_t)std::min<unsigned long long>(n, (unsigned long long)capacity);
    if (got) SGZ_HIP(ctx, hipMemcpy(out, ctx->d_cand, sizeof(sgz_candidate) * got, hipMemcpyDeviceToHost));
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

int sgz_corr_last_work(const sgz_context *ctx, int64_t *windows, double *flops)
{
    if (!ctx) return SGZ_E_ARG;
    if (windows) *windows = ctx->last_windows;
    if (flops) *flops = ctx->last_flops;
    return SGZ_OK;
}

}  // extern "C"
