// Internal declarations shared by the translation units of libstrugatzki_amd.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "strugatzki_amd.h"

namespace sgz {

// ---- geometry of the correlation kernel ------------------------------------
constexpr int kR = 16;             // windows per lane == lags per step == frames per chunk
constexpr int kTileW = 64 * kR;    // windows per workgroup tile
constexpr int kWaves = 8;          // waves per workgroup; they split the lag range
constexpr int kMT = 240;           // LDS row stride (chunks) of one staged channel block
constexpr int kSBMax = kMT - 64;   // most lag steps one staged block can serve

// One database file as the kernels see it.
struct FileDesc {
    const float *raw;        // interleaved raw frames [frames][numCh]
    long long frames;
    long long M;             // chunks per planar row (padded)
    long long planar_off;    // offset in doubles into the planar buffer, layout [numCh][kR][M]
    long long prefix_off;    // offset in doubles into the prefix buffer, 4 arrays of `frames` in-tile sums
    long long tbase_off;     // offset in doubles into the tile-base buffer, 4 arrays of nptiles
    long long nptiles;       // prep tiles that hold frames of this file
    long long win_off;       // offset of this file's windows in the dense sim / boost arrays
    long long num_windows;
    // centring constants of the file's half-precision image (normalised mean of the loudness channel and of the
    // other channels over the whole file): written on the device by file_const_kernel, never read on the host
    float cT, cS;
};

struct TileDesc {
    int file;
    int o0;  // first window of the tile
};

// A tile of the matrix-core screening kernel: everything a workgroup needs of its file, so that one (scalar)
// load of this record is all that stands between the start of a workgroup and its first staging loads.
struct HalfTile {
    int file;
    int o0;                  // first window of the tile
    int valid;               // windows of the file in this tile (>= 1)
    int pad;
    long long planar;        // offset (elements) of the tile's first frame in channel 0 of the half-precision image
    long long rowLen;        // elements per channel row of the file's image
    long long win;           // index of the tile's first window in the dense per-window arrays
};

struct PrepTile {
    int file;
    int chunk0;  // first chunk of the tile
};

struct NormParams {
    int enabled;
    float mn[SGZ_MAX_CHANNELS];
    float d[SGZ_MAX_CHANNELS];  // max - min, fp32 as MathUtil.normalize computes it
};

struct CorrParams {
    int numCh;
    int L;        // target frames
    int S;        // lag steps (even), S * kR >= L
    int nb;       // staged blocks per channel
    int SB;       // steps per staged block (even, <= kSBMax)
    int prepTileShift;  // log2(frames per prep tile)
    double kT, kS;            // centring constants of the planar data
    double aStdT, aStdS;      // target std-devs
    double sumAT, sumAS;      // sum of the centred target (rounding residue)
    double lnAvgIn;           // ln(avg loudness of the target)
    float w;                  // temporal weight
    float maxBoost;
};

// ---- geometry of the half-precision matrix-core screening kernel ----------------
constexpr int kHNB = 8;                 // 16x16 output tiles per wave (columns of 16 chunks each): the default
constexpr int kHWaves = 8;              // waves per workgroup; wave q takes the 32-lag steps s = q, q + 8, ...

struct HalfGeom {
    int NB;    // column tiles per wave of the variant in use: a workgroup tile is 256 * NB windows
    int S2;    // 32-lag steps: 32 * S2 >= L + 15
    int LAd;   // dwords per shifted copy of one target channel (LAd % 32 == 16)
    int XU;    // 16-byte units of one staged channel block of database frames
    double sumAT, sumAS;  // sums of the half-precision target values
    // relative error of a cross term formed from half-precision operands with fp32 sums (loudness channel, other
    // channels): see half_rel_err
    float relErrT, relErrS;
};

// ---- spectral (FFT) screening: the cross terms of N - L + 1 consecutive windows from one N-point transform ----
struct FftGeom {
    int N;      // transform length (0: spectral screening is not in use for this layout)
    int hop;    // windows per block: N - L + 1
    int q16;    // the database spectra are kept as 16-bit integers with one scale per 64 bins (else fp32)
    // What the kernel needs to bound the error of a block's cross terms (it combines them with the block's norms):
    // epsT: norm-wise relative error of one transform (forward or inverse); epsUT / epsUS: the roundings outside the
    // transforms (operands, products, channel sums); Euclidean norms of the fp32 target values -- loudness channel,
    // other channels, channels 0 and 1 together (they share a packed transform), sqrt(|a_T|^2 + 2 |a_S|^2) (an upper
    // bound of the norm over all packed pairs an other channel sits in) -- each rounded up; kT / kS = 1 / (aStd n).
    float epsT, epsUT, epsUS;
    // normalised values are formed as (x - min) * (1 / range) in fp32, rounded once at the size of the value itself,
    // which for a file far from zero is its level, not its spread: levelErr * |centring constant| / sd per window
    // (0 without normalisation, where the centred value is one exact-or-correctly-rounded subtraction)
    float levelErr;
    float aT, aS, a01, aP;
    float kT, kS;
    float relErr;          // fp32 rounding of the target values, relative to |a'| |x_window|
    double sumAT, sumAS;   // sums of the fp32 target values
};

// Where the spectral kernel lists the windows that can hold their file's maximum, when it runs for a search over
// bounds that needs nothing else (sgz_corr_set_screen_output): records, their count (which may pass the capacity:
// then the list is incomplete and the search takes another path), the capacity.  list == nullptr: dense output.
struct PreList {
    sgz_candidate *list;
    unsigned long long *count;
    long long capacity;
};

// One block of the spectral screening kernel: N frames of one file from frame o0 on, `valid` windows.
struct FftBlock {
    int file;
    int o0;
    int valid;
    int pad;
    long long spec;  // offset (complex values: 8 bytes as fp32, 4 as 16-bit integers) of the block's spectra [numCh][N / 2]
    long long win;   // index of the block's first window in the dense per-window arrays
};

struct DbFile {
    float *d_raw = nullptr;
    long long frames = 0;
    bool owned = false;
};

}  // namespace sgz

struct sgz_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    mutable std::string err;

    int numCh = 0;
    std::vector<sgz::DbFile> files;
    std::string db_tag;                      // the caller's label of the loaded database (sgz_db_set_tag)
    unsigned long long db_serial = 0;        // bumped by sgz_db_reset and every sgz_db_add_*: the files changed
    int multi_exchange = 0;                  // sgz_multi_set_exchange: 0 merge in host memory, 1 RCCL all-gather
    int multi_exchange_used = 0;             // what the latest sgz_corr_search_multi used
    void *multi_comm = nullptr;              // RCCL communicators and buffers of the contexts it led (sgz_multi.hip)
    void (*multi_comm_free)(void *) = nullptr;
    std::string multi_refused_key, multi_refused_why;  // the set of contexts RCCL last refused, and its reason
    std::string multi_ring_key;              // sgz_corr_search_multi (on its first context): what the ring images of
                                             // the short files were last worked out for
    sgz::NormParams norm{};
    bool norms_set = false;                  // sgz_set_norms has been called (else: no normalisation)

    // target
    bool have_target = false;
    sgz::CorrParams cp{};
    // the matrix-core kernel's target image is built when that kernel first needs it, from the fp64 target fetched
    // back into this pinned staging area
    double *half_stage_ap = nullptr;
    size_t half_stage_lpad = 0;
    uint16_t *half_stage_img = nullptr;
    size_t half_stage_halfwords = 0;
    bool half_img_dirty = false;
    hipEvent_t tgt_event = nullptr;        // the target's upload on `stream`, for the other stream that reads it
    // database files of up to 8 MB are carved from slabs of 32 MB (a hipMalloc per file costs 70 us: a third of
    // what loading a five-minute feature file takes); released by sgz_db_reset and sgz_destroy
    std::vector<void *> db_slabs;
    char *db_slab_cur = nullptr;
    size_t db_slab_left = 0;
    void *d_pcm_stage = nullptr;           // device: a file's samples as the file holds them, before sgz_db_add_file_pcm decodes them
    size_t pcm_stage_cap = 0;
    unsigned char *h_tgt_stage = nullptr;  // pinned: what sgz_corr_set_target uploads, written in place and copied
    size_t tgt_stage_cap = 0;              // asynchronously (the stream is idle when the next target overwrites it)
    double *d_aprime = nullptr;  // [numCh][S * kR] centred target, fp64
    size_t aprime_cap = 0;
    float *d_tnorm = nullptr;     // [L][numCh] normalised target, fp32 (exact re-scoring)
    size_t tnorm_cap = 0;
    double aMeanT = 0.0, aMeanS = 0.0;
    unsigned long long tgt_seq = 0;          // sequence number of the latest report of the target preparation
    double *d_tgt_state = nullptr;           // the partial sums its first pass leaves on the device (target_norm_kernel)
    size_t tgt_state_cap = 0;
    unsigned long long tgt_serial = 0;       // targets set so far
    float *d_aprime_f = nullptr;  // the same, rounded to fp32 (screening path)
    size_t aprime_f_cap = 0;
    uint16_t *d_aimg_h = nullptr; // half-precision target image [numCh][2 shifts][2 * LAd] (matrix-core screening)
    size_t aimg_h_cap = 0;
    sgz::HalfGeom hg{};

    // layout (depends on db + target length)
    bool layout_dirty = true;
    long long tail_trim = 0;                 // frames at the end of every file that the passes leave out
    std::vector<sgz::FileDesc> h_files;
    std::vector<sgz::TileDesc> h_tiles;      // correlation tiles, grouped by file
    std::vector<int> tile_begin;             // first tile of each file (+ sentinel)
    std::vector<sgz::PrepTile> h_ptiles;
    std::vector<int> ptile_begin;
    std::vector<float *> stale_images;       // L-frame images of files shorter than the target (see build_layout)
    std::map<int, std::vector<float>> ring_images;  // caller-supplied ring contents behind a short file's own frames
    std::vector<sgz::HalfTile> h_tiles_h;    // tiles of the matrix-core screening kernel
    std::vector<int> tile_h_begin;
    sgz::HalfTile *d_tiles_h = nullptr;
    size_t d_tiles_h_cap = 0;
    int prep_cb = 64;                        // chunks per prep tile
    sgz::FileDesc *d_files = nullptr;
    sgz::TileDesc *d_tiles = nullptr;
    sgz::PrepTile *d_ptiles = nullptr;
    size_t d_files_cap = 0, d_tiles_cap = 0, d_ptiles_cap = 0;
    long long total_windows = 0;

    double *d_planar = nullptr;
    size_t planar_cap = 0;
    float *d_planar_f = nullptr;             // planar image of the screening path
    size_t planar_f_cap = 0;
    uint16_t *d_planar_h = nullptr;          // half-precision image [numCh][16 * M], frames in natural order
    size_t planar_h_cap = 0;
    int *d_nonfinite = nullptr;              // per file: != 0 = the file holds a value that is not a finite number
    size_t nonfinite_cap = 0;                // once normalised (scanned when the layout is built; host copy beside it)
    std::vector<int> h_nonfinite;
    int *d_hflags = nullptr;                 // per file: 1 = a value left the half-precision range
    size_t hflags_cap = 0;
    double *d_prefix = nullptr;
    size_t prefix_cap = 0;
    double *d_tbase = nullptr;
    size_t tbase_cap = 0;
    float *d_sim = nullptr;
    float *d_boost = nullptr;
    size_t dense_cap = 0, boost_cap = 0;
    int lo_begin = 0, lo_end = 0;            // files whose d_sim holds upper bounds of the latest screening run over bounds
    // per file: largest lower bound of a half-precision run, as an ordered integer key.  Two arrays of one entry
    // per file: a run folds its bounds into one of them and resets the other for the run after it.
    int *d_lokey = nullptr;
    size_t lokey_cap = 0;
    int lokey_parity = 0;                    // array the next run writes
    bool lokey_clean[2] = {false, false};    // that array holds the key of -inf everywhere
    int *lokey_cur = nullptr;                // array of the latest run
    // The half-precision preparation (image, prefix records, tile bases, range flags, centring constants) depends on
    // database, norms and target LENGTH only: kept for files [hc_begin, hc_end) until the layout is rebuilt.
    float *d_prefix_h = nullptr;             // per frame: in-tile inclusive sums of (t, t^2, sum s, sum s^2), fp32
    size_t prefix_h_cap = 0;
    double *d_tbase_h = nullptr;             // tile bases of those sums
    size_t tbase_h_cap = 0;
    float *d_wstat_h = nullptr;              // per window: the target-independent part of its bounds (8 floats,
    size_t wstat_h_cap = 0;                  // sgz::WindowStatic), built with the image and kept as long as it
    int hc_begin = 0, hc_end = 0;
    // Spectral screening (sgz_corr_fft.hip): per block of N frames the spectra of all channels, built with the
    // half-precision preparation and kept as long as it; the target's spectra are rebuilt per target.
    int screen_kernel = 0;                   // sgz_corr_set_screen_kernel: 0 auto, 1 matrix cores, 2 spectral
    sgz::FftGeom fg{};
    std::vector<sgz::FftBlock> h_fblocks;
    std::vector<int> fblock_begin;           // first block of each file (+ sentinel)
    sgz::FftBlock *d_fblocks = nullptr;
    size_t d_fblocks_cap = 0;
    float2 *d_spec = nullptr;                // [block][numCh][N / 2] complex, bin 0 holds (DC, Nyquist)
    size_t spec_cap = 0;
    float2 *d_aspec = nullptr;               // conjugated target spectra / N, same packing
    size_t aspec_cap = 0;
    float2 *d_fft_tab = nullptr;             // (cos, sin)(2 pi j / kFftTab)
    float *d_blksq = nullptr;                // [block][numCh] sum of squares of the block's centred values
    size_t blksq_cap = 0;
    float4 *d_blknorm = nullptr;             // per block, Euclidean norms: loudness channel, the others, channels 0 and 1,
                                             // sqrt(|x_T|^2 + 2 |x_S|^2)
    size_t blknorm_cap = 0;
    int spectra_format = 0;                  // sgz_corr_set_spectra_format: 0 auto, 1 fp32, 2 16-bit integers
    float *d_specscale = nullptr;            // 16-bit spectra: [block][numCh][N / 128] the value of one integer step
    size_t specscale_cap = 0;
    float *d_blkdq = nullptr;                // 16-bit spectra: [block][numCh] sum of squares of the rounding to 16 bits
    size_t blkdq_cap = 0;
    float *d_aspecscale = nullptr;           // 16-bit spectra: the target's scales [numCh][N / 128]
    size_t aspecscale_cap = 0;
    float2 *d_tgt_z = nullptr;               // a new target's transforms before they are encoded: [pair][N]
    size_t tgt_z_cap = 0;
    float *d_tgtdq = nullptr;                // 16-bit spectra: [numCh][4] sums of squares of the target spectra's rounding, in four parts
    size_t tgtdq_cap = 0;
    float2 *d_blkq = nullptr;                // per block: that rounding as a norm of time-domain values, loudness / others
    size_t blkq_cap = 0;
    bool aspec_dirty = true;
    int fc_begin = 0, fc_end = 0;            // files whose spectra are current
    float *d_blklnmin = nullptr;             // per block: smallest ln(average loudness) of its windows
    size_t blklnmin_cap = 0;
    float4 *d_blkgu = nullptr;               // per block: largest g and u of both channel groups over its windows ...
    size_t blkgu_cap = 0;
    float2 *d_wrec2 = nullptr;               // ... and per window 1 / sd of both groups (fft_blkgu_kernel)
    size_t wrec2_cap = 0;
    // The spectral kernel does not write the boosts of its windows (4 of the bytes it would move per window; a
    // search needs the boost of the few windows it re-scores only).  boost_lazy[f] != 0: d_boost of file f is not
    // filled in; whoever reads it (dense reads, compaction over a mix of files) has boost_materialize fill it first
    // from the statics with the target value the run used (boost_lazy_ln).
    std::vector<char> boost_lazy;
    double boost_lazy_ln = 0.0;
    // sgz_corr_set_screen_output(1): a spectral run writes candidate lists instead of dense upper bounds.
    // no_dense[f] != 0: the latest run over file f was of that kind (dense reads and threshold compaction refuse);
    // the list of the latest such run covers files [pre_fb, pre_fe) and sits behind counter pre_parity ^ 1.
    int screen_output = 0;
    std::vector<char> no_dense;
    sgz_candidate *d_prelist = nullptr;
    size_t prelist_cap = 0;
    unsigned long long *d_precount = nullptr;  // two counters, used in turn (the consumer of one clears the other)
    size_t precount_cap = 0;
    int pre_parity = 0, pre_run_parity = 0;  // counter the next run takes; counter the latest run used
    bool pre_clean[2] = {false, false};
    bool pre_valid = false;
    int pre_fb = 0, pre_fe = 0;
    long long pre_capacity = 0;
    double tgt_sumT = 0.0, tgt_sumS = 0.0;   // sums and Euclidean norms of the fp32 target values
    double tgt_nrmT = 0.0, tgt_nrmS = 0.0, tgt_nrm1 = 0.0;  // (nrm1: channel 1 alone)
    float *d_thr = nullptr;                  // smallest of the head files' bounds (filter of the tail files)
    size_t thr_cap = 0;
    sgz_candidate *h_pinned = nullptr;       // pinned staging of the first candidate transfer
    float *d_normtab = nullptr;              // norms as a device table: min[64], then max - min[64]
    size_t normtab_cap = 0;
    double *d_rs_part = nullptr;             // partial sums of the exact re-scoring [window][slice][6]
    size_t rs_part_cap = 0;
    void *noalloc_scratch = nullptr;         // this device's copy of the file-scope scratch array (sgz_corr.hip)

    // sgz_selfsim_rows: the similarity matrix of the latest call, kept on the device while bands of it are fetched
    float *d_ss_sim = nullptr;
    unsigned long long ss_key = 0;           // digest of the arguments it was computed for (0: none)

    sgz_candidate *d_cand = nullptr;         // [two count headers][records]
    sgz_candidate *d_suspect = nullptr;      // [count header][windows an exact run hands to the two-pass kernels]
    size_t suspect_cap = 0;
    unsigned int *d_done = nullptr;          // workgroup counter of the host report (rescore_final_kernel)
    size_t done_cap = 0;
    bool done_init = false;
    unsigned long long report_seq = 0;       // sequence number of the latest host report
    size_t report_prev_n = 0;                // records the latest host report filled in the pinned buffer
    int cand_parity = 0;                     // which header the next compaction counts in
    bool cand_clean = false;                 // that header is known to be zero
    size_t cand_cap = 0;
    std::vector<sgz_candidate> h_cand;

    // completion markers of the sgz_corr_run calls whose results are still current
    struct Run {
        int fb, fe;
        hipEvent_t done;
        hipStream_t stream;  // where it ran: later work on the same stream needs no wait
        bool recorded;       // `done` is recorded only when work on another stream first asks for it
    };
    int wait_for_run(Run &run, hipStream_t consumer);  // makes `consumer` wait for the run (no-op on its stream)
    std::vector<Run> runs;
    std::vector<hipEvent_t> event_pool;
    hipStream_t copy_stream = nullptr;       // device-to-host result copies
    float *h_dense_pin = nullptr;            // pinned staging of dense read-backs (two halves: sims, boosts;
    size_t dense_pin_floats = 0;             // floats per half), followed by the sequence word of a prefetch
    // sgz_corr_prefetch_dense: a copy by kernel, in order on `stream`, that the next read of the same files takes
    bool prefetch_valid = false;
    int prefetch_fb = 0, prefetch_fe = 0;
    unsigned long long prefetch_seq = 0;
    unsigned int *d_prefetch_done = nullptr;
    size_t prefetch_done_cap = 0;
    struct PinSlot {                         // staging of large copies from / to pageable memory (copy_large)
        void *pin = nullptr;                 // two halves: one is filled or emptied by the host while the
        hipStream_t stream = nullptr;        // other is in flight
        hipEvent_t done[2] = {nullptr, nullptr};
    };
    std::vector<PinSlot> pin_pool;
    hipStream_t screen_stream = nullptr;     // screening runs (overlap with exact runs on `stream`)
    bool screen_stream_used = false;         // a run has been queued there (until then nobody has to wait for it)
    std::vector<const void *> lds_attr_set;  // kernels whose dynamic LDS limit has been raised (set_lds)

    // Kernel timing: a ring of event pairs, one pair per run (ev0 / ev1 = the pair of the latest run), so that
    // a caller can read the times of many runs after the fact instead of waiting on events between them.
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_ring;
    size_t ev_pos = 0, ev_runs = 0;          // ring slot of the latest run; runs recorded since creation
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    int timing_every = 1;                    // 0: runs record no events; n: every n-th run does
    unsigned long long timing_count = 0;
    int next_event_pair();                   // advances the ring (created on first use); SGZ_* code
    long long last_windows = 0;
    double last_flops = 0.0;

    int fail(int code, const std::string &msg) const
    {
        err = msg;
        return code;
    }
};

#define SGZ_HIP(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return (ctx)->fail(SGZ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace sgz {
// sgz_corr_mfma.hip
bool half_screen_fits(const sgz_context *ctx);
int half_screen_pick_tile(sgz_context *ctx, const long long *numWindows, int numFiles);
size_t half_image_halfwords(int L, int numCh);
int half_screen_set_target(sgz_context *ctx, bool in_half_range);
int half_screen_launch(sgz_context *ctx, int tile_begin, int tile_end, hipStream_t stream, hipEvent_t start,
                       hipEvent_t stop);

// sgz_corr_fft.hip
int fft_screen_layout(sgz_context *ctx);  // after the file table of a layout is built: block table, buffers
int fft_screen_prepare(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream);
int fft_screen_launch(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream, hipEvent_t start,
                      hipEvent_t stop);
void fft_screen_free(sgz_context *ctx);
// A new target for the layout in force: its spectra can be queued before the host knows the target's statistics
// (fft_target_spectra); once it does, fft_target_ready derives the error-bound factors and marks the spectra current.
// centre_args (a sgz::TargetCentreArgs, or null): the launch carries the second pass of the target preparation along,
// in fft_target_centre_workgroups() further workgroups, and forms its own input from the normalised target.
int fft_target_spectra(sgz_context *ctx, hipStream_t stream, const void *centre_args);
int fft_target_centre_workgroups(const sgz_context *ctx, int numCh, int Lpad);
void fft_target_ready(sgz_context *ctx);
// Fills in the boosts the spectral kernel left out for files [file_begin, file_end) (kernels on `stream`);
// *launched tells whether there was anything to do.
int boost_materialize(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream, bool *launched = nullptr);
// The statics array the compaction kernels compute boosts from when every file of the range is of that kind (else
// null, after filling in what was missing).
int boost_source(sgz_context *ctx, int file_begin, int file_end, hipStream_t stream, const float **lnAvg);
// Fails with SGZ_E_STATE when the latest screening run over one of the files kept candidate lists only.
int require_dense(sgz_context *ctx, int file_begin, int file_end);

// sgz_corr.hip: address of the device's scratch array of sgz_corr_rescore_device_noalloc (device must be current)
int noalloc_scratch_init(sgz_context *ctx);

// sgz_select.cpp: sgz_corr_ring_runs as a vector
std::vector<sgz_ring_run> ring_runs(const long long *frames, int num_files, int L);

// sgz_context.hip: synchronous copy between pageable host memory and the device, staged in parallel when large
int copy_large(sgz_context *ctx, void *dst, const void *src, size_t bytes, bool to_host);
int pcm_format(int sample_format, int *sample_bytes, int *big_endian, int *is_float);
int pcm_decode_launch(sgz_context *ctx, const void *raw, long long count, int sample_format, float *out, hipStream_t stream);

// Grows a device buffer to at least `count` elements (contents are not kept).
template <typename T>
int ensure(const sgz_context *ctx, T *&ptr, size_t &cap, size_t count)
{
    if (count <= cap && ptr) return SGZ_OK;
    if (ptr) {
        SGZ_HIP(ctx, hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
    const size_t n = count ? count : 1;
    SGZ_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
    cap = n;
    return SGZ_OK;
}
}  // namespace sgz
