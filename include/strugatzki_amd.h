/*
 * strugatzki_amd.h -- C ABI of the MI355X (gfx950) implementation of
 * Strugatzki's similarity hot path.
 *
 * This is the surface a JNI shim (or any other FFI) binds.  The reference has
 * no native boundary of its own: each entry point below names the lines of the
 * reference's Scala implementation whose work it takes over.  Paths are
 * relative to src/main/scala/de/sciss/strugatzki/ of the reference.
 *
 * Conventions
 *   - every function returns SGZ_OK (0) or a negative SGZ_E_* code; the text of
 *     the last failure of a context is available from sgz_last_error();
 *   - feature frames are "interleaved": frame-major float32 [numFrames][numCh]
 *     with numCh = numCoeffs + 1 and channel 0 = loudness, exactly the sample
 *     order of a feature AIFF (impl/NonRealtimeProcessor.scala:164-165,
 *     impl/FeatureExtractionImpl.scala:52-53);
 *   - the caller owns every host buffer passed in or out; the library owns all
 *     device memory behind the opaque context; a context is bound to one GPU
 *     and one HIP stream and must not be used from two threads at once;
 *   - nothing here falls back to the CPU: without a usable GPU every compute
 *     entry point fails with SGZ_E_HIP.
 */
#ifndef STRUGATZKI_AMD_H
#define STRUGATZKI_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGZ_API __attribute__((visibility("default")))

#define SGZ_OK 0
#define SGZ_E_ARG (-1)      /* illegal argument (IllegalArgumentException on the JVM side) */
#define SGZ_E_HIP (-2)      /* HIP runtime failure / no device */
#define SGZ_E_STATE (-3)    /* call order violated (e.g. correlate before set_target) */
#define SGZ_E_OVERFLOW (-4) /* an output buffer was too small; counts are still reported */
#define SGZ_E_IO (-5)       /* file could not be read / written / parsed (IOException) */

#define SGZ_MAX_CHANNELS 64

typedef struct sgz_context sgz_context;

/* One database window that passed the compaction threshold. */
typedef struct sgz_candidate {
    int32_t file;    /* index of the database file (order of sgz_db_add_*) */
    int32_t offset;  /* window start in feature frames = logicalOff of the reference loop */
    float sim;
    float boost;
} sgz_candidate;

/* Mirrors FeatureCorrelation.Match (FeatureCorrelation.scala:54) with the file
 * given as database index. */
typedef struct sgz_match {
    float sim;
    float boost_in;
    float boost_out;
    int32_t file;
    int64_t start; /* punch.start, sample frames */
    int64_t stop;  /* punch.stop */
} sgz_match;

/* Mirrors FeatureSegmentation.Break (FeatureSegmentation.scala:47). */
typedef struct sgz_break {
    float sim;
    int32_t pad_;
    int64_t pos; /* sample frames */
} sgz_break;

/* Region of a synthetic database file that is a noisy copy of another file. */
typedef struct sgz_plant {
    int64_t file;
    int64_t offset;
    int64_t src_file;
    int64_t src_offset;
    int64_t length;
    float amp;
    float gain;
} sgz_plant;

/* ---- life cycle --------------------------------------------------------- */

/* Creates a context on GPU `device`.  `hip_stream` is a hipStream_t to run on
 * (NULL: the context creates its own).  No counterpart in the reference; it
 * stands for the per-processor state a `new impl.FeatureCorrelationImpl(config)`
 * holds (FeatureCorrelation.scala:81). */
SGZ_API int sgz_create(int device, void *hip_stream, sgz_context **out);
SGZ_API void sgz_destroy(sgz_context *ctx);
SGZ_API const char *sgz_last_error(const sgz_context *ctx);
SGZ_API const char *sgz_version(void);
/* The HIP stream (hipStream_t) the context queues its kernels on, for callers that want to order their own
 * device work (copies, collectives) after the library's without a host synchronisation. */
SGZ_API void *sgz_get_stream(sgz_context *ctx);
/* Blocks until everything queued on the context's stream has finished. */
SGZ_API int sgz_synchronize(sgz_context *ctx);

/* ---- database ----------------------------------------------------------- */

/* Drops all database files; the next files must have num_channels channels.
 * Replaces the per-query opening of every feature file,
 * impl/FeatureCorrelationImpl.scala:161-169. */
SGZ_API int sgz_db_reset(sgz_context *ctx, int num_channels);
/* Copies one feature file (raw, un-normalised) to the GPU and appends it.
 * Replaces AudioFile.openRead + afExtr.read, impl/FeatureCorrelationImpl.scala:169,195. */
SGZ_API int sgz_db_add_file(sgz_context *ctx, const float *frames, int64_t num_frames);
/* sgz_db_add_file for frames as an audio file holds them (sample_format: SGZ_PCM_*, e.g. SGZ_PCM_F32_BE for the
 * big-endian floats of the feature AIFFs the reference writes, impl/NonRealtimeProcessor.scala:164-165): the file's
 * own bytes are uploaded and decoded on the device, the host makes no pass over the samples. */
SGZ_API int sgz_db_add_file_pcm(sgz_context *ctx, const void *samples, int sample_format, int64_t num_frames);
/* Appends a feature file that already sits in device memory (not copied, not
 * freed; must stay valid until the next sgz_db_reset; 16-byte aligned). */
SGZ_API int sgz_db_add_file_device(sgz_context *ctx, const float *device_frames, int64_t num_frames);
/* Appends a synthetic feature file generated on the GPU: moving-average
 * filtered counter-based noise, optionally overlaid with noisy copies
 * (`plants`, those whose .file == file_id apply).  Benchmark / test input only;
 * no counterpart in the reference. */
SGZ_API int sgz_db_add_synthetic(sgz_context *ctx, uint64_t seed, int64_t file_id, int64_t num_frames,
                         const sgz_plant *plants, int num_plants);
SGZ_API int sgz_db_num_files(const sgz_context *ctx);
SGZ_API int64_t sgz_db_file_frames(const sgz_context *ctx, int file);
/* Copies the raw frames of a database file back to the host. */
SGZ_API int sgz_db_read_file(sgz_context *ctx, int file, float *frames_out);

/* A label the caller attaches to the database it has loaded into the context (for instance the list of files
 * with their sizes and modification times), so that a later call can tell whether the resident files are the
 * ones it needs.  sgz_db_reset and every sgz_db_add_* clear it.  sgz_db_tag returns "" when there is none. */
SGZ_API int sgz_db_set_tag(sgz_context *ctx, const char *tag);
SGZ_API const char *sgz_db_tag(const sgz_context *ctx);
/* Copies `count` raw frames of a database file, starting at frame `first`, back to the host. */
SGZ_API int sgz_db_read_frames(sgz_context *ctx, int file, int64_t first, int64_t count, float *frames_out);

/* Per-channel (min, max) pairs, norms[2*ch], norms[2*ch+1]; NULL switches
 * normalisation off.  Replaces the feat_norms.aif read of
 * impl/FeatureCorrelationImpl.scala:61-71 and MathUtil.normalize
 * (impl/MathUtil.scala:132-152), which the kernels apply on the fly. */
SGZ_API int sgz_set_norms(sgz_context *ctx, const float *norms);

/* ---- FeatureCorrelation ------------------------------------------------- */

/* Sets the punch-in target: `frames` are the raw interleaved feature frames of
 * the target span (L = num_frames >= 1).  Normalises them, computes the
 * temporal / spectral mean and std-dev and ln(avg loudness)
 * (readInBuffer, impl/FeatureCorrelationImpl.scala:83-98) -- on the device: the host
 * hands over the raw frames only.  The fp64 sums of MathUtil.stat
 * (impl/MathUtil.scala:29-62) are taken in a fixed tree order, not sequentially: the
 * statistics agree with the reference's to a few units of the last place of a double. */
SGZ_API int sgz_corr_set_target(sgz_context *ctx, const float *frames, int num_frames,
                        float temporal_weight, float max_boost);
/* The same for raw frames that are in the memory of the context's GPU already -- what sgz_extract leaves there
 * for a query whose audio was analysed on the device, or a set of targets a service keeps resident: the first
 * kernel reads them in place, nothing crosses the bus but the statistics on their way back.  (The reference reads
 * the target's frames from the input's feature file, impl/FeatureCorrelationImpl.scala:73-98; where they come from
 * does not change what is computed.)  The buffer must stay valid until the call returns. */
SGZ_API int sgz_corr_set_target_device(sgz_context *ctx, const float *device_frames, int num_frames,
                                       float temporal_weight, float max_boost);
/* The statistics of the current target as readInBuffer keeps them
 * (impl/FeatureCorrelationImpl.scala:90-97): stats_out[0..4] = mean and std-dev of the
 * loudness row, mean and std-dev of the other rows, ln(average loudness). */
SGZ_API int sgz_corr_target_stats(const sgz_context *ctx, double *stats_out);

/* Number of sliding windows the reference loop visits for database file `file`
 * with the current target (impl/FeatureCorrelationImpl.scala:182-190,242-246):
 * frames - L + 1.  A non-empty file shorter than the target has 1 window, scored
 * as the reference does over the file's frames followed by what earlier files of
 * this context left in the L-frame ring (:154, :190-195); an empty file has 0. */
SGZ_API int64_t sgz_corr_num_windows(const sgz_context *ctx, int file);

/* What the reference's one ring of L = target_frames frames (impl/FeatureCorrelationImpl.scala:154) holds when the
 * loop reaches a file with fewer than L frames: such a file overwrites the first positions only and is scored
 * once over its own frames followed by whatever earlier files left behind (:190-195).  For files of the given
 * frame counts, visited in this order, the function lists where the frames behind each short file's own come
 * from: run = `count` ring positions of file `file` starting at position `dest`, holding frames `src_frame` ..
 * of file `src_file` (-1: never written, i.e. the zeros of the fresh buffer).  Pure host arithmetic.  Returns the
 * number of runs, also when it exceeds `capacity` (then only the first `capacity` are written).  A database
 * that is spread over several contexts calls it with the frame counts of ALL files in global order and hands
 * each short file's frames to the context that holds it (sgz_corr_set_ring_image), so that every context
 * scores the file as a single loop over all files would. */
typedef struct sgz_ring_run {
    int32_t file;
    int32_t dest;
    int32_t src_file;
    int32_t count;
    int64_t src_frame;
} sgz_ring_run;
SGZ_API int64_t sgz_corr_ring_runs(const int64_t *num_frames, int num_files, int target_frames, sgz_ring_run *out,
                                   int64_t capacity);
/* Ring contents for database file `file` of this context, which has fewer frames than the target: `frames`
 * holds the raw frames of ring positions [own frames, L), num_frames = L - own frames of them.  Replaces what
 * the context would derive from its own earlier files; NULL removes it.  Forgotten by sgz_db_reset.  A run
 * whose target length does not fit fails with SGZ_E_STATE. */
SGZ_API int sgz_corr_set_ring_image(sgz_context *ctx, int file, const float *frames, int num_frames);
/* One raw frame (num_channels values) that normalises to exactly 0 under the context's norms: what a ring position
 * nobody has written yet holds. */
SGZ_API int sgz_norm_zero_frame(const sgz_context *ctx, float *frame_out);

/* The sliding passes that follow stop `frames` frames before the end of every
 * database file: what the punch-in pass of a search with punch-out does with
 * frames = minPunch (impl/FeatureCorrelationImpl.scala:182-185: it runs over
 * numFrames - minPunch frames).  Window counts, the one window of a file that
 * becomes shorter than the target and the ring contents later files see follow
 * the shortened files.  0 (the state after sgz_db_reset) = whole files. */
SGZ_API int sgz_corr_set_tail_trim(sgz_context *ctx, int64_t frames);

/* Computes boost and similarity of every window of files [file_begin, file_end)
 * on the GPU and keeps them in device memory.  Asynchronous on the context's
 * stream.  Replaces the sliding loop of impl/FeatureCorrelationImpl.scala:190-210
 * (calcBoost :75-78, correlate :414-421, MathUtil.stat/correlate/avg). */
SGZ_API int sgz_corr_run(sgz_context *ctx, int file_begin, int file_end);
/* Screening variant of sgz_corr_run: the cross terms are formed with fp32 products
 * and fp32 partial sums (folded into fp64 after every channel block); window
 * statistics, boost and the final division stay fp64.  Same outputs, similarities
 * within a few 1e-7 of sgz_corr_run's; about twice the throughput.  Meant for
 * the files of a search whose windows are afterwards filtered by a threshold and
 * re-scored exactly (sgz_corr_rescore). */
SGZ_API int sgz_corr_run_screen(sgz_context *ctx, int file_begin, int file_end);
/* Second screening variant: the cross terms (the same sums of
 * impl/MathUtil.scala:177-196) are formed on the matrix cores from half-precision
 * copies of target and database (fp32 sums); window statistics, boost and the
 * final division stay fp64.  The similarity written for a window is an UPPER
 * BOUND of sgz_corr_run's value: the half-precision result plus the largest
 * error the operand roundings can cause for that window (about 1e-3 * sqrt(1 +
 * (mean offset / std-dev)^2); observed errors are near 1e-5), so that a
 * threshold test never loses a window; boosts are fp32 evaluations (about 1e-6).  Several times
 * the throughput of sgz_corr_run_screen.  For targets whose staged channel
 * fits LDS (up to about 12 500 frames) and whose centred values stay within
 * the half-precision range (with or without normalisation: every file's image
 * is centred on the file's own mean level): sgz_corr_screen_f16_available
 * tells (1 / 0), and the run fails with SGZ_E_STATE otherwise.  A file whose
 * values leave the half-precision range makes sgz_corr_compact* over it
 * report SGZ_E_OVERFLOW. */
SGZ_API int sgz_corr_run_screen_f16(sgz_context *ctx, int file_begin, int file_end);
SGZ_API int sgz_corr_screen_f16_available(const sgz_context *ctx);
/* Which kernel produces the bounds of sgz_corr_run_screen_f16 (same sums, impl/MathUtil.scala:177-196; the
 * reference has one loop, this choice has no counterpart there):
 *   1  the matrix-core kernel described above (direct sums, half-precision products);
 *   2  the spectral kernel: the cross terms of N - L + 1 consecutive windows are one circular correlation of
 *      an N-frame block with the target, evaluated from fp32 spectra of the blocks that are kept with the
 *      database (N = 4096, 8192 or 16384; about 1.3 x the database's size in HBM).  HBM-bound instead of
 *      compute-bound, bounds about ten times tighter; needs a target of at most 3/4 of the largest N;
 *   0  (default) the spectral kernel for targets of 1024 frames and more, else the matrix cores.
 * Bounds from either kernel enclose sgz_corr_run's value, and every search result is re-scored exactly, so the
 * choice never changes a result.  Takes effect with the next run (the layout is rebuilt).  The environment
 * variable SGZ_SCREEN_KERNEL=matrix|spectral overrides the setting.
 * sgz_corr_screen_kernel_in_use: 1 or 2 once a run has built the layout, 0 if neither applies, -1 before. */
SGZ_API int sgz_corr_set_screen_kernel(sgz_context *ctx, int kernel);
SGZ_API int sgz_corr_screen_kernel_in_use(const sgz_context *ctx);
/* What a run of the spectral kernel leaves behind: 0 (default) an upper bound per window, which sgz_corr_read_dense*
 * and every compaction call can read; 1 only the list of the windows whose upper bound reaches the largest lower
 * bound of their own block -- all a search with numPerFile = 1 needs (sgz_corr_compact_top*, sgz_corr_compact_bounded*
 * with head_end covering the run's files, which then filter that list instead of scanning every window), at 4
 * bytes per window less traffic.  After such a run the other readers fail with SGZ_E_STATE until the files are run
 * again; runs of the other kernels are not affected.  sgz_corr_search sets and resets it around its own run.  No
 * counterpart in the reference (impl/FeatureCorrelationImpl.scala:234-239 keeps one running maximum per file). */
SGZ_API int sgz_corr_set_screen_output(sgz_context *ctx, int lists_only);
/* Geometry of the spectral kernel for the current layout: transform length N, windows per block (N - L + 1) and
 * the number of blocks over all files; zeros when the spectral kernel is not in use.  Any pointer may be NULL.
 * For reports (a block moves 4 * numCh * N bytes of fp32 spectra, 2 * numCh * N + numCh * N / 32 of 16-bit ones);
 * no counterpart in the reference. */
SGZ_API int sgz_corr_screen_geometry(const sgz_context *ctx, int *n_out, int *hop_out, long long *blocks_out);
/* How the spectral kernel keeps the database's spectra: 1 fp32; 2 signed 16-bit integers with one scale per band of
 * 64 bins -- half the bytes the kernel streams, and the distance of every stored block to its fp32 spectrum is
 * measured when it is built and added to the block's certified error, so the bounds stay bounds (they widen by
 * about half); 0 (default) leaves the choice to the library (16-bit).  Takes effect with the next run (the spectra
 * are rebuilt).  sgz_corr_spectra_format_in_use: 1 or 2 once a run has built the layout with the spectral kernel, 0
 * if it is not in use, -1 before.  No counterpart in the reference (which holds no spectra). */
SGZ_API int sgz_corr_set_spectra_format(sgz_context *ctx, int format);
SGZ_API int sgz_corr_spectra_format_in_use(const sgz_context *ctx);
/* The N-point complex fp32 transform of the spectral kernel (N = 4096, 8192, 16384; forward or inverse without the
 * 1 / N factor; interleaved re, im), run thread by thread on the host with the kernels' own index arithmetic and
 * operations: test support, no GPU needed, no counterpart in the reference. */
SGZ_API int sgz_fft_reference_host(int n, int inverse, const float *in, float *out);

/* Starts the transfer of sim / boost of all windows of files [file_begin, file_end) to the host, in order
 * behind the runs queued so far (a kernel writes them to pinned memory); the next sgz_corr_read_dense_range over
 * exactly these files waits for it and copies from there.  For a caller that queues more GPU work between a run
 * and the read of its results (sgz_corr_search: the screening of the other files behind the exact run over the
 * first ones) -- a plain read would be carried out behind that work.  Ranges of more than 4 Mi windows are left
 * to the plain read.  No counterpart in the reference (its results are in host memory to begin with). */
SGZ_API int sgz_corr_prefetch_dense(sgz_context *ctx, int file_begin, int file_end);
/* Copies sim / boost of all windows of one file to the host (either pointer
 * may be NULL).  Synchronises the stream. */
SGZ_API int sgz_corr_read_dense(sgz_context *ctx, int file, float *sim_out, float *boost_out);
/* Same for the consecutive files [file_begin, file_end): their windows are
 * written back to back in file order.  Waits only for the sgz_corr_run calls
 * that produced these files, not for later work on the stream. */
SGZ_API int sgz_corr_read_dense_range(sgz_context *ctx, int file_begin, int file_end, float *sim_out,
                                      float *boost_out);

/* Collects the windows of files [file_begin, file_end) whose sim is > threshold
 * (the strict comparison of impl/FeatureCorrelationImpl.scala:234) into `out`
 * (host memory, `capacity` records), sorted by (file, offset).  *count receives
 * the number found even when it exceeds capacity (then SGZ_E_OVERFLOW). */
SGZ_API int sgz_corr_compact(sgz_context *ctx, int file_begin, int file_end, float threshold,
                     sgz_candidate *out, int64_t capacity, int64_t *count);
/* Same, but writes unsorted records and the 64-bit count into caller-provided
 * device memory and does not synchronise; for callers that exchange the
 * candidates between GPUs (RCCL all-gather) before reading them. */
SGZ_API int sgz_corr_compact_device(sgz_context *ctx, int file_begin, int file_end, float threshold,
                            sgz_candidate *device_out, int64_t capacity,
                            unsigned long long *device_count);

/* Re-evaluates the listed windows exactly (fp64, the reference's two-pass
 * MathUtil.stat + MathUtil.correlate order of operations, straight from the raw
 * frames) and overwrites their sim / boost.  `device_cands` / `device_count` are
 * device memory as filled by sgz_corr_compact_device.  Asynchronous. */
SGZ_API int sgz_corr_rescore_device(sgz_context *ctx, sgz_candidate *device_cands, int64_t capacity,
                                    const unsigned long long *device_count);
/* sgz_corr_rescore_device for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): same inputs, same results, but the partial sums of
 * the re-scoring live in a fixed 4 MiB array that is part of the library's device code instead of a buffer
 * allocated on demand, and the call makes no hipMalloc / hipFree.  No counterpart in the reference.
 * Limit: capacity * slices * 56 bytes <= 4 MiB with slices = ceil(L * numCh / 4096) (at most 64) for the
 * current target of L frames -- 8 322 records for a 30 s target of 14 channels; beyond it SGZ_E_ARG, nothing
 * is launched.  All calls on one GPU (of any context) share that GPU's array: calls that overlap in time on
 * one GPU race; calls queued on one stream do not.  Needs a layout (an sgz_corr_run* since the last change
 * of database, norms or target length), as sgz_corr_rescore_device does. */
SGZ_API int sgz_corr_rescore_device_noalloc(sgz_context *ctx, sgz_candidate *device_cands, int64_t capacity,
                                            const unsigned long long *device_count);
/* sgz_corr_compact followed by that exact re-scoring: the records returned carry
 * exact similarities, selected by the (possibly screened) stored ones. */
SGZ_API int sgz_corr_compact_rescored(sgz_context *ctx, int file_begin, int file_end, float threshold,
                                      sgz_candidate *out, int64_t capacity, int64_t *count);

/* Search over the bounds of one sgz_corr_run_screen_f16 run that covers files
 * [0, max(file_end, head_end)), for numPerFile = 1.  A head file (index <
 * head_end) can only contribute its largest similarity, first occurrence (the
 * running strict maximum kept by impl/FeatureCorrelationImpl.scala:234-239 with
 * entryHasSpace :124-128 and addMatch :136-150), so only windows whose upper
 * bound reaches the file's largest lower bound are collected from it; from the
 * other files, windows whose upper bound exceeds the smallest of those head
 * bounds, below which the worst entry of a full match set cannot lie.  The
 * device variant appends to `device_out` / `*device_count` as
 * sgz_corr_compact_device does; sgz_corr_compact_bounded re-scores the
 * candidates exactly and returns them sorted by (file, offset).  Head files
 * that hold a NaN similarity or no window, and files outside the
 * half-precision range, make the count exceed the capacity (SGZ_E_OVERFLOW):
 * the caller then takes the exact path. */
SGZ_API int sgz_corr_compact_bounded_device(sgz_context *ctx, int file_begin, int file_end, int head_end,
                                            sgz_candidate *device_out, int64_t capacity,
                                            unsigned long long *device_count);
SGZ_API int sgz_corr_compact_bounded(sgz_context *ctx, int file_begin, int file_end, int head_end,
                                     sgz_candidate *out, int64_t capacity, int64_t *count);
/* sgz_corr_compact_bounded with every file of the context a head file, and one more filter: T = the
 * num_top-th largest of the files' lower bounds (-inf when the context holds fewer than num_top files).  With
 * numPerFile = 1 and num_top = numMatches the match set (impl/FeatureCorrelationImpl.scala:113-150: a sorted set
 * of numMatches similarities, the worst one dropped when a better one arrives) ends with num_top entries >= T
 * whenever num_top files reach T with distinct similarities; a file whose upper bounds all lie below T then
 * cannot be in it, and none of its windows is collected or re-scored.  *threshold_out = T; the caller feeds the
 * result through sgz_sel_feed_top, which checks that condition on the finished set.  The device variant writes
 * the count to the first 8 bytes of `device_header` (16 bytes) and T, as a float, to its bytes 8..11. */
SGZ_API int sgz_corr_compact_top_device(sgz_context *ctx, int file_begin, int file_end, int num_top,
                                        sgz_candidate *device_out, int64_t capacity,
                                        unsigned long long *device_header);
SGZ_API int sgz_corr_compact_top(sgz_context *ctx, int file_begin, int file_end, int num_top, sgz_candidate *out,
                                 int64_t capacity, int64_t *count, float *threshold_out);

/* Milliseconds the GPU spent in the last sgz_corr_run's correlation kernel
 * (HIP events on the context's stream); < 0 if none has completed. */
SGZ_API float sgz_corr_last_kernel_ms(sgz_context *ctx);
/* Same clock for whichever timed kernel ran last on the context (correlation or
 * extraction). */
SGZ_API float sgz_last_kernel_ms(sgz_context *ctx);
/* Which sgz_corr_run* calls carry a pair of timing events: every call (every = 1, the default), every n-th
 * call (every = n), none (every = 0).  The events of the matrix-core kernel travel with its dispatch, but
 * still cost a few microseconds of a 0.3 ms search.  No counterpart in the reference (instrumentation). */
SGZ_API int sgz_set_kernel_timing(sgz_context *ctx, int every);
/* The kernel times (ms) of the latest runs that record one (sgz_corr_run*, sgz_extract), oldest first: at most n,
 * at most 256; returns how many.  Lets a measuring caller read the events of many runs after its timed
 * region instead of waiting on them between runs.  No counterpart in the reference (instrumentation). */
SGZ_API int sgz_kernel_ms_history(sgz_context *ctx, float *out, int n);
/* Algorithmic work of the last sgz_corr_run: windows and floating point
 * operations (2 * numCh * L per window for the cross term). */
SGZ_API int sgz_corr_last_work(const sgz_context *ctx, int64_t *windows, double *flops);

/* ---- match selection (host) --------------------------------------------- */

/* The order-dependent match bookkeeping of impl/FeatureCorrelationImpl.scala:
 * allPrio / entryPrio / lastEntryMatch (:113-150), the offer gate (:234), the
 * per-file merge (:166-167, :399-400).  Pure host code. */
typedef struct sgz_selector sgz_selector;
SGZ_API sgz_selector *sgz_sel_create(int num_matches, int num_per_file, int64_t min_spacing,
                             int step_size, int target_frames);
SGZ_API void sgz_sel_destroy(sgz_selector *sel);
/* Feeds the windows of the next database file in iteration order.  `offsets`
 * may be NULL, meaning 0..n-1 (dense).  Windows must be in ascending offset
 * order.  Leaving out windows is legal iff they could not pass the gate. */
SGZ_API int sgz_sel_feed_file(sgz_selector *sel, int file, int64_t n, const int32_t *offsets,
                      const float *sims, const float *boosts);
/* One database file of a search with punch-out (impl/FeatureCorrelationImpl.scala:182-397).
 * in_sim / in_boost: similarity and boost of the punch-in target at every window offset of the file
 * (num_frames - in_frames + 1 values, e.g. sgz_corr_read_dense after sgz_corr_run with the punch-in
 * target); out_sim / out_boost: the same for the punch-out target (num_frames - out_frames + 1 values,
 * indexed by window offset in the file).  min_punch / max_punch in feature frames (fullToFeat of the
 * config values, :116-117).  Replays the reference's bookkeeping: the punch-in pass covers
 * num_frames - min_punch frames and only finds the first window that passes the gate (:212); every
 * punch-in window from there on whose sim exceeds lowestSim^2 is paired with the punch-out windows
 * min_punch .. max_punch frames later, sim = sqrt(inSim * outSim) (:362), match span = [in offset,
 * out offset) in sample frames, boostOut from the punch-out window.  Where the reference's two pairing
 * loops run past what its passes wrote to their temp files (their bounds count frames, not windows:
 * :322, :352), this replay stops at the last window.  The selector must have been created with
 * target_frames = in_frames. */
SGZ_API int sgz_sel_feed_file_punch(sgz_selector *sel, int file, int64_t num_frames, int in_frames, int out_frames,
                                    int min_punch, int max_punch, const float *in_sim, const float *in_boost,
                                    const float *out_sim, const float *out_boost);
/* Feeds a whole candidate list, sorted by (file, offset): one sgz_sel_feed_file per
 * file that occurs in it, in ascending file order (files without candidates
 * need no call: nothing of theirs could pass the gate). */
SGZ_API int sgz_sel_feed_candidates(sgz_selector *sel, const sgz_candidate *cands, int64_t n);
/* sgz_sel_feed_screened over the candidate blocks several GPUs exchanged (one process per GPU, every rank
 * holding its own files: the reference's single loop over all files, impl/FeatureCorrelationImpl.scala:152-246,
 * cut by file).  `blocks`: world x rows records in host memory; row 0 of a block holds that rank's record
 * count in its first 8 bytes, rows 1.. its records with rank-local file indices.  maps[map_offsets[r] + i] =
 * global index of rank r's local file i.  SGZ_E_OVERFLOW when a rank has more records than a block holds. */
SGZ_API int sgz_sel_feed_gathered(sgz_selector *sel, const sgz_candidate *blocks, int world, int64_t rows,
                                  const int32_t *maps, const int64_t *map_offsets, int head_files, int num_files);

/* Screening modes of sgz_corr_search. */
#define SGZ_SCREEN_NONE 0 /* exact fp64 kernel for every file */
#define SGZ_SCREEN_F32 1  /* sgz_corr_run_screen for the tail files */
#define SGZ_SCREEN_F16 2  /* sgz_corr_run_screen_f16 (SGZ_E_STATE where it is not available) */
#define SGZ_SCREEN_AUTO 3 /* F16 where available, else F32 */

/* One whole search over the files of the context with the current target (target_frames = its length):
 * the loop over database files of impl/FeatureCorrelationImpl.scala:152-246 and the final ordering
 * :399-411, without punch-out.  over_bounds != 0 lets numPerFile = 1 searches take the route of
 * sgz_corr_compact_bounded / sgz_sel_feed_screened (all files through the matrix-core kernel); everything
 * else, and every case that route cannot decide, goes: exact kernel + dense replay for the first
 * numMatches files (and further ones while the set is not full), screening + threshold
 * compaction + exact re-scoring for the rest, dense exact replay on
 * a candidate overflow.  matches_out receives at most min(out_capacity, num_matches) records in descending
 * similarity; file = index in the context.  `capacity` = candidate records fetched from the GPU at most.
 * `margin`: screened windows with sim > bound - margin are re-scored; negative = the default of the mode
 * (1e-5 for F32, many times its largest observed error; 1e-6 for F16, whose kernel writes upper bounds). */
SGZ_API int sgz_corr_search(sgz_context *ctx, int target_frames, int num_matches, int num_per_file,
                            int64_t min_spacing, int step_size, int screen_mode, int over_bounds, int64_t capacity,
                            float margin, sgz_match *matches_out, int out_capacity, int *count_out);

/* Several queries in flight at once on one GPU: num_queries times sgz_corr_set_target (targets_on_device = 0:
 * targets[q] in host memory) or sgz_corr_set_target_device (1: in the memory of the contexts' GPU) followed by
 * sgz_corr_search with the arguments given -- each query one FeatureCorrelation run (impl/FeatureCorrelationImpl.scala
 * :83-98 the target, :152-246 and :399-411 the search) -- dealt over num_ctx contexts that are replicas of one
 * another: the caller has given every one the same files, in the same order, and the same norms (the same context
 * twice is refused: SGZ_E_ARG).  A host thread per
 * context takes the next query as soon as its context is free, so the short kernels and host waits of one query
 * run beside the chip-filling kernel of another (a query alone leaves the GPU idle for about half of its
 * duration).  All targets have target_frames frames.  Query q's matches go to matches_out[q * out_capacity ...],
 * their count to counts_out[q]; every result is what sgz_corr_search on a single context returns.  The first failure
 * ends the batch and is returned (sgz_last_error of the contexts says more); results of the queries completed
 * until then stay valid.  The reference runs one search at a time; its counterpart is a caller that starts
 * several FeatureCorrelation processors concurrently. */
SGZ_API int sgz_corr_search_batch(sgz_context *const *ctxs, int num_ctx, const float *const *targets,
                                  int targets_on_device, int num_queries, int target_frames, float temporal_weight,
                                  float max_boost, int num_matches, int num_per_file, int64_t min_spacing,
                                  int step_size, int screen_mode, int over_bounds, int64_t capacity, float margin,
                                  sgz_match *matches_out, int out_capacity, int *counts_out);

/* sgz_corr_search over a database that is spread over several contexts, one per GPU of the node (the loop over
 * all files of impl/FeatureCorrelationImpl.scala:152-246, cut by file; one host thread drives each context).
 * Every context holds its own files, has the same norms and the same target set; global_ids[c][i] = position of
 * context c's file i in the iteration order over all files (ascending in i; all contexts together hold
 * 0 .. num_files_total - 1 once each).  Files shorter than the target are first given the ring contents the single
 * loop would show them (sgz_corr_ring_runs, frames fetched from the contexts that hold them).  numPerFile = 1 with
 * half-precision screening available: every context finds the exact maxima of its files, the host merges and
 * replays; otherwise, and whenever that route cannot decide: exact kernel everywhere, dense replay of the first
 * numMatches files (more while the set is not full), threshold compaction of the rest.  matches_out: file = global index.  The result equals that
 * of one context holding all files in global order. */
SGZ_API int sgz_corr_search_multi(sgz_context *const *ctxs, int num_ctx, const int32_t *const *global_ids,
                                  int num_files_total, int target_frames, int num_matches, int num_per_file,
                                  int64_t min_spacing, int step_size, int64_t capacity, sgz_match *matches_out,
                                  int out_capacity, int *count_out);
/* How sgz_corr_search_multi led by `lead` (its first context) exchanges the contexts' candidates for numPerFile = 1:
 * 0 (default) merged in host memory; 1: every context sends a 16 KiB block (header row: count and the context's
 * threshold; 1023 records) through one RCCL all-gather -- a communicator rank per context, ncclCommInitAll over
 * their devices, each rank's call queued on its context's stream behind the kernels -- and the selection replays
 * over the first context's copy: the exchange of a search with one process per GPU, from one process.  RCCL is
 * loaded on first use, from the ROCm installation of this library's HIP runtime.  Where it cannot serve (contexts
 * that share a GPU: RCCL wants distinct devices; no library) the merge in host memory is taken and
 * sgz_last_error(lead) says why.  The environment variable SGZ_MULTI_EXCHANGE = rccl | host overrides the setting.
 * sgz_multi_last_exchange: what the latest search used (0 / 1).  No counterpart in the reference (single GPU-less
 * process). */
SGZ_API int sgz_multi_set_exchange(sgz_context *lead, int mode);
SGZ_API int sgz_multi_last_exchange(const sgz_context *lead);

/* Feeds the exactly re-scored candidates of sgz_corr_compact_bounded (sorted by
 * (file, offset), files 0 .. num_files-1, the first head_files of them head
 * files): per head file its largest similarity, first occurrence, then the
 * rest through sgz_sel_feed_candidates.  Needs numPerFile = 1.  Returns
 * SGZ_E_STATE when the shortcut does not apply (a head file without candidate,
 * a NaN similarity, equal similarities that leave the set short of numMatches
 * before the tail): the selector is then partly fed and must be discarded, and
 * the caller takes the exact path. */
SGZ_API int sgz_sel_feed_screened(sgz_selector *sel, const sgz_candidate *cands, int64_t n, int head_files,
                                  int num_files);
/* Feeds the exactly re-scored candidates of sgz_corr_compact_top (sorted by (file, offset)): per file that
 * occurs its largest similarity, first occurrence, in file order.  Files without candidates were left out
 * because everything of theirs lies below `threshold`; that is only sound when the finished set is full and
 * its worst entry is not below `threshold` (equal similarities among the best files collapse into one entry
 * and can leave room for a lesser file), which this call checks: SGZ_E_STATE otherwise, and for a NaN
 * similarity; the selector must then be discarded and the caller takes sgz_corr_compact_bounded.
 * threshold = -inf: no file was left out, nothing to check. */
SGZ_API int sgz_sel_feed_top(sgz_selector *sel, const sgz_candidate *cands, int64_t n, float threshold);
/* sgz_sel_feed_top over the blocks of a candidate exchange (layout as for sgz_sel_feed_gathered; bytes 8..11
 * of a block's header row hold that rank's threshold, as sgz_corr_compact_top_device leaves it: a rank leaves
 * out files below the num_top-th best of its own files, and the check uses the largest of the thresholds). */
SGZ_API int sgz_sel_feed_gathered_top(sgz_selector *sel, const sgz_candidate *blocks, int world, int64_t rows,
                                      const int32_t *maps, const int64_t *map_offsets);
/* 1 once allPrio holds num_matches entries: from then on only windows with
 * sim > sgz_sel_threshold() can still be offered. */
SGZ_API int sgz_sel_is_full(const sgz_selector *sel);
SGZ_API float sgz_sel_threshold(const sgz_selector *sel);
/* Writes the matches in descending sim order; returns their number. */
SGZ_API int sgz_sel_result(const sgz_selector *sel, sgz_match *out, int capacity);

/* ---- FeatureSegmentation ------------------------------------------------ */

/* Novelty curve of one feature file: for every window start o in
 * [af_start, af_stop - 2*half_win] the correlateHalf similarity of
 * impl/FeatureSegmentationImpl.scala:107-120 (MathUtil.correlateHalf,
 * impl/MathUtil.scala:80-99).  `frames` is host memory (raw interleaved,
 * num_frames x num_channels); `norms` is [num_channels][2] (min, max) or NULL
 * for no normalisation; sim_out receives af_stop - af_start - 2*half_win + 1
 * values; a non-empty span shorter than the window gives 1 value, the span's
 * frames followed by the zeros of the reference's fresh buffer (:86, :107-113);
 * an empty span gives none. */
SGZ_API int sgz_segm_curve(sgz_context *ctx, const float *frames, int64_t num_frames, int num_channels,
                           const float *norms, int af_start, int af_stop, int half_win,
                           float temporal_weight, float *sim_out, int64_t *num_out);

/* Break bookkeeping of impl/FeatureSegmentationImpl.scala:55-83,121-125 over a
 * dense novelty curve.  Host code.  Returns the number of breaks written
 * (ascending sim). */
SGZ_API int sgz_segm_select(const float *sims, int64_t n, int af_start, int half_win, int step_size,
                    int num_breaks, int64_t min_spacing, sgz_break *out, int capacity);

/* ---- SelfSimilarity ----------------------------------------------------- */

/* The decimated similarity matrix of impl/SelfSimilarityImpl.scala:123-165:
 * entry (x, y) compares the half window of file 1 at frame af_start + x*decim
 * with that of file 2 at af_start + y*decim; img_ext = numCorrs / decim
 * (:81-91).  frames2 may equal frames1; norms as for sgz_segm_curve.  sim_out (nullable) receives
 * img_ext*img_ext floats, [x][y] with x = left; img_out (nullable) receives
 * the TYPE_INT_RGB raster of :150-155 (gray colour scheme only: colors must be
 * 0).  Both triangles are filled from the upper one, as the reference does.
 * Similarities are within 1e-5 of the reference's fp64 evaluation (product
 * sums on the fp32 matrix cores, statistics in fp64; pixels for which that
 * bound cannot be certified are evaluated in the reference's own order). */
SGZ_API int sgz_selfsim_run(sgz_context *ctx, const float *frames1, int64_t num_frames1,
                    const float *frames2, int64_t num_frames2, int num_channels,
                    const float *norms, int af_start, int half_win, int decim, int img_ext,
                    float temporal_weight, int colors, float color_warp, float color_ceil,
                    int color_inv, float *sim_out, int32_t *img_out);

/* Rows [x_begin, x_end) of the same matrix: sim_out receives (x_end - x_begin) * img_ext
 * floats, [x - x_begin][y].  For callers that apply a colour scheme of their own to the
 * similarities (the reference's default PsychoOptical palette, SelfSimilarity.scala:30-56,
 * impl/SelfSimilarityImpl.scala:109-110,150) and cannot hold img_ext * img_ext floats in one
 * buffer: at the reference's largest raster (img_ext = 0xB504, :81-91) the matrix is 8.6 GB,
 * a JVM direct buffer ends at 2 GB.  The matrix is computed by the first call and stays on
 * the context's GPU (img_ext * img_ext * 4 bytes) while later calls with the same inputs --
 * recognised by a digest of frames, norms and geometry -- fetch further bands; inputs that
 * differ, sgz_selfsim_release or sgz_destroy free it.  The bands put together equal the
 * sim_out of sgz_selfsim_run bit for bit. */
SGZ_API int sgz_selfsim_rows(sgz_context *ctx, const float *frames1, int64_t num_frames1,
                     const float *frames2, int64_t num_frames2, int num_channels,
                     const float *norms, int af_start, int half_win, int decim, int img_ext,
                     float temporal_weight, int x_begin, int x_end, float *sim_out);
SGZ_API int sgz_selfsim_release(sgz_context *ctx);

/* ---- FeatureExtraction --------------------------------------------------- */

/* Feature frames an input of num_frames sample frames yields:
 * ceil(num_frames / step) - 1 with step = fft_size / fft_overlap
 * (impl/NonRealtimeProcessor.scala:93,107-108). */
SGZ_API int64_t sgz_extract_num_frames(int64_t num_frames, int fft_size, int fft_overlap);

/* FeatureStats over the database files [file_begin, file_end) of the context
 * (impl/FeatureStatsImpl.scala:25-141): per file and channel the 1st and the 99th
 * percentile from a 2048-bin histogram over [min, max] of the channel, warped by
 * the power law that puts the channel mean at the middle bin; minmax_out
 * receives, per channel, the smallest 1st and the largest 99th percentile over
 * the files ([num_channels][2] doubles) -- the pairs the reference's command
 * line writes to feat_norms.aif (Strugatzki.scala:414-429).  per_file_out (may
 * be NULL): [files][num_channels][2].  Files must have at least one frame. */
SGZ_API int sgz_feature_stats(sgz_context *ctx, int file_begin, int file_end, double *minmax_out,
                              double *per_file_out);

/* Extracts [loudness / 32, mfcc 0 .. num_coeffs-1] per hop from interleaved PCM
 * (host memory, num_frames x num_channels).  Takes the place of the scsynth
 * non-realtime render of impl/FeatureExtractionImpl.scala:40-69 and
 * impl/NonRealtimeProcessor.scala:57-160: same framing (mix-down per
 * channels_behavior 0 = Mix / 1 = First / 2 = Last, Hann window, hop, channel
 * order, frame count), own spectral model (42 mel bands, dB, DCT-II, A-weighted
 * sones) since scsynth's is not part of the reference sources.  features_out
 * receives *num_out frames of num_coeffs + 1 floats. */
SGZ_API int sgz_extract(sgz_context *ctx, const float *pcm, int64_t num_frames, int num_channels,
                        double sample_rate, int fft_size, int fft_overlap, int num_coeffs,
                        int channels_behavior, float *features_out, int64_t capacity_frames,
                        int64_t *num_out);

/* The same from samples as audio files hold them (AudioFile.openRead of FeatureExtractionImpl decodes them on the
 * JVM, impl/FeatureExtractionImpl.scala:34; here the device does): 16-, 24- or 32-bit integers -- v / 2^(bits-1) --
 * or 32-bit floats, in the host's (little endian) or in big-endian byte order, interleaved, tightly packed.  Half
 * the bytes of float samples cross the bus for 16-bit audio, and no host loop touches a sample.
 * sgz_extract(...) == sgz_extract_pcm(..., SGZ_PCM_F32, ...). */
enum {
    SGZ_PCM_F32 = 0,
    SGZ_PCM_F32_BE = 1,
    SGZ_PCM_S16 = 2,
    SGZ_PCM_S16_BE = 3,
    SGZ_PCM_S24 = 4,
    SGZ_PCM_S24_BE = 5,
    SGZ_PCM_S32 = 6,
    SGZ_PCM_S32_BE = 7
};
SGZ_API int sgz_extract_pcm(sgz_context *ctx, const void *samples, int sample_format, int64_t num_frames,
                            int num_channels, double sample_rate, int fft_size, int fft_overlap, int num_coeffs,
                            int channels_behavior, float *features_out, int64_t capacity_frames, int64_t *num_out);

/* ---- whole processors from their XML configs (host side) ------------------ */

/* Progress observer: called with a value in [0, 1] between GPU launches; a
 * return of 0 aborts the run (the reference polls checkAborted() and reports
 * `progress`, e.g. impl/FeatureCorrelationImpl.scala:164,402). */
typedef int (*sgz_progress_fn)(double progress, void *user);

/* FeatureCorrelation.run: `config_xml` is a <correlate> document
 * (FeatureCorrelation.scala:231-244).  Scans the database folder, loads the
 * feature files, runs the search; *result_xml receives "<matches>" holding one
 * <match> element per result (FeatureCorrelation.scala:55-63), to be released
 * with sgz_free_string.  Replaces impl/FeatureCorrelationImpl.body() (:32-412). */
SGZ_API int sgz_run_correlation_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user,
                                    char **result_xml);
/* sgz_run_correlation_xml with the database dealt over num_ctx contexts (one per GPU of the node; files round
 * robin in iteration order) and searched by sgz_corr_search_multi: the same <matches>.  Like the single-context
 * call it leaves the files resident: a later call over the same database (same feature files, sizes and
 * modification times; sgz_db_tag) skips reading and uploading them.  A search with <punchOut> runs on ctxs[0]. */
SGZ_API int sgz_run_correlation_xml_multi(sgz_context *const *ctxs, int num_ctx, const char *config_xml,
                                          sgz_progress_fn cb, void *user, char **result_xml);
/* FeatureSegmentation.run: <segmentation> in (FeatureSegmentation.scala:179-189),
 * "<breaks>" of <break> elements out.  Replaces impl/FeatureSegmentationImpl.body(). */
SGZ_API int sgz_run_segmentation_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user,
                                     char **result_xml);
/* SelfSimilarity.run: <selfsimilarity> in (SelfSimilarity.scala:223-238); writes the
 * PNG named by <output>.  Replaces impl/SelfSimilarityImpl.body(). */
SGZ_API int sgz_run_selfsimilarity_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user);
/* Rows [x_begin, x_end) of the similarity matrix that <selfsimilarity> document describes -- inputs, norms, span,
 * corrLen, decimation and the automatic decimation worked out as impl/SelfSimilarityImpl.scala:32-96 does -- as
 * (x_end - x_begin) * img_ext floats, and the matrix extent in *img_ext_out (x_begin == x_end: the extent alone,
 * sim_out may be NULL).  No colour scheme is applied: the route of a caller that renders the raster with a palette
 * of its own (colors = psycho, the reference's default: SelfSimilarity.scala:30-56,178) band by band, see
 * sgz_selfsim_rows. */
SGZ_API int sgz_run_selfsimilarity_rows_xml(sgz_context *ctx, const char *config_xml, int x_begin, int x_end,
                                            float *sim_out, int *img_ext_out);
/* FeatureExtraction.run: <feature> in (FeatureExtraction.scala:163-172); reads <input>,
 * writes the feature AIFF <output> and, if <meta> is set, the config XML there.
 * Replaces impl/FeatureExtractionImpl.body() (:31-78). */
SGZ_API int sgz_run_extraction_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user);
/* CrossSimilarity.run with the reference's <crosssimilarity> document (CrossSimilarity.scala:182-246,
 * impl/CrossSimilarityImpl.scala:32-173): the shorter span is correlated with every offset of the longer
 * one (exact kernel); the similarities are written to <output> as a 1-channel fp32 AIFF at the feature
 * rate of input 1.  The ring of the reference (8192 frames wide, offsets wrapping at the target length:
 * :118-119, 140, 165) is taken as the target length.  Replaces the files of the context's database. */
SGZ_API int sgz_run_crosssimilarity_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user);
/* FeatureStats.run (FeatureStats.scala:22-34, impl/FeatureStatsImpl.scala:31-54): the configuration is the
 * list of feature files (AIFF / AIFC / WAVE, as written by the extraction); they replace the context's
 * database.  minmax_out receives [channels][2] doubles, *num_channels the channel count (SGZ_E_OVERFLOW when
 * it exceeds capacity_channels). */
SGZ_API int sgz_run_feature_stats(sgz_context *ctx, const char *const *feature_files, int num_files,
                                  sgz_progress_fn cb, void *user, double *minmax_out, int capacity_channels,
                                  int *num_channels);
/* Writes the pairs as the normalisation file the searches read: 2 frames (min, max) x num_channels, fp32
 * AIFF at 44.1 kHz, what the reference's command line does with the result (Strugatzki.scala:414-429). */
SGZ_API int sgz_write_norms_file(const char *path, const double *minmax, int num_channels);
/* Text of the last failure of an sgz_run_* / sgz_*_xml / file call on this thread. */
SGZ_API const char *sgz_host_last_error(void);
SGZ_API void sgz_free_string(char *s);

/* Parses a config / result document of the given kind ("feature", "correlate",
 * "match", "segmentation", "break", "selfsimilarity") and prints it again:
 * X.fromXML(x.toXML) of the reference (StrugatzkiSuite.scala:25-29,53-58,64-68,86-91). */
SGZ_API int sgz_config_roundtrip_xml(const char *kind, const char *xml, char **out_xml);
/* The default configuration of a processor as XML (ConfigBuilder defaults,
 * e.g. FeatureCorrelation.scala:168-206). */
SGZ_API int sgz_default_config_xml(const char *kind, char **out_xml);

/* secsToFrames of the reference's command line (Strugatzki.scala:143): (secs * sample_rate + 0.5).toLong. */
SGZ_API int64_t sgz_secs_to_frames(double secs, double sample_rate);
/* The seconds-based options of `strugatzki -c` (Strugatzki.scala:101-176) as the <correlate> document
 * sgz_run_correlation_xml takes: spans, minimum / maximum fill duration and spacing converted with
 * sgz_secs_to_frames at the sample rate of the input's audio file (sample_rate <= 0: read from the file the
 * meta file names, :140-141), with the reference's checks ("Punch in span is empty", "Minimum duration is
 * zero", ...: SGZ_E_ARG).  has_out = 0: no punch-out span. */
SGZ_API int sgz_correlate_config_from_seconds(const char *database_dir, const char *meta_input, double sample_rate,
                                              double in_start, double in_stop, double in_temp, int has_out,
                                              double out_start, double out_stop, double out_temp, double dur_min,
                                              double dur_max, double boost_max, int num_matches, int num_per_file,
                                              double spacing, int normalize, char **out_xml);
/* Writes interleaved float frames as 32-bit float audio of the given type: "aiff" (AIFF-C fl32), "wav" (RIFF-WAVE,
 * IEEE float) or "next" (NeXT/Sun): the output types CrossSimilarity's <outputType> can name
 * (CrossSimilarity.scala:55,213; impl/CrossSimilarityImpl.scala:87-88). */
SGZ_API int sgz_write_float_audio(const char *path, const char *type, const float *frames, int64_t num_frames,
                                  int num_channels, double sample_rate);
/* Writes interleaved float frames as a 32-bit float AIFF-C, the feature file
 * format (impl/NonRealtimeProcessor.scala:164-165). */
SGZ_API int sgz_write_float_aiff(const char *path, const float *frames, int64_t num_frames, int num_channels,
                                 double sample_rate);
/* Reads an AIFF / AIFF-C / WAVE file; frames_out may be NULL to query the size
 * (AudioFile.openRead + read, e.g. impl/FeatureCorrelationImpl.scala:81-89). */
SGZ_API int sgz_read_audio_file(const char *path, float *frames_out, int64_t capacity_frames,
                                int64_t *num_frames, int *num_channels, double *sample_rate);

#ifdef __cplusplus
}
#endif
#endif /* STRUGATZKI_AMD_H */
