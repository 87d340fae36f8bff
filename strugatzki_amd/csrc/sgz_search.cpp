// sgz_corr_search: one database search over all files of a context -- GPU similarities, then the host-side
// replay of the reference's match bookkeeping in file order (impl/FeatureCorrelationImpl.scala:113-150, 234,
// 399-411) -- as one call through the fine-grained entry points of strugatzki_amd.h.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "strugatzki_amd.h"

namespace {

struct SelGuard {
    sgz_selector *s;
    ~SelGuard() { sgz_sel_destroy(s); }
};

}  // namespace

extern "C" int sgz_corr_search(sgz_context *ctx, int target_frames, int num_matches, int num_per_file,
                               int64_t min_spacing, int step_size, int screen_mode, int over_bounds, int64_t capacity,
                               float margin, sgz_match *matches_out, int out_capacity, int *count_out)
{
    if (!ctx || !count_out || out_capacity < 0 || (out_capacity > 0 && !matches_out) || capacity < 1) return SGZ_E_ARG;
    if (screen_mode < SGZ_SCREEN_NONE || screen_mode > SGZ_SCREEN_AUTO) return SGZ_E_ARG;
    *count_out = 0;
    const int nf = sgz_db_num_files(ctx);
    bool f16 = false;
    if (screen_mode == SGZ_SCREEN_F16 || screen_mode == SGZ_SCREEN_AUTO) f16 = sgz_corr_screen_f16_available(ctx) == 1;
    if (screen_mode == SGZ_SCREEN_F16 && !f16) return SGZ_E_STATE;
    const bool screen = screen_mode != SGZ_SCREEN_NONE;
    int rc;
    // candidate records fetched from the GPU; kept between calls (a fresh megabyte per search costs more
    // in page faults than the replay itself)
    static thread_local std::vector<sgz_candidate> cands;

    // numPerFile = 1: a file can only contribute its largest similarity, first occurrence.  All files go through
    // the half-precision kernel; per file the windows whose upper bound reaches the file's largest lower bound
    // are re-scored exactly and the exact maxima are fed in file order.
    if (over_bounds && f16 && num_per_file == 1 && num_matches >= 1 && nf > 0) {
        if (cands.size() < (size_t)capacity) cands.resize((size_t)capacity);
        int64_t n = 0;
        // (this path reads nothing of the run but its candidates: the spectral kernel then lists them itself)
        sgz_corr_set_screen_output(ctx, 1);
        rc = sgz_corr_run_screen_f16(ctx, 0, nf);
        sgz_corr_set_screen_output(ctx, 0);
        if (rc) return rc;
        // first with the files left out that cannot reach the numMatches-th best of the per-file bounds
        float top = 0.0f;
        rc = sgz_corr_compact_top(ctx, 0, nf, num_matches, cands.data(), capacity, &n, &top);
        if (rc != SGZ_OK && rc != SGZ_E_OVERFLOW) return rc;
        if (rc == SGZ_OK) {
            SelGuard q{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
            if (!q.s) return SGZ_E_ARG;
            rc = sgz_sel_feed_top(q.s, cands.data(), n, top);
            if (rc == SGZ_OK) {
                *count_out = sgz_sel_result(q.s, matches_out, std::min(out_capacity, num_matches));
                return *count_out < 0 ? *count_out : SGZ_OK;
            }
            if (rc != SGZ_E_STATE) return rc;  // SGZ_E_STATE: that filter does not apply; every file, then
            rc = sgz_corr_compact_bounded(ctx, 0, nf, nf, cands.data(), capacity, &n);
        }  // (an overflow with the filter is one without it, too)
        if (rc != SGZ_OK && rc != SGZ_E_OVERFLOW) return rc;
        if (rc == SGZ_OK) {
            SelGuard q{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
            if (!q.s) return SGZ_E_ARG;
            rc = sgz_sel_feed_screened(q.s, cands.data(), n, nf, nf);
            if (rc == SGZ_OK) {
                *count_out = sgz_sel_result(q.s, matches_out, std::min(out_capacity, num_matches));
                return *count_out < 0 ? *count_out : SGZ_OK;
            }
            if (rc != SGZ_E_STATE) return rc;  // SGZ_E_STATE: the shortcut does not apply
        }
    }

    // General case.  Head files: exact fp64 kernel, fed densely.  Tail files: a screening kernel (half-precision
    // matrix-core kernel, which writes upper bounds, or fp32 products); the windows within `margin` of the bound
    // are re-scored exactly on the GPU before they are fed.
    SelGuard g{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
    if (!g.s) return SGZ_E_ARG;
    sgz_selector *sel = g.s;
    // The first numMatches files are fed densely from exact values.  (The set can be full after
    // ceil(numMatches / numPerFile) of them, but a head that short leaves a low threshold and with it thousands
    // of windows to re-score: measured 1.15 - 1.9 ms against 0.97 ms per search at numPerFile 2 and 5.)  If the
    // set is still short afterwards (files gave fewer entries: minSpacing, equal sims, few windows), further
    // files follow the same way.
    const int head = std::min(nf, std::max(num_matches, 1));
    if (!(margin >= 0.0f)) margin = f16 ? 1e-6f : 1e-5f;
    bool screened = screen && head < nf;
    if ((rc = sgz_corr_run(ctx, 0, head))) return rc;
    if (head < nf && (rc = sgz_corr_prefetch_dense(ctx, 0, head))) return rc;  // on its way before the tail's run
    if (head < nf) {
        rc = !screened ? sgz_corr_run(ctx, head, nf)
                       : (f16 ? sgz_corr_run_screen_f16(ctx, head, nf) : sgz_corr_run_screen(ctx, head, nf));
        if (rc) return rc;
    }
    std::vector<float> sim, boost;
    auto feedDense = [&](int f0, int f1) -> int {
        int64_t total = 0;
        for (int f = f0; f < f1; f++) total += sgz_corr_num_windows(ctx, f);
        sim.resize((size_t)std::max<int64_t>(total, 1));
        boost.resize(sim.size());
        if (f1 > f0) {
            const int r = sgz_corr_read_dense_range(ctx, f0, f1, sim.data(), boost.data());
            if (r) return r;
        }
        size_t pos = 0;
        for (int f = f0; f < f1; f++) {
            const int64_t n = sgz_corr_num_windows(ctx, f);
            const int r = sgz_sel_feed_file(sel, f, n, nullptr, sim.data() + pos, boost.data() + pos);
            if (r) return r;
            pos += (size_t)n;
        }
        return SGZ_OK;
    };
    if ((rc = feedDense(0, head))) return rc;
    int f = head;
    // the set is not full yet -> keep feeding densely, a head's worth of files at a time; dense feeding needs
    // exact values (the screened values of the files behind them stay as they are)
    while (f < nf && !sgz_sel_is_full(sel)) {
        const int f1 = std::min(nf, f + head);
        if (screened && (rc = sgz_corr_run(ctx, f, f1))) return rc;
        if ((rc = feedDense(f, f1))) return rc;
        f = f1;
    }
    if (f < nf) {
        if (cands.size() < (size_t)capacity) cands.resize((size_t)capacity);
        int64_t n = 0;
        rc = screened ? sgz_corr_compact_rescored(ctx, f, nf, sgz_sel_threshold(sel) - margin, cands.data(), capacity, &n)
                      : sgz_corr_compact(ctx, f, nf, sgz_sel_threshold(sel), cands.data(), capacity, &n);
        if (rc == SGZ_E_OVERFLOW) {  // more candidates than the buffer holds: dense replay of the tail
            if (screened && (rc = sgz_corr_run(ctx, f, nf))) return rc;
            for (int k = f; k < nf; k++)
                if ((rc = feedDense(k, k + 1))) return rc;
        } else {
            if (rc) return rc;
            if ((rc = sgz_sel_feed_candidates(sel, cands.data(), n))) return rc;
        }
    }
    *count_out = sgz_sel_result(sel, matches_out, std::min(out_capacity, num_matches));
    return *count_out < 0 ? *count_out : SGZ_OK;
}

// sgz_corr_search_batch: queries in flight at once on one GPU.  One query is a chain of dependent launches -- the
// target's preparation and spectra, the screening kernel, three short kernels that filter and re-score a handful
// of candidates -- of which only the screening kernel fills the chip; the host waits twice per query.  With
// several queries in flight, each on a context (stream, target state, result buffers) of its own, the short
// kernels and the host's waits of one query run beside the screening kernel of another.  The contexts are
// replicas: every one holds the same files and norms.  A thread per context takes queries off a shared counter.
extern "C" int sgz_corr_search_batch(sgz_context *const *ctxs, int num_ctx, const float *const *targets,
                                     int targets_on_device, int num_queries, int target_frames, float temporal_weight,
                                     float max_boost, int num_matches, int num_per_file, int64_t min_spacing,
                                     int step_size, int screen_mode, int over_bounds, int64_t capacity, float margin,
                                     sgz_match *matches_out, int out_capacity, int *counts_out)
{
    if (!ctxs || num_ctx < 1 || num_queries < 0 || (num_queries > 0 && (!targets || !counts_out))) return SGZ_E_ARG;
    if (out_capacity < 0 || (out_capacity > 0 && !matches_out)) return SGZ_E_ARG;
    for (int c = 0; c < num_ctx; c++) {
        if (!ctxs[c]) return SGZ_E_ARG;
        for (int d = 0; d < c; d++)
            if (ctxs[d] == ctxs[c]) return SGZ_E_ARG;  // (a context serves one thread at a time)
    }
    for (int q = 0; q < num_queries; q++) {
        if (!targets[q]) return SGZ_E_ARG;
        counts_out[q] = 0;
    }
    std::atomic<int> next{0}, failed{SGZ_OK};
    auto lane = [&](int c) {
        sgz_context *ctx = ctxs[c];
        for (;;) {
            const int q = next.fetch_add(1, std::memory_order_relaxed);
            if (q >= num_queries || failed.load(std::memory_order_relaxed) != SGZ_OK) return;
            int rc = targets_on_device
                         ? sgz_corr_set_target_device(ctx, targets[q], target_frames, temporal_weight, max_boost)
                         : sgz_corr_set_target(ctx, targets[q], target_frames, temporal_weight, max_boost);
            if (rc == SGZ_OK)
                rc = sgz_corr_search(ctx, target_frames, num_matches, num_per_file, min_spacing, step_size, screen_mode,
                                     over_bounds, capacity, margin,
                                     out_capacity > 0 ? matches_out + (size_t)q * (size_t)out_capacity : nullptr,
                                     out_capacity, counts_out + q);
            if (rc != SGZ_OK) {
                int none = SGZ_OK;
                failed.compare_exchange_strong(none, rc);  // (the first failure is the one reported)
                return;
            }
        }
    };
    const int lanes = std::min(num_ctx, std::max(num_queries, 1));
    std::vector<std::thread> others;
    for (int c = 1; c < lanes; c++) others.emplace_back(lane, c);
    lane(0);
    for (auto &t : others) t.join();
    return failed.load();
}
