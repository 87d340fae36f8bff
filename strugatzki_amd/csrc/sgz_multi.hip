// sgz_corr_search_multi: one search over a database that is spread over several contexts (one per GPU of a
// node), driven from one host process -- what a JVM caller of FeatureCorrelation.run reaches through the C ABI.
// The reference's single loop over all files (impl/FeatureCorrelationImpl.scala:152-246) is cut by file: every
// context computes the similarities of its own files, the host merges per-file candidates and replays the
// selection in global file order (:113-150, 234, 399-411).  The exchange is a few records per file in host
// memory; no device-to-device traffic is needed for it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "sgz_internal.hpp"

namespace {

struct SelGuard {
    sgz_selector *s;
    ~SelGuard() { sgz_sel_destroy(s); }
};

// Runs fn(c) for every context on a thread of its own; returns the first failure code.
template <typename F>
int for_each_context(int n, F fn)
{
    std::vector<int> rc((size_t)n, SGZ_OK);
    std::vector<std::thread> pool;
    for (int c = 1; c < n; c++) pool.emplace_back([&, c] { rc[(size_t)c] = fn(c); });
    rc[0] = fn(0);
    for (auto &t : pool) t.join();
    for (int c = 0; c < n; c++)
        if (rc[(size_t)c] != SGZ_OK) return rc[(size_t)c];
    return SGZ_OK;
}

}  // namespace

extern "C" int sgz_corr_search_multi(sgz_context *const *ctxs, int num_ctx, const int32_t *const *global_ids,
                                     int num_files_total, int target_frames, int num_matches, int num_per_file,
                                     int64_t min_spacing, int step_size, int64_t capacity, sgz_match *matches_out,
                                     int out_capacity, int *count_out)
{
    if (!ctxs || num_ctx < 1 || !global_ids || !count_out || out_capacity < 0 || (out_capacity > 0 && !matches_out) ||
        capacity < 1 || num_files_total < 0)
        return SGZ_E_ARG;
    *count_out = 0;
    sgz_context *lead = ctxs[0];
    if (!lead) return SGZ_E_ARG;
    // where every global file lives
    struct Place {
        int ctx, local;
    };
    std::vector<Place> place((size_t)num_files_total, Place{-1, -1});
    for (int c = 0; c < num_ctx; c++) {
        if (!ctxs[c]) return lead->fail(SGZ_E_ARG, "null context in the list");
        if (!ctxs[c]->have_target || ctxs[c]->cp.L != target_frames || ctxs[c]->numCh != lead->numCh)
            return lead->fail(SGZ_E_STATE, "every context needs the same target (and channel count) set");
        const int nloc = (int)ctxs[c]->files.size();
        if (nloc > 0 && !global_ids[c]) return lead->fail(SGZ_E_ARG, "missing global ids of a context");
        for (int i = 0; i < nloc; i++) {
            const int g = global_ids[c][i];
            if (g < 0 || g >= num_files_total || place[(size_t)g].ctx >= 0 || (i > 0 && g <= global_ids[c][i - 1]))
                return lead->fail(SGZ_E_ARG, "global ids must be ascending per context and cover every file once");
            place[(size_t)g] = Place{c, i};
        }
    }
    for (const Place &p : place)
        if (p.ctx < 0) return lead->fail(SGZ_E_ARG, "global ids do not cover every file");
    const int L = target_frames, numCh = lead->numCh;

    // Files shorter than the target: the ring contents the single loop over all files would show them.  They
    // depend on the placement, the frame counts, the target length and the norms (positions no file has written
    // hold raw values that normalise to 0): worked out when one of these changes, not per search -- a resident
    // database answers its queries without reading frames back.
    {
        std::vector<long long> frames((size_t)num_files_total);
        bool anyShort = false;
        for (int g = 0; g < num_files_total; g++) {
            frames[(size_t)g] = ctxs[place[(size_t)g].ctx]->files[(size_t)place[(size_t)g].local].frames;
            anyShort |= frames[(size_t)g] > 0 && frames[(size_t)g] < L;
        }
        std::string key;
        if (anyShort) {
            auto put = [&key](const void *p, size_t n) { key.append(reinterpret_cast<const char *>(p), n); };
            put(&L, sizeof L);
            put(&num_ctx, sizeof num_ctx);
            for (int c = 0; c < num_ctx; c++) {
                put(&ctxs[c], sizeof ctxs[c]);
                put(&ctxs[c]->db_serial, sizeof ctxs[c]->db_serial);
                put(&ctxs[c]->norm, sizeof ctxs[c]->norm);
            }
            for (int g = 0; g < num_files_total; g++) {
                put(&place[(size_t)g], sizeof(Place));
                put(&frames[(size_t)g], sizeof(long long));
            }
        }
        if (anyShort && key != lead->multi_ring_key) {
            const std::vector<sgz_ring_run> runs = sgz::ring_runs(frames.data(), num_files_total, L);
            std::vector<float> zero((size_t)numCh), img;
            for (size_t k = 0; k < runs.size();) {
                const int g = runs[k].file;
                const Place dst = place[(size_t)g];
                const long long own = frames[(size_t)g];
                img.assign((size_t)(L - own) * numCh, 0.0f);
                sgz_norm_zero_frame(ctxs[dst.ctx], zero.data());
                for (; k < runs.size() && runs[k].file == g; k++) {
                    const sgz_ring_run &r = runs[k];
                    float *to = img.data() + (size_t)(r.dest - own) * numCh;
                    if (r.src_file < 0) {
                        for (int i = 0; i < r.count; i++) std::copy(zero.begin(), zero.end(), to + (size_t)i * numCh);
                    } else {
                        const Place src = place[(size_t)r.src_file];
                        const int rc = sgz_db_read_frames(ctxs[src.ctx], src.local, r.src_frame, r.count, to);
                        if (rc) return lead->fail(rc, sgz_last_error(ctxs[src.ctx]));
                    }
                }
                const int rc = sgz_corr_set_ring_image(ctxs[dst.ctx], dst.local, img.data(), (int)(L - own));
                if (rc) return lead->fail(rc, sgz_last_error(ctxs[dst.ctx]));
            }
            lead->multi_ring_key = key;  // (only now: a failure on the way leaves the work to be done again)
        }
    }

    std::vector<std::vector<sgz_candidate>> cands((size_t)num_ctx);
    std::vector<int64_t> counts((size_t)num_ctx, 0);
    auto merged = [&](std::vector<sgz_candidate> &all) {
        all.clear();
        for (int c = 0; c < num_ctx; c++)
            for (int64_t i = 0; i < counts[(size_t)c]; i++) {
                sgz_candidate r = cands[(size_t)c][(size_t)i];
                r.file = global_ids[c][r.file];
                all.push_back(r);
            }
        std::sort(all.begin(), all.end(), [](const sgz_candidate &a, const sgz_candidate &b) {
            return a.file != b.file ? a.file < b.file : a.offset < b.offset;
        });
    };
    std::vector<sgz_candidate> all;

    // numPerFile = 1: a file contributes its largest similarity, first occurrence; every context finds the exact
    // maxima of its own files through the bounds of the matrix-core kernel.
    bool f16 = num_per_file == 1 && num_matches >= 1 && num_files_total > 0;
    for (int c = 0; c < num_ctx && f16; c++) f16 = sgz_corr_screen_f16_available(ctxs[c]) == 1;
    // First with every context leaving out the files that cannot reach the numMatches-th best of its own files'
    // bounds (sgz_corr_compact_top), then, where the finished set shows that this filter does not apply, with
    // every file.
    for (int pass = 0; pass < 2 && f16; pass++) {
        std::vector<int> over((size_t)num_ctx, 0);
        std::vector<float> tops((size_t)num_ctx, -INFINITY);
        const int rc = for_each_context(num_ctx, [&](int c) -> int {
            sgz_context *ctx = ctxs[c];
            const int nloc = (int)ctx->files.size();
            counts[(size_t)c] = 0;
            if (nloc == 0) return SGZ_OK;
            cands[(size_t)c].resize((size_t)capacity);
            int r = SGZ_OK;
            if (pass == 0) {  // (nothing of the run is read but its candidates)
                sgz_corr_set_screen_output(ctx, 1);
                r = sgz_corr_run_screen_f16(ctx, 0, nloc);
                sgz_corr_set_screen_output(ctx, 0);
            }
            if (r) return r;
            r = pass == 0 ? sgz_corr_compact_top(ctx, 0, nloc, num_matches, cands[(size_t)c].data(), capacity,
                                                 &counts[(size_t)c], &tops[(size_t)c])
                          : sgz_corr_compact_bounded(ctx, 0, nloc, nloc, cands[(size_t)c].data(), capacity,
                                                     &counts[(size_t)c]);
            if (r == SGZ_E_OVERFLOW) {
                over[(size_t)c] = 1;
                return SGZ_OK;
            }
            return r;
        });
        if (rc) return rc;
        if (std::any_of(over.begin(), over.end(), [](int o) { return o != 0; })) break;
        merged(all);
        SelGuard q{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
        if (!q.s) return SGZ_E_ARG;
        const int r = pass == 0 ? sgz_sel_feed_top(q.s, all.data(), (int64_t)all.size(),
                                                   *std::max_element(tops.begin(), tops.end()))
                                : sgz_sel_feed_screened(q.s, all.data(), (int64_t)all.size(), num_files_total,
                                                        num_files_total);
        if (r == SGZ_OK) {
            *count_out = sgz_sel_result(q.s, matches_out, std::min(out_capacity, num_matches));
            return *count_out < 0 ? *count_out : SGZ_OK;
        }
        if (r != SGZ_E_STATE) return r;  // SGZ_E_STATE: the shortcut does not apply
    }

    // General case: the exact kernel over every file; the first numMatches files of the global order are replayed
    // densely (from then on the worst entry of the full set bounds what can still be offered, and that bound
    // never falls), the rest through threshold compaction on every context.
    int rc = for_each_context(num_ctx, [&](int c) -> int {
        const int nloc = (int)ctxs[c]->files.size();
        return nloc ? sgz_corr_run(ctxs[c], 0, nloc) : SGZ_OK;
    });
    if (rc) return rc;
    SelGuard g{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
    if (!g.s) return SGZ_E_ARG;
    std::vector<float> sim, boost;
    int next = 0;  // first global file that has not been fed
    const int head = std::min(num_files_total, std::max(num_matches, 1));
    while (next < num_files_total && (next < head || !sgz_sel_is_full(g.s))) {
        const Place p = place[(size_t)next];
        const int64_t n = sgz_corr_num_windows(ctxs[p.ctx], p.local);
        sim.resize((size_t)std::max<int64_t>(n, 1));
        boost.resize(sim.size());
        if ((rc = sgz_corr_read_dense(ctxs[p.ctx], p.local, sim.data(), boost.data())))
            return lead->fail(rc, sgz_last_error(ctxs[p.ctx]));
        if ((rc = sgz_sel_feed_file(g.s, next, n, nullptr, sim.data(), boost.data()))) return rc;
        next++;
    }
    if (next < num_files_total) {
        const float thr = sgz_sel_threshold(g.s);
        rc = for_each_context(num_ctx, [&](int c) -> int {
            sgz_context *ctx = ctxs[c];
            const int nloc = (int)ctx->files.size();
            // the files already fed are a prefix of this context's (ascending) list
            int first = 0;
            while (first < nloc && global_ids[c][first] < next) first++;
            counts[(size_t)c] = 0;
            if (first >= nloc) return SGZ_OK;
            cands[(size_t)c].resize((size_t)capacity);
            return sgz_corr_compact(ctx, first, nloc, thr, cands[(size_t)c].data(), capacity, &counts[(size_t)c]);
        });
        if (rc == SGZ_E_OVERFLOW) return lead->fail(rc, "more candidates than `capacity` on one context");
        if (rc) return rc;
        merged(all);
        if ((rc = sgz_sel_feed_candidates(g.s, all.data(), (int64_t)all.size()))) return rc;
    }
    *count_out = sgz_sel_result(g.s, matches_out, std::min(out_capacity, num_matches));
    return *count_out < 0 ? *count_out : SGZ_OK;
}
