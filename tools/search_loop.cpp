// The search step of the benchmark driven from C++ (no interpreter between searches): microseconds per
// sgz_corr_search on 120 synthetic files of 25 839 frames and a target of 2 584.
// Build: g++ -O2 -Iinclude tools/search_loop.cpp -Lstrugatzki_amd/lib -lstrugatzki_amd -Wl,-rpath,$PWD/strugatzki_amd/lib -o tools/search_loop
#include <chrono>
#include <cstdio>
#include <vector>

#include "strugatzki_amd.h"

int main()
{
    const int L = 2584, FR = 25839, NF = 120, CH = 14;
    sgz_context *ctx = nullptr;
    if (sgz_create(0, nullptr, &ctx) || !ctx) return 1;
    sgz_db_reset(ctx, CH);
    for (int g = 0; g < NF; g++)
        if (sgz_db_add_synthetic(ctx, 1234, g, FR, nullptr, 0)) return 2;
    std::vector<float> norms(2 * CH);
    for (int c = 0; c < CH; c++) {
        norms[2 * c] = c == 0 ? 0.0f : 0.01f * c;
        norms[2 * c + 1] = c == 0 ? 1.5f : 1.0f + 0.02f * c;
    }
    sgz_set_norms(ctx, norms.data());
    std::vector<float> file((size_t)FR * CH);
    sgz_db_read_file(ctx, 7, file.data());
    if (sgz_corr_set_target(ctx, file.data() + 1000 * CH, L, 0.5f, 8.0f)) return 3;
    sgz_match m[10];
    int n = 0;
    for (int i = 0; i < 600; i++)
        if (sgz_corr_search(ctx, L, 10, 1, 0, 512, SGZ_SCREEN_AUTO, 1, 1 << 16, -1.0f, m, 10, &n)) return 4;
    const int steps = 2000;
    for (int every : {1, 0, 8, 1}) {  // kernel timing events with every launch, never, with one launch in 8
        sgz_set_kernel_timing(ctx, every);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; i++)
            sgz_corr_search(ctx, L, 10, 1, 0, 512, SGZ_SCREEN_AUTO, 1, 1 << 16, -1.0f, m, 10, &n);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        std::printf("timing events every %d launches: %.2f us per search, %d matches, best %.6f in file %d\n", every, us, n,
                    m[0].sim, m[0].file);
    }
    // the same through the fine-grained calls, with the host time of each
    std::vector<sgz_candidate> cands(1 << 16);
    double tRun = 0, tCompact = 0, tFeed = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto usSince = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::micro>(now() - a).count();
    };
    const auto t1 = now();
    for (int i = 0; i < steps; i++) {
        auto a = now();
        sgz_corr_run_screen_f16(ctx, 0, NF);
        tRun += usSince(a);
        a = now();
        int64_t cnt = 0;
        float top = 0;
        sgz_corr_compact_top(ctx, 0, NF, 10, cands.data(), 1 << 16, &cnt, &top);
        tCompact += usSince(a);
        a = now();
        sgz_selector *sel = sgz_sel_create(10, 1, 0, 512, L);
        sgz_sel_feed_top(sel, cands.data(), cnt, top);
        n = sgz_sel_result(sel, m, 10);
        sgz_sel_destroy(sel);
        tFeed += usSince(a);
    }
    std::printf("fine-grained: %.2f us per search = launch of the matrix kernel %.2f + compaction, re-scoring and wait %.2f"
                " + selection %.2f\n", usSince(t1) / steps, tRun / steps, tCompact / steps, tFeed / steps);
    // the general path (numPerFile 2), phase by phase
    {
        const int head = 10;
        std::vector<float> sim((size_t)head * (FR - L + 1)), boost(sim.size());
        double tLaunch = 0, tRead = 0, tReplay = 0, tTail = 0;
        const int reps = 200;
        const auto tg = now();
        for (int i = 0; i < reps; i++) {
            auto a = now();
            sgz_corr_run(ctx, 0, head);
            sgz_corr_run_screen_f16(ctx, head, NF);
            tLaunch += usSince(a);
            a = now();
            sgz_corr_read_dense_range(ctx, 0, head, sim.data(), boost.data());
            tRead += usSince(a);
            a = now();
            sgz_selector *sel = sgz_sel_create(10, 2, 0, 512, L);
            size_t pos = 0;
            for (int f = 0; f < head; f++) {
                const int64_t nw = sgz_corr_num_windows(ctx, f);
                sgz_sel_feed_file(sel, f, nw, nullptr, sim.data() + pos, boost.data() + pos);
                pos += (size_t)nw;
            }
            tReplay += usSince(a);
            a = now();
            int64_t cnt = 0;
            sgz_corr_compact_rescored(ctx, head, NF, sgz_sel_threshold(sel) - 1e-6f, cands.data(), 1 << 16, &cnt);
            sgz_sel_feed_candidates(sel, cands.data(), cnt);
            n = sgz_sel_result(sel, m, 10);
            sgz_sel_destroy(sel);
            tTail += usSince(a);
        }
        std::printf("general path: %.1f us per search = launches %.1f + wait for and read the dense head %.1f + replay %.1f"
                    " + tail compaction, re-scoring, replay %.1f\n", usSince(tg) / reps, tLaunch / reps, tRead / reps,
                    tReplay / reps, tTail / reps);
    }
    sgz_destroy(ctx);
    return 0;
}
