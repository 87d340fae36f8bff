// Queries in flight at once on one GPU, driven from C++: K contexts (each with its own copy of the database), a host
// thread per context, every thread running queries (a new target, then sgz_corr_search) back to back.
// Build: g++ -O2 -pthread -Iinclude tools/query_threads.cpp -Lstrugatzki_amd/lib -lstrugatzki_amd -Wl,-rpath,$PWD/strugatzki_amd/lib -o tools/query_threads
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "strugatzki_amd.h"

int main(int argc, char **argv)
{
    const int L = 2584, FR = 25839, CH = 14;
    const int steps = argc > 1 ? std::atoi(argv[1]) : 4000, NF = argc > 2 ? std::atoi(argv[2]) : 120;
    const int kmax = argc > 3 ? std::atoi(argv[3]) : 4;
    std::vector<sgz_context *> ctxs(kmax, nullptr);
    std::vector<float> norms(2 * CH);
    for (int c = 0; c < CH; c++) {
        norms[2 * c] = c == 0 ? 0.0f : 0.01f * c;
        norms[2 * c + 1] = c == 0 ? 1.5f : 1.0f + 0.02f * c;
    }
    for (auto &ctx : ctxs) {
        if (sgz_create(0, nullptr, &ctx) || !ctx) return 1;
        sgz_db_reset(ctx, CH);
        for (int g = 0; g < NF; g++)
            if (sgz_db_add_synthetic(ctx, 1234, g, FR, nullptr, 0)) return 2;
        sgz_set_norms(ctx, norms.data());
    }
    std::vector<float> file((size_t)FR * CH);
    sgz_db_read_file(ctxs[0], 7, file.data());
    const int NT = 8;
    auto target = [&](int k) { return file.data() + (size_t)(1000 + 2600 * k) * CH; };
    int expect[NT];
    for (auto ctx : ctxs)
        for (int k = 0; k < NT; k++) {
            sgz_match m[10];
            int n = 0;
            if (sgz_corr_set_target(ctx, target(k), L, 0.5f, 8.0f)) return 3;
            if (sgz_corr_search(ctx, L, 10, 1, 0, 512, SGZ_SCREEN_AUTO, 1, 1 << 16, -1.0f, m, 10, &n)) return 4;
            expect[k] = m[0].start;
        }
    for (int K = 1; K <= kmax; K++)
        for (int rep = 0; rep < 2; rep++) {
            std::atomic<int> go{0}, wrong{0};
            std::vector<std::thread> th;
            for (int j = 0; j < K; j++)
                th.emplace_back([&, j] {
                    while (!go.load()) {
                    }
                    sgz_match m[10];
                    int n = 0;
                    for (int i = 0; i < steps; i++) {
                        const int k = (i + 3 * j) % NT;
                        sgz_corr_set_target(ctxs[j], target(k), L, 0.5f, 8.0f);
                        sgz_corr_search(ctxs[j], L, 10, 1, 0, 512, SGZ_SCREEN_AUTO, 1, 1 << 16, -1.0f, m, 10, &n);
                        if (n != 10 || m[0].start != expect[k]) wrong++;
                    }
                });
            const auto t0 = std::chrono::steady_clock::now();
            go = 1;
            for (auto &t : th) t.join();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%d in flight: %.2f us per query (%d queries, %d wrong)\n", K, us / (steps * K), steps * K, wrong.load());
            std::fflush(stdout);
        }
    for (auto ctx : ctxs) sgz_destroy(ctx);
    return 0;
}
