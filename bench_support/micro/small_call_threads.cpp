// Aggregate rate of CONCURRENT small calls at the C ABI, without Python in the loop: T threads, a context each, K calls of `rows` rows each
// through strsim_pairs_host (the in-place small-call path: one kernel launch + one synchronise per call).  The question (SURVEY 8 f3,
// "batching of concurrent small calls"): does the device, or the runtime's launch path, saturate before the host's threads do?
//   g++ -O2 -std=c++17 -pthread -Iinclude bench_support/micro/small_call_threads.cpp -o bench_support/micro/small_call_threads \
//       polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,$PWD/polars-strsim_amd/polars_strsim
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "strsim_amd.h"

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? atoi(argv[1]) : 2000;
    for (int rows : {100, 4000}) {
        std::vector<uint32_t> ao(rows + 1), bo(rows + 1);
        std::vector<uint8_t> av, bv;
        for (int i = 0; i < rows; ++i) {
            const int la = 1 + (i * 7) % 32, lb = 1 + (i * 11) % 32;
            ao[i] = (uint32_t)av.size(); bo[i] = (uint32_t)bv.size();
            for (int k = 0; k < la; ++k) av.push_back('a' + (i + k) % 26);
            for (int k = 0; k < lb; ++k) bv.push_back('a' + (i * 3 + k) % 26);
        }
        ao[rows] = (uint32_t)av.size(); bo[rows] = (uint32_t)bv.size();
        for (int T : {1, 2, 4, 8, 16, 32}) {
            std::atomic<int> ready{0}, failed{0};
            std::atomic<bool> go{false};
            std::vector<double> secs(T, 0.0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    strsim_ctx_t *ctx = nullptr;
                    std::vector<double> out(rows);
                    if (strsim_ctx_create(0, nullptr, &ctx) != STRSIM_OK) { failed++; ready++; return; }
                    for (int w = 0; w < 20; ++w) // warm-up: workspace, code object
                        (void)strsim_pairs_host(ctx, 0, ao.data(), av.data(), rows, bo.data(), bv.data(), rows, out.data(), rows);
                    ready++;
                    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int c = 0; c < calls; ++c)
                        if (strsim_pairs_host(ctx, 0, ao.data(), av.data(), rows, bo.data(), bv.data(), rows, out.data(), rows) != STRSIM_OK) { failed++; break; }
                    secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    strsim_ctx_destroy(ctx);
                });
            while (ready.load() < T) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            go.store(true, std::memory_order_release);
            for (auto &x : th) x.join();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            double mean = 0;
            for (double s : secs) mean += s / T;
            printf("rows %5d  threads %2d: %8.0f calls/s in all (%6.1f us per call and thread; %7.2f M pairs/s)%s\n", rows, T, (double)T * calls / wall, mean / calls * 1e6,
                   (double)T * calls * rows / wall / 1e6, failed.load() ? "  FAILED" : "");
        }
    }
    return 0;
}
