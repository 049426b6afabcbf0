// Aggregate rate of CONCURRENT small calls THROUGH THE PLUGIN ABI (what an engine's worker threads do in a group_by of many small groups),
// without Python in the loop: T threads, K calls of `rows` rows each into _polars_plugin_levenshtein, inputs as Arrow "u" (Utf8, i32
// offsets) series built once per thread.  Run with POLARS_STRSIM_COALESCE=0 and =1 to see what combining concurrent calls buys.
//   g++ -O2 -std=c++17 -pthread -Iinclude bench_support/micro/plugin_small_threads.cpp -o bench_support/micro/plugin_small_threads \
//       polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,$PWD/polars-strsim_amd/polars_strsim
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "polars_plugin_abi.h"

namespace {

struct Column { // one Utf8 column, exported afresh for every call (the callee owns and releases what it is handed)
    std::vector<int32_t> off;
    std::vector<uint8_t> val;
    const void *bufs[3];
    ArrowArray arr;
    ArrowArray *arrp;
    ArrowSchema sch;
    static void rel_a(ArrowArray *a) { a->release = nullptr; }
    static void rel_s(ArrowSchema *s) { s->release = nullptr; }
    static void rel_e(SeriesExport *e) { e->release = nullptr; }
    void export_to(SeriesExport *e, const char *name)
    {
        bufs[0] = nullptr; bufs[1] = off.data(); bufs[2] = val.data();
        memset(&arr, 0, sizeof arr);
        arr.length = (int64_t)off.size() - 1; arr.null_count = 0; arr.offset = 0; arr.n_buffers = 3; arr.n_children = 0;
        arr.buffers = bufs; arr.release = rel_a;
        arrp = &arr;
        memset(&sch, 0, sizeof sch);
        sch.format = "u"; sch.name = name; sch.flags = 2; sch.release = rel_s;
        e->field = &sch; e->arrays = &arrp; e->len = 1; e->release = rel_e; e->private_data = nullptr;
    }
};

} // namespace

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? atoi(argv[1]) : 2000;
    uint64_t st[4];
    for (int rows : {100, 1000, 4000}) {
        for (int T : {1, 4, 8, 16, 32}) {
            std::atomic<int> ready{0}, failed{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    Column a, b;
                    a.off.push_back(0); b.off.push_back(0);
                    for (int i = 0; i < rows; ++i) {
                        const int la = 1 + (i * 7 + t) % 32, lb = 1 + (i * 11 + t) % 32;
                        for (int k = 0; k < la; ++k) a.val.push_back((uint8_t)('a' + (i + k) % 26));
                        for (int k = 0; k < lb; ++k) b.val.push_back((uint8_t)('a' + (i * 3 + k) % 26));
                        a.off.push_back((int32_t)a.val.size()); b.off.push_back((int32_t)b.val.size());
                    }
                    a.val.resize(a.val.size() + 64); b.val.resize(b.val.size() + 64);
                    CallerContext cc{1}; // PARALLEL: the engine is already parallel (a group_by's calls)
                    double first = -1.0;
                    auto one = [&]() -> bool {
                        SeriesExport in[2], ret;
                        a.export_to(&in[0], "a");
                        b.export_to(&in[1], "b");
                        memset(&ret, 0, sizeof ret);
                        _polars_plugin_levenshtein(in, 2, nullptr, 0, &ret, &cc);
                        if (!ret.release) { fprintf(stderr, "%s\n", _polars_plugin_get_last_error_message()); return false; }
                        const double v = static_cast<const double *>(ret.arrays[0]->buffers[1])[rows - 1];
                        if (first < 0) first = v; else if (v != first) return false; // (the same rows every call: the same last value)
                        for (size_t c = 0; c < ret.len; ++c) if (ret.arrays[c]->release) ret.arrays[c]->release(ret.arrays[c]);
                        if (ret.field && ret.field->release) ret.field->release(ret.field);
                        ret.release(&ret);
                        return true;
                    };
                    for (int w = 0; w < 20; ++w) if (!one()) failed++;
                    ready++;
                    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                    for (int c = 0; c < calls; ++c) if (!one()) { failed++; break; }
                });
            while (ready.load() < T) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            go.store(true, std::memory_order_release);
            for (auto &x : th) x.join();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            _polars_plugin_strsim_coalesce_stats(st);
            printf("rows %5d  threads %2d: %8.0f calls/s in all (%6.1f us per call and thread; %7.2f M pairs/s)  [combined launches %llu carrying %llu calls, most %llu; direct %llu]%s\n",
                   rows, T, (double)T * calls / wall, wall / calls * 1e6, (double)T * calls * rows / wall / 1e6, (unsigned long long)st[0], (unsigned long long)st[1],
                   (unsigned long long)st[2], (unsigned long long)st[3], failed.load() ? "  FAILED" : "");
        }
    }
    return 0;
}
