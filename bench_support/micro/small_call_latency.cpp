// Per-call latency of the C ABI for tiny calls, without any Python in the loop.
//   g++ -O2 -Iinclude bench_support/micro/small_call_latency.cpp -o bench_support/micro/small_call_latency \
//       polars-strsim_amd/polars_strsim/libpolars_strsim_amd.so -Wl,-rpath,'$ORIGIN/../../polars-strsim_amd/polars_strsim'
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "strsim_amd.h"

int main()
{
    strsim_ctx_t *ctx = nullptr;
    if (strsim_ctx_create(0, nullptr, &ctx) != STRSIM_OK) { printf("%s\n", strsim_last_error_message()); return 1; }
    for (int rows : {1, 100, 1000, 10000}) {
        std::vector<uint32_t> ao(rows + 1), bo(rows + 1);
        std::vector<uint8_t> av, bv;
        for (int i = 0; i < rows; ++i) {
            const int la = 1 + (i * 7) % 32, lb = 1 + (i * 11) % 32;
            ao[i] = (uint32_t)av.size(); bo[i] = (uint32_t)bv.size();
            for (int k = 0; k < la; ++k) av.push_back('a' + (i + k) % 26);
            for (int k = 0; k < lb; ++k) bv.push_back('a' + (i * 3 + k) % 26);
        }
        ao[rows] = (uint32_t)av.size(); bo[rows] = (uint32_t)bv.size();
        std::vector<double> out(rows);
        for (int m : {0, 1}) {
            std::vector<double> ts;
            for (int it = 0; it < 300; ++it) {
                const auto t0 = std::chrono::steady_clock::now();
                if (strsim_pairs_host(ctx, m, ao.data(), av.data(), rows, bo.data(), bv.data(), rows, out.data(), rows) != STRSIM_OK) {
                    printf("%s\n", strsim_last_error_message());
                    return 1;
                }
                ts.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            std::sort(ts.begin(), ts.end());
            printf("measure %d rows %6d: strsim_pairs_host median %.1f us, min %.1f us (out[0] = %.4f, rows on the wave kernel %llu)\n", m, rows, ts[ts.size() / 2], ts[0], out[0], (unsigned long long)strsim_ctx_last_wave_rows(ctx));
        }
    }
    strsim_ctx_destroy(ctx);
    return 0;
}
