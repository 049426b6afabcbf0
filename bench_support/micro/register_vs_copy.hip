// f1 evidence (SURVEY 8f / VERDICT r1 item 6): how should pageable Arrow buffers reach the GPU?
//   (a) hipHostRegister the caller's pageable buffer, hipMemcpyAsync H2D straight from it, hipHostUnregister
//   (b) hipMemcpy H2D from the pageable buffer (the runtime stages it itself)
//   (c) memcpy on T host threads into pinned staging, then hipMemcpyAsync H2D   (what the plugin does, minus the gather)
// for a buffer of `MB` megabytes (default 1600 = the views of one 100 M-row column).
//   hipcc -O3 -std=c++17 bench_support/micro/register_vs_copy.hip -o bench_support/micro/register_vs_copy -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1600;
    const unsigned T = argc > 2 ? atoi(argv[2]) : 16;
    const size_t bytes = mb << 20;
    char *src = static_cast<char *>(aligned_alloc(4096, bytes));
    {   // first touch by several threads (as a producer like Polars would leave it)
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back([=] { memset(src + bytes * t / T, (int)t + 1, bytes * (t + 1) / T - bytes * t / T); });
        for (auto &x : th) x.join();
    }
    void *dev, *pin;
    CK(hipMalloc(&dev, bytes));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipMemcpy(dev, pin, 1 << 20, hipMemcpyHostToDevice)); // warm-up
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipHostUnregister(src));
        double t3 = now();
        printf("(a) register %.1f ms + H2D %.1f ms (%.1f GB/s) + unregister %.1f ms = %.1f ms -> %.1f GB/s overall\n", (t1 - t0) * 1e3,
               (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3, bytes / (t3 - t0) / 1e9);
        t0 = now();
        CK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice));
        t1 = now();
        printf("(b) hipMemcpy from pageable: %.1f ms -> %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        t0 = now();
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t)
                th.emplace_back([=] { memcpy((char *)pin + bytes * t / T, src + bytes * t / T, bytes * (t + 1) / T - bytes * t / T); });
            for (auto &x : th) x.join();
        }
        t1 = now();
        CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t2 = now();
        printf("(c) %u-thread memcpy to pinned %.1f ms (%.1f GB/s) + H2D %.1f ms (%.1f GB/s) = %.1f ms -> %.1f GB/s if serial, %.1f GB/s pipelined\n",
               T, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t2 - t0) * 1e3,
               bytes / (t2 - t0) / 1e9, bytes / ((t1 - t0) > (t2 - t1) ? (t1 - t0) : (t2 - t1)) / 1e9);
    }
    return 0;
}
