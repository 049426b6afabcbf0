// Micro-benchmark: what does rocprofv3's FETCH_SIZE count for k_lane_wide's access pattern?  
//
// The guide's gfx950 rule -- FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports exactly HALF the bytes of a wide coalesced 16 B/lane stream
// (128-byte requests tallied at 64) -- was applied to k_lane_wide too, whose row fetches are scattered 16-byte pieces: cfg3's
// "29.0 GB against 6.8 GB algorithmic" may be overstated by up to 2 x.  Every kernel here reads a KNOWN set of bytes ONCE, from a
// buffer far larger than the 256 MiB Infinity Cache, in one access shape each:
//   stream     lane i of the grid reads 16 bytes at 16 i                                   (the guide's case: expect 0.5)
//   rows64     4 consecutive lanes read the 4 x 16 bytes of one 64-byte window, windows at a random UNALIGNED byte offset inside
//              their own 256-byte slot (k_lane_wide's two-word pattern: coop_fetch<4>)
//   rows128    8 consecutive lanes, 128-byte windows inside 512-byte slots                 (the four-word pattern: coop_fetch<8>)
//   lines16    every lane reads 16 aligned bytes of its OWN 128-byte line, lines in a random order within a 1 MiB neighbourhood
// Per shape the program prints the bytes REQUESTED, and the bytes in the 32- / 64- / 128-byte aligned blocks those requests touch
// (what the memory side must have moved if it fetches at that granularity).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE ...      and      --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
// (bench_support/jobs/r6_fetch_calib.sh) and compare: the factor that turns FETCH_SIZE into bytes for each shape.
//   hipcc --offload-arch=gfx950 -O3 bench_support/micro/fetch_calib.hip -o bench_support/micro/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_calib_stream(const uint8_t *buf, uint64_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(buf + 16u * i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// LPR lanes per window of LPR x 16 bytes; window w starts at off[w] (any byte)
template <int LPR>
__global__ __launch_bounds__(256) void k_calib_rows(const uint8_t *buf, const uint64_t *off, uint64_t nwin, uint32_t *sink)
{
    uint32_t acc = 0;
    const uint32_t c = threadIdx.x & (LPR - 1);
    for (uint64_t w = ((uint64_t)blockIdx.x * 256u + threadIdx.x) / LPR; w < nwin; w += (uint64_t)gridDim.x * 256u / LPR) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(buf + off[w] + 16u * c);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void k_calib_lines16(const uint8_t *buf, const uint64_t *off, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(buf + off[i]);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

static uint64_t rng(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static void blocks(const char *name, const std::vector<uint64_t> &off, uint64_t len, double ms)
{
    // bytes in the aligned g-byte blocks the windows [off, off + len) touch (windows never share a block here: own slots)
    double req = (double)off.size() * len, b[3] = {0, 0, 0};
    const uint64_t g[3] = {32, 64, 128};
    for (uint64_t o : off)
        for (int q = 0; q < 3; ++q) b[q] += (double)(((o + len - 1) / g[q] - o / g[q] + 1) * g[q]);
    printf("%-8s requested %.0f B; touched at 32 B %.0f, at 64 B %.0f, at 128 B %.0f; kernel %.3f ms = %.0f GB/s of requested bytes\n", name, req, b[0], b[1], b[2], ms,
           req / ms / 1e6);
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "all";
    const uint64_t bytes = (argc > 2 ? strtoull(argv[2], nullptr, 10) : 4096ull) << 20; // 4 GiB: 16 x the Infinity Cache
    uint8_t *buf;
    uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes + 4096));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes + 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](auto launch) {
        launch(); // (warm-up: code object, clocks)
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms;
    };
    const int grid = 256 * 8;
    uint64_t seed = 42;
    if (!strcmp(which, "all") || !strcmp(which, "stream")) {
        const uint64_t n16 = bytes / 16;
        const double ms = timed([&] { hipLaunchKernelGGL(k_calib_stream, dim3(grid), dim3(256), 0, 0, buf, n16, sink); });
        printf("%-8s requested %.0f B (= touched at any granularity); kernel %.3f ms = %.0f GB/s\n", "stream", (double)n16 * 16, ms, (double)n16 * 16 / ms / 1e6);
    }
    auto rows = [&](const char *name, uint64_t len, uint64_t slot, auto kernel) {
        if (strcmp(which, "all") && strcmp(which, name)) return;
        const uint64_t nwin = bytes / slot;
        std::vector<uint64_t> off(nwin);
        for (uint64_t w = 0; w < nwin; ++w) off[w] = (w * slot + rng(seed) % (slot - len + 1)) & (len == 16 ? ~15ull : ~0ull);
        // a round's rows are not consecutive: shuffle inside neighbourhoods of 8 192 windows (a super's rows in key order)
        for (uint64_t base = 0; base < nwin; base += 8192) {
            const uint64_t m = nwin - base < 8192 ? nwin - base : 8192;
            for (uint64_t i = m - 1; i > 0; --i) { const uint64_t j = rng(seed) % (i + 1); std::swap(off[base + i], off[base + j]); }
        }
        uint64_t *d;
        CHECK(hipMalloc(&d, nwin * 8));
        CHECK(hipMemcpy(d, off.data(), nwin * 8, hipMemcpyHostToDevice));
        const double ms = timed([&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, buf, d, nwin, sink); });
        blocks(name, off, len, ms);
        printf("         (+ the offset list itself: %.0f B streamed, 8 bytes per window)\n", (double)nwin * 8);
        CHECK(hipFree(d));
    };
    rows("rows64", 64, 256, k_calib_rows<4>);
    rows("rows128", 128, 512, k_calib_rows<8>);
    rows("lines16", 16, 128, k_calib_lines16); // (16 aligned bytes somewhere in the lane's own line: slot 128, len 16 -- offsets rounded below)
    CHECK(hipDeviceSynchronize());
    return 0;
}
