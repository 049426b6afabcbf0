// Micro-benchmark: cost of fetching a 32-byte window at an arbitrary byte offset from LDS on gfx950, per wave-round
// (64 lanes, one window each, random offsets), for the ways the lane kernel could do it.
//   hipcc --offload-arch=gfx950 -O3 bench_support/micro/lds_window.hip -o bench_support/micro/lds_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4_a __attribute__((ext_vector_type(4), aligned(16)));
typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32_u __attribute__((aligned(1)));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[10240 + 64];
    for (uint32_t i = threadIdx.x; i < (10240 + 64) / 4; i += 256) reinterpret_cast<uint32_t *>(buf)[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc = 0, x = seed * 977u + threadIdx.x * 7919u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t off = (x >> 8) % 10200u; // random byte offset
        const uint8_t *p = buf + off;
        if (MODE == 0) { // two unaligned 16-byte reads (what k_lane_stage does)
            const u32x4_u a = *reinterpret_cast<const u32x4_u *>(p), b = *reinterpret_cast<const u32x4_u *>(p + 16);
            acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
        } else if (MODE == 1) { // four unaligned 8-byte reads
            uint32_t t = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const u32x2_u a = *reinterpret_cast<const u32x2_u *>(p + 8 * q); t ^= a.x ^ a.y; }
            acc += t;
        } else if (MODE == 2) { // eight unaligned 4-byte reads
            uint32_t t = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) t ^= *reinterpret_cast<const u32_u *>(p + 4 * q);
            acc += t;
        } else if (MODE == 3) { // three aligned 16-byte reads covering the window
            const uint8_t *q = buf + (off & ~15u);
            const u32x4_a a = *reinterpret_cast<const u32x4_a *>(q), b = *reinterpret_cast<const u32x4_a *>(q + 16), c = *reinterpret_cast<const u32x4_a *>(q + 32);
            acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w;
        } else if (MODE == 4) { // nine dword-aligned dwords (then 8 v_alignbyte would follow)
            const uint8_t *q = buf + (off & ~3u);
            uint32_t t = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const u32x2_a4 a = *reinterpret_cast<const u32x2_a4 *>(q + 8 * j); t ^= a.x ^ a.y; }
            t ^= *reinterpret_cast<const uint32_t *>(q + 32);
            acc += t;
        } else if (MODE == 6) { // two 16-byte reads + one dword at a DWORD-aligned (not 16-byte aligned) address: 9 dwords
            const uint8_t *q = buf + (off & ~3u);
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            u32x4_a4 a, b;
            uint32_t c;
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)q) : "memory");
            acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c;
        } else if (MODE == 7) { // four 8-byte reads + one dword at a dword-aligned address
            const uint8_t *q = buf + (off & ~3u);
            typedef uint32_t u32x2_a4b __attribute__((ext_vector_type(2), aligned(4)));
            u32x2_a4b a0, a1, a2, a3;
            uint32_t c;
            asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:8\n\tds_read_b64 %2, %5 offset:16\n\tds_read_b64 %3, %5 offset:24\n\tds_read_b32 %4, %5 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(c) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)q) : "memory");
            acc += a0.x ^ a0.y ^ a1.x ^ a1.y ^ a2.x ^ a2.y ^ a3.x ^ a3.y ^ c;
        } else { // two aligned 16-byte reads (the floor: what an aligned window would cost)
            const uint8_t *q = buf + (off & ~15u);
            const u32x4_a a = *reinterpret_cast<const u32x4_a *>(q), b = *reinterpret_cast<const u32x4_a *>(q + 16);
            acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
static void run(int wg_per_cu, const char *name)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * wg_per_cu;
    uint32_t *d; unsigned long long *c;
    (void)hipMalloc(&d, (size_t)blocks * 1024); (void)hipMalloc(&c, (size_t)blocks * 16);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, c, 12345u, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, c, 12345u, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long *h = new unsigned long long[2 * blocks];
    (void)hipMemcpy(h, c, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double ghz = cyc / rt * 0.1;
    // CU cycles per wave-window-fetch: wall cycles / (iters * waves per CU)
    printf("%-44s WG/CU=%d: %7.1f CU cycles per 64-lane window fetch\n", name, wg_per_cu, ms * 1e-3 * ghz * 1e9 / ((double)iters * wg_per_cu * 4));
    delete[] h; (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    for (int w : {1, 4}) {
        run<0>(w, "2 x ds_read_b128, byte-aligned");
        run<1>(w, "4 x ds_read_b64, byte-aligned");
        run<2>(w, "8 x ds_read_b32, byte-aligned");
        run<3>(w, "3 x ds_read_b128, 16-byte aligned");
        run<4>(w, "4 x ds_read2_b32 + ds_read_b32, dword-aligned");
        run<5>(w, "2 x ds_read_b128, 16-byte aligned (floor)");
        run<6>(w, "2 x ds_read_b128 + ds_read_b32, dword-aligned");
        run<7>(w, "4 x ds_read_b64 + ds_read_b32, dword-aligned");
    }
    return 0;
}
