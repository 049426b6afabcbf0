// Micro-benchmark: the Levenshtein lane core with match masks from the per-lane LDS tables (strsim_lane_lut.h) against the
// bit-fill masks (strsim_lane_core.h), everything else in registers, at 2 / 3 / 4 / 5 waves per SIMD on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipolars-strsim_amd/csrc bench_support/micro/core_lut.hip -o /tmp/core_lut && /tmp/core_lut
// Prints cycles of SIMD time per iteration (= per 64 pairs of 32 columns, planes + tables included) and checks that both
// cores return the same distances.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "strsim_lane_lut.h"

template <int CORE>
__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters)
{
    using namespace strsim;
    __shared__ __attribute__((aligned(4096))) uint32_t s_lut[LUT_ENTRIES * 64];
    uint32_t wa[8], wb[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        wa[d] = (seed * (2 * d + 3) + threadIdx.x * 0x01010101u) & 0x1F1F1F1Fu;
        wb[d] = (seed * (2 * d + 5) + threadIdx.x * 0x01000193u) & 0x1F1F1F1Fu;
    }
    EqLut t;
    const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void *)(&s_lut[0]));
    t.lane4 = threadIdx.x * 4u;
    t.krep = (base >> 8) * 0x01010101u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        uint32_t P[5];
        build_planes<5>(wb, P);
        uint32_t dist;
        if (CORE == 1) {
            dist = lev_myers32_snap<5>(wa, 29u + (threadIdx.x & 3u), 29u, 32u, P, 32u);
        } else {
            lut_build<5>(t, P, 0xFFFFFFFFu);
            dist = lev_myers32_lut<5>(t, wa, 29u + (threadIdx.x & 3u), 29u, 32u, P, 32u);
        }
        acc += dist;
        wa[0] ^= dist & 0x1Fu; // the next iteration depends on this one
        wb[7] = (wb[7] + dist) & 0x1F1F1F1Fu;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int CORE>
static uint32_t run(int cus, int w, uint32_t *first)
{
    const int blocks = cus * 4 * w; // one wave per block
    uint32_t *d;
    unsigned long long *c;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    (void)hipMalloc(&c, (size_t)blocks * 16);
    const int iters = 4000;
    hipLaunchKernelGGL(k<CORE>, dim3(blocks), dim3(64), 0, 0, d, c, 12345u, 200);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<CORE>, dim3(blocks), dim3(64), 0, 0, d, c, 12345u, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long *h = new unsigned long long[2 * blocks];
    (void)hipMemcpy(h, c, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    uint32_t ho[64];
    (void)hipMemcpy(ho, d, sizeof ho, hipMemcpyDeviceToHost);
    uint32_t sum = 0;
    for (int i = 0; i < 64; ++i) sum = sum * 31u + ho[i];
    *first = sum;
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    cyc /= blocks; rt /= blocks;
    const double ghz = cyc / rt * 0.1;
    printf("core %s waves/SIMD=%d: %.3f ms, clock %.2f GHz, %.0f cycles of SIMD time per iteration (64 pairs x 32 columns)\n",
           CORE == 1 ? "bit-fill" : "lut     ", w, ms, ghz, ms * 1e-3 * ghz * 1e9 / ((double)iters * w));
    delete[] h;
    (void)hipFree(d); (void)hipFree(c);
    return sum;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int w : {2, 3, 4, 5}) {
        uint32_t s1, s2;
        run<1>(cus, w, &s1);
        run<2>(cus, w, &s2);
        printf("   results %s (%08x %08x)\n", s1 == s2 ? "agree" : "DIFFER", s1, s2);
    }
    return 0;
}
