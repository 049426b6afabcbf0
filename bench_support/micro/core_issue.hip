// Micro-benchmark: issue rate of the REAL Levenshtein lane core (build_planes<5> + lev_myers32<5>, 32 columns, everything in
// registers, no memory traffic) at 1..8 waves per SIMD on gfx950.  Answers "what can a SIMD issue of THIS instruction mix".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipolars-strsim_amd/csrc bench_support/micro/core_issue.hip -o /tmp/core_issue && /tmp/core_issue
// VALU_PER_ITER: count the v_* instructions of one loop iteration in the disassembly (printed by the Makefile-less recipe in
// DESIGN.md) and pass it with -DVALU_PER_ITER=...; the program prints wave-instructions per cycle per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "strsim_lane_core.h"

#ifndef CORE
#define CORE 0
#endif
#ifndef TMIN
#define TMIN 29u
#endif
#ifndef VALU_PER_ITER
#define VALU_PER_ITER 800
#endif

__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters)
{
    using namespace strsim;
    uint32_t wa[8], wb[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        wa[d] = (seed * (2 * d + 3) + threadIdx.x * 0x01010101u) & 0x1F1F1F1Fu;
        wb[d] = (seed * (2 * d + 5) + threadIdx.x * 0x01000193u) & 0x1F1F1F1Fu;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        uint32_t P[5];
        build_planes<5>(wb, P);
#if CORE == 0
        const uint32_t dist = lev_myers32<5>(wa, 32u, 32u, P, 32u);
#elif CORE == 1
        const uint32_t dist = lev_myers32_snap<5>(wa, 29u + (threadIdx.x & 3u), TMIN, 32u, P, 32u);
#else
        const uint32_t dist = P[0] ^ P[1] ^ P[2] ^ P[3] ^ P[4]; // planes only
#endif
        acc += dist;
        wa[0] ^= dist; // the next iteration depends on this one
        wb[7] += dist;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int w : {2, 4, 8}) {
        const int blocks = cus * 4 * w; // one wave per block
        uint32_t *d;
        unsigned long long *c;
        (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
        (void)hipMalloc(&c, (size_t)blocks * 16);
        const int iters = 4000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, c, 12345u, 200);
        (void)hipDeviceSynchronize();
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, c, 12345u, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        unsigned long long *h = new unsigned long long[2 * blocks];
        (void)hipMemcpy(h, c, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
        cyc /= blocks; rt /= blocks;
        const double ghz = cyc / rt * 0.1;
        const double ops = (double)blocks * iters * (double)VALU_PER_ITER;
        printf("core %d waves/SIMD=%d: %.3f ms, clock %.2f GHz, %.0f cycles of SIMD time per iteration (= per 64 pairs of 32 columns), %.3f VALU per cycle per SIMD\n",
               CORE, w, ms, ghz, ms * 1e-3 * ghz * 1e9 / ((double)iters * w), ops / (cus * 4.0) / (ms * 1e-3 * ghz * 1e9));
        delete[] h;
        (void)hipFree(d); (void)hipFree(c);
    }
    return 0;
}
