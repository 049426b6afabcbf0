// Micro-benchmark: VALU issue rate of dependent vs independent integer chains at 1..8 waves per SIMD (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t x[CHAINS], y = seed + threadIdx.x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = seed * (c + 3) + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_amdgcn_bitop3_b32(x[c], y, x[c] + (uint32_t)u, 0x96) + x[c]; // 3 dependent ops
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r ^= x[c];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int CHAINS>
static void run(int waves_per_simd)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 4 * waves_per_simd; // one wave per block
    uint32_t *d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    const int iters = 20000 / CHAINS;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, 12345u, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, d, 12345u, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * iters * 8.0 * CHAINS * 3.0; // wave-level VALU ops (add, bitop3, add)
    const double per_simd_per_cycle = ops / (cus * 4.0) / (ms * 1e-3 * 2.4e9);
    printf("chains=%d waves/SIMD=%d: %.3f ms, %.3f VALU ops per cycle per SIMD (at 2.4 GHz)\n", CHAINS, waves_per_simd, ms, per_simd_per_cycle);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 2, 3, 4, 6, 8}) run<1>(w);
    for (int w : {1, 2, 4, 8}) run<2>(w);
    for (int w : {1, 2, 4}) run<4>(w);
    return 0;
}
