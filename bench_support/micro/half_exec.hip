// Micro-benchmark: does a wave64 vector instruction cost less when one half of EXEC is empty?  (gfx950: four SIMD-32 units per CU,
// a wave64 instruction takes two passes.)  A chain of full-rate ops (v_add / v_bitop3) and a chain of half-rate ones (v_bfe_i32)
// run under EXEC = all lanes, lanes 0..31 only, lanes 32..63 only, even lanes only, one lane.
// hipcc --offload-arch=gfx950 -O3 half_exec.hip -o half_exec && ./half_exec
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, int iters, unsigned long long mask)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x0 = seed + lane, x1 = seed * 3u + lane, x2 = seed * 5u + lane, x3 = seed * 7u + lane, y = seed ^ lane;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll 16
            for (int u = 0; u < 16; ++u) {
                if (KIND == 0) { // full-rate: add + bitop3 on four independent chains
                    x0 = __builtin_amdgcn_bitop3_b32(x0, y, x0 + (uint32_t)u, 0x96);
                    x1 = __builtin_amdgcn_bitop3_b32(x1, y, x1 + (uint32_t)u, 0x96);
                    x2 = __builtin_amdgcn_bitop3_b32(x2, y, x2 + (uint32_t)u, 0x96);
                    x3 = __builtin_amdgcn_bitop3_b32(x3, y, x3 + (uint32_t)u, 0x96);
                } else { // half-rate: v_bfe_i32 + xor
                    x0 ^= (uint32_t)__builtin_amdgcn_sbfe((int)x1, (unsigned)(u & 31), 1u);
                    x1 ^= (uint32_t)__builtin_amdgcn_sbfe((int)x2, (unsigned)((u + 5) & 31), 1u);
                    x2 ^= (uint32_t)__builtin_amdgcn_sbfe((int)x3, (unsigned)((u + 9) & 31), 1u);
                    x3 ^= (uint32_t)__builtin_amdgcn_sbfe((int)x0, (unsigned)((u + 13) & 31), 1u);
                }
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

template <int KIND>
static void run(const char *what, unsigned long long mask, int waves_per_simd)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4 * waves_per_simd, iters = 4000;
    uint32_t *d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 12345u, 50, mask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 12345u, iters, mask);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double insts = (double)waves_per_simd * iters * 16.0 * (KIND == 0 ? 8.0 : 8.0); // wave instructions per SIMD
    printf("%-10s exec %-12s waves/SIMD %d: %8.3f ms  -> %.2f ns per wave instruction per SIMD\n", KIND == 0 ? "full-rate" : "half-rate", what,
           waves_per_simd, ms, ms * 1e6 / insts);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 4, 8}) {
        run<0>("all", ~0ull, w);
        run<0>("low half", 0x00000000FFFFFFFFull, w);
        run<0>("high half", 0xFFFFFFFF00000000ull, w);
        run<0>("even lanes", 0x5555555555555555ull, w);
        run<0>("one lane", 1ull, w);
        run<1>("all", ~0ull, w);
        run<1>("low half", 0x00000000FFFFFFFFull, w);
        run<1>("one lane", 1ull, w);
    }
    return 0;
}
