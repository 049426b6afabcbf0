// Micro-benchmark: is the VALU issue rate of long straight-line code bound by instruction fetch?  Same dependent integer
// work as (a) a small loop, (b) the same loop unrolled to ~40 KB of code, with 4-byte (VOP2) or 8-byte (VOP3) encodings.
//   hipcc --offload-arch=gfx950 -O3 bench_support/micro/fetch_issue.hip -o bench_support/micro/fetch_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// one step = 4 VALU ops on x (dependent chain), y is loop-invariant
template <int ENC>
__device__ __forceinline__ uint32_t step(uint32_t x, uint32_t y)
{
    if (ENC == 2) { // VOP2: 4-byte encodings
        asm volatile("v_xor_b32_e32 %0, %0, %1\n\tv_add_u32_e32 %0, %0, %1\n\tv_and_b32_e32 %0, %0, %1\n\tv_or_b32_e32 %0, %0, %1" : "+v"(x) : "v"(y));
    } else if (ENC == 3) { // VOP3: 8-byte encodings, three register operands
        asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96\n\tv_alignbit_b32 %0, %0, %1, 31\n\tv_bfe_i32 %0, %0, 3, 9\n\tv_bitop3_b32 %0, %0, %1, %1 bitop3:0xbe" : "+v"(x) : "v"(y));
    } else { // mixed like the Levenshtein column: 2 VOP3 + 2 VOP2
        asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96\n\tv_add_u32_e32 %0, %0, %1\n\tv_bfe_i32 %0, %0, 3, 9\n\tv_and_b32_e32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    return x;
}

template <int ENC, int UNROLL>
__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters)
{
    uint32_t x = seed + threadIdx.x, y = seed * 7u + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x = step<ENC>(x, y);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int ENC, int UNROLL>
static void run(int w)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4 * w;
    uint32_t *d; unsigned long long *c;
    (void)hipMalloc(&d, (size_t)blocks * 256); (void)hipMalloc(&c, (size_t)blocks * 16);
    const int iters = 400000 / UNROLL;
    hipLaunchKernelGGL((k<ENC, UNROLL>), dim3(blocks), dim3(64), 0, 0, d, c, 12345u, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<ENC, UNROLL>), dim3(blocks), dim3(64), 0, 0, d, c, 12345u, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long *h = new unsigned long long[2 * blocks];
    (void)hipMemcpy(h, c, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double ghz = cyc / rt * 0.1;
    const double ops = (double)blocks * iters * UNROLL * 4.0;
    printf("enc=%d code=%6d B waves/SIMD=%d: %.3f ms clock %.2f GHz -> %.3f VALU per cycle per SIMD\n", ENC,
           UNROLL * 4 * (ENC == 2 ? 4 : ENC == 3 ? 8 : 6), w, ms, ghz, ops / (cus * 4.0) / (ms * 1e-3 * ghz * 1e9));
    delete[] h; (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    for (int w : {1, 2, 4, 8}) { run<2, 4>(w); run<2, 2048>(w); run<3, 4>(w); run<3, 1024>(w); run<1, 4>(w); run<1, 1024>(w); }
    return 0;
}
