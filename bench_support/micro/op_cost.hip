// Micro-benchmark: issue cost (cycles of SIMD time per wave64 instruction) of the individual VALU opcodes the lane cores use,
// gfx950.  Eight waves per SIMD, 512 copies of the instruction unrolled (dependent on the previous result through %0), no memory.
//   hipcc --offload-arch=gfx950 -O3 bench_support/micro/op_cost.hip -o bench_support/micro/op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OPS(X) \
    X(0, "v_xor_b32_e32 %0, %0, %1") \
    X(1, "v_add_u32_e32 %0, %0, %1") \
    X(2, "v_lshlrev_b32_e32 %0, 1, %0") \
    X(3, "v_and_b32_e32 %0, %0, %1") \
    X(4, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96") \
    X(5, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96") \
    X(6, "v_bfe_i32 %0, %0, 3, 1") \
    X(7, "v_bfe_u32 %0, %0, 3, 9") \
    X(8, "v_alignbit_b32 %0, %0, %1, 31") \
    X(9, "v_alignbyte_b32 %0, %0, %1, 3") \
    X(10, "v_lshl_or_b32 %0, %0, 1, %1") \
    X(11, "v_and_or_b32 %0, %0, %1, %2") \
    X(12, "v_lshl_add_u32 %0, %0, 1, %1") \
    X(13, "v_add3_u32 %0, %0, %1, %2") \
    X(14, "v_perm_b32 %0, %0, %1, %2") \
    X(15, "v_bcnt_u32_b32 %0, %0, %1") \
    X(16, "v_cndmask_b32_e32 %0, %0, %1, vcc") \
    X(17, "v_mov_b32_e32 %0, %1") \
    X(18, "v_or3_b32 %0, %0, %1, %2") \
    X(19, "v_xad_u32 %0, %0, %1, %2") \
    X(20, "v_add_lshl_u32 %0, %0, %1, 1") \
    X(21, "v_bfi_b32 %0, %0, %1, %2") \
    X(22, "v_ashrrev_i32_e32 %0, 31, %0") \
    X(23, "v_lshrrev_b32_e32 %0, 1, %0") \
    X(24, "v_sub_u32_e32 %0, %0, %1") \
    X(25, "v_not_b32_e32 %0, %0") \
    X(26, "v_pk_add_u16 %0, %0, %1") \
    X(27, "v_mad_u32_u24 %0, %0, %1, %2") \
    X(28, "v_mul_u32_u24_e32 %0, %0, %1") \
    X(29, "v_lshlrev_b64 %3, 1, %3") \
    X(30, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf") \
    X(31, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96") \
    X(32, "v_ashrrev_i32_sdwa %0, %2, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(33, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD") \
    X(34, "v_cmp_eq_u32_e32 vcc, %0, %1") \
    X(35, "v_cndmask_b32_e64 %0, %0, %1, s[2:3]") \
    X(36, "v_min_u32_e32 %0, %0, %1") \
    X(37, "v_or_b32_e32 %0, %0, %1") \
    X(38, "v_subrev_u32_e32 %0, %0, %1") \
    X(39, "v_lshrrev_b32_e32 %0, %1, %0") \
    X(40, "v_lshlrev_b32_e32 %0, %1, %0") \
    X(41, "v_lshlrev_b16_e32 %0, 1, %0") \
    X(42, "v_mov_b32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1") \
    X(43, "v_readfirstlane_b32 s4, %0") \
    X(44, "v_bfe_i32 %0, %1, 3, 1")

template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *clk, uint32_t seed, int iters)
{
    uint32_t x = seed + threadIdx.x, y = seed * 7u + threadIdx.x, z = seed * 13u + threadIdx.x * 3u;
    unsigned long long q = ((unsigned long long)x << 32) | y;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 512; ++u) {
#define X(ID, TXT) if (OP == ID) asm volatile(TXT : "+v"(x), "+v"(y), "+v"(z), "+v"(q) : : "vcc", "s4", "s2", "s3");
            OPS(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = x ^ y ^ z ^ (uint32_t)q ^ (uint32_t)(q >> 32);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static const char *names[] = {
#define X(ID, TXT) TXT,
    OPS(X)
#undef X
};

template <int OP>
static void run(int w)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4 * w;
    uint32_t *d; unsigned long long *c;
    (void)hipMalloc(&d, (size_t)blocks * 256); (void)hipMalloc(&c, (size_t)blocks * 16);
    const int iters = 200;
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, d, c, 12345u, 4);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, d, c, 12345u, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long *h = new unsigned long long[2 * blocks];
    (void)hipMemcpy(h, c, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double ghz = cyc / rt * 0.1;
    const double ops = (double)blocks * iters * 512.0;
    const double ipc = ops / (cus * 4.0) / (ms * 1e-3 * ghz * 1e9);
    printf("w=%d  %-58s %6.2f cycles of SIMD time per instruction (%.3f per cycle)\n", w, names[OP], 1.0 / ipc, ipc);
    delete[] h; (void)hipFree(d); (void)hipFree(c);
}

template <int OP>
static void all()
{
    run<OP>(8);
    if constexpr (OP + 1 < 45) all<OP + 1>();
}

int main()
{
    all<0>();
    run<4>(2); run<0>(2); run<8>(2); run<6>(2);
    return 0;
}
