// synth.hip -- device kernels + host loops for the synthetic column generator (see synth.h).
// C ABI (ctypes from bench.py / tests):
//   synth_lengths_device(cfg..., row0, n, lenA*, lenB*, stream)  -> uint32 lengths per row (device)
//   synth_fill_device(cfg..., row0, n, offA*, valA*, offB*, valB*, stream)  (offsets = exclusive scan of lengths, device)
//   synth_lengths_host / synth_fill_host: same on host memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "synth.h"

using namespace synth;

__global__ void k_lengths(Config c, uint64_t row0, uint64_t n, uint32_t *lenA, uint32_t *lenB)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t la, lb;
        row_lengths(c, row0 + i, la, lb);
        lenA[i] = la;
        lenB[i] = lb;
    }
}

__global__ void k_fill(Config c, uint64_t row0, uint64_t n, const uint32_t *offA, uint8_t *valA, const uint32_t *offB,
                       uint8_t *valB)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t a[MAX_LEN + 4], b[MAX_LEN + 4];
        row_fill(c, row0 + i, a, b);
        const uint32_t a0 = offA[i], la = offA[i + 1] - a0;
        const uint32_t b0 = offB[i], lb = offB[i + 1] - b0;
        for (uint32_t k = 0; k < la; ++k) valA[(uint64_t)a0 + k] = a[k];
        for (uint32_t k = 0; k < lb; ++k) valB[(uint64_t)b0 + k] = b[k];
    }
}

#define SYNTH_API extern "C" __attribute__((visibility("default")))

SYNTH_API int synth_lengths_device(uint64_t seed, int law, uint32_t lo, uint32_t hi, uint64_t row0, uint64_t n,
                                   uint32_t *lenA, uint32_t *lenB, void *stream)
{
    if (hi > (uint32_t)MAX_LEN || lo > hi) return -1;
    if (n == 0) return 0;
    Config c{seed, law, lo, hi};
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_lengths, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c, row0, n, lenA, lenB);
    return (int)hipGetLastError();
}

SYNTH_API int synth_fill_device(uint64_t seed, int law, uint32_t lo, uint32_t hi, uint64_t row0, uint64_t n,
                                const uint32_t *offA, uint8_t *valA, const uint32_t *offB, uint8_t *valB, void *stream)
{
    if (hi > (uint32_t)MAX_LEN || lo > hi) return -1;
    if (n == 0) return 0;
    Config c{seed, law, lo, hi};
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c, row0, n, offA, valA, offB, valB);
    return (int)hipGetLastError();
}

SYNTH_API int synth_lengths_host(uint64_t seed, int law, uint32_t lo, uint32_t hi, uint64_t row0, uint64_t n,
                                 uint32_t *lenA, uint32_t *lenB)
{
    if (hi > (uint32_t)MAX_LEN || lo > hi) return -1;
    Config c{seed, law, lo, hi};
    for (uint64_t i = 0; i < n; ++i) row_lengths(c, row0 + i, lenA[i], lenB[i]);
    return 0;
}

SYNTH_API int synth_fill_host(uint64_t seed, int law, uint32_t lo, uint32_t hi, uint64_t row0, uint64_t n,
                              const uint32_t *offA, uint8_t *valA, const uint32_t *offB, uint8_t *valB)
{
    if (hi > (uint32_t)MAX_LEN || lo > hi) return -1;
    Config c{seed, law, lo, hi};
    uint8_t a[MAX_LEN + 4], b[MAX_LEN + 4];
    for (uint64_t i = 0; i < n; ++i) {
        row_fill(c, row0 + i, a, b);
        const uint32_t a0 = offA[i], la = offA[i + 1] - a0;
        const uint32_t b0 = offB[i], lb = offB[i + 1] - b0;
        for (uint32_t k = 0; k < la; ++k) valA[(uint64_t)a0 + k] = a[k];
        for (uint32_t k = 0; k < lb; ++k) valB[(uint64_t)b0 + k] = b[k];
    }
    return 0;
}
