// strsim_lane_common.h -- small device helpers shared by the one-pair-per-lane kernels (strsim_kernels.hip,
// strsim_lane_stage.h).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "strsim_lane_core.h"

namespace strsim {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// 32-byte window vals[off, off+32) into w[0..7]; bytes at or beyond `total` read as 0.
__device__ __forceinline__ void load_window32(const uint8_t *__restrict__ vals, uint32_t off, uint32_t total,
                                              uint32_t (&w)[8])
{
    const uint8_t *p = vals + off;
    if (total - off >= 32u) {
        const u32x4_unaligned lo = *reinterpret_cast<const u32x4_unaligned *>(p);
        const u32x4_unaligned hi = *reinterpret_cast<const u32x4_unaligned *>(p + 16);
        w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w;
        w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
    } else {
        const uint32_t avail = total - off; // < 32
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            uint32_t v = 0u;
            if (avail >= 4u * d + 4u) {
                v = *reinterpret_cast<const u32_unaligned *>(p + 4 * d);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (avail > 4u * d + k) v |= (uint32_t)p[4 * d + k] << (8 * k);
            }
            w[d] = v;
        }
    }
}

// COLS_PER_TEST * ceil(max over the wave of v / COLS_PER_TEST), at least COLS_PER_TEST (v <= 32): a binary search
// with ballots, result in an SGPR
__device__ __forceinline__ uint32_t wave_max_rounded(uint32_t v)
{
    constexpr uint32_t C = (uint32_t)COLS_PER_TEST;
    uint32_t g = 0; // groups of C columns below the answer
#pragma unroll
    for (uint32_t half = 16u / C; half >= 1u; half >>= 1)
        if (__ballot(v > C * (g + half)) != 0ull) g += half;
    return C * (g + 1u);
}

#ifndef STRSIM_LANE_FULL_BARRIERS
#define STRSIM_LANE_FULL_BARRIERS 0
#endif
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's global loads and STORES
// (s_waitcnt vmcnt(0)), which k_lane_pairs does not need -- it communicates through LDS alone -- and which exposes the
// latency of a block's result stores at the next barrier.
__device__ __forceinline__ void lds_barrier()
{
#if STRSIM_LANE_FULL_BARRIERS
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

} // namespace strsim
