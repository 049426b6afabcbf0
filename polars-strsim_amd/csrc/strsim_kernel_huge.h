// strsim_kernel_huge.h -- k_huge_pairs<M>: rows with a string beyond WAVE_CAP bytes, scratch in a global workspace; launched from strsim_ctx_synchronize().
// Included by strsim_kernels.hip inside namespace strsim, after the kernels in front of it ([r5] split out of strsim_kernels.hip
// along its seams, VERDICT r4 item 8: no behaviour change -- the translation unit's ISA is byte-identical before and after).
// Reference semantics: /root/reference/src/expressions/strsim.rs:125-345 (the cores cite their lines).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_huge_pairs: rows with a string longer than WAVE_CAP bytes, same algorithms with the scratch arrays in a
// global-memory workspace (3 * (cap + 64) words per wave).  Launched only when k_wave_pairs counted such rows.
// ------------------------------------------------------------------------------------------------
template <int MEASURE>
__global__ __launch_bounds__(64) void k_huge_pairs(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA,
                                                   uint64_t rowsA, const uint32_t *__restrict__ offB,
                                                   const uint8_t *__restrict__ valB, uint64_t rowsB,
                                                   double *__restrict__ out, uint64_t n, uint32_t *__restrict__ ws,
                                                   uint32_t cap)
{
    const uint32_t lane = lane_id();
    const bool bcastA = rowsA == 1, bcastB = rowsB == 1;
    const uint32_t totalA = offA[rowsA], totalB = offB[rowsB];
    // per wave: 64 words of front pad (wave_lev_stripes reads a little in front of its text), then the three arrays
    uint32_t *sA = ws + (uint64_t)blockIdx.x * HUGE_WS_WORDS(cap) + 64u;
    uint32_t *sB = sA + cap + 64u;
    uint32_t *aux = sB + cap + 64u;
    __shared__ uint32_t s_tab[MEASURE == LEVENSHTEIN ? 32 * 64 : 1];
    for (uint64_t base = (uint64_t)blockIdx.x * 64u; base < n; base += (uint64_t)gridDim.x * 64u) {
        const uint64_t rmine = base + lane;
        bool big = false;
        if (rmine < n) {
            const uint64_t ra = bcastA ? 0 : rmine, rb = bcastB ? 0 : rmine;
            big = (offA[ra + 1] - offA[ra]) > (uint32_t)WAVE_CAP || (offB[rb + 1] - offB[rb]) > (uint32_t)WAVE_CAP;
        }
        unsigned long long pending = __ballot(big);
        while (pending != 0ull) {
            const uint32_t src = (uint32_t)__builtin_ctzll(pending);
            pending &= pending - 1ull;
            const uint64_t row = base + src;
            const uint64_t ra = bcastA ? 0 : row, rb = bcastB ? 0 : row;
            const uint32_t a0 = uniform(offA[ra]), a1 = uniform(offA[ra + 1]);
            const uint32_t b0 = uniform(offB[rb]), b1 = uniform(offB[rb + 1]);
            double r;
            if constexpr (MEASURE == LEVENSHTEIN) {
                if (a1 != a0 && b1 != b0 && lev_fits_symbols(valA + a0, a1 - a0, valB + b0, b1 - b0)) continue; // k_wave_pairs did it
                r = huge_levenshtein(valA + a0, a1 - a0, valB + b0, b1 - b0, sA, sB, aux, s_tab);
            }
            else
                r = wave_row<MEASURE>(valA, a0, a1 - a0, totalA, valB, b0, b1 - b0, totalB, sA, sB, aux, cap, 4u * (cap + 64u));
            if (lane == 0u) out[row] = r;
        }
    }
}
