// CPU harness for polars-strsim_amd/csrc/strsim_lane_core.h: runs the exact per-lane arithmetic of
// the gfx950 lane-per-pair kernels on the host (match table in a plain array instead of an LDS
// column) so tests/ can compare it with the oracle without a GPU.  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include "strsim_lane_core.h"

using namespace strsim;

struct ArrayPeq {
    const uint32_t *tab;
    uint32_t operator()(uint32_t c) const { return tab[c & 127u]; }
};

template <int M>
static double run(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb)
{
    uint32_t wa[8] = {0}, wb[8] = {0};
    // garbage beyond the length, like the kernel's unmasked 32-byte window
    std::memset(wa, 0x5a, sizeof wa);
    std::memset(wb, 0x33, sizeof wb);
    std::memcpy(wa, a, la);
    std::memcpy(wb, b, lb);
    uint32_t tab[128] = {0};
    if (lane_needs_table(la, lb)) {
        const uint32_t s = lane_peq_shift<M>(lb);
        for (uint32_t j = 0; j < lb; ++j) tab[lane_byte(wb, (int)j)] |= (1u << j) << s;
    }
    ArrayPeq peq{tab};
    return lane_pair_result<M>(wa, la, wb, lb, peq);
}

extern "C" double harness_lane_pair(int measure, const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb)
{
    switch (measure) {
    case LEVENSHTEIN: return run<LEVENSHTEIN>(a, la, b, lb);
    case JARO: return run<JARO>(a, la, b, lb);
    case JARO_WINKLER: return run<JARO_WINKLER>(a, la, b, lb);
    case JACCARD: return run<JACCARD>(a, la, b, lb);
    default: return run<SORENSEN_DICE>(a, la, b, lb);
    }
}
