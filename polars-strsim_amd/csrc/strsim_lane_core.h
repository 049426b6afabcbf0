// strsim_lane_core.h -- per-lane (one string pair per SIMD lane) bit-parallel cores for strings of
// <= 32 ASCII bytes.  Host/device portable so the exact arithmetic the gfx950 kernels run can also be
// exercised by a CPU harness in tests/ (there is no GPU in the build container).
//
// Every routine sees one pair:
//   * the "pattern" string P (<= 32 bytes) has already been turned into a match table
//     peq(c) = bitmask of the positions of byte c in P   (an LDS column per lane on the GPU);
//   * the "text" string T sits in eight 32-bit registers wt[0..7] (little-endian bytes, bytes at or
//     beyond lt are don't-care).
//
// Semantics restated (reference = /root/reference/src/expressions/strsim.rs):
//   lev_myers32      -> integer edit distance of Levenshtein::compute      (:141-160)
//   jaro_match32     -> (m, t) of Jaro::compute                            (:200-237)
//   multiset_isect32 -> sum of min(countA[c], countB[c]) of Jaccard/Dice   (:297-305, :333-341)
// and the f64 epilogues, in the reference's operation order (compile with -ffp-contract=off).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define STRSIM_HD __host__ __device__ __forceinline__
#else
#define STRSIM_HD inline
#endif

namespace strsim {

enum Measure : int { LEVENSHTEIN = 0, JARO = 1, JARO_WINKLER = 2, JACCARD = 3, SORENSEN_DICE = 4 };

STRSIM_HD uint32_t lane_byte(const uint32_t (&w)[8], int j) { return (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu; }

STRSIM_HD uint32_t low_ones(uint32_t k) { return k >= 32u ? 0xFFFFFFFFu : ((1u << k) - 1u); }

STRSIM_HD uint32_t popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// ---------------------------------------------------------------------------------------------
// Levenshtein distance, Myers/Hyyro bit-vector recurrence on one 32-bit word.
// The pattern (length lp, 1..32) is LEFT-aligned: position j of P is bit j + (32 - lp), so the row
// whose score we track is always bit 31.  The 32 - lp low bits act as rows of a fictitious prefix
// that both strings share and that has already been consumed: their vertical deltas are -1
// (Mv ones), the real rows start at +1 (Pv ones), and the bottom-row score starts at lp.
// `peq(c)` must return the left-aligned mask.  lt >= 1.
// ---------------------------------------------------------------------------------------------
template <class Peq>
STRSIM_HD uint32_t lev_myers32(const uint32_t (&wt)[8], uint32_t lt, uint32_t lp, const Peq &peq)
{
    const uint32_t s = 32u - lp;
    uint32_t Pv = 0xFFFFFFFFu << s; // s <= 31 because lp >= 1
    uint32_t Mv = ~Pv;
    uint32_t score = lp;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if ((uint32_t)j < lt) {
            const uint32_t Eq = peq(lane_byte(wt, j));
            const uint32_t D0 = (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv;
            const uint32_t HP = Mv | ~(D0 | Pv);
            const uint32_t HN = Pv & D0;
            score += HP >> 31;
            score -= HN >> 31;
            const uint32_t X = (HP << 1) | 1u;
            Pv = (HN << 1) | ~(D0 | X);
            Mv = D0 & X;
        }
    }
    return score;
}

// ---------------------------------------------------------------------------------------------
// Jaro matching (strsim.rs:200-237).  Pattern = b (peq gives positions in b, bit j = b[j]), text =
// a.  Iterates a in order; for each a_i takes the LOWEST unflagged equal position of b inside
// [i-bound, min(i+bound, lb-1)] -- the reference's inner `for j in lower..=upper { .. break }`.
// Transpositions: the k-th flagged char of a vs the k-th flagged char of b (ascending positions);
// they are equal iff bit j_k of peq(a_{i_k}) is set, so no byte of b is ever extracted.
// la, lb >= 1.  Returns m (matches) and t (unequal zipped pairs, NOT halved).
// ---------------------------------------------------------------------------------------------
template <class Peq>
STRSIM_HD void jaro_match32(const uint32_t (&wa)[8], uint32_t la, uint32_t lb, const Peq &peq, uint32_t &m_out,
                            uint32_t &t_out)
{
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u; // mx/2 - 1 (:200); mx == 1 only for the 1x1 case, window {i}
    const uint32_t lbmask = low_ones(lb);
    uint32_t himask = low_ones((bound + 1u) < lb ? (bound + 1u) : lb); // ones at [0, min(i+bound, lb-1)]
    uint32_t lomask = 0u;                                              // ones below max(0, i-bound)
    uint32_t fb = 0u, fa = 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if ((uint32_t)i < la) {
            const uint32_t Eq = peq(lane_byte(wa, i));
            const uint32_t cand = Eq & himask & ~(lomask | fb);
            const uint32_t bit = cand & (0u - cand);
            fb |= bit;
            fa |= (bit ? 1u : 0u) << i;
            himask = ((himask << 1) | 1u) & lbmask;
            if ((uint32_t)i >= bound) lomask = (lomask << 1) | 1u;
        }
    }
    uint32_t t = 0u;
    uint32_t rest = fb;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if ((uint32_t)i < la && ((fa >> i) & 1u)) {
            const uint32_t jbit = rest & (0u - rest);
            rest ^= jbit;
            if ((peq(lane_byte(wa, i)) & jbit) == 0u) ++t;
        }
    }
    m_out = popc32(fb);
    t_out = t;
}

// Common leading bytes, capped at 4 and at both lengths (strsim.rs:261-266; ASCII so bytes = chars).
STRSIM_HD uint32_t common_prefix4(uint32_t a0, uint32_t la, uint32_t b0, uint32_t lb)
{
    const uint32_t x = a0 ^ b0;
    uint32_t p = 4u;
    if (x & 0x000000FFu) p = 0u;
    else if (x & 0x0000FF00u) p = 1u;
    else if (x & 0x00FF0000u) p = 2u;
    else if (x & 0xFF000000u) p = 3u;
    if (p > la) p = la;
    if (p > lb) p = lb;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Character-multiset intersection size I = sum_c min(countA[c], countB[c]).
// Any maximal matching of equal characters has exactly I edges: walk a, give each a_i the lowest
// still-unused equal position of b.
// ---------------------------------------------------------------------------------------------
template <class Peq>
STRSIM_HD uint32_t multiset_isect32(const uint32_t (&wa)[8], uint32_t la, const Peq &peq)
{
    uint32_t used = 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if ((uint32_t)i < la) {
            const uint32_t cand = peq(lane_byte(wa, i)) & ~used;
            used |= cand & (0u - cand);
        }
    }
    return popc32(used);
}

// ---------------------------------------------------------------------------------------------
// f64 epilogues -- same IEEE-754 operations in the same order as the Rust source.
// ---------------------------------------------------------------------------------------------
// strsim.rs:160   1.0 - (dist as f64 / max(la,lb) as f64); both-empty handled by the caller (:128)
STRSIM_HD double epilogue_levenshtein(uint64_t dist, uint64_t la, uint64_t lb)
{
    const uint64_t den = la > lb ? la : lb;
    return 1.0 - ((double)dist / (double)den);
}

// strsim.rs:238-243   integer t/2, three divisions summed left to right, then / 3.0
STRSIM_HD double epilogue_jaro(uint64_t m, uint64_t t, uint64_t la, uint64_t lb)
{
    if (m == 0) return 0.0;
    const double dm = (double)m;
    return (dm / (double)la + dm / (double)lb + (double)(m - t / 2) / dm) / 3.0;
}

// strsim.rs:260-270   strict > 0.7; jaro + ((prefix * 0.1) * (1.0 - jaro)); never fused
STRSIM_HD double epilogue_jaro_winkler(double jaro, uint32_t prefix)
{
    if (jaro > 0.7) {
        const double pl = (double)prefix;
        return jaro + (pl * 0.1 * (1.0 - jaro));
    }
    return jaro;
}

// strsim.rs:301-306   sum(min) / sum(max); sum(max) = la + lb - sum(min) as integers
STRSIM_HD double epilogue_jaccard(uint64_t isect, uint64_t la, uint64_t lb)
{
    return (double)isect / (double)(la + lb - isect);
}

// strsim.rs:337-343   (2.0 * I) / (la + lb)
STRSIM_HD double epilogue_sorensen_dice(uint64_t isect, uint64_t la, uint64_t lb)
{
    return 2.0 * (double)isect / (double)(la + lb);
}

// ---------------------------------------------------------------------------------------------
// One lane's result for one pair, given the match table of b (left-aligned for Levenshtein, see
// lane_peq_shift).  Handles the reference's early-outs (:128-130, :182-186, :288-292, :324-328); the
// `a == b` early-out needs no code: every formula below yields exactly 1.0 for equal strings.
// ---------------------------------------------------------------------------------------------
template <int MEASURE>
STRSIM_HD uint32_t lane_peq_shift(uint32_t lb)
{
    return MEASURE == LEVENSHTEIN ? 32u - lb : 0u; // only used when lb >= 1
}

// true when the pair needs the match table / bit-parallel loops at all
STRSIM_HD bool lane_needs_table(uint32_t la, uint32_t lb) { return la != 0u && lb != 0u; }

template <int MEASURE, class Peq>
STRSIM_HD double lane_pair_result(const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8], uint32_t lb,
                                  const Peq &peq)
{
    if (la == 0u && lb == 0u) return 1.0;
    if (la == 0u || lb == 0u) return 0.0; // Levenshtein: 1 - max/max = 0.0 as well (:160)
    if (MEASURE == LEVENSHTEIN) {
        const uint32_t dist = lev_myers32(wa, la, lb, peq);
        return epilogue_levenshtein(dist, la, lb);
    } else if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        uint32_t m, t;
        jaro_match32(wa, la, lb, peq, m, t);
        const double j = epilogue_jaro(m, t, la, lb);
        if (MEASURE == JARO) return j;
        return epilogue_jaro_winkler(j, common_prefix4(wa[0], la, wb[0], lb));
    } else {
        const uint32_t isect = multiset_isect32(wa, la, peq);
        return MEASURE == JACCARD ? epilogue_jaccard(isect, la, lb) : epilogue_sorensen_dice(isect, la, lb);
    }
}

} // namespace strsim
