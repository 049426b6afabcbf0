// lane_core_textbook.h -- the textbook arrangements of the 32-bit per-lane cores (round 1's kernels ran these): left-aligned
// Myers/Hyyro with per-column history words, two-pass Jaro, stand-alone multiset intersection, and a per-pair result built from
// them.  TEST INFRASTRUCTURE: the CPU harness checks the product's rearranged cores (polars-strsim_amd/csrc/strsim_lane_core.h:
// lev_myers32_snap, lane_cores32, build_planes) against these and against the oracle.  Nothing in the product includes this file.
#pragma once
#include "strsim_lane_core.h"

namespace strsim {

// The round-1 arrangement (kept for the CPU harness to check the two against each other).
template <int NP>
STRSIM_HD void build_planes_r1(const uint32_t (&w)[8], uint32_t (&P)[NP])
{
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        lo[g] = w[2 * g];
        hi[g] = w[2 * g + 1];
        transpose8x8(lo[g], hi[g]);
    }
    // 4x4 byte transposes: plane k = byte k of group 0..3 (planes 0-3 from lo[], 4-7 from hi[])
    {
        const uint32_t u0 = perm_b32(lo[1], lo[0], 0x05010400u); // [lo0.b0, lo1.b0, lo0.b1, lo1.b1]
        const uint32_t u2 = perm_b32(lo[3], lo[2], 0x05010400u);
        P[0] = perm_b32(u2, u0, 0x05040100u);
        if (NP > 1) P[1 < NP ? 1 : 0] = perm_b32(u2, u0, 0x07060302u);
        if (NP > 2) {
            const uint32_t u1 = perm_b32(lo[1], lo[0], 0x07030602u); // [lo0.b2, lo1.b2, lo0.b3, lo1.b3]
            const uint32_t u3 = perm_b32(lo[3], lo[2], 0x07030602u);
            P[2 < NP ? 2 : 0] = perm_b32(u3, u1, 0x05040100u);
            if (NP > 3) P[3 < NP ? 3 : 0] = perm_b32(u3, u1, 0x07060302u);
        }
    }
    if (NP > 4) {
        const uint32_t u0 = perm_b32(hi[1], hi[0], 0x05010400u);
        const uint32_t u2 = perm_b32(hi[3], hi[2], 0x05010400u);
        P[4 < NP ? 4 : 0] = perm_b32(u2, u0, 0x05040100u);
        if (NP > 5) P[5 < NP ? 5 : 0] = perm_b32(u2, u0, 0x07060302u);
        if (NP > 6) {
            const uint32_t u1 = perm_b32(hi[1], hi[0], 0x07030602u);
            const uint32_t u3 = perm_b32(hi[3], hi[2], 0x07030602u);
            P[6 < NP ? 6 : 0] = perm_b32(u3, u1, 0x05040100u);
            if (NP > 7) P[7 < NP ? 7 : 0] = perm_b32(u3, u1, 0x07060302u);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Levenshtein distance, Myers/Hyyro bit-vector recurrence on one 32-bit word.
// The pattern (length lp, 1..32) is LEFT-aligned: position j of P is bit j + (32 - lp), so the row
// whose score we track is always bit 31.  The 32 - lp low bits act as rows of a fictitious prefix
// that both strings share and that has already been consumed: their vertical deltas are -1
// (Mv ones), the real rows start at +1 (Pv ones), and the bottom-row score starts at lp.
// No step is predicated: every lane runs the same `nit` columns (the text's bytes past lt are
// don't-care) while the bottom-row deltas of each column are shifted into two history words; the score
// after exactly lt columns is lp + popc(+1 history) - popc(-1 history) over the first lt columns.
// `P` must already be shifted left by 32 - lp.  lt >= 1.  tmax: lane-uniform bound >= lt.
// ---------------------------------------------------------------------------------------------
template <int NP>
STRSIM_HD uint32_t lev_myers32(const uint32_t (&wt)[8], uint32_t lt, uint32_t tmax, const uint32_t (&P)[NP], uint32_t lp)
{
    const uint32_t s = 32u - lp;
    const uint32_t valid = 0xFFFFFFFFu << s; // s <= 31 because lp >= 1
    uint32_t Pv = valid;
    uint32_t Mv = ~Pv;
    uint32_t hp = 0u, hn = 0u;
    uint32_t nit = 0u;
#pragma unroll
    for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
        if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
#pragma unroll
        for (int jj = 0; jj < COLS_PER_TEST; ++jj) {
            const int j = COLS_PER_TEST * g + jj;
            const uint32_t Eq = eq_mask<NP>(P, valid, wt[j >> 2], j & 3);
            const uint32_t D0 = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq | Mv); // (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv
            const uint32_t HP = bitop3<0xF1>(Mv, D0, Pv);                    // Mv | ~(D0 | Pv)
            const uint32_t HN = Pv & D0;
            hp = (hp << 1) | (HP >> 31);
            hn = (hn << 1) | (HN >> 31);
            const uint32_t X = (HP << 1) | 1u;
            Pv = bitop3<0xF1>(HN << 1, D0, X);                               // (HN << 1) | ~(D0 | X)
            Mv = D0 & X;
        }
        nit += (uint32_t)COLS_PER_TEST;
    }
    // column j sits at history bit nit-1-j; keep columns 0..lt-1
    const uint32_t cols = low_ones(lt) << (nit - lt);
    return lp + popc32(hp & cols) - popc32(hn & cols);
}


// ---------------------------------------------------------------------------------------------
// Jaro matching (strsim.rs:200-237).  Pattern = b (planes of b, bit j = b[j]), text = a.
// Iterates a in order; for each a_i takes the LOWEST unflagged equal position of b inside
// [i-bound, min(i+bound, lb-1)] -- the reference's inner `for j in lower..=upper { .. break }`.
// Transpositions: the k-th flagged char of a vs the k-th flagged char of b (ascending positions);
// they are equal iff bit j_k of Eq(a_{i_k}) is set, so no byte of b is ever extracted.
// la, lb >= 1.  Returns m (matches) and t (unequal zipped pairs, NOT halved).
// ---------------------------------------------------------------------------------------------
template <int NP>
STRSIM_HD void jaro_match32(const uint32_t (&wa)[8], uint32_t la, uint32_t tmax, uint32_t lb, const uint32_t (&P)[NP],
                            uint32_t &m_out, uint32_t &t_out)
{
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u; // mx/2 - 1 (:200); mx == 1 only for the 1x1 case, window {i}
    const uint32_t lbmask = low_ones(lb);
    const uint32_t live = low_ones(la);                                 // bit i set while i < la
    uint32_t himask = low_ones((bound + 1u) < lb ? (bound + 1u) : lb); // ones at [0, min(i+bound, lb-1)]
    uint32_t lomask = 0u;                                              // ones below max(0, i-bound)
    uint32_t fb = 0u, fa = 0u;
#pragma unroll
    for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
        if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
#pragma unroll
        for (int ii = 0; ii < COLS_PER_TEST; ++ii) {
            const int i = COLS_PER_TEST * g + ii;
            const uint32_t Eq = eq_mask<NP>(P, lbmask, wa[i >> 2], i & 3);
            const uint32_t cand = Eq & himask & ~(lomask | fb) & bit_fill(live, i);
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
    for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
        if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
#pragma unroll
        for (int ii = 0; ii < COLS_PER_TEST; ++ii) {
            const int i = COLS_PER_TEST * g + ii;
            const uint32_t on = bit_fill(fa, i);               // a_i was matched
            const uint32_t jbit = rest & (0u - rest) & on;     // its partner in the zip: lowest remaining flag of b
            rest ^= jbit;
            const uint32_t Eq = eq_mask<NP>(P, lbmask, wa[i >> 2], i & 3);
            t += ((jbit & ~Eq) != 0u) ? 1u : 0u;
        }
    }
    m_out = popc32(fb);
    t_out = t;
}


// ---------------------------------------------------------------------------------------------
// Character-multiset intersection size I = sum_c min(countA[c], countB[c]).
// Any maximal matching of equal characters has exactly I edges: walk a, give each a_i the lowest
// still-unused equal position of b.
// ---------------------------------------------------------------------------------------------
template <int NP>
STRSIM_HD uint32_t multiset_isect32(const uint32_t (&wa)[8], uint32_t la, uint32_t tmax, uint32_t lb,
                                    const uint32_t (&P)[NP])
{
    const uint32_t lbmask = low_ones(lb);
    const uint32_t live = low_ones(la);
    uint32_t used = 0u;
#pragma unroll
    for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
        if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
#pragma unroll
        for (int ii = 0; ii < COLS_PER_TEST; ++ii) {
            const int i = COLS_PER_TEST * g + ii;
            const uint32_t cand = eq_mask<NP>(P, lbmask, wa[i >> 2], i & 3) & ~used & bit_fill(live, i);
            used |= cand & (0u - cand);
        }
    }
    return popc32(used);
}


// ---------------------------------------------------------------------------------------------
// One lane's result for one pair (both strings <= 32 ASCII bytes, in registers).  Handles the
// reference's early-outs (:128-130, :182-186, :288-292, :324-328); the `a == b` early-out needs no
// code: every formula below yields exactly 1.0 for equal strings.
// NP: number of low bits that tell the pair's bytes apart (planes_needed); tmax: uniform bound >= la.
// ---------------------------------------------------------------------------------------------
// levtab: optional 33x33 table of 1.0 - dist/den (index dist*33 + den), else nullptr.
template <int MEASURE, int NP>
STRSIM_HD double lane_pair_result(const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8], uint32_t lb,
                                  uint32_t tmax, const double *levtab = nullptr, const double *qtab = nullptr)
{
    uint32_t P[NP];
    build_planes<NP>(wb, P); // pattern = b
    const bool live = la != 0u && lb != 0u;
    const uint32_t la1 = live ? la : 1u, lb1 = live ? lb : 1u; // keep every shift amount in range on dead lanes
    double r;
    if (MEASURE == LEVENSHTEIN) {
        const uint32_t s = 32u - lb1;
#pragma unroll
        for (int k = 0; k < NP; ++k) P[k] <<= s;
        const uint32_t dist = lev_myers32<NP>(wa, la1, tmax, P, lb1);
        r = levtab ? levtab[dist * 33u + (la1 > lb1 ? la1 : lb1)] : epilogue_levenshtein(dist, la1, lb1);
    } else if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        uint32_t m, t;
        jaro_match32<NP>(wa, la1, tmax, lb1, P, m, t);
        r = qtab ? epilogue_jaro_q(qtab, m, t, la1, lb1) : epilogue_jaro(m, t, la1, lb1);
        if (MEASURE == JARO_WINKLER) r = epilogue_jaro_winkler(r, common_prefix4(wa[0], la1, wb[0], lb1));
    } else {
        const uint32_t isect = multiset_isect32<NP>(wa, la1, tmax, lb1, P);
        r = MEASURE == JACCARD ? epilogue_jaccard(isect, la1, lb1) : epilogue_sorensen_dice(isect, la1, lb1);
    }
    if (!live) r = (la == 0u && lb == 0u) ? 1.0 : 0.0; // Levenshtein with one empty side: 1 - max/max = 0.0 (:160)
    return r;
}

// ---------------------------------------------------------------------------------------------
// All five measures of one pair from one set of bit-planes (BASELINE config 4): the planes of b are built once,
// Jaro's matching serves both Jaro and Jaro-Winkler, the multiset intersection serves Jaccard and Dice.
// r[] is indexed by Measure.  No role swap here: Jaro walks a, so every measure does.
// ---------------------------------------------------------------------------------------------
template <int NP>
STRSIM_HD void lane_all_results(const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8], uint32_t lb,
                                uint32_t tmax, const double *levtab, double (&r)[5], const double *qtab = nullptr)
{
    uint32_t P[NP];
    build_planes<NP>(wb, P);
    const bool live = la != 0u && lb != 0u;
    const uint32_t la1 = live ? la : 1u, lb1 = live ? lb : 1u;
    uint32_t m, t;
    jaro_match32<NP>(wa, la1, tmax, lb1, P, m, t);
    const double j = qtab ? epilogue_jaro_q(qtab, m, t, la1, lb1) : epilogue_jaro(m, t, la1, lb1);
    r[JARO] = j;
    r[JARO_WINKLER] = epilogue_jaro_winkler(j, common_prefix4(wa[0], la1, wb[0], lb1));
    const uint32_t isect = multiset_isect32<NP>(wa, la1, tmax, lb1, P);
    r[JACCARD] = epilogue_jaccard(isect, la1, lb1);
    r[SORENSEN_DICE] = epilogue_sorensen_dice(isect, la1, lb1);
    const uint32_t s = 32u - lb1;
#pragma unroll
    for (int k = 0; k < NP; ++k) P[k] <<= s;
    const uint32_t dist = lev_myers32<NP>(wa, la1, tmax, P, lb1);
    r[LEVENSHTEIN] = levtab ? levtab[dist * 33u + (la1 > lb1 ? la1 : lb1)] : epilogue_levenshtein(dist, la1, lb1);
    if (!live) {
        const double v = (la == 0u && lb == 0u) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = v;
    }
}


} // namespace strsim
