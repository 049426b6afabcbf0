// strsim_lane_lut.h -- match masks from a per-lane lookup table in LDS instead of bit fills.
//
// The bit-plane match mask of strsim_lane_core.h (eq_mask) spends two VALU instructions per plane and column: a
// v_bfe_i32 that replicates one bit of the text byte (a half-rate opcode on gfx950, 4.1 cycles of SIMD time) and the
// v_bitop3 that consumes it -- 33 of the 59 cycles of a Levenshtein column, the largest single item of k_lane_stage
// (DESIGN 3.0).  Here the planes are turned ONCE per pair into two small tables
//       L[l] = positions whose low three bits are l  (8 masks, one v_bitop3 of planes 0..2 each)
//       M[m] = positions whose bits 3..4 are m       (4 masks, one v_bitop3 of planes 3..4 each, `valid` folded in)
// and a column's mask is  Eq(c) = L[c & 7] & M[(c >> 3) & 3]:  two ds_read_b32 whose addresses cost one v_perm_b32 each,
// and the AND folds into the three-input ops that consume the mask.  Planes 5 and 6 (rounds whose bytes differ there:
// mixed case, digits) stay in registers and are applied with bit fills as before.
//
// LDS layout (device): entry-major, one dword per lane -- address = lane * 4 + (K + e) * 256, K = (wave's table base) >> 8,
// a multiple of 16 -- so the 64 lanes of a read always hit 64 different banks whatever entries they ask for, and the
// address of entry e for byte j of a text dword is ONE v_perm_b32: byte 0 = lane * 4, byte 1 = the byte of a word that
// holds (K | e) for each of its four characters (two or three ops per FOUR characters to make that word).
// 12 entries x 256 B = 3 KB per wave.
//
// Host build (tests/cpu_harness): the same functions on a plain array.
//
// Semantics are those of strsim_lane_core.h (reference strsim.rs:141-160, :200-237, :297-305, :333-341): only how a
// column's match mask is produced differs; the recurrences are the same instructions.
#pragma once
#include <stdint.h>
#include <type_traits>

#include "strsim_bounds.h"
#include "strsim_lane_core.h"

namespace strsim {

#ifndef STRSIM_LUT_AHEAD
#define STRSIM_LUT_AHEAD 1 // groups of COLS_PER_TEST columns between a table read and its use
#endif
constexpr int LUT_AHEAD = STRSIM_LUT_AHEAD;
#ifndef STRSIM_JARO_KEEP_EQ
#define STRSIM_JARO_KEEP_EQ 1 // Jaro's zip pass reads the first pass's match masks from registers instead of the tables again ([r5]: up
#endif                        // to 32 registers that the four-waves-per-SIMD instantiations have: cfg2 Jaro 51.5 -> 56.6 G pairs/s, five outputs 34.5 -> 37.2)

constexpr int LUT_ENTRIES = 12;               // L: 0..7, M: 8..11
constexpr int LUT_WAVE_BYTES = LUT_ENTRIES * 256;

struct EqLut {
    uint32_t lane4; // device: lane * 4 (byte 0 of every table address of this lane)
    uint32_t krep;  // device: K replicated to the four bytes (K = table base >> 8; low four bits clear)
#if !defined(__HIP_DEVICE_COMPILE__)
    uint32_t tab[LUT_ENTRIES]; // host: the table itself
#endif
};

// a text dword's four characters as table coordinates: byte j of .l / .m = (K | entry) of character j
struct LutIndex {
    uint32_t l, m;
};

STRSIM_HD LutIndex lut_index(const EqLut &t, uint32_t w)
{
    LutIndex x;
#if defined(__HIP_DEVICE_COMPILE__)
    x.l = bitop3<0xEA>(w, 0x07070707u, t.krep);                      // (w & 7s) | K
    x.m = bitop3<0xEA>(w >> 3, 0x03030303u, t.krep | 0x08080808u);  // ((w >> 3) & 3s) | (K + 8)
#else
    (void)t;
    x.l = w & 0x07070707u;
    x.m = ((w >> 3) & 0x03030303u) | 0x08080808u;
#endif
    return x;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t lut_lds_read(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
}
__device__ __forceinline__ void lut_lds_write(uint32_t addr, uint32_t v)
{
    *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)addr) = v;
}
#endif

// L / M entry of character `byte` (0..3, static) of the indexed dword
STRSIM_HD uint32_t lut_read(const EqLut &t, uint32_t idx, int byte)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // v_perm_b32 {S0 = idx (bytes 4..7), S1 = lane4 (bytes 0..3)}: [lane4.b0, idx.b<byte>, 0, 0]
    const uint32_t addr = __builtin_amdgcn_perm(idx, t.lane4, 0x0C0C0000u | ((4u + (uint32_t)byte) << 8));
    // (lab: inside this wave's twelve table entries)
    STRSIM_CHECK_RANGE(K_STAGE, 50, ~0ull, addr, (t.krep & 0xFFu) << 8, ((t.krep & 0xFFu) << 8) + (uint32_t)LUT_WAVE_BYTES - 4u);
    return lut_lds_read(addr);
#else
    return t.tab[(idx >> (8 * byte)) & 0xFFu];
#endif
}

// The tables of a pattern from its planes 0..4 (`valid`: the positions that may match at all -- all-ones for the
// Levenshtein arrangement, the pattern's length mask for Jaro and the multiset intersection).
template <int NP>
STRSIM_HD void lut_build(EqLut &t, const uint32_t (&P)[NP], uint32_t valid)
{
    static_assert(NP >= 5, "tables cover planes 0..4");
    uint32_t e[LUT_ENTRIES];
    // L[l]: P0 == l0 && P1 == l1 && P2 == l2 -- truth-table bit (a << 2 | b << 1 | c) with a = P0, b = P1, c = P2
    e[0] = bitop3<0x01>(P[0], P[1], P[2]); // l = 0: a0 b0 c0
    e[1] = bitop3<0x10>(P[0], P[1], P[2]); // l = 1: a1 b0 c0
    e[2] = bitop3<0x04>(P[0], P[1], P[2]); // l = 2: a0 b1 c0
    e[3] = bitop3<0x40>(P[0], P[1], P[2]); // l = 3: a1 b1 c0
    e[4] = bitop3<0x02>(P[0], P[1], P[2]); // l = 4: a0 b0 c1
    e[5] = bitop3<0x20>(P[0], P[1], P[2]); // l = 5: a1 b0 c1
    e[6] = bitop3<0x08>(P[0], P[1], P[2]); // l = 6: a0 b1 c1
    e[7] = bitop3<0x80>(P[0], P[1], P[2]); // l = 7
    // M[m]: P3 == m0 && P4 == m1 && valid  (a = P3, b = P4, c = valid)
    e[8] = bitop3<0x02>(P[3], P[4], valid);
    e[9] = bitop3<0x20>(P[3], P[4], valid);
    e[10] = bitop3<0x08>(P[3], P[4], valid);
    e[11] = bitop3<0x80>(P[3], P[4], valid);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t base = t.lane4 + ((t.krep & 0xFFu) << 8);
#pragma unroll
    for (int q = 0; q < LUT_ENTRIES; ++q) lut_lds_write(base + 256u * (uint32_t)q, e[q]);
#else
#pragma unroll
    for (int q = 0; q < LUT_ENTRIES; ++q) t.tab[q] = e[q];
#endif
}

// what planes 5.. contribute to the mask of character `byte` of dword w (all-ones when NP == 5)
template <int NP>
STRSIM_HD uint32_t lut_high_planes(const uint32_t (&P)[NP], uint32_t w, int byte)
{
    uint32_t acc = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 5; k < NP; ++k) acc = bitop3<0x90>(acc, P[k < NP ? k : 0], bit_fill(w, 8 * byte + k));
    return acc;
}

// ---------------------------------------------------------------------------------------------
// lev_myers32_snap (strsim_lane_core.h) with table masks.  The two table entries of a column are fetched one group of
// columns ahead of their use (software pipeline: the loads of group g + 1 are issued before the recurrence of group g), and
// L & M is never formed: Eq & Pv is one three-input AND, and D0 takes L and M as two of the inputs of a three-input op.
// lt, lp >= 1; tmin <= lt <= tmax, both lane-uniform.  `t` must hold the tables of THIS pattern (lut_build, valid = ~0).
// ---------------------------------------------------------------------------------------------
template <int NP>
STRSIM_HD uint32_t lev_myers32_lut(const EqLut &t, const uint32_t (&wt)[8], uint32_t lt, uint32_t tmin, uint32_t tmax,
                                   const uint32_t (&P)[NP], uint32_t lp)
{
    constexpr int CPT = COLS_PER_TEST, NG = 32 / COLS_PER_TEST;
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0u;
    uint32_t L[NG + LUT_AHEAD][CPT], M[NG + LUT_AHEAD][CPT]; // (every index is a compile-time constant: registers, two groups live at a time)
    LutIndex ix = lut_index(t, wt[0]);
    auto fetch = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int jj = 0; jj < CPT; ++jj) {
            const int j = CPT * g + jj;
            if ((j & 3) == 0 && j != 0) ix = lut_index(t, wt[(j >> 2) & 7]);
            L[g][jj] = lut_read(t, ix.l, j & 3);
            M[g][jj] = lut_read(t, ix.m, j & 3);
        }
    };
    unrolled_until<0, LUT_AHEAD>([&](auto gc) { fetch(gc); return true; });
    unrolled_until<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if ((uint32_t)(CPT * g) >= tmax) return false;
        if constexpr (g + LUT_AHEAD < NG) fetch(std::integral_constant<int, g + LUT_AHEAD>{});
        auto column = [&](int jj) {
            const int j = CPT * g + jj;
            const uint32_t l = L[g][jj], m = M[g][jj];
            uint32_t D0;
            if (NP > 5) {
                const uint32_t Eq = bitop3<0x80>(l, m, lut_high_planes<NP>(P, wt[j >> 2], j & 3));
                D0 = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq) | Mv;
            } else {
                const uint32_t X = bitop3<0xBE>(bitop3<0x80>(l, m, Pv) + Pv, Pv, Mv); // (((Eq & Pv) + Pv) ^ Pv) | Mv
                D0 = bitop3<0xF8>(X, l, m);                                           // | Eq
            }
            const uint32_t nHP = bitop3<0x0E>(Mv, D0, Pv);                 // ~HP = ~Mv & (D0 | Pv)
            const uint32_t nX = twice(nHP);                                 // ~((HP << 1) | 1)
            const uint32_t HN2 = twice(D0 & Pv);                            // HN << 1
            Pv = bitop3<0xF2>(HN2, D0, nX);                                 // (HN << 1) | ~(D0 | X)
            Mv = bitop3<0x50>(D0, D0, nX);                                  // D0 & X
        };
        if ((uint32_t)(CPT * (g + 1)) >= tmin) {                           // (uniform) some lane's text may end in this group
#pragma unroll
            for (int jj = 0; jj < CPT; ++jj)
                if ((uint32_t)(CPT * g + jj) < lt) column(jj);
        } else {
#pragma unroll
            for (int jj = 0; jj < CPT; ++jj) column(jj);
        }
        return true;
    });
    const uint32_t rows = low_ones(lp);
    return lt + popc32(Pv & rows) - popc32(Mv & rows);
}

// ---------------------------------------------------------------------------------------------
// lane_cores32 (strsim_lane_core.h: Levenshtein recurrence, Jaro's first pass and the multiset intersection in one column
// loop over the text a, Jaro's zip pass behind it) with table masks, fetched one group of columns ahead.  The zip pass
// takes the first pass's masks from registers ([r5] STRSIM_JARO_KEEP_EQ; rounds 3-4 read the tables again: two LDS reads and
// one AND per column, where the bit fills had rebuilt the mask with ten instructions).
// la, lb >= 1; tmin <= la <= tmax, both lane-uniform; `t` holds the tables of THIS pattern (lut_build, valid = ~0).
// ---------------------------------------------------------------------------------------------
template <int NP, bool DO_LEV, bool DO_JARO, bool DO_ISECT>
STRSIM_HD void lane_cores32_lut(const EqLut &t, const uint32_t (&wa)[8], uint32_t la, uint32_t tmin, uint32_t tmax, uint32_t lb,
                                const uint32_t (&P)[NP], uint32_t &dist, uint32_t &m_out, uint32_t &t_out, uint32_t &isect)
{
    const uint32_t lbmask = low_ones(lb);
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0u;                                 // Levenshtein
    const uint32_t mx = la > lb ? la : lb;                               // Jaro, first pass
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u;                      // mx/2 - 1 (:200); mx == 1 only for the 1x1 case, window {i}
    uint32_t win = low_ones(bound + 1u);                                 // ones at [max(i - bound, 0), i + bound] (lane_cores32)
    uint32_t fb = 0u, fa = 0u;
    uint32_t used = 0u;                                                  // multiset intersection
    constexpr int CPT = COLS_PER_TEST, NG = 32 / COLS_PER_TEST;
    uint32_t E[NG + LUT_AHEAD][CPT]; // (every index is a compile-time constant: registers, two groups live at a time)
    LutIndex ix = lut_index(t, wa[0]);
    auto fetch = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int jj = 0; jj < CPT; ++jj) {
            const int j = CPT * g + jj;
            if ((j & 3) == 0 && j != 0) ix = lut_index(t, wa[(j >> 2) & 7]);
            const uint32_t l = lut_read(t, ix.l, j & 3), m = lut_read(t, ix.m, j & 3);
            // (the masks end at lb when Jaro is on, as in lane_cores32: one more input of the same three-input AND)
            if (NP > 5) E[g][jj] = bitop3<0x80>(l, m, lut_high_planes<NP>(P, wa[(j >> 2) & 7], j & 3)) & (DO_JARO ? lbmask : 0xFFFFFFFFu);
            else E[g][jj] = DO_JARO ? bitop3<0x80>(l, m, lbmask) : (l & m);
        }
    };
    unrolled_until<0, LUT_AHEAD>([&](auto gc) { fetch(gc); return true; });
    unrolled_until<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if ((uint32_t)(CPT * g) >= tmax) return false;
        if constexpr (g + LUT_AHEAD < NG) fetch(std::integral_constant<int, g + LUT_AHEAD>{});
        auto column = [&](int jj) {
            const int i = CPT * g + jj;
            const uint32_t Eq = E[g][jj];
            if (DO_LEV) {
                const uint32_t D0 = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq) | Mv; // (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv
                const uint32_t nHP = bitop3<0x0E>(Mv, D0, Pv);                 // ~HP = ~Mv & (D0 | Pv)
                const uint32_t nX = twice(nHP);                                 // ~((HP << 1) | 1)
                const uint32_t HN2 = twice(D0 & Pv);                            // HN << 1
                Pv = bitop3<0xF2>(HN2, D0, nX);                                 // (HN << 1) | ~(D0 | X)
                Mv = bitop3<0x50>(D0, D0, nX);                                  // D0 & X
            }
            if (DO_JARO) {
                // candidates: equal, inside [max(i - bound, 0), min(i + bound, lb - 1)], not flagged yet
                const uint32_t cand = bitop3<0x40>(Eq, win, fb);                          // Eq & win & ~fb
                fb = bitop3<0xF8>(fb, cand, 0u - cand);                                   // fb | lowest candidate
                fa |= cand ? (1u << i) : 0u;
                win = win + win + ((uint32_t)i < bound ? 1u : 0u);
            }
            if (DO_ISECT) {
                const uint32_t cand = bitop3<0x08>(used, Eq, lbmask);                     // ~used & Eq & lbmask
                used = bitop3<0xF8>(used, cand, 0u - cand);
            }
        };
        if ((uint32_t)(CPT * (g + 1)) >= tmin) {                           // (uniform) some lane's text may end in this group
#pragma unroll
            for (int jj = 0; jj < CPT; ++jj)
                if ((uint32_t)(CPT * g + jj) < la) column(jj);
        } else {
#pragma unroll
            for (int jj = 0; jj < CPT; ++jj) column(jj);
        }
        return true;
    });
    if (DO_LEV) dist = la + popc32(Pv & lbmask) - popc32(Mv & lbmask);
    if (DO_ISECT) isect = popc32(used);
    if (DO_JARO) {
        // second pass: the k-th flagged character of a against the k-th flagged character of b (ascending positions); they
        // are equal iff bit j_k of Eq(a_{i_k}) is set.  Columns at or beyond la have no flag, so no predicate is needed.
        uint32_t unequal = 0u, rest = fb; // (collected as bits of b, counted once: lane_cores32)
#if !STRSIM_JARO_KEEP_EQ
        ix = lut_index(t, wa[0]);
        unrolled_until<0, LUT_AHEAD>([&](auto gc) { fetch(gc); return true; });
#endif
        unrolled_until<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if ((uint32_t)(CPT * g) >= tmax) return false;
#if !STRSIM_JARO_KEEP_EQ
            if constexpr (g + LUT_AHEAD < NG) fetch(std::integral_constant<int, g + LUT_AHEAD>{});
#endif
#pragma unroll
            for (int ii = 0; ii < CPT; ++ii) {
                const int i = CPT * g + ii;
                const uint32_t on = bit_fill(fa, i);               // a_i was matched
                const uint32_t jbit = rest & (0u - rest) & on;     // its partner in the zip: lowest remaining flag of b
                rest ^= jbit;
                unequal = bitop3<0xF4>(unequal, jbit, E[g][ii]);   // unequal | (jbit & ~Eq)
            }
            return true;
        });
        m_out = popc32(fb);
        t_out = popc32(unequal);
    }
}

} // namespace strsim
