// strsim_lane_core.h -- per-lane (one string pair per SIMD lane) bit-parallel cores for strings of
// <= 32 ASCII bytes.  Host/device portable so the exact arithmetic the gfx950 kernels run can also be
// exercised by a CPU harness in tests/ (there is no GPU in the build container).
//
// Every routine sees one pair, both strings in eight 32-bit registers each (little-endian bytes,
// bytes at or beyond the length are don't-care):
//   * the "pattern" string (b) is transposed into bit-planes, from which the match mask of any byte
//     is two VALU ops per plane (eq_mask) -- no lookup table, no LDS;
//   * the "text" string (a) is walked byte by byte.
//
// Semantics restated (reference = /root/reference/src/expressions/strsim.rs):
//   lev_myers32_snap / lane_cores32<DO_LEV>   -> integer edit distance of Levenshtein::compute      (:141-160)
//   lane_cores32<DO_JARO>                      -> (m, t) of Jaro::compute                            (:200-237)
//   lane_cores32<DO_ISECT>                     -> sum of min(countA[c], countB[c]) of Jaccard/Dice   (:297-305, :333-341)
// and the f64 epilogues, in the reference's operation order (compile with -ffp-contract=off).  The textbook arrangements of
// the same three cores (left-aligned Myers with history words, two-pass Jaro, stand-alone intersection) live in
// tests/cpu_harness/lane_core_textbook.h: the CPU harness checks these against them and against the oracle.
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#define STRSIM_HD __host__ __device__ __forceinline__
#else
#define STRSIM_HD inline
#endif

namespace strsim {

enum Measure : int { LEVENSHTEIN = 0, JARO = 1, JARO_WINKLER = 2, JACCARD = 3, SORENSEN_DICE = 4 };

// EVERY measure is symmetric in its two strings, bit for bit -- so every kernel is free to choose which string its column
// loop walks (the "text": the shorter one) and which one is transposed into bit-planes (the "pattern").  For Levenshtein and the
// multiset measures that is plain.  [r5] For the reference's Jaro (strsim.rs:200-243) it holds too, although its loop walks a and
// picks the lowest free partner in b:
//   * an edge (i, j) exists iff a_i == b_j, |i - j| <= bound, i < la, j < lb (`take(b.len() + bound)` and `min(i + bound, lb - 1)`,
//     :208-210, only cut what those four conditions cut), and bound = max(la, lb) / 2 - 1 (:200) does not care about the order;
//   * the greedy matching is the same from either side.  Per character (edges only join equal characters): let a1 / b1 be the
//     lowest position of that character in a / b.  If |a1 - b1| <= bound both loops pair them first (b1 is the lowest candidate
//     of a1 and vice versa); if b1 < a1 - bound no position of a reaches b1 (they all lie at or above a1), from either side; if
//     a1 < b1 - bound likewise.  Remove what was decided and repeat: the two loops flag the same positions;
//   * t zips the flagged characters of both strings in order (:220-237), m / la + m / lb is an IEEE addition, which commutes, and
//     the common prefix of Jaro-Winkler (:261-266) does not know which string came first.
// tests/test_oracle_golden.py::test_jaro_is_symmetric_bit_for_bit holds the oracle to it (small alphabets, lengths 0..400); the
// GPU parity suites compare every kernel against jaro(a, b) as the reference computes it.

// The per-column loops of the cores are fully unrolled and leave at a lane-uniform bound (tmax) that is tested every
// COLS_PER_TEST columns: a round of 64 pairs runs its longest text rounded up to that (cfg2: 17.5 columns per pair
// with 4, 16.3 with 2, 13.9 needed).
#ifndef STRSIM_COLS_PER_TEST
#define STRSIM_COLS_PER_TEST 2
#endif
constexpr int COLS_PER_TEST = STRSIM_COLS_PER_TEST;

STRSIM_HD uint32_t lane_byte(const uint32_t (&w)[8], int j) { return (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu; }

STRSIM_HD uint32_t low_ones(uint32_t k) { return k >= 32u ? 0xFFFFFFFFu : ((1u << k) - 1u); }

// for (g = G; g < NG; ++g) if (!f(integral_constant<g>)) break;  -- unrolled by construction (the column loops index
// registers with g: left to `#pragma unroll` a loop with an early exit may stay rolled, and then every register array becomes
// a chain of selects)
template <int G, int NG, typename F>
STRSIM_HD void unrolled_until(F &&f)
{
    if constexpr (G < NG) {
        if (f(std::integral_constant<int, G>{})) unrolled_until<G + 1, NG>(f);
    }
}


// A value the caller KNOWS to be the same in every lane of the wave, made so for the compiler too (v_readfirstlane; folded away
// when the value already lives in a scalar register).  Operands handed to inline asm through an "s" constraint go through this:
// should a future caller pass a divergent value, every lane visibly gets lane 0's -- instead of the compiler inserting the same
// readfirstlane silently.
STRSIM_HD uint32_t wave_uniform(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}

STRSIM_HD uint32_t popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}

// ---------------------------------------------------------------------------------------------
// Bit-sliced match masks.  The "pattern" string (<= 32 bytes, in eight registers) is transposed once
// into bit-planes: bit i of plane k = bit k of pattern byte i.  The positions where the pattern equals a
// byte c are then   Eq(c) = valid & AND_k ~(P_k ^ m_k),  m_k = bit k of c replicated to all 32 bits
// -- two VALU ops per plane (v_bfe_i32 + v_bitop3_b32) and no memory at all, so the lane-per-pair
// kernels need no LDS and run at full occupancy.  Only the low NP bits of the bytes are compared:
// the caller guarantees that bits >= NP are the same in every byte of the pair (a-z: NP = 5).
// ---------------------------------------------------------------------------------------------
STRSIM_HD uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    // v_perm_b32: result byte n = byte sel[n] of the 8 bytes {s0 = bytes 4..7, s1 = bytes 0..3}
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int n = 0; n < 4; ++n) {
        const uint32_t sl = (sel >> (8 * n)) & 0xFFu;
        const uint32_t b = sl < 8u ? (uint32_t)((src >> (8 * sl)) & 0xFFu) : (sl >= 0x0Du ? 0xFFu : 0u);
        r |= b << (8 * n);
    }
    return r;
#endif
}

// bit `off` of w replicated to all 32 bits (0 or ~0)
STRSIM_HD uint32_t bit_fill(uint32_t w, int off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_sbfe((int)w, (unsigned)off, 1u);
#else
    return 0u - ((w >> off) & 1u);
#endif
}

// Three-input boolean function with truth table TT (a = 0xF0, b = 0xCC, c = 0xAA): one v_bitop3_b32.
template <int TT>
STRSIM_HD uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    if (TT & 0x01) r |= ~a & ~b & ~c;
    if (TT & 0x02) r |= ~a & ~b & c;
    if (TT & 0x04) r |= ~a & b & ~c;
    if (TT & 0x08) r |= ~a & b & c;
    if (TT & 0x10) r |= a & ~b & ~c;
    if (TT & 0x20) r |= a & ~b & c;
    if (TT & 0x40) r |= a & b & ~c;
    if (TT & 0x80) r |= a & b & c;
    return r;
#endif
}

// x + x as an add the compiler cannot turn back into a shift (v_lshlrev_b32 costs two issue slots on gfx950, v_add_u32 one)
STRSIM_HD uint32_t twice(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_add_u32_e32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x + x;
#endif
}

// 8x8 bit-matrix transpose of the eight bytes {lo, hi}: afterwards byte k holds bit k of the eight
// source bytes (bit i of byte k = bit k of source byte i).  Three block-swap rounds (1, 2, 4 bits).
// (written with explicit three-input ops: (a ^ b) & c = 0x28, a ^ b ^ c = 0x96 -- hipcc does not form v_bitop3_b32 from
// plain C, and on gfx950 v_bitop3_b32 issues at the full rate like v_xor while a left shift costs two slots,
// bench_support/micro/op_cost.hip)
STRSIM_HD void transpose8x8(uint32_t &lo, uint32_t &hi)
{
    uint32_t t;
    t = bitop3<0x28>(lo, lo >> 7, 0x00AA00AAu); lo = bitop3<0x96>(lo, t, t << 7);
    t = bitop3<0x28>(hi, hi >> 7, 0x00AA00AAu); hi = bitop3<0x96>(hi, t, t << 7);
    t = bitop3<0x28>(lo, lo >> 14, 0x0000CCCCu); lo = bitop3<0x96>(lo, t, t << 14);
    t = bitop3<0x28>(hi, hi >> 14, 0x0000CCCCu); hi = bitop3<0x96>(hi, t, t << 14);
    t = bitop3<0x28>(lo, hi << 4, 0xF0F0F0F0u);
    lo ^= t;
    hi ^= t >> 4;
}

// Planes 0..NP-1 of the 32 pattern bytes w[0..7] (NP in 1..8).
// Two 4x4 byte transposes first (sixteen v_perm_b32): register e then holds bytes e, e + 8, e + 16, e + 24, i.e. its bit
// 8y + k is bit k of byte 8y + e.  What remains is to exchange the register index e with the in-byte bit index k, for the
// four byte lanes at once: three rounds of delta swaps BETWEEN registers (bit t of e against bit t of k), each half of a
// swap one shift and one three-input select -- and the halves that would only feed planes NP .. 7 are never computed.
// 58 instructions for five planes where the within-register transposes (build_planes_r1 below: three rounds on each pair of
// dwords, then the byte transposes of the planes) take 91, and 191 instead of 293 cycles of SIMD time (gfx950 issue costs,
// bench_support/micro/op_cost.hip).
template <int NP>
STRSIM_HD void build_planes(const uint32_t (&w)[8], uint32_t (&P)[NP])
{
    uint32_t r[8];
    {
        // r[e] byte y = w[2y + (e >> 2)] byte (e & 3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t t0 = perm_b32(w[2 + h], w[0 + h], 0x05010400u); // [w0.b0, w2.b0, w0.b1, w2.b1]
            const uint32_t t1 = perm_b32(w[2 + h], w[0 + h], 0x07030602u); // [w0.b2, w2.b2, w0.b3, w2.b3]
            const uint32_t t2 = perm_b32(w[6 + h], w[4 + h], 0x05010400u);
            const uint32_t t3 = perm_b32(w[6 + h], w[4 + h], 0x07030602u);
            r[4 * h + 0] = perm_b32(t2, t0, 0x05040100u);
            r[4 * h + 1] = perm_b32(t2, t0, 0x07060302u);
            r[4 * h + 2] = perm_b32(t3, t1, 0x05040100u);
            r[4 * h + 3] = perm_b32(t3, t1, 0x07060302u);
        }
    }
    // round t: registers e (bit t clear) and e | 2^t; M = the bit positions whose in-byte index has bit t clear
    //   low'  = M ? low : high << 2^t        high' = M ? low >> 2^t : high          (bitop3 0xCA = a ? b : c)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const uint32_t a = r[e], b = r[e + 1];
        r[e] = bitop3<0xCA>(0x55555555u, a, twice(b));
        r[e + 1] = bitop3<0xCA>(0x55555555u, a >> 1, b);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = (q & 1) | ((q & 2) << 1); // 0, 1, 4, 5
        const uint32_t a = r[e], b = r[e + 2];
        r[e] = bitop3<0xCA>(0x33333333u, a, b << 2);
        r[e + 2] = bitop3<0xCA>(0x33333333u, a >> 2, b);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t a = r[e], b = r[e + 4];
        if (e < NP) P[e < NP ? e : 0] = bitop3<0xCA>(0x0F0F0F0Fu, a, b << 4);
        if (e + 4 < NP) P[e + 4 < NP ? e + 4 : 0] = bitop3<0xCA>(0x0F0F0F0Fu, a >> 4, b);
    }
}

// Positions (within `valid`) where the pattern equals byte `byte` (0..3, static) of dword w.
template <int NP>
STRSIM_HD uint32_t eq_mask(const uint32_t (&P)[NP], uint32_t valid, uint32_t w, int byte)
{
    uint32_t acc = valid;
#pragma unroll
    for (int k = 0; k < NP; ++k) acc = bitop3<0x90>(acc, P[k], bit_fill(w, 8 * byte + k)); // acc & ~(P_k ^ m_k)
    return acc;
}

// Smallest NP in {5,6,7} that separates every byte of the pair: `vary` = OR of (byte ^ byte') over the
// pair's bytes folded to 8 bits.  (Bit 7 set = non-ASCII = not a lane-path row at all.)
STRSIM_HD int planes_needed(uint32_t vary) { return (vary & 0x40u) ? 7 : ((vary & 0x20u) ? 6 : 5); }

// Varying bits of the 64 bytes of two windows, folded to the low 8 bits; `any` gets the OR of all bytes.
STRSIM_HD uint32_t window_vary(const uint32_t (&wa)[8], const uint32_t (&wb)[8], uint32_t &any)
{
    // three-input OR (0xFE) / AND (0x80): 8 + 8 ops for the sixteen dwords
    uint32_t o = bitop3<0xFE>(wa[0], wa[1], wa[2]), n = bitop3<0x80>(wa[0], wa[1], wa[2]);
    o = bitop3<0xFE>(o, wa[3], wa[4]); n = bitop3<0x80>(n, wa[3], wa[4]);
    o = bitop3<0xFE>(o, wa[5], wa[6]); n = bitop3<0x80>(n, wa[5], wa[6]);
    o = bitop3<0xFE>(o, wa[7], wb[0]); n = bitop3<0x80>(n, wa[7], wb[0]);
    o = bitop3<0xFE>(o, wb[1], wb[2]); n = bitop3<0x80>(n, wb[1], wb[2]);
    o = bitop3<0xFE>(o, wb[3], wb[4]); n = bitop3<0x80>(n, wb[3], wb[4]);
    o = bitop3<0xFE>(o, wb[5], wb[6]); n = bitop3<0x80>(n, wb[5], wb[6]);
    o |= wb[7]; n &= wb[7];
    // fold the four byte lanes: OR of ORs, AND of ANDs
    uint32_t o8 = o | (o >> 16); o8 |= o8 >> 8;
    uint32_t n8 = n & (n >> 16); n8 &= n8 >> 8;
    any = o8 & 0xFFu;
    return (o8 ^ n8) & 0xFFu;
}

// ---------------------------------------------------------------------------------------------
// The same recurrence arranged for what gfx950 issues at the full rate (v_and / v_or / v_add / v_bitop3; shifts by
// v_add x, x; no v_alignbit, no left shift: bench_support/micro/op_cost.hip) -- 59 instead of 73 cycles of SIMD time per
// column:
//   * the pattern stays RIGHT-aligned (position j = bit j, planes as build_planes leaves them, no per-pair shift); rows at
//     and above lp belong to the neighbour string or are zero and never influence the rows below them (carries and
//     shifts only travel upwards);
//   * no per-column history: the distance after exactly lt columns is lt + (vertical +1 deltas) - (vertical -1 deltas)
//     over the lp pattern rows of THAT column, so each lane keeps a copy of (Pv, Mv) from the column where its own text
//     ends; a wave only pays for that (a compare and two bit-selects) in the columns tmin .. tmax where some lane ends;
//   * ~HP is carried instead of HP: (HP << 1) | 1 = ~(~HP + ~HP), which folds into the two v_bitop3 that consume it.
// lt, lp >= 1; tmin <= lt <= tmax, both lane-uniform.
// ---------------------------------------------------------------------------------------------
template <int NP>
STRSIM_HD uint32_t lev_myers32_snap(const uint32_t (&wt)[8], uint32_t lt, uint32_t tmin, uint32_t tmax, const uint32_t (&P)[NP],
                                    uint32_t lp)
{
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0u;
#pragma unroll
    for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
        if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
        // one straight-line block per group of columns (the match masks of the whole group are independent work the
        // scheduler can put between the dependent steps of the recurrence)
        // A lane whose text has ended keeps its (Pv, Mv): in the groups where that can happen (from tmin on) every column
        // runs under `column < lt` -- one compare per column and the exec mask, instead of copying the state aside.
        auto column = [&](int j) {
            const uint32_t Eq = eq_mask<NP>(P, 0xFFFFFFFFu, wt[j >> 2], j & 3);
            const uint32_t D0 = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq) | Mv; // (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv
            const uint32_t nHP = bitop3<0x0E>(Mv, D0, Pv);                 // ~HP = ~Mv & (D0 | Pv)
            const uint32_t nX = twice(nHP);                                 // ~((HP << 1) | 1)
            const uint32_t HN2 = twice(D0 & Pv);                            // HN << 1
            Pv = bitop3<0xF2>(HN2, D0, nX);                                 // (HN << 1) | ~(D0 | X)
            Mv = bitop3<0x50>(D0, D0, nX);                                  // D0 & X
        };
        if ((uint32_t)(COLS_PER_TEST * (g + 1)) >= tmin) {                 // (uniform) some lane's text may end in this group
#pragma unroll
            for (int jj = 0; jj < COLS_PER_TEST; ++jj)
                if ((uint32_t)(COLS_PER_TEST * g + jj) < lt) column(COLS_PER_TEST * g + jj);
        } else {
#pragma unroll
            for (int jj = 0; jj < COLS_PER_TEST; ++jj) column(COLS_PER_TEST * g + jj);
        }
    }
    const uint32_t rows = low_ones(lp);
    return lt + popc32(Pv & rows) - popc32(Mv & rows);
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
// The cores in ONE column loop over the text a (pattern = b, right-aligned planes): each column's match mask is built once
// and feeds the Levenshtein recurrence (lev_myers32_snap's arrangement), Jaro's first pass (strsim.rs:200-230) and the
// multiset intersection; Jaro's second pass (the zip of the flagged characters, :231-237) runs behind it and is the only
// other place a match mask is built -- two per column instead of four when all three are wanted (BASELINE config 4), and
// the single-measure kernels use the same loop with one core switched on.  What differs from jaro_match32 /
// multiset_isect32 above, all of it to spend fewer and cheaper instructions (bench_support/micro/op_cost.hip):
//   * no `live` mask: in the column groups below tmin every lane's text is still running, from tmin on a column runs under
//     `column < la` (the exec mask), so a lane whose text has ended simply stops updating ALL of its state;
//   * Jaro's window is ONE mask, ones at [max(i - bound, 0), i + bound], moved along by an add with carry-in (i < bound); its
//     upper end is not clamped to lb, the match masks are ([r4]; before: a mask for the upper edge, shifted and clamped, and a
//     variable shift of all-ones for the lower one);
//   * "a_i found a partner" is `cand != 0`, and the lowest candidate goes into the flags with one three-input op.
// la, lb >= 1; tmin <= la <= tmax, both lane-uniform.  Outputs: dist (edit distance), m / t (Jaro matches, unequal zipped
// pairs NOT halved), isect (sum of min counts); only those of the cores switched on are written.
// KEEP_EQ [r5]: Jaro's zip pass takes the match masks of the first pass from registers (one per column walked: up to 32)
// instead of building them again -- 5 instead of 15 instructions per column of that pass, for kernels that have the registers.
// ---------------------------------------------------------------------------------------------
template <int NP, bool DO_LEV, bool DO_JARO, bool DO_ISECT, bool KEEP_EQ = false>
STRSIM_HD void lane_cores32(const uint32_t (&wa)[8], uint32_t la, uint32_t tmin, uint32_t tmax, uint32_t lb,
                            const uint32_t (&P)[NP], uint32_t &dist, uint32_t &m_out, uint32_t &t_out, uint32_t &isect)
{
    const uint32_t lbmask = low_ones(lb);
    // Levenshtein
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0u;
    // Jaro, first pass
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u; // mx/2 - 1 (:200); mx == 1 only for the 1x1 case, window {i}
    uint32_t win = low_ones(bound + 1u); // Jaro's window, ones at [max(i - bound, 0), i + bound]: (win << 1) | (i < bound) moves it along
    uint32_t fb = 0u, fa = 0u;
    // (the match masks end at lb when Jaro is on: its window does not; the recurrence of Levenshtein never looks down from there)
    const uint32_t valid = DO_JARO ? lbmask : 0xFFFFFFFFu;
    // multiset intersection
    uint32_t used = 0u;
    uint32_t E[(DO_JARO && KEEP_EQ) ? 32 : 1]; // (every index is a compile-time constant: registers)
#pragma unroll
    for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
        if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
        auto column = [&](int i, bool live) {
            const uint32_t Eq = eq_mask<NP>(P, valid, wa[i >> 2], i & 3);
            // KEEP_EQ: the zip pass reads E[i] of EVERY column of a visited group (masked by a clear flag for columns past the end
            // of a), so it is written for every such column too -- outside the lane's `i < la` predicate (an
            // indeterminate read otherwise; at most COLS_PER_TEST - 1 extra masks per pair, in the text's last group only)
            if (DO_JARO && KEEP_EQ) E[(DO_JARO && KEEP_EQ) ? i : 0] = Eq;
            if (!live) return;
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
        if ((uint32_t)(COLS_PER_TEST * (g + 1)) >= tmin) {                 // (uniform) some lane's text may end in this group
#pragma unroll
            for (int jj = 0; jj < COLS_PER_TEST; ++jj) {
                const bool live = (uint32_t)(COLS_PER_TEST * g + jj) < la;
                if ((DO_JARO && KEEP_EQ) || live) column(COLS_PER_TEST * g + jj, live);
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < COLS_PER_TEST; ++jj) column(COLS_PER_TEST * g + jj, true);
        }
    }
    if (DO_LEV) {
        const uint32_t rows = lbmask;
        dist = la + popc32(Pv & rows) - popc32(Mv & rows);
    }
    if (DO_ISECT) isect = popc32(used);
    if (DO_JARO) {
        // second pass: the k-th flagged character of a against the k-th flagged character of b (ascending positions); they
        // are equal iff bit j_k of Eq(a_{i_k}) is set.  Columns at or beyond la have no flag, so no predicate is needed.
        // [r4] The unequal pairs are COLLECTED as bits of b (every partner is a different bit) and counted once at the end: one
        // three-input op per column where a compare and an add with carry stood.
        uint32_t unequal = 0u, rest = fb;
#pragma unroll
        for (int g = 0; g < 32 / COLS_PER_TEST; ++g) {
            if ((uint32_t)(COLS_PER_TEST * g) >= tmax) break;
#pragma unroll
            for (int ii = 0; ii < COLS_PER_TEST; ++ii) {
                const int i = COLS_PER_TEST * g + ii;
                const uint32_t on = bit_fill(fa, i);               // a_i was matched
                const uint32_t jbit = rest & (0u - rest) & on;     // its partner in the zip: lowest remaining flag of b
                rest ^= jbit;
                // (the first pass's mask ends at lb; jbit is a flagged position of b, so that makes no difference)
                const uint32_t Eq = (DO_JARO && KEEP_EQ) ? E[(DO_JARO && KEEP_EQ) ? i : 0] : eq_mask<NP>(P, 0xFFFFFFFFu, wa[i >> 2], i & 3);
                unequal = bitop3<0xF4>(unequal, jbit, Eq);         // unequal | (jbit & ~Eq)
            }
        }
        m_out = popc32(fb);
        t_out = popc32(unequal);
    }
}

// x / 3.0 for the sums Jaro's epilogue divides (strsim.rs:241-242), without the division: q = x * RN(1/3), one Newton correction with
// the exact residual -- r = fma(-3, q, x); q' = fma(r, RN(1/3), q) -- which is the correctly rounded quotient (Markstein's
// theorem: RN(1/3) is the correctly rounded reciprocal and q is within an ulp) and therefore the same double the reference's `/ 3.0`
// produces.  Three instructions where the IEEE division sequence is about a dozen, once per row in the store phase.  Not taken on
// faith: tests/test_lane_core_cpu.py::test_jaro_division_by_three_is_exact compares it with x / 3.0 for EVERY sum the table
// epilogues can see (m / la + m / lb + (m - t / 2) / m, lengths up to 64: 810 160 values).  These are explicit fused
// multiply-adds of a division algorithm, not contractions of the reference's arithmetic (-ffp-contract=off stays).
STRSIM_HD double div3_exact(double x)
{
    const double y = 1.0 / 3.0; // RN(1/3) = 0x3FD5555555555555
    const double q = x * y;
    const double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, y, q);
}

// The Jaro epilogue with its three integer quotients read from a table: q[a * QTAB_N + b] = (double)a / (double)b for
// 0 <= a, b <= 64 (b = 0: unused), computed once with the same IEEE division -- an f64 division is ~40 VALU-equivalents
// on CDNA.  Valid for strings of at most 32 characters.  (Measured on cfg2: Jaro +1.4 %; the single division of
// Jaccard / Dice is cheaper than a dependent table load, -6 %, so those keep dividing.)
constexpr int QTAB_N = 65;
STRSIM_HD double epilogue_jaro_q(const double *q, uint32_t m, uint32_t t, uint32_t la, uint32_t lb)
{
    if (m == 0) return 0.0;
    return div3_exact(q[m * QTAB_N + la] + q[m * QTAB_N + lb] + q[(m - t / 2) * QTAB_N + m]);
}
} // namespace strsim
