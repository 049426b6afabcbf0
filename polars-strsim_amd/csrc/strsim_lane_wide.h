// strsim_lane_wide.h -- per-lane cores for strings of up to 32*W bytes (W = 2: 64, W = 3: 96, W = 4: 128), ASCII.
//
// Same bit-sliced formulation as strsim_lane_core.h, with every mask W words wide.  The text is not
// kept in registers: it is read one dword (4 bytes) at a time through `txt(g)` (an LDS column per lane
// on the GPU, a plain array in the CPU harness), so the column loop is a runtime loop of 4-step bodies
// and the code stays small for any W.  Host/device portable; bit-exact against the oracle by
// construction (tests/test_lane_core_cpu.py).
//
// Semantics restated (reference = /root/reference/src/expressions/strsim.rs):
//   lev_wide       -> integer edit distance of Levenshtein::compute      (:141-160)
//   jaro_wide      -> (m, t) of Jaro::compute                            (:200-237)
//   isect_wide     -> sum of min(countA[c], countB[c]) of Jaccard/Dice   (:297-305, :333-341)
#pragma once
#include "strsim_lane_core.h"

namespace strsim {

// ---- W-word helpers (word 0 = least significant) -------------------------------------------------
template <int W>
STRSIM_HD void low_ones_wide(uint32_t k, uint32_t (&m)[W]) // ones at bits [0, k), k <= 32*W
{
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t lo = 32u * (uint32_t)w;
        m[w] = k <= lo ? 0u : low_ones(k - lo);
    }
}

// x = (x << 1) | in  (in = 0 or 1), as an add-with-carry chain
template <int W>
STRSIM_HD void shl1_in(uint32_t (&x)[W], uint32_t in)
{
    uint32_t c = in;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t top = x[w] >> 31;
        x[w] = (x[w] << 1) | c;
        c = top;
    }
}

// lowest set bit of the W-word integer x (all zero if x == 0): x & ~(x - 1)
template <int W>
STRSIM_HD void lowest_bit_wide(const uint32_t (&x)[W], uint32_t (&bit)[W])
{
    uint32_t borrow = 1u;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const uint32_t d = x[w] - borrow;
        borrow = (x[w] < borrow) ? 1u : 0u;
        bit[w] = x[w] & ~d;
    }
}

template <int W>
STRSIM_HD uint32_t any_wide(const uint32_t (&x)[W])
{
    uint32_t o = x[0];
#pragma unroll
    for (int w = 1; w < W; ++w) o |= x[w];
    return o;
}

// ---- W-word shifts and decrements as carry chains --------------------------------------------------
// On gfx950 an add-with-carry costs one full-rate issue slot and a 32-bit left shift two (bench_support/micro/op_cost.hip),
// so "shift the W-word mask left by one and bring a bit in" is W v_addc_co_u32 with the incoming bit as the first carry, and
// "x - 1" (for the lowest set bit, x & ~(x - 1)) is one v_subrev_co_u32 + W - 1 v_subbrev_co_u32.  A chain lives in ONE asm
// statement (VCC must survive from one instruction to the next); the host versions are the plain arithmetic.
#if defined(__HIP_DEVICE_COMPILE__)
// x = (x << 1) | 1
STRSIM_HD void shl1_one(uint32_t (&x)[1]) { x[0] = x[0] + x[0] + 1u; }
STRSIM_HD void shl1_one(uint32_t (&x)[2])
{
    asm("s_mov_b64 vcc, -1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "+v"(x[0]), "+v"(x[1]) : : "vcc");
}
STRSIM_HD void shl1_one(uint32_t (&x)[3])
{
    asm("s_mov_b64 vcc, -1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : : "vcc");
}
STRSIM_HD void shl1_one(uint32_t (&x)[4])
{
    asm("s_mov_b64 vcc, -1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, %3, %3, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : : "vcc");
}
// x = (x << 1) | (a >= b)
STRSIM_HD void shl1_ge(uint32_t (&x)[1], uint32_t a, uint32_t b) { x[0] = x[0] + x[0] + (a >= b ? 1u : 0u); }
STRSIM_HD void shl1_ge(uint32_t (&x)[2], uint32_t a, uint32_t b)
{
    asm("v_cmp_ge_u32_e32 vcc, %2, %3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "+v"(x[0]), "+v"(x[1]) : "s"(a), "v"(b) : "vcc"); // (a: the column index, wave-uniform)
}
STRSIM_HD void shl1_ge(uint32_t (&x)[3], uint32_t a, uint32_t b)
{
    asm("v_cmp_ge_u32_e32 vcc, %3, %4\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "s"(a), "v"(b) : "vcc");
}
STRSIM_HD void shl1_ge(uint32_t (&x)[4], uint32_t a, uint32_t b)
{
    asm("v_cmp_ge_u32_e32 vcc, %4, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, %3, %3, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "s"(a), "v"(b) : "vcc");
}
// x = (x << 1) | (a <= b)
STRSIM_HD void shl1_le(uint32_t (&x)[1], uint32_t a, uint32_t b) { x[0] = x[0] + x[0] + (a <= b ? 1u : 0u); }
STRSIM_HD void shl1_le(uint32_t (&x)[2], uint32_t a, uint32_t b)
{
    asm("v_cmp_le_u32_e32 vcc, %2, %3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "+v"(x[0]), "+v"(x[1]) : "s"(a), "v"(b) : "vcc"); // (a: the column index + 1, wave-uniform)
}
STRSIM_HD void shl1_le(uint32_t (&x)[3], uint32_t a, uint32_t b)
{
    asm("v_cmp_le_u32_e32 vcc, %3, %4\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "s"(a), "v"(b) : "vcc");
}
STRSIM_HD void shl1_le(uint32_t (&x)[4], uint32_t a, uint32_t b)
{
    asm("v_cmp_le_u32_e32 vcc, %4, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, %3, %3, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "s"(a), "v"(b) : "vcc");
}
// (x << 1) | (v != 0), one word
STRSIM_HD uint32_t shl1_nz(uint32_t x, uint32_t v)
{
    asm("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(v) : "vcc");
    return x;
}
// x + (v != 0)
STRSIM_HD uint32_t add_nz(uint32_t x, uint32_t v)
{
    asm("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(x) : "v"(v) : "vcc");
    return x;
}
// d = x - 1 over W words
STRSIM_HD void minus1_wide(const uint32_t (&x)[1], uint32_t (&d)[1]) { d[0] = x[0] - 1u; }
STRSIM_HD void minus1_wide(const uint32_t (&x)[2], uint32_t (&d)[2])
{
    asm("v_subrev_co_u32_e32 %0, vcc, 1, %2\n\tv_subbrev_co_u32_e32 %1, vcc, 0, %3, vcc"
        : "=&v"(d[0]), "=&v"(d[1]) : "v"(x[0]), "v"(x[1]) : "vcc");
}
STRSIM_HD void minus1_wide(const uint32_t (&x)[3], uint32_t (&d)[3])
{
    asm("v_subrev_co_u32_e32 %0, vcc, 1, %3\n\tv_subbrev_co_u32_e32 %1, vcc, 0, %4, vcc\n\tv_subbrev_co_u32_e32 %2, vcc, 0, %5, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]) : "v"(x[0]), "v"(x[1]), "v"(x[2]) : "vcc");
}
STRSIM_HD void minus1_wide(const uint32_t (&x)[4], uint32_t (&d)[4])
{
    asm("v_subrev_co_u32_e32 %0, vcc, 1, %4\n\tv_subbrev_co_u32_e32 %1, vcc, 0, %5, vcc\n\t"
        "v_subbrev_co_u32_e32 %2, vcc, 0, %6, vcc\n\tv_subbrev_co_u32_e32 %3, vcc, 0, %7, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "vcc");
}
#else
template <int W> STRSIM_HD void shl1_one(uint32_t (&x)[W]) { shl1_in<W>(x, 1u); }
template <int W> STRSIM_HD void shl1_ge(uint32_t (&x)[W], uint32_t a, uint32_t b) { shl1_in<W>(x, a >= b ? 1u : 0u); }
template <int W> STRSIM_HD void shl1_le(uint32_t (&x)[W], uint32_t a, uint32_t b) { shl1_in<W>(x, a <= b ? 1u : 0u); }
STRSIM_HD uint32_t shl1_nz(uint32_t x, uint32_t v) { return (x << 1) | (v ? 1u : 0u); }
STRSIM_HD uint32_t add_nz(uint32_t x, uint32_t v) { return x + (v ? 1u : 0u); }
template <int W> STRSIM_HD void minus1_wide(const uint32_t (&x)[W], uint32_t (&d)[W])
{
    uint32_t borrow = 1u;
    for (int w = 0; w < W; ++w) {
        d[w] = x[w] - borrow;
        borrow = (x[w] < borrow) ? 1u : 0u;
    }
}
#endif

// s = x + y over W words; x = x << 1 over W words
#if defined(__HIP_DEVICE_COMPILE__)
STRSIM_HD void add_wide(const uint32_t (&x)[1], const uint32_t (&y)[1], uint32_t (&s)[1]) { s[0] = x[0] + y[0]; }
STRSIM_HD void add_wide(const uint32_t (&x)[2], const uint32_t (&y)[2], uint32_t (&s)[2])
{
    asm("v_add_co_u32_e32 %0, vcc, %2, %4\n\tv_addc_co_u32_e32 %1, vcc, %3, %5, vcc"
        : "=&v"(s[0]), "=&v"(s[1]) : "v"(x[0]), "v"(x[1]), "v"(y[0]), "v"(y[1]) : "vcc");
}
STRSIM_HD void add_wide(const uint32_t (&x)[3], const uint32_t (&y)[3], uint32_t (&s)[3])
{
    asm("v_add_co_u32_e32 %0, vcc, %3, %6\n\tv_addc_co_u32_e32 %1, vcc, %4, %7, vcc\n\tv_addc_co_u32_e32 %2, vcc, %5, %8, vcc"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(y[0]), "v"(y[1]), "v"(y[2]) : "vcc");
}
STRSIM_HD void add_wide(const uint32_t (&x)[4], const uint32_t (&y)[4], uint32_t (&s)[4])
{
    asm("v_add_co_u32_e32 %0, vcc, %4, %8\n\tv_addc_co_u32_e32 %1, vcc, %5, %9, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %6, %10, vcc\n\tv_addc_co_u32_e32 %3, vcc, %7, %11, vcc"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]) : "vcc");
}
STRSIM_HD void shl1_zero(uint32_t (&x)[1]) { x[0] = x[0] + x[0]; }
STRSIM_HD void shl1_zero(uint32_t (&x)[2])
{
    asm("v_add_co_u32_e32 %0, vcc, %0, %0\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc" : "+v"(x[0]), "+v"(x[1]) : : "vcc");
}
STRSIM_HD void shl1_zero(uint32_t (&x)[3])
{
    asm("v_add_co_u32_e32 %0, vcc, %0, %0\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\tv_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : : "vcc");
}
STRSIM_HD void shl1_zero(uint32_t (&x)[4])
{
    asm("v_add_co_u32_e32 %0, vcc, %0, %0\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\tv_addc_co_u32_e32 %3, vcc, %3, %3, vcc"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : : "vcc");
}
#else
template <int W> STRSIM_HD void add_wide(const uint32_t (&x)[W], const uint32_t (&y)[W], uint32_t (&s)[W])
{
    uint32_t carry = 0u;
    for (int w = 0; w < W; ++w) {
        const uint64_t t = (uint64_t)x[w] + y[w] + carry;
        s[w] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
}
template <int W> STRSIM_HD void shl1_zero(uint32_t (&x)[W]) { shl1_in<W>(x, 0u); }
#endif

// 0xFFFFFFFF where the signed x is negative, else 0 (v_ashrrev_i32: full rate)
STRSIM_HD uint32_t sign_fill(uint32_t x) { return (uint32_t)((int32_t)x >> 31); }

// Planes of a 32*W-byte window (dwords wp[0 .. 8W)): P[k][w] bit i = bit k of byte 32w + i.
template <int NP, int W>
STRSIM_HD void build_planes_wide(const uint32_t (&wp)[8 * W], uint32_t (&P)[NP][W])
{
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint32_t grp[8], pl[NP];
#pragma unroll
        for (int d = 0; d < 8; ++d) grp[d] = wp[8 * w + d];
        build_planes<NP>(grp, pl);
#pragma unroll
        for (int k = 0; k < NP; ++k) P[k][w] = pl[k];
    }
}

// match mask of byte `byte` (static 0..3) of dword `c4` against all 32*W pattern positions
template <int NP, int W>
STRSIM_HD void eq_wide(const uint32_t (&P)[NP][W], const uint32_t (&valid)[W], uint32_t c4, int byte, uint32_t (&Eq)[W])
{
    uint32_t m[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) m[k] = bit_fill(c4, 8 * byte + k);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint32_t acc = valid[w];
#pragma unroll
        for (int k = 0; k < NP; ++k) acc = bitop3<0x90>(acc, P[k][w], m[k]);
        Eq[w] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Levenshtein, W-word Myers/Hyyro in the arrangement of lev_myers32_snap (strsim_lane_core.h): the pattern b is RIGHT-aligned
// (bit j = b[j], planes of the window that starts at b; rows at and above lb belong to whatever follows b and never
// influence the rows below them), ~HP is carried, every shift and the W-word addition are carry chains, and a lane whose
// text has ended keeps its (Pv, Mv): text dwords below gfull (lane-uniform: every lane's text is still running there) run
// as they are, the ones from gfull on run each column under `column < la`.  The distance after exactly la columns is
// la + popc(Pv) - popc(Mv) over the pattern's rows.  la, lb >= 1; ng4 = text dwords the wave walks (uniform).
// ---------------------------------------------------------------------------------------------
template <int NP, int W, class Txt>
STRSIM_HD uint32_t lev_wide(const Txt &txt, uint32_t la, uint32_t gfull, uint32_t ng4, const uint32_t (&P)[NP][W], uint32_t lb)
{
    uint32_t all[W], Pv[W], Mv[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { all[w] = 0xFFFFFFFFu; Pv[w] = 0xFFFFFFFFu; Mv[w] = 0u; }
    auto column = [&](uint32_t c4, int jj) {
        uint32_t Eq[W], X[W], S[W], D0[W], nX[W], HN2[W];
        eq_wide<NP, W>(P, all, c4, jj, Eq);
#pragma unroll
        for (int w = 0; w < W; ++w) X[w] = Eq[w] & Pv[w];
        add_wide(X, Pv, S);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            D0[w] = bitop3<0xBE>(S[w], Pv[w], Eq[w]) | Mv[w]; // (((Eq & Pv) + Pv) ^ Pv) | Eq | Mv
            nX[w] = bitop3<0x0E>(Mv[w], D0[w], Pv[w]);        // ~HP = ~Mv & (D0 | Pv)
            HN2[w] = D0[w] & Pv[w];                            // HN
        }
        shl1_zero(nX);  // ~((HP << 1) | 1)
        shl1_zero(HN2); // HN << 1
#pragma unroll
        for (int w = 0; w < W; ++w) {
            Pv[w] = bitop3<0xF2>(HN2[w], D0[w], nX[w]); // (HN << 1) | ~(D0 | X)
            Mv[w] = bitop3<0x50>(D0[w], D0[w], nX[w]);  // D0 & X
        }
    };
    uint32_t g = 0;
    for (; g < gfull && g < ng4; ++g) {
        const uint32_t c4 = txt(g);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) column(c4, jj);
    }
    for (; g < ng4; ++g) {
        const uint32_t c4 = txt(g);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            if (4u * g + (uint32_t)jj < la) column(c4, jj);
    }
    uint32_t rows[W];
    low_ones_wide<W>(lb, rows);
    uint32_t up = 0u, down = 0u;
#pragma unroll
    for (int w = 0; w < W; ++w) { up += popc32(Pv[w] & rows[w]); down += popc32(Mv[w] & rows[w]); }
    return la + up - down;
}

// ---------------------------------------------------------------------------------------------
// Jaro matching, W words.  Pattern = b (right-aligned planes: bit j = b[j]; wp = its bytes), text = a through txt().
// la, lb >= 1; nb4 = dwords of b to walk in the second pass (>= ceil(lb / 4), uniform).  Returns m and t (not halved).
//
// Transpositions [r3]: the characters of the matched a_i are collected, in the order of a, as a byte string SA -- `sa.put(k, c,
// hit)` stores c at position k if hit, `sa.get(k)` reads position k; on the GPU SA overwrites the front of the text column in
// LDS (there are never more matches than columns walked, and a group's four characters are in a register before its columns
// run) --, and the second pass walks b: the k-th flagged b_j against SA[k] (strsim.rs:222-233 zips exactly these two
// sequences; `put` may ignore `hit`: a character stored at k without a match is overwritten by the next match, k not having
// moved; `get4(k)`, k a multiple of 4, reads positions k .. k + 3 as one dword; `zip_over_matches(m, W, nb4)` -> 0, or the
// wave's largest m when the zip pass should walk SA instead of b: see below).  Rounds 1-2 walked a again and rebuilt every column's match mask to test one bit of it: 9 + 9 W instructions
// per column of a against about ten per position of b here, and the flags of a (kept in LDS between the passes) are gone.
// ---------------------------------------------------------------------------------------------
template <int NP, int W, class Txt, class Sa>
STRSIM_HD void jaro_wide(const Txt &txt, uint32_t la, uint32_t gfull, uint32_t ng4, uint32_t lb, uint32_t nb4, const uint32_t (&P)[NP][W],
                         const uint32_t (&wp)[8 * W], const Sa &sa, uint32_t &m_out, uint32_t &t_out)
{
    // Instruction diet: the window mask moves along as a carry chain, "this column is past the end of a" is the
    // sign of a running counter (and only looked at from the text dword on in which some lane's text ends: gfull), the lowest
    // candidate is folded into the flags with one three-input op per word.
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u;
    // the match window as ONE mask: ones at [i - bound, i + bound] (from 0 on while i < bound; not clamped to lb -- the match
    // masks already are): (win << 1) | (i < bound) moves it along.  [r4] Two masks ([0, i + bound] and [0, i - bound)), two carry
    // chains and two mask operations per word and column before.
    uint32_t lbmask[W], win[W], fb[W];
    low_ones_wide<W>(lb, lbmask);
    low_ones_wide<W>(bound + 1u, win);
#pragma unroll
    for (int w = 0; w < W; ++w) fb[w] = 0u;
    uint32_t m = 0u;
    // a column: TAIL = some lane's text may have ended (dead: all ones once i >= la); the text dwords below gfull run without it
    auto column = [&](uint32_t c4, int ii, uint32_t i, uint32_t dead, auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        uint32_t Eq[W], cand[W], d[W];
        eq_wide<NP, W>(P, lbmask, c4, ii, Eq);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            cand[w] = bitop3<0x40>(Eq[w], win[w], fb[w]); // Eq & win & ~fb
            if (TAIL) cand[w] &= ~dead;
        }
        minus1_wide(cand, d);
#pragma unroll
        for (int w = 0; w < W; ++w) fb[w] = bitop3<0xF4>(fb[w], cand[w], d[w]); // fb | (cand & ~(cand - 1))
        const uint32_t hit = any_wide<W>(cand);
        sa.put(m, (c4 >> (8 * ii)) & 0xFFu, hit);
        m = add_nz(m, hit);
        shl1_le(win, wave_uniform(i + 1u), bound); // (the column index: uniform -- shl1_le takes it from a scalar register)
    };
    uint32_t g = 0;
    for (; g < gfull && g < ng4; ++g) {
        const uint32_t c4 = txt(g);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) column(c4, ii, 4u * g + (uint32_t)ii, 0u, std::false_type{});
    }
    uint32_t left = la - 1u - 4u * g; // la - 1 - i: negative from column la on
    for (; g < ng4; ++g) {
        const uint32_t c4 = txt(g);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const uint32_t dead = sign_fill(left);
            left -= 1u;
            column(c4, ii, 4u * g + (uint32_t)ii, dead, std::true_type{});
        }
    }
    // [r5] The zip pass, one of two ways -- the wave decides (sa.zip_over_matches: uniform):
    //   * over the MATCHED characters of a (SA[0 .. m)): the k-th one against the lowest flag of b not used yet, equal iff that
    //     bit is in the character's match mask (what the 32-byte cores do) -- 5 + 9 W instructions per match;
    //   * over the positions of b (below): 7 instructions per position.
    // The text is the shorter string and m <= its length, so the first way wins whenever the pattern is much longer than the text:
    // half of cfg3's rows of 33..128 bytes are an independent pair with a short side of a dozen characters (7 x 65 positions
    // against 23 x 12 matches).  Lanes with fewer matches than the wave's maximum have no flag left: their steps change nothing.
    const uint32_t kmax = sa.zip_over_matches(m, (uint32_t)W, nb4);
    if (kmax != 0u) {
        uint32_t all[W], rest[W], uneq[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { all[w] = 0xFFFFFFFFu; rest[w] = fb[w]; uneq[w] = 0u; }
        for (uint32_t k0 = 0u; k0 < kmax; k0 += 4u) {
            const uint32_t c4 = sa.get4(k0); // SA[k0 .. k0 + 3]
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                uint32_t Eq[W], d[W];
                eq_wide<NP, W>(P, all, c4, jj, Eq);
                minus1_wide(rest, d);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uneq[w] |= bitop3<0x10>(rest[w], d[w], Eq[w]); // lowest flag left (rest & ~(rest - 1)), if not in Eq
                    rest[w] &= d[w];                                // ... is used up
                }
            }
        }
        uint32_t tt = 0u;
#pragma unroll
        for (int w = 0; w < W; ++w) tt += popc32(uneq[w]);
        m_out = m;
        t_out = tt;
        return;
    }
    uint32_t t = 0u, k = 0u;
    unrolled_until<0, 8 * W>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if ((uint32_t)g >= nb4) return false;
        const uint32_t word = wp[g];
        constexpr int j0 = 4 * g;
        // the four positions of this dword: flags and SA positions first, so that the four reads are in flight together.  A flag
        // is kept as a MASK (all ones: b_j was matched): k - mask steps the position, (character ^ byte) & mask is non-zero for
        // an unequal pair -- seven instructions per position where the compiler made ten of the 0 / 1 form ([r4]).
        uint32_t flm[4], ach[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            flm[jj] = bit_fill(fb[j0 >> 5], (j0 & 31) + jj);
            ach[jj] = sa.get(k);
            k -= flm[jj];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) t = add_nz(t, bitop3<0x28>(ach[jj], (word >> (8 * jj)) & 0xFFu, flm[jj])); // (a ^ b) & c
        return true;
    });
    m_out = m;
    t_out = t;
}

// Multiset intersection size, W words (see multiset_isect32).
template <int NP, int W, class Txt>
STRSIM_HD uint32_t isect_wide(const Txt &txt, uint32_t la, uint32_t gfull, uint32_t ng4, uint32_t lb, const uint32_t (&P)[NP][W])
{
    uint32_t lbmask[W], used[W];
    low_ones_wide<W>(lb, lbmask);
#pragma unroll
    for (int w = 0; w < W; ++w) used[w] = 0u;
    auto column = [&](uint32_t c4, int ii, uint32_t dead, auto tail) { // (TAIL: as in jaro_wide)
        constexpr bool TAIL = decltype(tail)::value;
        uint32_t Eq[W], cand[W], d[W];
        eq_wide<NP, W>(P, lbmask, c4, ii, Eq);
#pragma unroll
        for (int w = 0; w < W; ++w) cand[w] = TAIL ? bitop3<0x10>(Eq[w], used[w], dead) : (Eq[w] & ~used[w]); // Eq & ~used (& ~dead)
        minus1_wide(cand, d);
#pragma unroll
        for (int w = 0; w < W; ++w) used[w] = bitop3<0xF4>(used[w], cand[w], d[w]); // used | lowest candidate
    };
    uint32_t g = 0;
    for (; g < gfull && g < ng4; ++g) {
        const uint32_t c4 = txt(g);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) column(c4, ii, 0u, std::false_type{});
    }
    uint32_t left = la - 1u - 4u * g; // la - 1 - i: negative from column la on
    for (; g < ng4; ++g) {
        const uint32_t c4 = txt(g);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const uint32_t dead = sign_fill(left);
            left -= 1u;
            column(c4, ii, dead, std::true_type{});
        }
    }
    uint32_t n = 0u;
#pragma unroll
    for (int w = 0; w < W; ++w) n += popc32(used[w]);
    return n;
}

// ---------------------------------------------------------------------------------------------
// One lane's result, strings of 1..32*W ASCII bytes each (the empty cases never reach the wide path).
// wp = the pattern window: the 32W bytes that START at b.  a0w / b0w = first dwords of a and b (Jaro-Winkler prefix).
// gfull <= la / 4 for every lane of the wave (text dwords in which no lane's text ends), ng4 = text dwords to walk, nb4 = pattern
// dwords to walk (Jaro's second pass); sa: see jaro_wide.
// ---------------------------------------------------------------------------------------------
template <int MEASURE, int NP, int W, class Txt, class Sa>
STRSIM_HD double lane_wide_result(const Txt &txt, uint32_t la, uint32_t gfull, uint32_t ng4, const uint32_t (&wp)[8 * W], uint32_t lb,
                                  uint32_t nb4, uint32_t a0w, uint32_t b0w, const Sa &sa)
{
    uint32_t P[NP][W];
    build_planes_wide<NP, W>(wp, P);
    if (MEASURE == LEVENSHTEIN) {
        return epilogue_levenshtein(lev_wide<NP, W>(txt, la, gfull, ng4, P, lb), la, lb);
    } else if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        uint32_t m, t;
        jaro_wide<NP, W>(txt, la, gfull, ng4, lb, nb4, P, wp, sa, m, t);
        const double j = epilogue_jaro(m, t, la, lb);
        return MEASURE == JARO ? j : epilogue_jaro_winkler(j, common_prefix4(a0w, la, b0w, lb));
    } else {
        const uint32_t isect = isect_wide<NP, W>(txt, la, gfull, ng4, lb, P);
        return MEASURE == JACCARD ? epilogue_jaccard(isect, la, lb) : epilogue_sorensen_dice(isect, la, lb);
    }
}

} // namespace strsim
