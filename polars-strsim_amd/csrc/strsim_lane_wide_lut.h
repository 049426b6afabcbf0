// strsim_lane_wide_lut.h -- match masks of the W-word cores (strsim_lane_wide.h) from per-lane LDS tables.
//
// A column's match mask costs the bit-fill form five v_bfe_i32 and 5 W three-input ops (25 of the ~57 vector instructions of a
// four-word Jaro column).  The split tables of strsim_lane_lut.h -- Eq(c) = L[c & 7] & M[(c >> 3) & 3], built once per pair from
// the planes with one three-input op per entry and word -- bring that to two address forms (one v_perm_b32 each), two LDS reads
// and W ANDs.  The W words of an entry of a lane lie together (entry e of lane l at (e << ES) + l * 4 WS bytes from the tables'
// start, WS = 2 words for W = 2, 4 for W = 3 and 4), so an entry is ONE ds_read_b64 / ds_read_b128 -- 256 bytes per LDS clock,
// conflict-free whatever entries the lanes ask for (the entry index moves the address by whole multiples of the bank period).
// The tables must start at a multiple of 16 << ES bytes (k_wide_bins: one wave per workgroup, the tables first in its LDS).
//
// Semantics are those of strsim_lane_wide.h (reference strsim.rs:141-160, :200-237, :297-305, :333-341): only how a column's
// match mask is produced differs.  Host/device portable: the CPU harness runs the same cores with the tables in an array.
#pragma once
#include "strsim_lane_wide.h"

namespace strsim {

template <int W>
struct WideLut {
    static constexpr int WS = W == 2 ? 2 : 4;              // words of an entry as stored
    static constexpr int ES = W == 2 ? 9 : 10;             // log2(bytes of one entry of the 64 lanes)
    static constexpr int ENTRIES = 12;                     // L: 0..7 (planes 0..2), M: 8..11 (planes 3..4, and the valid mask)
    static constexpr int BYTES = ENTRIES << ES;            // 6 KB / 12 KB per wave
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lanereg; // byte 0: low byte of the lane's offset inside an entry (lane * 4 WS), byte 2: bits 16.. of the tables' address
    uint32_t k1rep;   // bits 8..15 of (tables' address + the lane's offset), in all four bytes
    uint32_t mine;    // LDS byte address of the lane's words of entry 0
#else
    uint32_t tab[ENTRIES][W];
#endif
};

// the handle of the calling lane for tables at LDS byte address `base` (a multiple of 16 << ES); host: an empty table
template <int W>
STRSIM_HD WideLut<W> wide_lut_at(uint32_t base, uint32_t lane)
{
    WideLut<W> t{};
#if defined(__HIP_DEVICE_COMPILE__)
    t.mine = base + lane * 4u * (uint32_t)WideLut<W>::WS;
    t.lanereg = (t.mine & 0xFFu) | (t.mine & 0x00FF0000u);
    t.k1rep = ((t.mine >> 8) & 0xFFu) * 0x01010101u;
#else
    (void)base; (void)lane;
#endif
    return t;
}

// the tables of a pattern from its planes 0..4; `valid`: the positions that may match at all
template <int NP, int W>
STRSIM_HD void wide_lut_build(WideLut<W> &t, const uint32_t (&P)[NP][W], const uint32_t (&valid)[W])
{
    static_assert(NP >= 5, "the tables cover planes 0..4");
    uint32_t e[WideLut<W>::ENTRIES][W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        // L[l]: P0 == l0 && P1 == l1 && P2 == l2 -- truth-table bit (a << 2 | b << 1 | c) with a = P0, b = P1, c = P2
        e[0][w] = bitop3<0x01>(P[0][w], P[1][w], P[2][w]);
        e[1][w] = bitop3<0x10>(P[0][w], P[1][w], P[2][w]);
        e[2][w] = bitop3<0x04>(P[0][w], P[1][w], P[2][w]);
        e[3][w] = bitop3<0x40>(P[0][w], P[1][w], P[2][w]);
        e[4][w] = bitop3<0x02>(P[0][w], P[1][w], P[2][w]);
        e[5][w] = bitop3<0x20>(P[0][w], P[1][w], P[2][w]);
        e[6][w] = bitop3<0x08>(P[0][w], P[1][w], P[2][w]);
        e[7][w] = bitop3<0x80>(P[0][w], P[1][w], P[2][w]);
        // M[m]: P3 == m0 && P4 == m1 && valid  (a = P3, b = P4, c = valid)
        e[8][w] = bitop3<0x02>(P[3][w], P[4][w], valid[w]);
        e[9][w] = bitop3<0x20>(P[3][w], P[4][w], valid[w]);
        e[10][w] = bitop3<0x08>(P[3][w], P[4][w], valid[w]);
        e[11][w] = bitop3<0x80>(P[3][w], P[4][w], valid[w]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int q = 0; q < WideLut<W>::ENTRIES; ++q) {
        const uint32_t addr = t.mine + ((uint32_t)q << WideLut<W>::ES);
        if (W == 2) {
            *reinterpret_cast<__attribute__((address_space(3))) uint2 *>((uintptr_t)addr) = make_uint2(e[q][0], e[q][1]);
        } else {
            *reinterpret_cast<__attribute__((address_space(3))) uint4 *>((uintptr_t)addr) =
                make_uint4(e[q][0], e[q][1], e[q][2], W == 4 ? e[q][W - 1] : 0u);
        }
    }
#else
#pragma unroll
    for (int q = 0; q < WideLut<W>::ENTRIES; ++q)
#pragma unroll
        for (int w = 0; w < W; ++w) t.tab[q][w] = e[q][w];
#endif
}

template <int NP, int W>
struct EqTab {
    const WideLut<W> &t;
    const uint32_t (&P)[NP][W]; // (planes 5.. for the rounds that need them)
    // a text dword's four characters as table coordinates: byte j of l / m = bits 8..15 of the address of character j's entry
    struct Group { uint32_t l, m, c4; };
    STRSIM_HD Group group(uint32_t c4) const
    {
        Group g;
        g.c4 = c4;
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int S = WideLut<W>::ES - 8;             // an entry index moves address bits ES.., i.e. bits S.. of byte 1
        g.l = bitop3<0xEA>(c4 << S, 0x07070707u << S, t.k1rep);                              // ((c & 7) << S) | K1
        g.m = bitop3<0xEA>(S == 2 ? c4 >> 1 : c4 >> 2, 0x03030303u << S, t.k1rep | (0x08080808u << S)); // ((8 + (c >> 3 & 3)) << S) | K1
#else
        g.l = c4 & 0x07070707u;
        g.m = ((c4 >> 3) & 0x03030303u) | 0x08080808u;
#endif
        return g;
    }
    STRSIM_HD void entry(uint32_t idx, int byte, uint32_t (&v)[W]) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // v_perm_b32 {S0 = idx (bytes 4..7), S1 = lanereg (bytes 0..3)}: [lanereg.b0, idx.b<byte>, lanereg.b2, 0]
        const uint32_t addr = __builtin_amdgcn_perm(idx, t.lanereg, 0x0C020000u | ((4u + (uint32_t)byte) << 8));
        if (W == 2) {
            const uint2 r = *reinterpret_cast<const __attribute__((address_space(3))) uint2 *>((uintptr_t)addr);
            v[0] = r.x; v[1] = r.y;
        } else {
            const uint4 r = *reinterpret_cast<const __attribute__((address_space(3))) uint4 *>((uintptr_t)addr);
            v[0] = r.x; v[1] = r.y; v[2 < W ? 2 : 0] = r.z;
            if (W == 4) v[W - 1] = r.w;
        }
#else
        const uint32_t e = (idx >> (8 * byte)) & 0xFFu;
#pragma unroll
        for (int w = 0; w < W; ++w) v[w] = t.tab[e][w];
#endif
    }
    struct Col { uint32_t L[W], M[W]; };
    STRSIM_HD Col fetch(const Group &g, int byte) const
    {
        Col c;
        entry(g.l, byte, c.L);
        entry(g.m, byte, c.M);
        return c;
    }
    STRSIM_HD void mask(const Group &g, const Col &c, int byte, uint32_t (&Eq)[W]) const
    {
        if (NP > 5) { // planes 5.. by bit fills on top of the two entries
            uint32_t m[NP];
#pragma unroll
            for (int k = 5; k < NP; ++k) m[k] = bit_fill(g.c4, 8 * byte + k);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint32_t acc = c.L[w] & c.M[w];
#pragma unroll
                for (int k = 5; k < NP; ++k) acc = bitop3<0x90>(acc, P[k < NP ? k : 0][w], m[k < NP ? k : 0]);
                Eq[w] = acc;
            }
        } else {
#pragma unroll
            for (int w = 0; w < W; ++w) Eq[w] = c.L[w] & c.M[w];
        }
    }
};

// lane_wide_result (strsim_lane_wide.h) with the match masks out of `lut` (its LDS must not hold anything else of this lane)
template <int MEASURE, int NP, int W, class Txt, class Sa>
STRSIM_HD double lane_wide_result_lut(WideLut<W> &lut, const Txt &txt, uint32_t la, uint32_t gfull, uint32_t ng4, const uint32_t (&wp)[8 * W],
                                      uint32_t lb, uint32_t nb4, uint32_t a0w, uint32_t b0w, const Sa &sa)
{
    uint32_t P[NP][W], valid[W];
    build_planes_wide<NP, W>(wp, P);
    low_ones_wide<W>(MEASURE == LEVENSHTEIN ? 32u * W : lb, valid);
    wide_lut_build<NP, W>(lut, P, valid);
    const EqTab<NP, W> eq{lut, P};
    return lane_wide_from<MEASURE, W>(txt, la, gfull, ng4, eq, wp, lb, nb4, a0w, b0w, sa);
}

} // namespace strsim
