// strsim_lane_sym.h -- per-lane cores for short NON-ASCII strings: up to 32 Unicode scalar values per string,
// every one of them in the Basic Multilingual Plane (<= 0xFFFF), i.e. names and words in any script.
//
// The reference compares `char`s, not bytes (strsim.rs:133,189,297), so each lane first decodes its two UTF-8
// strings (<= 128 bytes, in registers) into 16-bit symbols stored in an LDS column per lane; from there the cores
// are the bit-sliced ones of strsim_lane_core.h with up to 16 bit-planes per pattern instead of 7:
//     Eq(sym) = valid & AND_k ~(P_k ^ bit_k(sym)),   k over the bits that vary inside the pair.
// Host/device portable (the CPU harness in tests/ runs the same code).
#pragma once
#include "strsim_lane_core.h"

namespace strsim {

// ---------------------------------------------------------------------------------------------
// `str::chars()` for one lane: walk the first len8 bytes of the 32*NW-byte window w[] and hand the scalar values to `emit`.
// nd4 = number of dwords to walk (lane-uniform bound >= ceil(len8 / 4)).  Valid UTF-8 only.  Returns the number of scalar
// values; `big` is set when one exceeds 0xFFFF, `ovar`/`avar` accumulate OR / AND of the values (for the plane-count choice).
//
// Branch-free, four byte positions per trip (the per-byte state machine this replaces cost ~5 divergent branches per byte):
// with the next two bytes of every position lined up beside it (two v_alignbyte), the value a position WOULD start is formed
// for all four positions at once, byte lane by byte lane -- its low byte and its high byte separately, which is also how the
// plane build wants the symbols:
//     0xxxxxxx                      lo = b0                               hi = 0
//     110xxxxx 10xxxxxx             lo = (b0 & 3) << 6 | (b1 & 0x3F)       hi = (b0 >> 2) & 7
//     1110xxxx 10xxxxxx 10xxxxxx    lo = (b1 & 3) << 6 | (b2 & 0x3F)       hi = (b0 & 15) << 4 | (b1 >> 2) & 15
// (a lead byte 1111xxxx starts a value beyond 0xFFFF: `big`).  EVERY position is then emitted at the running count of lead
// bytes in front of it: a lead byte lands at its value's index, a continuation byte (or a byte behind the string) writes a
// don't-care at the index of the value that follows and is overwritten by it -- emit(k, v) must accept any k (the callers
// drop what is beyond their capacity) and the same k more than once, and the entry at index `count` may be left holding a
// don't-care (so a caller with room for N values must drop index N, not wrap it onto index 0).
// ---------------------------------------------------------------------------------------------
STRSIM_HD uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t sh) // ({hi, lo} >> 8 * sh), sh = 1 .. 3
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * sh));
#endif
}
STRSIM_HD uint32_t byte_masks(uint32_t bits01) { return (bits01 << 8) - bits01; } // bytes 0 / 1 -> 0x00 / 0xFF
STRSIM_HD uint32_t fold_or8(uint32_t x) { x |= x >> 16; x |= x >> 8; return x & 0xFFu; }
STRSIM_HD uint32_t fold_and8(uint32_t x) { x &= x >> 16; x &= x >> 8; return x & 0xFFu; }

template <int NDW, class Emit>
STRSIM_HD uint32_t utf8_decode_lane(const uint32_t (&w)[NDW], uint32_t len8, uint32_t nd4, const Emit &emit, bool &big,
                                    uint32_t &ovar, uint32_t &avar)
{
    uint32_t k = 0;                               // lead bytes (= scalar values) so far
    uint32_t o_lo = 0u, o_hi = 0u, a_lo = 0xFFFFFFFFu, a_hi = 0xFFFFFFFFu, four = 0u;
#pragma unroll
    for (int d = 0; d < NDW; ++d) {
        if ((uint32_t)d >= nd4) break;
        const uint32_t cur = w[d], nxt = d + 1 < NDW ? w[d + 1 < NDW ? d + 1 : d] : 0u;
        const uint32_t w1 = align_bytes(nxt, cur, 1u), w2 = align_bytes(nxt, cur, 2u);
        const uint32_t b7 = (cur >> 7) & 0x01010101u, b6 = (cur >> 6) & 0x01010101u, b5 = (cur >> 5) & 0x01010101u;
        const uint32_t m7 = byte_masks(b7), m5 = byte_masks(b5);
        // positions of this dword inside the string
        const int32_t rem = (int32_t)len8 - 4 * d;
        const uint32_t cut = rem >= 4 ? 0u : (rem <= 0 ? 4u : (uint32_t)(4 - rem));
        const uint32_t vb = cut >= 4u ? 0u : (0x01010101u >> (8u * cut));
        const uint32_t leadb = bitop3<0x8A>(b7, b6, vb); // (~b7 | b6) & vb: not a continuation byte, inside the string
        const uint32_t ml = byte_masks(leadb);
        const uint32_t lo2 = ((cur & 0x03030303u) << 6) | (w1 & 0x3F3F3F3Fu);
        const uint32_t lo3 = ((w1 & 0x03030303u) << 6) | (w2 & 0x3F3F3F3Fu);
        const uint32_t hi2 = (cur >> 2) & 0x07070707u;
        const uint32_t hi3 = ((cur & 0x0F0F0F0Fu) << 4) | ((w1 >> 2) & 0x0F0F0F0Fu);
        const uint32_t lo = bitop3<0xCA>(m7, bitop3<0xCA>(m5, lo3, lo2), cur);
        const uint32_t hi = m7 & bitop3<0xCA>(m5, hi3, hi2);
        four |= cur & (cur << 1) & (cur << 2) & (cur << 3) & (leadb << 7); // lead byte 1111xxxx
        o_lo = bitop3<0xF8>(o_lo, lo, ml); o_hi = bitop3<0xF8>(o_hi, hi, ml); // o | (v & ml)
        a_lo = bitop3<0xD0>(a_lo, lo, ml); a_hi = bitop3<0xD0>(a_hi, hi, ml); // a & (v | ~ml)
        const uint32_t s01 = perm_b32(hi, lo, 0x05010400u); // [lo.b0, hi.b0, lo.b1, hi.b1]: positions 0 and 1 as 16-bit values
        const uint32_t s23 = perm_b32(hi, lo, 0x07030602u);
        emit(k, s01 & 0xFFFFu); k += leadb & 1u;
        emit(k, s01 >> 16);     k += (leadb >> 8) & 1u;
        emit(k, s23 & 0xFFFFu); k += (leadb >> 16) & 1u;
        emit(k, s23 >> 16);     k += leadb >> 24;
    }
    big = big || four != 0u;
    ovar |= fold_or8(o_lo) | (fold_or8(o_hi) << 8);
    avar &= (fold_and8(a_lo) | (fold_and8(a_hi) << 8)) | 0xFFFF0000u;
    if (k != 0u) avar &= 0x0000FFFFu; // (as the per-value AND of 16-bit values left it)
    return k;
}

// planes 0..NP-1 (NP <= 16) of up to 32 16-bit symbols; sym(k) returns symbol k (k < 32; don't-care past the length)
template <int NP, class Sym>
STRSIM_HD void build_planes_sym(const Sym &sym, uint32_t (&P)[NP])
{
    uint32_t lo[8], hi[8]; // low bytes / high bytes of the 32 symbols, packed 4 per dword
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const uint32_t s0 = sym(4 * g), s1 = sym(4 * g + 1), s2 = sym(4 * g + 2), s3 = sym(4 * g + 3);
        lo[g] = (s0 & 0xFFu) | ((s1 & 0xFFu) << 8) | ((s2 & 0xFFu) << 16) | ((s3 & 0xFFu) << 24);
        hi[g] = ((s0 >> 8) & 0xFFu) | (((s1 >> 8) & 0xFFu) << 8) | (((s2 >> 8) & 0xFFu) << 16) | (((s3 >> 8) & 0xFFu) << 24);
    }
    constexpr int NLO = NP < 8 ? NP : 8;
    uint32_t plo[NLO];
    build_planes<NLO>(lo, plo);
#pragma unroll
    for (int k = 0; k < NLO; ++k) P[k] = plo[k];
    if (NP > 8) {
        constexpr int NHI = NP > 8 ? NP - 8 : 1;
        uint32_t phi[NHI];
        build_planes<NHI>(hi, phi);
#pragma unroll
        for (int k = 0; k < NHI; ++k) P[(8 + k) < NP ? (8 + k) : 0] = phi[k];
    }
}

template <int NP>
STRSIM_HD uint32_t eq_sym(const uint32_t (&P)[NP], uint32_t valid, uint32_t s)
{
    uint32_t acc = valid;
#pragma unroll
    for (int k = 0; k < NP; ++k) acc = bitop3<0x90>(acc, P[k], bit_fill(s, k));
    return acc;
}

// number of planes needed: position of the highest varying bit + 1, rounded to the instantiated sizes
STRSIM_HD int planes_needed_sym(uint32_t vary) { return (vary >> 11) ? 16 : ((vary >> 8) ? 11 : 8); }

// ---- the three cores; text symbols through txt(j), steps = lane-uniform loop bound >= lt -----------------
template <int NP, class Txt>
STRSIM_HD uint32_t lev_sym(const Txt &txt, uint32_t lt, uint32_t steps, const uint32_t (&P)[NP], uint32_t lp)
{
    const uint32_t s = 32u - lp;
    const uint32_t valid = 0xFFFFFFFFu << s;
    uint32_t Pv = valid, Mv = ~Pv, hp = 0u, hn = 0u;
    const uint32_t steps4 = (steps + 3u) & ~3u; // whole groups of four columns: four symbol reads in flight at a time
    for (uint32_t j = 0; j < steps4; j += 4u) {
        uint32_t sy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sy[q] = txt(j + (uint32_t)q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t Eq = eq_sym<NP>(P, valid, sy[q]);
            const uint32_t D0 = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq | Mv);
            const uint32_t HP = bitop3<0xF1>(Mv, D0, Pv);
            const uint32_t HN = Pv & D0;
            hp = (hp << 1) | (HP >> 31);
            hn = (hn << 1) | (HN >> 31);
            const uint32_t X = (HP << 1) | 1u;
            Pv = bitop3<0xF1>(HN << 1, D0, X);
            Mv = D0 & X;
        }
    }
    const uint32_t cols = low_ones(lt) << (steps4 - lt);
    return lp + popc32(hp & cols) - popc32(hn & cols);
}

template <int NP, class Txt>
STRSIM_HD void jaro_sym(const Txt &txt, uint32_t la, uint32_t steps, uint32_t lb, const uint32_t (&P)[NP], uint32_t &m_out,
                        uint32_t &t_out)
{
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u;
    const uint32_t lbmask = low_ones(lb);
    uint32_t himask = low_ones((bound + 1u) < lb ? (bound + 1u) : lb), lomask = 0u, fb = 0u, fa = 0u;
    const uint32_t steps4 = (steps + 3u) & ~3u; // steps <= 32, so steps4 <= 32 as well
    for (uint32_t i0 = 0; i0 < steps4; i0 += 4u) {
        uint32_t sy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sy[q] = txt(i0 + (uint32_t)q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = i0 + (uint32_t)q;
            const uint32_t live = i < la ? 0xFFFFFFFFu : 0u;
            const uint32_t Eq = eq_sym<NP>(P, lbmask, sy[q]);
            const uint32_t cand = bitop3<0x20>(Eq & himask, lomask | fb, live);
            const uint32_t bit = cand & (0u - cand);
            fb |= bit;
            fa |= (bit ? 1u : 0u) << i;
            himask = ((himask << 1) | 1u) & lbmask;
            if (i >= bound) lomask = (lomask << 1) | 1u;
        }
    }
    uint32_t unequal = 0u, rest = fb; // (the unequal pairs collected as bits of b, counted once: lane_cores32)
    for (uint32_t i0 = 0; i0 < steps4; i0 += 4u) {
        uint32_t sy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sy[q] = txt(i0 + (uint32_t)q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = i0 + (uint32_t)q;
            const uint32_t on = 0u - ((fa >> i) & 1u);
            const uint32_t jbit = rest & (0u - rest) & on;
            rest ^= jbit;
            const uint32_t Eq = eq_sym<NP>(P, lbmask, sy[q]);
            unequal = bitop3<0xF4>(unequal, jbit, Eq); // unequal | (jbit & ~Eq)
        }
    }
    m_out = popc32(fb);
    t_out = popc32(unequal);
}

template <int NP, class Txt>
STRSIM_HD uint32_t isect_sym(const Txt &txt, uint32_t la, uint32_t steps, uint32_t lb, const uint32_t (&P)[NP])
{
    const uint32_t lbmask = low_ones(lb);
    uint32_t used = 0u;
    const uint32_t steps4 = (steps + 3u) & ~3u;
    for (uint32_t i0 = 0; i0 < steps4; i0 += 4u) {
        uint32_t sy[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sy[q] = txt(i0 + (uint32_t)q);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t live = (i0 + (uint32_t)q) < la ? 0xFFFFFFFFu : 0u;
            const uint32_t cand = bitop3<0x20>(eq_sym<NP>(P, lbmask, sy[q]), used, live);
            used |= cand & (0u - cand);
        }
    }
    return popc32(used);
}

// One lane's result: text symbols txt(0 .. la), pattern symbols pat(0 .. lb), 1 <= la, lb <= 32.
// For Jaro-Winkler the common prefix is counted on the symbols (strsim.rs:261-266).
template <int MEASURE, int NP, class Txt, class Pat>
STRSIM_HD double lane_sym_result(const Txt &txt, uint32_t la, uint32_t steps, const Pat &pat, uint32_t lb)
{
    uint32_t P[NP];
    build_planes_sym<NP>(pat, P);
    if (MEASURE == LEVENSHTEIN) {
        const uint32_t s = 32u - lb;
#pragma unroll
        for (int k = 0; k < NP; ++k) P[k] <<= s;
        return epilogue_levenshtein(lev_sym<NP>(txt, la, steps, P, lb), la, lb);
    } else if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        uint32_t m, t;
        jaro_sym<NP>(txt, la, steps, lb, P, m, t);
        const double j = epilogue_jaro(m, t, la, lb);
        if (MEASURE == JARO) return j;
        uint32_t p = 0;
        const uint32_t lim = la < lb ? (la < 4u ? la : 4u) : (lb < 4u ? lb : 4u);
        while (p < lim && txt(p) == pat(p)) ++p;
        return epilogue_jaro_winkler(j, p);
    } else {
        const uint32_t isect = isect_sym<NP>(txt, la, steps, lb, P);
        return MEASURE == JACCARD ? epilogue_jaccard(isect, la, lb) : epilogue_sorensen_dice(isect, la, lb);
    }
}

} // namespace strsim
