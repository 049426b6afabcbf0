// CPU harness for polars-strsim_amd/csrc/strsim_lane_core.h: runs the exact per-lane arithmetic of
// the gfx950 lane-per-pair kernels on the host so tests/ can compare it with the oracle without a
// GPU.  Test infrastructure only.
#include <cstdint>
#include <cstring>
#include <algorithm>
#include "strsim_lane_core.h"
#include "strsim_lane_lut.h"
#include "strsim_lane_wide.h"
#include "strsim_lane_sym.h"
#include "lane_core_textbook.h"

using namespace strsim;

template <int M, int NP>
static double run_np(const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8], uint32_t lb)
{
    const uint32_t tmax = (la + 3u) & ~3u;
    return lane_pair_result<M, NP>(wa, la, wb, lb, tmax ? tmax : 4u);
}

// force_np: 0 = choose like the kernel does (from the windows' varying bits), else 5/6/7/8
template <int M>
static double run(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, int force_np, uint8_t fill)
{
    uint32_t wa[8], wb[8];
    // bytes past the length: what a neighbouring string could hold (the kernel loads a 32-byte window)
    std::memset(wa, fill, sizeof wa);
    std::memset(wb, fill, sizeof wb);
    std::memcpy(wa, a, la);
    std::memcpy(wb, b, lb);
    uint32_t any;
    const uint32_t vary = window_vary(wa, wb, any);
    int np = force_np ? force_np : planes_needed(vary);
    switch (np) {
    case 5: return run_np<M, 5>(wa, la, wb, lb);
    case 6: return run_np<M, 6>(wa, la, wb, lb);
    case 7: return run_np<M, 7>(wa, la, wb, lb);
    default: return run_np<M, 8>(wa, la, wb, lb);
    }
}

extern "C" double harness_lane_pair(int measure, const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb,
                                    int force_np, int fill)
{
    switch (measure) {
    case LEVENSHTEIN: return run<LEVENSHTEIN>(a, la, b, lb, force_np, (uint8_t)fill);
    case JARO: return run<JARO>(a, la, b, lb, force_np, (uint8_t)fill);
    case JARO_WINKLER: return run<JARO_WINKLER>(a, la, b, lb, force_np, (uint8_t)fill);
    case JACCARD: return run<JACCARD>(a, la, b, lb, force_np, (uint8_t)fill);
    default: return run<SORENSEN_DICE>(a, la, b, lb, force_np, (uint8_t)fill);
    }
}

// the full-rate form of the Levenshtein core (right-aligned pattern, copies of the vertical deltas at the lane's own last
// column): plain distance, text = a walked in tmax columns with lanes' ends expected from column tmin on
extern "C" uint32_t harness_lev_snap(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, uint32_t tmin, uint32_t tmax,
                                     int np, int fill)
{
    uint32_t wa[8], wb[8];
    std::memset(wa, fill, sizeof wa);
    std::memset(wb, fill, sizeof wb);
    std::memcpy(wa, a, la);
    std::memcpy(wb, b, lb);
    // the bit-fill form and the table form (strsim_lane_lut.h) must agree; 0xFFFFFFFF when they do not
    EqLut t{};
    if (np == 5) {
        uint32_t P[5];
        build_planes<5>(wb, P);
        lut_build<5>(t, P, 0xFFFFFFFFu);
        const uint32_t d = lev_myers32_snap<5>(wa, la, tmin, tmax, P, lb);
        return lev_myers32_lut<5>(t, wa, la, tmin, tmax, P, lb) == d ? d : 0xFFFFFFFFu;
    }
    uint32_t P[7];
    build_planes<7>(wb, P);
    lut_build<7>(t, P, 0xFFFFFFFFu);
    const uint32_t d = lev_myers32_snap<7>(wa, la, tmin, tmax, P, lb);
    return lev_myers32_lut<7>(t, wa, la, tmin, tmax, P, lb) == d ? d : 0xFFFFFFFFu;
}

// the one-loop cores (lane_cores32) with all three cores on, and each of them on its own: out[0..3] = dist, m, t, isect of
// the fused run; returns 0 when the single-core runs agree with it, else a code
extern "C" int harness_cores32(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, uint32_t tmin, uint32_t tmax,
                               int np, int fill, uint32_t *out)
{
    uint32_t wa[8], wb[8];
    std::memset(wa, fill, sizeof wa);
    std::memset(wb, fill, sizeof wb);
    std::memcpy(wa, a, la);
    std::memcpy(wb, b, lb);
    uint32_t d = 0, m = 0, t = 0, is = 0, d1 = 0, m1 = 0, t1 = 0, is1 = 0, x = 0;
    if (np == 5) {
        uint32_t P[5];
        build_planes<5>(wb, P);
        lane_cores32<5, true, true, true>(wa, la, tmin, tmax, lb, P, d, m, t, is);
        lane_cores32<5, true, false, false>(wa, la, tmin, tmax, lb, P, d1, x, x, x);
        lane_cores32<5, false, true, false>(wa, la, tmin, tmax, lb, P, x, m1, t1, x);
        lane_cores32<5, false, false, true>(wa, la, tmin, tmax, lb, P, x, x, x, is1);
        uint32_t mk = 0, tk = 0, ma = 0, ta_ = 0, da = 0, ia = 0; // (KEEP_EQ: the zip pass on the first pass's masks)
        lane_cores32<5, false, true, false, true>(wa, la, tmin, tmax, lb, P, x, mk, tk, x);
        lane_cores32<5, true, true, true, true>(wa, la, tmin, tmax, lb, P, da, ma, ta_, ia);
        if (mk != m || tk != t || ma != m || ta_ != t || da != d || ia != is) return 6;
    } else {
        uint32_t P[7];
        build_planes<7>(wb, P);
        lane_cores32<7, true, true, true>(wa, la, tmin, tmax, lb, P, d, m, t, is);
        lane_cores32<7, true, false, false>(wa, la, tmin, tmax, lb, P, d1, x, x, x);
        lane_cores32<7, false, true, false>(wa, la, tmin, tmax, lb, P, x, m1, t1, x);
        lane_cores32<7, false, false, true>(wa, la, tmin, tmax, lb, P, x, x, x, is1);
        uint32_t mk = 0, tk = 0;
        lane_cores32<7, false, true, false, true>(wa, la, tmin, tmax, lb, P, x, mk, tk, x);
        if (mk != m || tk != t) return 6;
    }
    out[0] = d; out[1] = m; out[2] = t; out[3] = is;
    if (d1 != d) return 1;
    if (m1 != m || t1 != t) return 2;
    if (is1 != is) return 3;
    // the same cores with table masks (strsim_lane_lut.h): fused and one at a time
    uint32_t d2 = 0, m2 = 0, t2 = 0, is2 = 0, d3 = 0, m3 = 0, t3 = 0, is3 = 0;
    EqLut tb{};
    if (np == 5) {
        uint32_t P[5];
        build_planes<5>(wb, P);
        lut_build<5>(tb, P, 0xFFFFFFFFu);
        lane_cores32_lut<5, true, true, true>(tb, wa, la, tmin, tmax, lb, P, d2, m2, t2, is2);
        lane_cores32_lut<5, true, false, false>(tb, wa, la, tmin, tmax, lb, P, d3, x, x, x);
        lane_cores32_lut<5, false, true, false>(tb, wa, la, tmin, tmax, lb, P, x, m3, t3, x);
        lane_cores32_lut<5, false, false, true>(tb, wa, la, tmin, tmax, lb, P, x, x, x, is3);
    } else {
        uint32_t P[7];
        build_planes<7>(wb, P);
        lut_build<7>(tb, P, 0xFFFFFFFFu);
        lane_cores32_lut<7, true, true, true>(tb, wa, la, tmin, tmax, lb, P, d2, m2, t2, is2);
        lane_cores32_lut<7, true, false, false>(tb, wa, la, tmin, tmax, lb, P, d3, x, x, x);
        lane_cores32_lut<7, false, true, false>(tb, wa, la, tmin, tmax, lb, P, x, m3, t3, x);
        lane_cores32_lut<7, false, false, true>(tb, wa, la, tmin, tmax, lb, P, x, x, x, is3);
    }
    if (d2 != d || m2 != m || t2 != t || is2 != is) return 4;
    if (d3 != d || m3 != m || t3 != t || is3 != is) return 5;
    return 0;
}

// plane build vs the definition, for any 32 bytes
extern "C" int harness_check_planes(const uint8_t *bytes32)
{
    uint32_t w[8];
    std::memcpy(w, bytes32, 32);
    uint32_t P[8];
    build_planes<8>(w, P);
    for (int k = 0; k < 8; ++k) {
        uint32_t ref = 0;
        for (int i = 0; i < 32; ++i) ref |= (uint32_t)((bytes32[i] >> k) & 1u) << i;
        if (ref != P[k]) return k + 1;
    }
    uint32_t R[8];
    build_planes_r1<8>(w, R);
    for (int k = 0; k < 8; ++k) if (R[k] != P[k]) return 300 + k;
    uint32_t P5[5];
    build_planes<5>(w, P5);
    for (int k = 0; k < 5; ++k) if (P5[k] != P[k]) return 100 + k;
    uint32_t P7[7];
    build_planes<7>(w, P7);
    for (int k = 0; k < 7; ++k) if (P7[k] != P[k]) return 200 + k;
    return 0;
}

// div3_exact (strsim_lane_core.h) against x / 3.0 over every sum Jaro's table epilogues can divide: returns the mismatches, *count = sums
extern "C" long harness_div3_mismatches(long *count)
{
    long bad = 0, n = 0;
    for (int la = 1; la <= 64; ++la)
        for (int lb = 1; lb <= 64; ++lb)
            for (int m = 1; m <= (la < lb ? la : lb); ++m)
                for (int h = 0; h <= m / 2; ++h) {
                    const double x = (double)m / (double)la + (double)m / (double)lb + (double)(m - h) / (double)m;
                    const double a = x / 3.0, b = div3_exact(x);
                    ++n;
                    bad += std::memcmp(&a, &b, 8) != 0;
                }
    *count = n;
    return bad;
}

// ---- wide (W-word) cores -------------------------------------------------------------------------
struct ArrTxt { const uint32_t *w; uint32_t operator()(uint32_t g) const { return w[g]; } };
// Jaro's string of matched characters overwrites the front of the text, as on the GPU (the text column in LDS)
// How Jaro's zip pass runs (jaro_wide: a wave-uniform choice on the GPU): 0 = over the positions of b, 1 = over the matched
// characters as far as the lane's own m (rounded up to 4) + g_zip_slack more (the wave's largest m is some other lane's),
// 2 = the kernel's rule applied to this lane alone.  Set by harness_set_zip_mode().
static int g_zip_mode = 2;
static uint32_t g_zip_slack = 0;
extern "C" void harness_set_zip_mode(int mode, uint32_t slack) { g_zip_mode = mode; g_zip_slack = slack & ~3u; }
struct ArrSa {
    uint8_t *bytes;
    uint32_t cap; // bytes of the text buffer SA lives in (the GPU: 128 per lane)
    void put(uint32_t k, uint32_t c, uint32_t) const { bytes[k] = (uint8_t)c; } // unconditional, as on the GPU
    uint32_t get(uint32_t k) const { return bytes[k]; }
    uint32_t get4(uint32_t k) const { uint32_t v; std::memcpy(&v, bytes + k, 4); return v; }
    uint32_t zip_over_matches(uint32_t m, uint32_t W, uint32_t nb4) const
    {
        uint32_t k4 = (m + 3u) & ~3u;
        if (g_zip_mode == 0) return 0u;
        if (g_zip_mode == 1) { k4 = std::min(k4 + g_zip_slack, cap); return k4 ? k4 : 4u; }
        return (k4 != 0u && k4 * (5u + 9u * W) < 28u * nb4) ? k4 : 0u;
    }
};

template <int M, int NP, int W, uint32_t TCAP = 32u * W>
static double run_wide_np(uint32_t *ta, uint32_t la, const uint32_t (&wp)[8 * W], uint32_t lb, uint32_t b0w)
{
    const uint32_t ng4 = (la + 3u) / 4u;
    // (nb4: any value from ceil(lb / 4) up to the window: the kernel passes the wave's maximum)
    const uint32_t nb4 = std::min<uint32_t>(8u * W, (lb + 3u) / 4u + ((la * 7u + lb) % 3u));
    const uint32_t a0w = ta[0];
    // (gfull: any value up to la / 4; the kernel passes the wave's minimum -- sweep it through a few)
    const uint32_t gfull = (la / 4u) * ((la ^ lb) & 3u) / 3u;
    return lane_wide_result<M, NP, W>(ArrTxt{ta}, la, gfull, ng4, wp, lb, nb4, a0w, b0w, ArrSa{reinterpret_cast<uint8_t *>(ta), TCAP});
}

template <int M, int W>
static double run_wide(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, int force_np, uint8_t fill)
{
    uint32_t ta[8 * W], wp[8 * W], wbn[8 * W];
    uint8_t buf[32 * W];
    std::memset(ta, fill, sizeof ta);
    std::memcpy(ta, a, la);
    std::memset(buf, fill, sizeof buf);
    std::memcpy(buf, b, lb);
    std::memcpy(wbn, buf, sizeof buf);            // natural window (starts at b)
    std::memcpy(wp, buf, sizeof buf);
    // varying bits over both windows
    uint32_t o = 0, n = 0xFFFFFFFFu;
    for (int d = 0; d < 8 * W; ++d) { o |= ta[d] | wp[d]; n &= ta[d] & wp[d]; }
    uint32_t o8 = o | (o >> 16); o8 |= o8 >> 8;
    uint32_t n8 = n & (n >> 16); n8 &= n8 >> 8;
    const int np = force_np ? force_np : planes_needed((o8 ^ n8) & 0xFFu);
    switch (np) {
    case 5: return run_wide_np<M, 5, W>(ta, la, wp, lb, wbn[0]);
    case 6: return run_wide_np<M, 6, W>(ta, la, wp, lb, wbn[0]);
    default: return run_wide_np<M, 7, W>(ta, la, wp, lb, wbn[0]);
    }
}

template <int W>
static double run_wide_m(int measure, const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, int force_np, uint8_t fill)
{
    switch (measure) {
    case LEVENSHTEIN: return run_wide<LEVENSHTEIN, W>(a, la, b, lb, force_np, fill);
    case JARO: return run_wide<JARO, W>(a, la, b, lb, force_np, fill);
    case JARO_WINKLER: return run_wide<JARO_WINKLER, W>(a, la, b, lb, force_np, fill);
    case JACCARD: return run_wide<JACCARD, W>(a, la, b, lb, force_np, fill);
    default: return run_wide<SORENSEN_DICE, W>(a, la, b, lb, force_np, fill);
    }
}

// strings of 1..32*W bytes each
extern "C" double harness_lane_pair_wide(int measure, int W, const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb,
                                         int force_np, int fill)
{
    if (W == 1) return run_wide_m<1>(measure, a, la, b, lb, force_np, (uint8_t)fill);
    if (W == 2) return run_wide_m<2>(measure, a, la, b, lb, force_np, (uint8_t)fill);
    if (W == 3) return run_wide_m<3>(measure, a, la, b, lb, force_np, (uint8_t)fill);
    return run_wide_m<4>(measure, a, la, b, lb, force_np, (uint8_t)fill);
}

// Masks as wide as the PATTERN (lb <= 32 * W), the text of any length up to 128 bytes: k_lane_wide's Jaro / Jaro-Winkler rounds
// whose a is the long side (and what the cores allow for every measure)
template <int M, int W>
static double run_wide_tp(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, int force_np, uint8_t fill)
{
    uint32_t ta[32], wp[8 * W];
    uint8_t buf[32 * W];
    std::memset(ta, fill, sizeof ta);
    std::memcpy(ta, a, la);
    std::memset(buf, fill, sizeof buf);
    std::memcpy(buf, b, lb);
    std::memcpy(wp, buf, sizeof buf);
    const uint32_t b0w = wp[0];
    uint32_t o = 0, n = 0xFFFFFFFFu;
    for (int d = 0; d < 32; ++d) { o |= ta[d]; n &= ta[d]; }
    for (int d = 0; d < 8 * W; ++d) { o |= wp[d]; n &= wp[d]; }
    uint32_t o8 = o | (o >> 16); o8 |= o8 >> 8;
    uint32_t n8 = n & (n >> 16); n8 &= n8 >> 8;
    const int np = force_np ? force_np : planes_needed((o8 ^ n8) & 0xFFu);
    switch (np) {
    case 5: return run_wide_np<M, 5, W, 128u>(ta, la, wp, lb, b0w);
    case 6: return run_wide_np<M, 6, W, 128u>(ta, la, wp, lb, b0w);
    default: return run_wide_np<M, 7, W, 128u>(ta, la, wp, lb, b0w);
    }
}

template <int W>
static double run_wide_tp_m(int measure, const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb, int force_np, uint8_t fill)
{
    switch (measure) {
    case LEVENSHTEIN: return run_wide_tp<LEVENSHTEIN, W>(a, la, b, lb, force_np, fill);
    case JARO: return run_wide_tp<JARO, W>(a, la, b, lb, force_np, fill);
    case JARO_WINKLER: return run_wide_tp<JARO_WINKLER, W>(a, la, b, lb, force_np, fill);
    case JACCARD: return run_wide_tp<JACCARD, W>(a, la, b, lb, force_np, fill);
    default: return run_wide_tp<SORENSEN_DICE, W>(a, la, b, lb, force_np, fill);
    }
}

// a: 1..128 bytes (the text), b: 1..32*W bytes (the pattern)
extern "C" double harness_lane_pair_wide_tp(int measure, int W, const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb,
                                            int force_np, int fill)
{
    if (W == 1) return run_wide_tp_m<1>(measure, a, la, b, lb, force_np, (uint8_t)fill);
    if (W == 2) return run_wide_tp_m<2>(measure, a, la, b, lb, force_np, (uint8_t)fill);
    if (W == 3) return run_wide_tp_m<3>(measure, a, la, b, lb, force_np, (uint8_t)fill);
    return run_wide_tp_m<4>(measure, a, la, b, lb, force_np, (uint8_t)fill);
}

// ---- symbol (non-ASCII) cores ----------------------------------------------------------------------
struct SymArr { const uint16_t *s; uint32_t operator()(uint32_t k) const { return s[k]; } };
struct EmitArr { uint16_t *s; void operator()(uint32_t k, uint32_t cp) const { if (k < 40) s[k] = (uint16_t)cp; } };

template <int M, int NP>
static double run_sym_np(const uint16_t *ta, uint32_t la, const uint16_t *pb, uint32_t lb)
{
    return lane_sym_result<M, NP>(SymArr{ta}, la, la, SymArr{pb}, lb);
}

// utf8_decode_lane on its own: the scalar values of a string of n <= 128 bytes (garbage behind it in the window) -> out[0 .. 131],
// returns their count; flags[0] = big, flags[1] = OR of the values, flags[2] = AND of the values (low 16 bits)
extern "C" uint32_t harness_utf8_decode(const uint8_t *a, uint32_t n, int fill, uint16_t *out, uint32_t *flags)
{
    uint32_t w[32];
    std::memset(w, fill, sizeof w);
    std::memcpy(w, a, n);
    struct Emit { uint16_t *s; void operator()(uint32_t k, uint32_t cp) const { if (k < 132) s[k] = (uint16_t)cp; } };
    bool big = false;
    uint32_t ov = 0, av = 0xFFFFFFFFu;
    const uint32_t cnt = utf8_decode_lane<32>(w, n, (n + 3) / 4, Emit{out}, big, ov, av);
    flags[0] = big ? 1u : 0u;
    flags[1] = ov;
    flags[2] = av & 0xFFFFu;
    return cnt;
}

// returns -1.0 when the pair is not eligible (more than 32 scalar values, or one outside the BMP, or an empty side)
extern "C" double harness_lane_pair_sym(int measure, const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb, int force_np)
{
    if (na > 128 || nb > 128 || na == 0 || nb == 0) return -1.0;
    uint32_t wa[32], wb[32];
    std::memset(wa, 0xAB, sizeof wa);
    std::memset(wb, 0xCD, sizeof wb);
    std::memcpy(wa, a, na);
    std::memcpy(wb, b, nb);
    uint16_t sa[40], sb[40];
    std::memset(sa, 0x5A, sizeof sa);
    std::memset(sb, 0x33, sizeof sb);
    bool big = false;
    uint32_t ov = 0, av = 0xFFFFFFFFu;
    const uint32_t la = utf8_decode_lane<32>(wa, na, (na + 3) / 4, EmitArr{sa}, big, ov, av);
    const uint32_t lb = utf8_decode_lane<32>(wb, nb, (nb + 3) / 4, EmitArr{sb}, big, ov, av);
    if (big || la > 32 || lb > 32) return -1.0;
    const int np = force_np ? force_np : planes_needed_sym((ov ^ av) & 0xFFFFu);
#define SYM_DISPATCH(M)                                                       \
    switch (np) {                                                             \
    case 8: return run_sym_np<M, 8>(sa, la, sb, lb);                          \
    case 11: return run_sym_np<M, 11>(sa, la, sb, lb);                        \
    default: return run_sym_np<M, 16>(sa, la, sb, lb);                        \
    }
    switch (measure) {
    case LEVENSHTEIN: SYM_DISPATCH(LEVENSHTEIN)
    case JARO: SYM_DISPATCH(JARO)
    case JARO_WINKLER: SYM_DISPATCH(JARO_WINKLER)
    case JACCARD: SYM_DISPATCH(JACCARD)
    default: SYM_DISPATCH(SORENSEN_DICE)
    }
}
