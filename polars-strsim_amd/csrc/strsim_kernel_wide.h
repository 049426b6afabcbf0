// strsim_kernel_wide.h -- k_lane_wide<M>: one pair per lane, 33..128 ASCII bytes (masks of 2..4 words: strsim_lane_wide.h), and the span / list geometry
// it shares with k_lane_utf8.
// Included by strsim_kernels.hip inside namespace strsim, after the kernels in front of it ([r5] split out of strsim_kernels.hip
// along its seams, VERDICT r4 item 8: no behaviour change -- the translation unit's ISA is byte-identical before and after).
// Reference semantics: /root/reference/src/expressions/strsim.rs:125-345 (the cores cite their lines).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_lane_wide: one pair per lane for the rows k_lane_stage left behind whose longer string is 33..128 ASCII
// bytes: masks of W = 2..4 words, as wide as the PATTERN (strsim_lane_wide.h) -- the LONGER string; the columns walk the shorter
// one ([r5] for Jaro / Jaro-Winkler too: every measure is symmetric, strsim_lane_core.h; before, a Jaro row with a 100-byte a
// ran 100 columns).  A workgroup takes a super of up to 256 mask words (16 384 rows), sorts its flagged
// rows into an LDS list by (W, columns to run) and runs them 64 at a time; finished rows are cleared from the mask,
// the rest (non-ASCII, longer, empty side) stay for k_lane_utf8 / k_wave_pairs.  With nothing flagged a super costs
// one coalesced read and one barrier.
// ------------------------------------------------------------------------------------------------
constexpr int WIDE_BLOCK = 256;
constexpr int WIDE_WAVES = WIDE_BLOCK / 64;
#ifndef STRSIM_WIDE_SPAN
#define STRSIM_WIDE_SPAN 64 // cfg3: 6.20 -> 5.89 ms against 32 (fuller rounds, longer hand-out queues); 128 does not fit LDS three times
#endif
constexpr int WIDE_SPAN = STRSIM_WIDE_SPAN;  // mask words (64-row chunks) per span of k_lane_wide
constexpr int WIDE_ROWS = WIDE_SPAN * 64;    // 4096 rows
#ifndef STRSIM_WIDE_LIST
// rows on k_lane_wide's sorted list (at least WIDE_ROWS).  [r5] 4 096, where rounds 3-4 took 8 192 "because the LDS has it": a super
// whose candidates do not fit is listed two spans (8 192 rows) at a time instead of all four, and that is FASTER wherever it happens --
// cfg3 (5 860 candidates per super): k_lane_wide 3.76-3.94 -> 3.66-3.70 ms for Jaro-Winkler, 3.83 -> 3.43 for Levenshtein, 3.83 -> 3.21
// for Jaccard on the same frame (profiles/r5_rounds_ab.txt) -- the rows a workgroup's rounds fetch lie within 8 192 consecutive rows
// instead of 16 384 (two rows in most lines it touches), its collect pass is half as long a stretch without cores, and the rounds
// lose little: 2 930 rows over 384 keys are still alike.  Lists of one span (4 096 rows: STRSIM_WIDE_SPAN=32) lose 2-3 % again; frames
// whose supers fit (sparse wide rows) and frames of wide rows only (one span per list either way) do not change.  8 KB of LDS less.
#define STRSIM_WIDE_LIST 4096
#endif
constexpr int WIDE_LIST = STRSIM_WIDE_LIST;
static_assert(WIDE_LIST >= WIDE_ROWS && WIDE_BLOCK * 64 <= 65536, "one span always fits the list; list entries are 16-bit row indices of a super");
constexpr int U8_SPAN = 32;                  // ... of k_lane_utf8 (its symbol columns take the LDS: a larger span costs a workgroup per CU)
constexpr int U8_ROWS = U8_SPAN * 64;        // 2048 rows
constexpr int WIDE_MAXW = 4;
#ifndef STRSIM_WIDE_ONE_WORD
// 1: k_lane_wide also takes flagged ASCII rows whose LONGER string is <= 32 bytes (masks of one word) -- the rows k_lane_stage's LONG
// geometry leaves behind under STRSIM_STAGE_LONG_TEXT_MAX, and short rows a staging overflow left in the mask.  Round 6's experiment.
#define STRSIM_WIDE_ONE_WORD 0
#endif

// NDW dwords of vals[start, start + 4*NDW); bytes outside [0, total) read as 0.  start may be negative.
template <int NDW>
__device__ __forceinline__ void load_window_any(const uint8_t *__restrict__ vals, int64_t start, uint32_t total,
                                                uint32_t (&w)[NDW])
{
    if (start >= 0 && start + 4 * NDW <= (int64_t)total) {
        const uint8_t *p = vals + start;
        STRSIM_CHECK_SPAN(K_WIDE, 40, start, p, 4 * NDW, vals, total); // (lab; also k_lane_utf8's windows)
#pragma unroll
        for (int q = 0; q < NDW / 4; ++q) {
            const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned *>(p + 16 * q);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll 1
        for (int d = 0; d < NDW; ++d) {
            uint32_t v = 0u;
            for (int k = 0; k < 4; ++k) {
                const int64_t idx = start + 4 * d + k;
                if (idx >= 0 && idx < (int64_t)total) v |= (uint32_t)vals[idx] << (8 * k);
            }
            // static register index: select by compare chain
#pragma unroll
            for (int e = 0; e < NDW; ++e)
                if (e == d) w[e] = v;
        }
    }
}

// A lane's text in LDS: 128 bytes of its own ("row" = LDS byte address, a multiple of 128), byte k at row + (k ^ swz) with
// swz = (lane & 31) << 2 -- dword g of all 64 lanes lies in 32 different banks (two lanes per bank: the LDS's natural rate), and
// the address of a byte or a dword is ONE v_xad_u32 ((k ^ swz) + row).  [r4] Columns of dwords, [dword][lane], had the same
// banks and a three-instruction byte address; Jaro's string of matched characters is written once per column of a and read
// once per position of b (33..128-byte frame: Jaro / Jaro-Winkler +4 %).
__device__ __forceinline__ uint32_t xad(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t xad_s(uint32_t a, uint32_t b, uint32_t c) // (a: wave-uniform, from a scalar register)
{
    uint32_t r;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "s"(a), "v"(b), "v"(c));
    return r;
}
struct LdsTxt {
    uint32_t row, swz;
    // g: the text dword, the same in every lane (loop counters bounded by wave maxima) -- xad_s takes it from a scalar register
    __device__ __forceinline__ uint32_t at(uint32_t g) const
    {
        STRSIM_CHECK_INDEX(K_WIDE, 60, ~0ull, g, 32); // (lab: a dword of the lane's own 128 bytes)
        return xad_s(wave_uniform(g << 2), swz, row);
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t g) const
    {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)at(g));
    }
    __device__ __forceinline__ void put(uint32_t g, uint32_t v) const
    {
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)at(g)) = v;
    }
};
// Jaro's string of matched characters (strsim_lane_wide.h) overwrites the front of the lane's text
#ifndef STRSIM_WIDE_ZIP_MATCHES
#define STRSIM_WIDE_ZIP_MATCHES 0 // 1: Jaro's zip pass may walk the matched characters instead of b (jaro_wide): measured, off -- see DESIGN 3.3 [r5]
#endif
struct LdsSa {
    uint32_t row, swz;
    __device__ __forceinline__ uint32_t at(uint32_t k) const
    {
        STRSIM_CHECK_INDEX(K_WIDE, 61, ~0ull, k, 128); // (lab: a byte of the lane's own 128)
        return xad(k, swz, row);
    }
    // Unconditional: a character that found no partner is overwritten by the next one that does (k does not move), and what
    // is left behind the last match is never read.
    __device__ __forceinline__ void put(uint32_t k, uint32_t c, uint32_t) const
    {
        *reinterpret_cast<__attribute__((address_space(3))) uint8_t *>((uintptr_t)at(k)) = (uint8_t)c;
    }
    __device__ __forceinline__ uint32_t get(uint32_t k) const
    {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uintptr_t)at(k));
    }
    // positions k .. k + 3, k a multiple of 4 (the swizzle permutes whole dwords)
    __device__ __forceinline__ uint32_t get4(uint32_t k) const
    {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)at(k));
    }
    // Jaro's zip pass over the matched characters (5 + 9 W instructions per match, as far as the wave's largest m) or over
    // the positions of b (7 per position, nb4 dwords)?  -> 0, or the number of matches to walk.  Uniform.
    __device__ __forceinline__ uint32_t zip_over_matches(uint32_t m, uint32_t W, uint32_t nb4) const
    {
        uint32_t mm = m; // (m <= 128)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mm, d);
            mm = mm > o ? mm : o;
        }
        mm = uniform(mm);
        const uint32_t k4 = (mm + 3u) & ~3u;
        return (STRSIM_WIDE_ZIP_MATCHES && k4 != 0u && k4 * (5u + 9u * W) < 28u * nb4) ? k4 : 0u;
    }
};

// max over the wave of v (v <= 255), uniform
__device__ __forceinline__ uint32_t wave_max_u8(uint32_t v)
{
    uint32_t m = 0;
#pragma unroll
    for (int b = 7; b >= 0; --b)
        if (__ballot(v >= (m | (1u << b))) != 0ull) m |= 1u << b;
    return m;
}

#if defined(STRSIM_LAB) && defined(STRSIM_WIDE_STAMPS)
// lab build only (make EXTRA="-DSTRSIM_LAB -DSTRSIM_WIDE_STAMPS"): per-wave cycle sums of the phases of k_lane_wide, read back with
// strsim_debug_wide_stamps() (bench_support/wide_stamps.py).  [0] mask + collect (keys, scan, list) [1] a round's rows and
// offsets [2] windows -> registers / LDS, the tests [3] the cores [4] result + mask bit [5] the barrier behind the list
// [6] rounds [10] all [11] all (100 MHz)
__device__ unsigned long long g_wide_stamps[16384][16];
#define WIDE_STAMP(cat) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                             __builtin_amdgcn_sched_barrier(0); wst_acc[cat] += t_ - wst_last; wst_last = t_; } while (0)
#define WIDE_STAMP_PARAMS , unsigned long long (&wst_acc)[8], unsigned long long &wst_last
#define WIDE_STAMP_ARGS , wst_acc, wst_last
#else
#define WIDE_STAMP(cat) do { } while (0)
#define WIDE_STAMP_PARAMS
#define WIDE_STAMP_ARGS
#endif

// ---- a round's windows, fetched by the wave TOGETHER ------------------------------------------------------------------
// [r4] Each lane used to fetch its own row's windows: 16-byte loads from 64 unrelated places per instruction (the address unit
// works through them one cache line at a time; the 8 pieces of a 128-byte window came in 8 instructions, so lines were fetched
// from L2 again and again: 2.9-5.8 x the rows' bytes), and a wave sat 28 % of its time behind them.  Now LPR
// consecutive lanes fetch consecutive 16-byte pieces of ONE row (LPR = 2, 4 or 8 pieces per row; 64 / LPR rows per instruction,
// LPR instructions per 64 rows), the pieces go through the wave's LDS rows (128 bytes per lane, the text's home anyway), and
// every lane picks up its own row there.
typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));
template <int NCH> struct CoopGeom {
    static constexpr int LPR = NCH <= 2 ? 2 : (NCH <= 4 ? 4 : 8); // lanes per row
    static constexpr int RPI = 64 / LPR;                           // rows per instruction
    static constexpr int ITER = LPR;                               // instructions per 64 rows
};
// piece c (0 .. NCH-1; lanes with c >= NCH idle) of row it * RPI + lane / LPR, it = 0 .. ITER-1; `start`: the owner lane's byte
// offset of its window.  The caller has checked that all NCH pieces of every row lie inside the column (a round that touches
// the column's last bytes takes the lane-by-lane path, wide_text / load_window_any).
// (`vals` is the OWNER's column -- the symmetric measures take their text from either -- so its address travels with `start`)
#if STRSIM_BOUNDS_ON
// lab: the byte ranges the two columns may be read in (include/strsim_amd.h: the 16-byte chunks their strings touch)
struct ColumnRanges { unsigned long long lo[2], hi[2]; };
#define STRSIM_COOP_RANGES , const ColumnRanges &rg
#define STRSIM_COOP_RANGES_ARG , rg
#else
#define STRSIM_COOP_RANGES
#define STRSIM_COOP_RANGES_ARG
#endif
template <int NCH>
__device__ __forceinline__ void coop_fetch(const uint8_t *vals, uint32_t start, uint32_t lane, uint4 (&v)[CoopGeom<NCH>::ITER] STRSIM_COOP_RANGES)
{
    using G = CoopGeom<NCH>;
    const uint32_t c = lane & (uint32_t)(G::LPR - 1), rsub = lane / (uint32_t)G::LPR;
    const uint64_t mine = (uint64_t)reinterpret_cast<uintptr_t>(vals) + start;
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        const uint32_t r = (uint32_t)(it * G::RPI) + rsub;
        const uint64_t p = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(mine >> 32), (int)r) << 32) | (uint32_t)__shfl((int)(uint32_t)mine, (int)r);
#if STRSIM_BOUNDS_ON
        {   // every lane that is about to load, whatever row its address came from: inside one of the two columns
            const unsigned long long q = p + 16u * (c < (uint32_t)NCH ? c : 0u);
            if (!((q >= rg.lo[0] && q + 16u <= rg.hi[0]) || (q >= rg.lo[1] && q + 16u <= rg.hi[1])))
                bounds::hit(bounds::K_WIDE, 10, r, q, rg.lo[0], rg.hi[1]);
        }
#endif
        const u32x4_unaligned t = *reinterpret_cast<const u32x4_unaligned *>(reinterpret_cast<const uint8_t *>((uintptr_t)p) + 16u * (c < (uint32_t)NCH ? c : 0u));
        v[it] = make_uint4(t.x, t.y, t.z, t.w);
    }
}
// the pieces -> the wave's LDS rows, 16-byte groups swizzled by the row (group c of row r at r * 128 + ((c ^ r) & 7) * 16), dwords in
// order: the PATTERN's layout -- its owner reads it back into registers with wide_pattern_regs
template <int NCH>
__device__ __forceinline__ void coop_store_groups(uint32_t base, uint32_t lane, const uint4 (&v)[CoopGeom<NCH>::ITER])
{
    using G = CoopGeom<NCH>;
    const uint32_t c = lane & (uint32_t)(G::LPR - 1), rsub = lane / (uint32_t)G::LPR;
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        const uint32_t r = (uint32_t)(it * G::RPI) + rsub;
        if (c < (uint32_t)NCH)
        {
            u32x4_lds q;
            q.x = v[it].x; q.y = v[it].y; q.z = v[it].z; q.w = v[it].w;
            STRSIM_CHECK_RANGE(K_WIDE, 20, r, r * 128u + (((c ^ r) & 7u) << 4) + 16u, 0, 64u * 128u); // (lab: the wave's 64 rows of 128 bytes)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4_lds *>((uintptr_t)(base + r * 128u + (((c ^ r) & 7u) << 4))) = q;
        }
    }
}
template <int W>
__device__ __forceinline__ void wide_pattern_regs(uint32_t base, uint32_t lane, uint32_t (&wp)[8 * W])
{
#pragma unroll
    for (int c = 0; c < 2 * W; ++c) {
        const u32x4_lds q = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_lds *>((uintptr_t)(base + lane * 128u + ((((uint32_t)c ^ lane) & 7u) << 4)));
        wp[4 * c] = q.x; wp[4 * c + 1] = q.y; wp[4 * c + 2] = q.z; wp[4 * c + 3] = q.w;
    }
}
// the pieces -> the TEXT's layout (LdsTxt: dword g of row r at r * 128 + ((g ^ r) & 31) * 4)
template <int NCH>
__device__ __forceinline__ void coop_store_text(uint32_t base, uint32_t lane, const uint4 (&v)[CoopGeom<NCH>::ITER])
{
    using G = CoopGeom<NCH>;
    const uint32_t c = lane & (uint32_t)(G::LPR - 1), rsub = lane / (uint32_t)G::LPR;
#pragma unroll
    for (int it = 0; it < G::ITER; ++it) {
        const uint32_t r = (uint32_t)(it * G::RPI) + rsub;
        if (c < (uint32_t)NCH) {
            const uint32_t row = base + r * 128u, swz = (r & 31u) << 2;
            const uint32_t e[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                STRSIM_CHECK_RANGE(K_WIDE, 21, r, r * 128u + ((16u * c + 4u * (uint32_t)k) ^ swz) + 4u, 0, 64u * 128u);
                *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)(row + ((16u * c + 4u * (uint32_t)k) ^ swz))) = e[k];
            }
        }
    }
}
// a lane's own text row: OR / AND of its first 32 * WT bytes (whole 16-byte groups: the swizzle only permutes inside them)
template <int WT>
__device__ __forceinline__ void wide_text_or_and(const LdsTxt &txt, uint32_t &o, uint32_t &n)
{
    o = 0u; n = 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < 2 * WT; ++c) {
        const u32x4_lds q = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_lds *>((uintptr_t)(txt.row + (((uint32_t)c << 4) ^ (txt.swz & 0x70u))));
        o |= q.x | q.y | q.z | q.w;
        n &= q.x & q.y & q.z & q.w;
    }
}

// The lane-by-lane path (a round with a window that reaches past its column's last byte, a literal side): WT x 32 bytes from each
// lane's a0 on into its LDS row, byte by byte where the column ends.  o / n: OR / AND of the dwords (lanes without a row: 0),
// a0w: the first dword.
template <int WT>
__device__ __forceinline__ void wide_text(const uint8_t *__restrict__ valA, uint32_t totalA, bool has, uint32_t a0, const LdsTxt &txt,
                                          uint32_t &o, uint32_t &n, uint32_t &a0w)
{
    uint32_t ta[8 * WT];
#pragma unroll
    for (int d = 0; d < 8 * WT; ++d) ta[d] = 0u;
    if (has) load_window_any<8 * WT>(valA, (int64_t)a0, totalA, ta);
    o = ta[0]; n = ta[0];
#pragma unroll
    for (int d = 1; d < 8 * WT; ++d) { o |= ta[d]; n &= ta[d]; }
#pragma unroll
    for (int d = 0; d < 8 * WT; ++d) txt.put((uint32_t)d, ta[d]);
    a0w = ta[0];
}

#ifndef STRSIM_WIDE_PREFETCH
// 1: the windows of the NEXT two-word round are fetched into registers before the cores of the current two-word round run (rows two
// rounds ahead, windows one): round 6's experiment (LAB_NOTES 9.0 "~4 % of cfg3").  See profiles/r6_rounds_ab.txt.
#define STRSIM_WIDE_PREFETCH 0
#endif
// a two-word round's windows in flight: 4 x 16-byte pieces of pattern per row (coop_fetch<4>), 2 or 4 of text
struct WidePref {
    bool valid;
    uint4 vp[4], vt[4];
};
__device__ __forceinline__ void wide_prefetch2(const uint8_t *vT, const uint8_t *vP, uint32_t tstart, uint32_t pstart, uint32_t wtw, WidePref &pf
                                               STRSIM_COOP_RANGES)
{
    uint32_t lane = lane_id();
    asm volatile("" : "+v"(lane));
    coop_fetch<4>(vP, pstart, lane, pf.vp STRSIM_COOP_RANGES_ARG);
    if (STRSIM_WIDE_PREFETCH == 2) { // (the pattern alone: 16 registers instead of 32; the text is fetched when its round starts)
    } else if (wtw <= 1u) {
        uint4 t[2];
        coop_fetch<2>(vT, tstart, lane, t STRSIM_COOP_RANGES_ARG);
        pf.vt[0] = t[0]; pf.vt[1] = t[1];
    } else {
        coop_fetch<4>(vT, tstart, lane, pf.vt STRSIM_COOP_RANGES_ARG);
    }
    pf.valid = true;
}

// W: words of the masks (by the round's longest PATTERN), wtw: 32-byte units of text to fetch (uniform, 1..4)
// pf: the round's windows, already fetched (two-word rounds under STRSIM_WIDE_PREFETCH; valid implies the round is not an edge round)
template <int MEASURE, int W>
__device__ __forceinline__ void wide_round(const WidePref &pf, const uint8_t *__restrict__ valA, uint32_t totalA,
                                           const uint8_t *__restrict__ valB, uint32_t totalB, uint32_t firstA, uint32_t firstB, bool has, uint32_t a0,
                                           uint32_t la, uint32_t b0, uint32_t lb, uint32_t wtw, const LdsTxt &txt, bool &done, double &res WIDE_STAMP_PARAMS
                                           STRSIM_COOP_RANGES)
{
    uint32_t wp[8 * W];
    uint32_t b0w = 0u, a0w = 0u;
    bool fast = has;
    uint32_t vary = 0u;
    // (opaque: the LDS addresses below depend on the lane alone, and hoisted out of the kernel's loops -- eight instructions' worth
    //  for each of the twelve piece geometries -- they are a hundred registers that live in scratch)
    uint32_t lane = lane_id();
    asm volatile("" : "+v"(lane));
    const uint32_t base = txt.row - lane * 128u; // the wave's 64 rows of 128 bytes
    uint32_t o = 0u, n = 0u;
    // (windows: 32 W bytes of pattern, 32 wtw bytes of text; a lane without a row fetches its column's first bytes)
    // (valA / valB and their totals are the LANE's: the symmetric measures take their text from either column, and a lane without a
    //  row reads its columns' first bytes -- the test covers every lane and its result is uniform: a cooperative fetch with some
    //  lanes switched off hands out null addresses [r4, found by tests/fuzz_gpu.py: a 32-byte literal against 33..40-byte rows])
    // [r5] "first bytes" = the column's first byte, offsets[0] -- not values + 0: with offsets that do not start at 0 that address lies
    // in front of the column, outside what include/strsim_amd.h lets the kernels read (found by the bounds-checked lab build on the
    // 2^20-offset-base test; inside the caller's buffer there, but a caller is free to pass values = buffer - offsets[0]).
    const bool edge = __ballot((has && ((uint64_t)b0 + 32u * (uint32_t)W > totalB || (uint64_t)a0 + 32u * wtw > totalA)) ||
                               totalB - firstB < 32u * (uint32_t)W || totalA - firstA < 32u * wtw) != 0ull;
    if (!edge) {
        // both fetches in flight together; the pattern passes through the rows first, then the text moves in
        const uint32_t pstart = has ? b0 : firstB, tstart = has ? a0 : firstA;
        const bool have = W == 2 && STRSIM_WIDE_PREFETCH && pf.valid; // (uniform)
        uint4 vp[CoopGeom<2 * W>::ITER];
        if (have) {
#pragma unroll
            for (int q = 0; q < CoopGeom<2 * W>::ITER; ++q) vp[q] = pf.vp[q & 3];
        } else {
            coop_fetch<2 * W>(valB, pstart, lane, vp STRSIM_COOP_RANGES_ARG);
        }
        auto text = [&](auto wt) {
            constexpr int WT = decltype(wt)::value;
            uint4 vt[CoopGeom<2 * WT>::ITER];
            if (have && WT <= 2 && STRSIM_WIDE_PREFETCH == 1) {
#pragma unroll
                for (int q = 0; q < CoopGeom<2 * WT>::ITER; ++q) vt[q] = pf.vt[q & 3];
            } else {
                coop_fetch<2 * WT>(valA, tstart, lane, vt STRSIM_COOP_RANGES_ARG);
            }
            coop_store_groups<2 * W>(base, lane, vp);
            wide_pattern_regs<W>(base, lane, wp);
            coop_store_text<2 * WT>(base, lane, vt);
            wide_text_or_and<WT>(txt, o, n);
        };
        if (wtw <= 1u) text(std::integral_constant<int, 1>{});
        else if (wtw == 2u) text(std::integral_constant<int, 2>{});
        else if (wtw == 3u) text(std::integral_constant<int, 3>{});
        else text(std::integral_constant<int, 4>{});
        a0w = txt(0u);
    } else { // a window reaches past its column's last byte: lane by lane, byte by byte where it must
#pragma unroll
        for (int d = 0; d < 8 * W; ++d) wp[d] = 0u;
        if (has) load_window_any<8 * W>(valB, (int64_t)b0, totalB, wp);
        if (wtw <= 1u) wide_text<1>(valA, totalA, has, a0, txt, o, n, a0w);
        else if (wtw == 2u) wide_text<2>(valA, totalA, has, a0, txt, o, n, a0w);
        else if (wtw == 3u) wide_text<3>(valA, totalA, has, a0, txt, o, n, a0w);
        else wide_text<4>(valA, totalA, has, a0, txt, o, n, a0w);
    }
    if (has) {
        if (MEASURE == JARO_WINKLER) b0w = wp[0];
#pragma unroll
        for (int d = 0; d < 8 * W; ++d) { o |= wp[d]; n &= wp[d]; }
        uint32_t o8 = o | (o >> 16); o8 |= o8 >> 8;
        uint32_t n8 = n & (n >> 16); n8 &= n8 >> 8;
        vary = (o8 ^ n8) & 0xFFu;
        fast = (o8 & 0x80u) == 0u; // any high bit in the windows: leave the row to the code-point kernel
    }
    const uint32_t lae = fast ? la : 1u, lbe = fast ? lb : 1u;
    const uint32_t ng4 = (wave_max_u8(fast ? la : 0u) + 3u) >> 2;
    const uint32_t gfull = (255u - wave_max_u8(255u - (fast ? la : 255u))) >> 2; // text dwords before the shortest live text ends
    done = false;
    if (__ballot(fast) == 0ull) return;
    const bool need7 = __ballot(fast && (vary & 0x40u)) != 0ull;
    const bool need6 = __ballot(fast && (vary & 0x20u)) != 0ull;
    const LdsSa sa{txt.row, txt.swz};
    // Jaro's second pass walks b: as far as the round's longest one reaches
    const uint32_t nb4 = (MEASURE == JARO || MEASURE == JARO_WINKLER) ? (wave_max_u8(fast ? lb : 0u) + 3u) >> 2 : 0u;
    // two instantiations per width (a six-plane one only inflated the kernel's register allocation)
    __builtin_amdgcn_s_setprio(0); // the column loops yield to waves that have loads to issue
    WIDE_STAMP(2);
    if (need7 || need6) res = lane_wide_result<MEASURE, 7, W>(txt, lae, gfull, ng4, wp, lbe, nb4, a0w, b0w, sa);
    else res = lane_wide_result<MEASURE, 5, W>(txt, lae, gfull, ng4, wp, lbe, nb4, a0w, b0w, sa);
    __builtin_amdgcn_s_setprio(1);
    done = fast;
}

// Register budget: three waves per SIMD (168 VGPRs): Levenshtein / Jaccard / Dice take 162, Jaro / Jaro-Winkler 165-167, nothing
// spilled ([r4]: one window mask instead of two, and the text no longer waits in 32 registers beside the pattern's 32; rounds
// 1-3: ~40 spilled in the four-word Jaro path).  LDS (53.5 KB) allows three workgroups per CU as well.  Two kernels, one per
// width, so that the two-word rows run at 4 waves per SIMD were tried in round 2: -2.5 % on cfg3 and -10 % on a
// 33-128-byte frame (a second scan + collect pass, half-empty rounds).
#ifndef STRSIM_WIDE_WAVES_PER_EU
#define STRSIM_WIDE_WAVES_PER_EU 3
#endif

template <int MEASURE>
__global__ __launch_bounds__(WIDE_BLOCK) __attribute__((amdgpu_waves_per_eu(STRSIM_WIDE_WAVES_PER_EU))) void k_lane_wide(const uint32_t *__restrict__ offA,
                                                          const uint8_t *__restrict__ valA, uint64_t rowsA,
                                                          const uint32_t *__restrict__ offB,
                                                          const uint8_t *__restrict__ valB, uint64_t rowsB,
                                                          double *__restrict__ out, uint64_t n,
                                                          unsigned long long *__restrict__ slowmask, uint32_t sps)
{
    // A workgroup takes a SUPER-span of up to WIDE_BLOCK mask words (16 384 rows: sps spans of WIDE_SPAN words, one word per
    // thread), collects the flagged rows it can run into an LDS list sorted by (width class, columns to run), and runs the list
    // 64 rows at a time.  The longer the list, the more alike the rows of a round: cfg3's rounds ran 74 % of their
    // lane-columns on a row that needed them with lists of one span (4 096 rows, 24 keys of 8 / 16 columns), 90 % with the whole
    // super and keys of 4 columns.  A super whose candidates do not fit the list (WIDE_LIST entries) is done span by span.
    // 4 width classes (masks of 1 .. 4 words: by the pattern, the longer string -- the one-word class stays empty) x 32 column
    // counts (4 columns each) x 4 quarters of the class's pattern lengths (Jaro's second pass walks the pattern as far as the
    // round's longest one reaches)
    constexpr int NKEY = 512, KPL = NKEY / 64;
    __shared__ unsigned long long s_mask[WIDE_BLOCK];
    __shared__ uint32_t s_cnt[NKEY];            // rows per key, then the key's next free list position
    __shared__ uint32_t s_next, s_total;        // next round to hand out; rows on the list
    __shared__ uint16_t s_list[WIDE_LIST];      // candidate rows (index within the super), sorted by key
    __shared__ uint32_t s_txt[WIDE_WAVES][8 * WIDE_MAXW][64];

    constexpr int RPS = WIDE_ROWS / WIDE_BLOCK; // rows per thread and span in the collection phase
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    __builtin_amdgcn_s_setprio(1);
#if defined(STRSIM_LAB) && defined(STRSIM_WIDE_STAMPS)
    unsigned long long wst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long wst_t0 = __builtin_amdgcn_s_memtime(), wst_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wst_last = wst_t0;
#endif
    const uint32_t totalA = offA[rowsA], totalB = offB[rowsB];
    const uint32_t firstA = offA[0], firstB = offB[0];
#if STRSIM_BOUNDS_ON
    ColumnRanges rg;
    rg.lo[0] = ((unsigned long long)(uintptr_t)valA + offA[0]) & ~15ull; rg.hi[0] = ((unsigned long long)(uintptr_t)valA + totalA + 15ull) & ~15ull;
    rg.lo[1] = ((unsigned long long)(uintptr_t)valB + offB[0]) & ~15ull; rg.hi[1] = ((unsigned long long)(uintptr_t)valB + totalB + 15ull) & ~15ull;
#endif
    const bool bcastA = rowsA == 1, bcastB = rowsB == 1;
    const uint64_t nchunks = (n + 63u) >> 6;
    const uint64_t nspans = (nchunks + WIDE_SPAN - 1) / WIDE_SPAN;

    // sps spans per super (1 .. WIDE_BLOCK / WIDE_SPAN, chosen by the launcher so that a mid-size frame still makes
    // enough workgroups to fill the chip); the common all-clear super costs one coalesced read and one barrier
    const uint64_t nsuper = (nspans + sps - 1) / sps;
    const uint32_t super_words = sps * (uint32_t)WIDE_SPAN;
    for (uint64_t sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
      const uint64_t cw0 = sup * super_words;
      const unsigned long long myword = (tid < super_words && cw0 + tid < nchunks) ? slowmask[cw0 + tid] : 0ull;
      if (!__syncthreads_or(myword != 0ull)) continue;
      s_mask[tid] = myword;
      uint32_t gsz = sps; // spans per list: the whole super, or one at a time when that does not fit
      for (uint32_t g0 = 0; g0 < sps;) {
        for (uint32_t q = tid; q < (uint32_t)NKEY; q += (uint32_t)WIDE_BLOCK) s_cnt[q] = 0u;
        if (tid == 0u) s_next = 0u;
        lds_barrier();
        // ---- key of a flagged row: 33..128-byte strings on the longer side and a non-empty shorter side; width class, then the
        //      number of DP columns (the text length).  The keys wait for the second pass in the text columns' LDS, idle until
        //      the rounds start (64 keys per thread in registers cost more than the kernel has: 1.6 KB of scratch per lane).
        uint16_t *const s_key = reinterpret_cast<uint16_t *>(&s_txt[0][0][0]);
        static_assert(sizeof(s_txt) >= (size_t)WIDE_BLOCK * 64 * 2, "a 16-bit key per row of a super");
        // The lengths of ALL rows of a span are loaded first, RPS rows per thread in flight at once (consecutive threads, consecutive
        // rows: whole lines), and only then looked at: loads behind the test of the row's mask bit wait for their data one row
        // at a time -- 64 round trips per thread and super, 23 % of the kernel's time on cfg3 ([r4], bench_support/wide_stamps.py).
        auto row_key = [&](uint32_t i, uint32_t la8, uint32_t lb8) -> uint32_t { // 0xFFFF: not a candidate
            if (!((s_mask[i >> 6] >> (i & 63u)) & 1ull)) return 0xFFFFu;
            const uint32_t mx = la8 > lb8 ? la8 : lb8, mn = la8 < lb8 ? la8 : lb8;
            if (!(mx > (STRSIM_WIDE_ONE_WORD ? 0u : 32u) && mx <= 128u && mn >= 1u)) return 0xFFFFu;
            const uint32_t steps = mn, pat = mx; // text = the shorter string, pattern = the longer one
            const uint32_t cls = (pat - 1u) >> 5; // masks of cls + 1 words: as wide as the PATTERN is long
            return (cls * 32u + ((steps - 1u) >> 2)) * 4u + (pat - 1u - 32u * cls) / 8u;
        };
#pragma unroll 1
        for (uint32_t sp = g0; sp < g0 + gsz; ++sp) {
            uint32_t la8[RPS], lb8[RPS];
#pragma unroll
            for (int kk = 0; kk < RPS; ++kk) {
                const uint32_t i = (sp * RPS + (uint32_t)kk) * WIDE_BLOCK + tid;
                const uint64_t r = cw0 * 64u + i, row = r < n ? r : n - 1u; // (rows behind the frame: their mask bits are clear)
                const uint64_t ra = bcastA ? 0 : row, rb = bcastB ? 0 : row;
                STRSIM_CHECK_INDEX(K_WIDE, 1, row, ra + 1u, rowsA + 1u);
                STRSIM_CHECK_INDEX(K_WIDE, 2, row, rb + 1u, rowsB + 1u);
                la8[kk] = offA[ra + 1] - offA[ra];
                lb8[kk] = offB[rb + 1] - offB[rb];
            }
#pragma unroll
            for (int kk = 0; kk < RPS; ++kk) {
                const uint32_t i = (sp * RPS + (uint32_t)kk) * WIDE_BLOCK + tid;
                const uint32_t key = row_key(i, la8[kk], lb8[kk]);
                STRSIM_CHECK_INDEX(K_WIDE, 3, cw0 * 64u + i, i, WIDE_BLOCK * 64);
                STRSIM_CHECK_RANGE(K_WIDE, 4, cw0 * 64u + i, key, 0, key == 0xFFFFu ? 0xFFFFu : (uint32_t)NKEY - 1u);
                s_key[i] = (uint16_t)key;
                if (key != 0xFFFFu) atomicAdd(&s_cnt[key], 1u);
            }
        }
        lds_barrier();
        // ---- list positions: exclusive prefix sum of the counters (wave 0, KPL keys per lane)
        if (wv == 0u) {
            uint32_t c[KPL], sum = 0u;
#pragma unroll
            for (int q = 0; q < KPL; ++q) { c[q] = s_cnt[(uint32_t)KPL * lane + (uint32_t)q]; sum += c[q]; }
            uint32_t inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
                if (lane >= (uint32_t)d) inc += up;
            }
            uint32_t base = inc - sum;
#pragma unroll
            for (int q = 0; q < KPL; ++q) { s_cnt[(uint32_t)KPL * lane + (uint32_t)q] = base; base += c[q]; }
            if (lane == 63u) s_total = inc;
        }
        lds_barrier();
        const uint32_t total = s_total;
        if (total > (uint32_t)WIDE_LIST) { // (only with gsz > 1: one span holds WIDE_ROWS <= WIDE_LIST rows)
            // as many spans per list as fit at this density, a power of two (checked again on the next trip)
            const uint32_t fit = gsz * (uint32_t)WIDE_LIST / total;
            gsz = fit >= 2u ? 2u : 1u;
            lds_barrier();
            continue;
        }
        if (total != 0u) {
#pragma unroll 1
            for (uint32_t sp = g0; sp < g0 + gsz; ++sp) {
#pragma unroll
                for (int kk = 0; kk < RPS; ++kk) {
                    const uint32_t i = (sp * RPS + (uint32_t)kk) * WIDE_BLOCK + tid;
                    const uint32_t key = s_key[i];
                    if (key != 0xFFFFu) {
                        const uint32_t pos = atomicAdd(&s_cnt[key], 1u);
                        STRSIM_CHECK_INDEX(K_WIDE, 5, cw0 * 64u + i, pos, WIDE_LIST);
                        s_list[pos] = (uint16_t)i;
                    }
                }
            }
            lds_barrier();
            // ---- rounds of 64 rows of similar width and length, longest first; a wave takes the next one when it is done
            //      (a four-word round costs four times a two-word one: dealing them out in turn leaves waves idle at the
            //      barrier behind the list)
            const uint32_t nrounds = (total + 63u) >> 6;
            // a round's rows and their offsets are fetched one round AHEAD: list entry -> offsets -> windows are three dependent
            // trips (LDS, then two scattered global loads), and three waves per SIMD do not hide two of the latter per round
            struct Rows { bool valid, has; uint32_t i, a0, la, b0, lb; };
            auto take = [&]() -> Rows {
                Rows q{false, false, 0u, 0u, 0u, 0u, 0u};
                uint32_t rr = 0u;
                if (lane == 0u) rr = atomicAdd(&s_next, 1u);
                rr = uniform(rr);
                if (rr >= nrounds) return q;
                q.valid = true;
                // (cut from the long end: the round that is not full holds the list's cheapest rows, not its dearest)
                const uint32_t hi = total - 64u * rr, first = hi >= 64u ? hi - 64u : 0u;
                const uint32_t li = first + lane;
                q.has = li < hi;
                if (q.has) STRSIM_CHECK_INDEX(K_WIDE, 6, cw0, li, WIDE_LIST);
                q.i = q.has ? s_list[li] : 0u;
                if (q.has) {
                    const uint64_t row = cw0 * 64u + q.i;
                    const uint64_t ra = bcastA ? 0 : row, rb = bcastB ? 0 : row;
                    STRSIM_CHECK_INDEX(K_WIDE, 7, row, row, n);
                    STRSIM_CHECK_INDEX(K_WIDE, 8, row, ra + 1u, rowsA + 1u);
                    STRSIM_CHECK_INDEX(K_WIDE, 9, row, rb + 1u, rowsB + 1u);
                    q.a0 = offA[ra]; q.la = offA[ra + 1] - q.a0;
                    q.b0 = offB[rb]; q.lb = offB[rb + 1] - q.b0;
                }
                return q;
            };
            WIDE_STAMP(0);
            // what a round's lanes fetch and how wide its masks are (uniform ballots over the round's rows)
            struct Cls {
                const uint8_t *vT, *vP; uint32_t tT, tP, fT, fP, t0, lt, p0, lp, wtw; bool pat2, pat3, pat4;
            };
            auto classify = [&](const Rows &q) -> Cls {
                Cls c;
                const bool swap = q.la > q.lb; // the columns walk the shorter string
                c.vT = swap ? valB : valA; c.vP = swap ? valA : valB;
                c.tT = swap ? totalB : totalA; c.tP = swap ? totalA : totalB;
                c.fT = swap ? firstB : firstA; c.fP = swap ? firstA : firstB; // (the columns' first bytes: offsets[0])
                c.t0 = swap ? q.b0 : q.a0; c.lt = swap ? q.lb : q.la; c.p0 = swap ? q.a0 : q.b0; c.lp = swap ? q.la : q.lb;
                c.pat2 = !STRSIM_WIDE_ONE_WORD || __ballot(q.has && c.lp > 32u) != 0ull;
                c.pat3 = __ballot(q.has && c.lp > 64u) != 0ull;
                c.pat4 = __ballot(q.has && c.lp > 96u) != 0ull;
                c.wtw = 1u + (__ballot(q.has && c.lt > 32u) != 0ull) + (__ballot(q.has && c.lt > 64u) != 0ull) + (__ballot(q.has && c.lt > 96u) != 0ull);
                return c;
            };
            const LdsTxt txt{STRSIM_LDS_ADDR(&s_txt[wv][0][0]) + lane * 128u, (lane & 31u) << 2};
            auto finish = [&](uint32_t i, bool done, double res) {
                const uint64_t row = cw0 * 64u + i;
                if (done) {
                    STRSIM_CHECK_INDEX(K_WIDE, 30, row, row, n);
                    STRSIM_CHECK_INDEX(K_WIDE, 31, row, i >> 6, WIDE_BLOCK);
                    out[row] = res;
                    atomicAnd(&s_mask[i >> 6], ~(1ull << (i & 63u)));
                }
            };
            WidePref none;
            none.valid = false;
            Rows cur = take();
            // ---- loop A: every round (with STRSIM_WIDE_PREFETCH: the rounds of three- and four-word masks -- they come first, the list
            //      is dealt from its long end -- and their registers leave no room for a deeper pipeline)
            while (cur.valid) {
                const Cls c = classify(cur);
                if (STRSIM_WIDE_PREFETCH && !c.pat3) break; // two-word rounds (and narrower) from here on: loop B
                const Rows nxt = take();
#if defined(STRSIM_LAB) && defined(STRSIM_WIDE_STAMPS)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                wst_acc[6] += 1;
#endif
                WIDE_STAMP(1);
                bool done = false;
                double res = 0.0;
                if (!c.pat2) { // (only with STRSIM_WIDE_ONE_WORD: a round whose longest pattern fits one word)
#if STRSIM_WIDE_ONE_WORD
                    wide_round<MEASURE, 1>(none, c.vT, c.tT, c.vP, c.tP, c.fT, c.fP, cur.has, c.t0, c.lt, c.p0, c.lp, c.wtw, txt, done, res WIDE_STAMP_ARGS STRSIM_COOP_RANGES_ARG);
#endif
                } else if (!c.pat3) // (the pattern is the longer string: two words or more)
                    wide_round<MEASURE, 2>(none, c.vT, c.tT, c.vP, c.tP, c.fT, c.fP, cur.has, c.t0, c.lt, c.p0, c.lp, c.wtw, txt, done, res WIDE_STAMP_ARGS STRSIM_COOP_RANGES_ARG);
                else if (!c.pat4)
                    wide_round<MEASURE, 3>(none, c.vT, c.tT, c.vP, c.tP, c.fT, c.fP, cur.has, c.t0, c.lt, c.p0, c.lp, c.wtw, txt, done, res WIDE_STAMP_ARGS STRSIM_COOP_RANGES_ARG);
                else
                    wide_round<MEASURE, 4>(none, c.vT, c.tT, c.vP, c.tP, c.fT, c.fP, cur.has, c.t0, c.lt, c.p0, c.lp, c.wtw, txt, done, res WIDE_STAMP_ARGS STRSIM_COOP_RANGES_ARG);
                WIDE_STAMP(3);
                finish(cur.i, done, res);
                cur = nxt;
                WIDE_STAMP(4);
            }
#if STRSIM_WIDE_PREFETCH
            // ---- loop B: the two-word rounds, one stage deeper: rows TWO rounds ahead, and the next round's windows go out into
            //      registers before this round's cores run (they land under the cores instead of in front of the next round)
            if (cur.valid) {
                Rows nxt = take();
                WidePref pf;
                pf.valid = false;
                while (cur.valid) {
                    const Rows nn = nxt.valid ? take() : nxt;
                    const Cls c = classify(cur);
                    WidePref pn;
                    pn.valid = false;
                    if (nxt.valid) {
                        const Cls cn = classify(nxt);
                        const bool edge_n = __ballot((nxt.has && ((uint64_t)cn.p0 + 64u > cn.tP || (uint64_t)cn.t0 + 32u * cn.wtw > cn.tT)) ||
                                                     cn.tP - cn.fP < 64u || cn.tT - cn.fT < 32u * cn.wtw) != 0ull;
                        if (cn.pat2 && !edge_n)
                            wide_prefetch2(cn.vT, cn.vP, nxt.has ? cn.t0 : cn.fT, nxt.has ? cn.p0 : cn.fP, cn.wtw, pn STRSIM_COOP_RANGES_ARG);
                    }
                    bool done = false;
                    double res = 0.0;
                    if (!c.pat2) {
#if STRSIM_WIDE_ONE_WORD
                        wide_round<MEASURE, 1>(none, c.vT, c.tT, c.vP, c.tP, c.fT, c.fP, cur.has, c.t0, c.lt, c.p0, c.lp, c.wtw, txt, done, res WIDE_STAMP_ARGS STRSIM_COOP_RANGES_ARG);
#endif
                    } else {
                        wide_round<MEASURE, 2>(pf, c.vT, c.tT, c.vP, c.tP, c.fT, c.fP, cur.has, c.t0, c.lt, c.p0, c.lp, c.wtw, txt, done, res WIDE_STAMP_ARGS STRSIM_COOP_RANGES_ARG);
                    }
                    finish(cur.i, done, res);
                    cur = nxt;
                    nxt = nn;
                    pf = pn;
                }
            }
#endif
        }
        lds_barrier();
        WIDE_STAMP(5);
        g0 += gsz;
      }
      if (tid < super_words && cw0 + tid < nchunks) slowmask[cw0 + tid] = s_mask[tid];
      lds_barrier();
    }
#if defined(STRSIM_LAB) && defined(STRSIM_WIDE_STAMPS)
    if (lane == 0u) {
        const uint32_t w = (blockIdx.x * WIDE_WAVES + wv) & 16383u;
        for (int q = 0; q < 8; ++q) g_wide_stamps[w][q] = wst_acc[q];
        g_wide_stamps[w][10] = __builtin_amdgcn_s_memtime() - wst_t0;
        g_wide_stamps[w][11] = __builtin_amdgcn_s_memrealtime() - wst_r0;
    }
#endif
}
