// strsim_lane_stage.h -- k_lane_stage<M>: the one-pair-per-lane kernel for strings of <= 32 ASCII bytes with the
// string bytes STAGED THROUGH LDS.  Included by strsim_kernels.hip inside namespace strsim, after its helpers.
//
// Per-pair arithmetic: strsim_lane_core.h (reference strsim.rs:125-162, :180-245, :257-272, :286-308, :322-345).
// How the bytes reach the lanes: a lane pulling its two 32-byte windows with unaligned 16-byte global loads in length-sorted
// (= scattered) order costs 4 load instructions of 64 different cache-line pairs each per 64 pairs, which keeps the CU's
// texture-address unit busy ~600 cycles per 64 pairs (round 1's kernel was bound by that unit).  Here
// a workgroup copies the CONTIGUOUS byte range of its block of rows from both value buffers into LDS with coalesced
// 16-byte-per-lane LDS-DMA loads (global_load_lds_dwordx4: every 128-byte line crosses the address unit once and
// never touches a VGPR) and the lanes then pick their windows out of LDS as nine aligned dwords + eight v_alignbyte.
//
//   per block of STAGE_ROWS consecutive rows (persistent 256-thread workgroup, grid-stride over blocks):
//     A  bytes DMA(j)      both columns' byte ranges of block j -> s_bytes (in flight during B and C)
//     B  store(j-1)        results of block j-1, staged in LDS by its rounds, leave as coalesced 8-byte stores
//     C  sortA/sortB(j)    offsets (in LDS since the previous block's rounds) -> lengths -> bucket by the number of
//                          DP columns -> rank by LDS atomics | barrier | wave scan of the counters -> 8-byte
//                          descriptors in length order
//     D  wait for the DMA, barrier
//     E  offsets DMA(j+1)  the next block's 2 x (STAGE_ROWS + 1) offsets -> s_off (in flight during F)
//     F  rounds(j)         wave w: rounds w and NR-1-w, ... of 64 pairs of similar length: descriptor -> two windows
//                          from LDS -> bit-parallel cores in registers -> result code (Levenshtein) / f64 into LDS
//     G  barrier
//   Three LDS-only barriers per block.  Blocks are cut to fit: a workgroup owns a contiguous range of 64-row chunks and
//   takes as many of the next chunks (at most STAGE_ROWS / 64) as have their bytes inside the staging area, so data
//   with longer rows simply runs in smaller blocks.  Rows longer than 32 bytes, non-ASCII rows and rows of a single
//   chunk that alone overflows the staging area stay in the mask for the later kernels.
#pragma once

#ifndef STRSIM_STAGE_ROWS
#define STRSIM_STAGE_ROWS 512
#endif
#ifndef STRSIM_STAGE_CAP
#define STRSIM_STAGE_CAP 10240 // staged bytes per column and block (cfg2: 512 rows are 8.4 KB +- 0.2 KB; a block that does not fit is cut)
#endif
#ifndef STRSIM_STAGE_CAP_LONG
#define STRSIM_STAGE_CAP_LONG 15104 // the same for the long-row geometry (40 KB of LDS: four workgroups per CU; cfg3: 13 312 2.02 ms, 14 336 1.95, 15 104 1.88)
#endif
#ifndef STRSIM_STAGE_CAP_LUT
#define STRSIM_STAGE_CAP_LUT 8960 // the same for the instantiations with match-mask tables (40 KB of LDS: four workgroups per CU)
#endif
#ifndef STRSIM_STAGE_WAVES_PER_EU
#define STRSIM_STAGE_WAVES_PER_EU 5
#endif
#ifndef STRSIM_STAGE_BUCKET_SHIFT
#define STRSIM_STAGE_BUCKET_SHIFT 1
#endif

#ifndef STRSIM_STAGE_LUT
#define STRSIM_STAGE_LUT 0x26 // bit m set: measure m (bit 5: the five-output pass) takes its match masks from per-lane LDS tables
#endif                       //   (strsim_lane_lut.h) instead of bit fills.  Default: Jaro, Jaro-Winkler, the five-output pass.
#ifndef STRSIM_JARO_KEEP_EQ_FILLS
#define STRSIM_JARO_KEEP_EQ_FILLS 1 // the bit-fill form of Jaro's cores (the long-row geometry) keeps the first pass's masks for the zip pass too
#endif
#ifndef STRSIM_STAGE_LONG_TEXT_MAX
// LONG geometry only (frames of long rows, cfg3): rows whose SHORTER string exceeds this many bytes are left to k_lane_wide's one-word
// class (STRSIM_WIDE_ONE_WORD).  32 = off.  Round 6's experiment: a block's critical path is its longest round,
// and on Zipf lengths the 64 longest of ~275 texts run 32 columns where the next round runs 14 (LAB 3.3 [r5]).
#define STRSIM_STAGE_LONG_TEXT_MAX 32
#endif
#ifndef STRSIM_STAGE_RANGE_MAX
#define STRSIM_STAGE_RANGE_MAX 64   // chunks of 64 rows a workgroup takes from the device-wide counter at a time, at most
#endif
#ifndef STRSIM_STAGE_RANGE_DIV
#define STRSIM_STAGE_RANGE_DIV 4    // ... = what is left / (this x workgroups) once less than RANGE_MAX x this x workgroups are left
#endif
#ifndef STRSIM_STAGE_RANGE_MIN_BLOCKS
#define STRSIM_STAGE_RANGE_MIN_BLOCKS 2 // ... and at least this many blocks (1: a 3 M-row call 132 us instead of 102, cfg2 +2 %: the counter is one contended address)
#endif

#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
// lab build only (make EXTRA="-DSTRSIM_LAB -DSTRSIM_STAGE_STAMPS"; never in the product library): per-wave cycle sums of the phases of k_lane_stage, read back with
// strsim_debug_stage_stamps().  [0] block cut + bytes DMA issue [1] store [2] sortA [3] barrier [4] sortB [5] DMA wait +
// barrier [6] offsets DMA issue [7] rounds: descriptor + windows [8] rounds: cores [9] barrier G [10] all [11] all (100 MHz)
__device__ unsigned long long g_stage_stamps[16384][16]; // [12], [13]: s_memrealtime at the wave's start and end
#define STAGE_STAMP(cat) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                              __builtin_amdgcn_sched_barrier(0); st_acc[cat] += t_ - st_last; st_last = t_; } while (0)
#else
#define STAGE_STAMP(cat) do { } while (0)
#endif

#ifndef STRSIM_STAGE_THREADS
#define STRSIM_STAGE_THREADS 256
#endif
constexpr int STAGE_BLOCK = STRSIM_STAGE_THREADS;     // threads per workgroup
constexpr int STAGE_WAVES = STAGE_BLOCK / 64;         // 4
constexpr int STAGE_ROWS = STRSIM_STAGE_ROWS;         // rows per block
constexpr int STAGE_RPT = STAGE_ROWS / STAGE_BLOCK;   // rows per thread in the coalesced phases
constexpr int STAGE_NR = STAGE_ROWS / 64;             // rounds per block
constexpr int STAGE_RPW = STAGE_NR / STAGE_WAVES;     // rounds per wave and block
// Which instantiations use the LDS match-mask tables.  The tables cost 12 KB of LDS per workgroup (4 workgroups per CU
// instead of 5) and pay where the kernel runs at 4 waves per SIMD or fewer anyway (Jaro's two passes; the five-output pass):
// Jaro 41 -> 45.6 G pairs/s on cfg2's lengths.  Levenshtein, Jaccard and Dice are issue-bound at 5 workgroups per CU and
// latency-bound at 4: 28 % fewer vector instructions bought nothing there.
template <int MEASURE> constexpr bool stage_uses_lut() { return ((STRSIM_STAGE_LUT >> MEASURE) & 1) != 0; } // (5: five outputs)

// staged bytes per column: Levenshtein keeps 16 bits per row for the store phase, the other measures 32 (64: five outputs)
// LONG: the geometry for frames of long strings (the launcher takes it when the context's last call left more than 1/16 of its
// rows to the later kernels): cfg3's rows average 26 bytes per column, 10 KB cut its blocks at 345 rows of which 185 are this
// kernel's -- 3.4 rounds for four waves and the per-block chain (copy, sort, store) for 2/3 of a block.  15 KB hold 512 rows
// (275 of this kernel's: 4.3 rounds) at four workgroups per CU: cfg3 2.24 -> 1.88 ms.
template <bool TABLES, bool LONG = false> struct StageGeom {
    static_assert(!(TABLES && LONG), "the long-row geometry has no match-mask tables");
    static constexpr int CAP = LONG ? STRSIM_STAGE_CAP_LONG : (TABLES ? STRSIM_STAGE_CAP_LUT : STRSIM_STAGE_CAP);
    static constexpr int COL = CAP + 96;              // LDS bytes per column: + 32 bytes behind the block, a chunk's rounding, 32 zeros
    static constexpr int DMA_ITERS = (CAP + 48 + 16 * STAGE_BLOCK - 1) / (16 * STAGE_BLOCK);
    static_assert(CAP % 16 == 0 && 2 * COL + 64 + 16 <= 65536, "staging area: whole wave-instructions, 16-bit LDS offsets");
};
constexpr int STAGE_BSH = STRSIM_STAGE_BUCKET_SHIFT;  // buckets of 2^BSH column counts
constexpr int STAGE_NBK = (32 >> STAGE_BSH) + 1;      // + one for the rows this kernel leaves to the later ones
static_assert(STAGE_RPT >= 1 && STAGE_RPT <= 4 && STAGE_RPW >= 1, "STAGE_ROWS is 256, 512 or 1024");
static_assert(STAGE_NBK <= 32, "the bucket scan runs on 32 lanes");

// LDS byte address of a __shared__ object (what M0 / a DS instruction's address operand hold)
#define STRSIM_LDS_ADDR(p) ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) void *)(p)))

// LDS-DMA: each active lane copies SIZE bytes from its own global address to lds_dst + lane * SIZE (lds_dst uniform).
// Inline asm on purpose: hipcc orders every later LDS access behind a __builtin_amdgcn_global_load_lds it knows about with
// s_waitcnt vmcnt(0) (it cannot tell the arrays apart), which would serialise exactly the overlap this kernel is built for.
// The kernel waits for its DMAs itself: s_waitcnt vmcnt(0) + workgroup barrier before the first read of the data.
// M0 (the LDS base of the DMA) is saved and restored inside the statement; s_nop 0: SALU write of M0 -> LDS-DMA read.
// (address = uniform base + a 32-bit byte offset per lane: no 64-bit vector arithmetic)
__device__ __forceinline__ void lds_dma_b128(const void *gbase, uint32_t goff, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(goff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void lds_dma_b32(const void *gbase, uint32_t goff, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(goff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

// a load the compiler may do on the scalar unit: the offsets are never written while the kernel runs (constant address
// space = invariant memory; a plain load behind the kernel's own stores becomes a vector load + s_waitcnt vmcnt(0))
__device__ __forceinline__ uint32_t load_invariant(const uint32_t *p)
{
    return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(reinterpret_cast<uintptr_t>(p));
}

// Descriptor of a row, 8 bytes, written in length order by sortB:
//   x = text window | pattern window << 16: byte index into s_bytes (staging areas of a and b, the literals' copies)
//   y = text length | pattern length << 8 | row index within the block << 16 | not mine << 31
// "text" is the string the columns of the bit-parallel cores walk: the shorter one when both sides are columns ([r5] for
// Jaro / Jaro-Winkler and the five-output pass too: the reference's greedy matching is symmetric, strsim_lane_core.h).
constexpr uint32_t STAGE_DEAD = 1u << 31;

#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STRIP)
// LAB ONLY (round 6, LAB_NOTES "an idea priced and not built"): Levenshtein on a pair's strings with their common prefix and suffix
// stripped (the distance does not change).  STRSIM_STAGE_STRIP = the most bytes stripped at either end.  Mode STRSIM_STAGE_STRIP_ORACLE:
// the strip lengths come from an array that a separate, UNTIMED kernel has filled (k_strip_oracle) -- the BENEFIT side of the idea
// alone: what the kernel gains from the shorter texts if knowing them were free.  A descriptor's y carries p + s in bits 25..30 (the
// row index needs 9 of its 11 bits): the denominator of the result is the ORIGINAL longer length = lp + p + s.
__device__ const uint16_t *g_strip_ptr; // per row: p | s << 8
__global__ __launch_bounds__(256) void k_strip_oracle(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, const uint32_t *__restrict__ offB,
                                                      const uint8_t *__restrict__ valB, uint64_t n, uint16_t *__restrict__ out)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t a0 = offA[r], la = offA[r + 1] - a0, b0 = offB[r], lb = offB[r + 1] - b0;
    const uint32_t m = la < lb ? la : lb, cap = (uint32_t)STRSIM_STAGE_STRIP;
    uint32_t p = 0, s = 0;
    while (p < m && p < cap && valA[a0 + p] == valB[b0 + p]) ++p;
    while (s < m - p && s < cap && valA[a0 + la - 1u - s] == valB[b0 + lb - 1u - s]) ++s;
    if (p + s > 63u) s = 63u - p;
    out[r] = (uint16_t)(p | (s << 8));
}
#define STRSIM_STRIP_ON 1
#else
#define STRSIM_STRIP_ON 0
#endif

// 32 bytes at byte offset `at` of `base` (an LDS array) into w[0..7]: nine dwords from the dword-aligned address below (56 LDS
// cycles per 64 lanes) + eight v_alignbyte_b32.  (gfx950 serves a wide LDS read that is not naturally aligned one lane at a time:
// two ds_read_b128 at the byte address cost 129 LDS cycles, bench_support/micro/lds_window.hip; cfg2 1.636 vs 1.587 ms.)
__device__ __forceinline__ void stage_window(const uint8_t *base, uint32_t at, uint32_t (&w)[8], uint32_t lds_bytes = 0u, uint32_t who = 0u)
{
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    (void)lds_bytes; (void)who;
    if (who == 1u) STRSIM_CHECK_RANGE(K_STAGE, 40, ~0ull, (at & ~3u) + 36u, 0, lds_bytes); // (nine dwords from the aligned address below `at`)
    if (who == 2u) STRSIM_CHECK_RANGE(K_LIT, 40, ~0ull, (at & ~3u) + 36u, 0, lds_bytes);
    const uint8_t *q = base + (at & ~3u);
    const u32x4_a4 lo = *reinterpret_cast<const u32x4_a4 *>(q);
    const u32x4_a4 hi = *reinterpret_cast<const u32x4_a4 *>(q + 16);
    const uint32_t top = *reinterpret_cast<const uint32_t *>(q + 32);
    const uint32_t sh = at & 3u;
    w[0] = __builtin_amdgcn_alignbyte(lo.y, lo.x, sh); w[1] = __builtin_amdgcn_alignbyte(lo.z, lo.y, sh);
    w[2] = __builtin_amdgcn_alignbyte(lo.w, lo.z, sh); w[3] = __builtin_amdgcn_alignbyte(hi.x, lo.w, sh);
    w[4] = __builtin_amdgcn_alignbyte(hi.y, hi.x, sh); w[5] = __builtin_amdgcn_alignbyte(hi.z, hi.y, sh);
    w[6] = __builtin_amdgcn_alignbyte(hi.w, hi.z, sh); w[7] = __builtin_amdgcn_alignbyte(top, hi.w, sh);
}

// inclusive prefix sum over lanes 0..31 (and, independently, 32..63) with five DPP adds: shifts by 1, 2, 4, 8 inside the
// rows of 16 lanes (lanes shifted in from outside a row read 0), then lane 15 of the even rows added to the odd rows
__device__ __forceinline__ uint32_t scan32_inclusive(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true); // row_bcast:15 into rows 1 and 3
    return x;
}

// Levenshtein result as an index into the table of integer quotients (dist * QTAB_N + den); 0xFFFF is never produced
template <int NP, bool USE_LUT> // match masks from bit fills / from the tables
__device__ __forceinline__ uint32_t stage_lev_code(const EqLut &lut, const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8],
                                                   uint32_t lb, uint32_t tmin, uint32_t tmax, uint32_t stripped = 0u)
{
    uint32_t P[NP];
    build_planes<NP>(wb, P);
    EqLut t = lut;
    if (USE_LUT) lut_build<NP>(t, P, 0xFFFFFFFFu);
    const bool live = la != 0u && lb != 0u;
    const uint32_t la1 = live ? la : 1u, lb1 = live ? lb : 1u;
    const uint32_t dist = USE_LUT ? lev_myers32_lut<NP>(t, wa, la1, tmin, tmax, P, lb1) : lev_myers32_snap<NP>(wa, la1, tmin, tmax, P, lb1);
    uint32_t code = dist * (uint32_t)QTAB_N + (la1 > lb1 ? la1 : lb1);
    // both empty: 1.0 = 1 - 0/1; one side empty: 0.0 = 1 - 1/1   (strsim.rs:128, :160)
    if (!live) code = (la == 0u && lb == 0u) ? 1u : (uint32_t)QTAB_N + 1u;
#if STRSIM_STRIP_ON
    // (lab) la / lb are what is left of the strings behind a common prefix and in front of a common suffix of `stripped` bytes in all;
    // the denominator is the original longer length; an empty side leaves the other side's length as the distance
    // (the longer one is lb between two columns -- the columns walk the shorter string -- but not with a literal side)
    const uint32_t longer = la > lb ? la : lb;
    code = (live ? dist : longer) * (uint32_t)QTAB_N + (longer + stripped);
#endif
    return code;
}

// All five measures of one pair from one set of bit-planes (BASELINE config 4), as the integers their epilogues need, packed
// into 64 bits: dist | m << 6 | t << 12 | I << 18 | common prefix << 24 | la << 27 | lb << 33 (all-ones is never produced).
// Jaro's matching serves Jaro and Jaro-Winkler, the multiset intersection Jaccard and Dice; "la" / "lb" are the lengths of the
// text and the pattern (the shorter and the longer string: every epilogue is symmetric in them).  The epilogues run in the store
// phase (stage_all_epilogues), once per row, on coalesced lanes.
template <int NP, bool USE_LUT>
__device__ __forceinline__ unsigned long long stage_all_ints(const EqLut &lut, const uint32_t (&wa)[8], uint32_t la,
                                                             const uint32_t (&wb)[8], uint32_t lb, uint32_t tmin, uint32_t tmax)
{
    uint32_t P[NP];
    build_planes<NP>(wb, P);
    const bool live = la != 0u && lb != 0u;
    const uint32_t la1 = live ? la : 1u, lb1 = live ? lb : 1u;
    uint32_t dist, m, t, isect; // one column loop, one match mask per column for the three cores (lane_cores32)
    if (USE_LUT) {
        EqLut tb = lut;
        lut_build<NP>(tb, P, 0xFFFFFFFFu);
        lane_cores32_lut<NP, true, true, true>(tb, wa, la1, tmin, tmax, lb1, P, dist, m, t, isect);
    } else {
        lane_cores32<NP, true, true, true>(wa, la1, tmin, tmax, lb1, P, dist, m, t, isect);
    }
    const uint32_t pre = common_prefix4(wa[0], la1, wb[0], lb1);
    const uint32_t lo = dist | (m << 6) | (t << 12) | (isect << 18) | (pre << 24) | (la << 27); // la: 5 of its 6 bits fit here
    return (unsigned long long)lo | ((unsigned long long)(la >> 5) << 32) | ((unsigned long long)lb << 33);
}

// the five f64 results of a packed row, the reference's operations in the reference's order (strsim.rs:160, :238-243,
// :260-270, :301-306, :337-343; early-outs :128-130, :182-186, :288-292, :324-328).  qa, qb, qc, qd: the integer quotients
// m / la, m / lb, (m - t / 2) / m and dist / max(la, lb) from the context's table (loaded by the caller, all rows at once).
__device__ __forceinline__ void stage_all_epilogues(unsigned long long pk, double qa, double qb, double qc, double qd, double (&r)[5])
{
    const uint32_t lo = (uint32_t)pk, hi = (uint32_t)(pk >> 32);
    const uint32_t m = (lo >> 6) & 63u, isect = (lo >> 18) & 63u, pre = (lo >> 24) & 7u;
    const uint32_t la = ((lo >> 27) & 31u) | ((hi & 1u) << 5), lb = (hi >> 1) & 63u;
    if (la == 0u || lb == 0u) {
        const double v = (la == 0u && lb == 0u) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) r[q] = v;
        return;
    }
    r[LEVENSHTEIN] = 1.0 - qd;
    const double j = m == 0u ? 0.0 : div3_exact(qa + qb + qc); // (strsim.rs:238-243; the division: strsim_lane_core.h)
    r[JARO] = j;
    r[JARO_WINKLER] = epilogue_jaro_winkler(j, pre);
    r[JACCARD] = epilogue_jaccard(isect, la, lb);
    r[SORENSEN_DICE] = epilogue_sorensen_dice(isect, la, lb);
}

// One of the other four measures as 32 bits of integers: Jaro / Jaro-Winkler: m | t << 6 | la << 12 | lb << 18 | common
// prefix << 24; Jaccard / Dice: I | la << 6 | lb << 12 (all-ones is never produced).  la / lb: text / pattern.
template <int MEASURE, int NP, bool USE_LUT>
__device__ __forceinline__ uint32_t stage_ints(const EqLut &lut, const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8],
                                               uint32_t lb, uint32_t tmin, uint32_t tmax)
{
    uint32_t P[NP];
    build_planes<NP>(wb, P);
    const bool live = la != 0u && lb != 0u;
    const uint32_t la1 = live ? la : 1u, lb1 = live ? lb : 1u;
    uint32_t dist, m, t, isect;
    EqLut tb = lut;
    if (USE_LUT) lut_build<NP>(tb, P, 0xFFFFFFFFu);
    if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        if (USE_LUT) lane_cores32_lut<NP, false, true, false>(tb, wa, la1, tmin, tmax, lb1, P, dist, m, t, isect);
        else lane_cores32<NP, false, true, false, STRSIM_JARO_KEEP_EQ_FILLS != 0>(wa, la1, tmin, tmax, lb1, P, dist, m, t, isect); // (the long-row geometry: four waves per SIMD too)
        const uint32_t pre = MEASURE == JARO_WINKLER ? common_prefix4(wa[0], la1, wb[0], lb1) : 0u;
        return m | (t << 6) | (la << 12) | (lb << 18) | (pre << 24);
    }
    if (USE_LUT) lane_cores32_lut<NP, false, false, true>(tb, wa, la1, tmin, tmax, lb1, P, dist, m, t, isect);
    else lane_cores32<NP, false, false, true>(wa, la1, tmin, tmax, lb1, P, dist, m, t, isect);
    return isect | (la << 6) | (lb << 12);
}

// its f64 epilogue, the reference's operations in the reference's order (strsim.rs:238-243, :260-270, :301-306, :337-343;
// early-outs :182-186, :288-292, :324-328).  qa, qb, qc: m / la, m / lb, (m - t / 2) / m from the table (Jaro, Jaro-Winkler).
template <int MEASURE>
__device__ __forceinline__ double stage_epilogue(uint32_t pk, double qa, double qb, double qc)
{
    if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        const uint32_t m = pk & 63u, la = (pk >> 12) & 63u, lb = (pk >> 18) & 63u, pre = (pk >> 24) & 7u;
        if (la == 0u || lb == 0u) return (la == 0u && lb == 0u) ? 1.0 : 0.0;
        const double j = m == 0u ? 0.0 : div3_exact(qa + qb + qc);
        return MEASURE == JARO_WINKLER ? epilogue_jaro_winkler(j, pre) : j;
    }
    const uint32_t isect = pk & 63u, la = (pk >> 6) & 63u, lb = (pk >> 12) & 63u;
    if (la == 0u || lb == 0u) return (la == 0u && lb == 0u) ? 1.0 : 0.0;
    return MEASURE == JACCARD ? epilogue_jaccard(isect, la, lb) : epilogue_sorensen_dice(isect, la, lb);
}

// `last`: the round's last lane that holds a row of this kernel (uniform)
template <int MEASURE, bool LUT>
__device__ __forceinline__ void stage_compute(const EqLut &lut, const uint32_t (&wt)[8], const uint32_t (&wp)[8], uint32_t meta,
                                              uint32_t last, uint16_t *s_code, uint32_t *s_word, double *s_val)
{
    bool fast = (meta & STAGE_DEAD) == 0u;
#if STRSIM_STRIP_ON
    const uint32_t lt = meta & 0x3Fu, lp = (meta >> 8) & 0x3Fu, idx = (meta >> 16) & 0x1FFu, stripped = (meta >> 25) & 0x3Fu;
#else
    const uint32_t lt = meta & 0x3Fu, lp = (meta >> 8) & 0x3Fu, idx = (meta >> 16) & 0x7FFu, stripped = 0u;
#endif
    // conservative tests on the whole 32-byte windows (bytes past a string belong to its neighbours): any high bit leaves
    // the row to the code-point kernels; the varying low bits decide how many bit-planes the match masks need
    uint32_t any;
    const uint32_t vary = window_vary(wt, wp, any);
    if (any & 0x80u) fast = false;
    if (__ballot(fast) == 0ull) return;
    const uint32_t la = fast ? lt : 0u, lb = fast ? lp : 0u;
    // the round's rows come in bucket order (2^STAGE_BSH text lengths per bucket): its last row's bucket bounds the longest
    // text from above (exactly, up to the rounding to whole column groups that the cores do anyway), its first row's bucket
    // the shortest from below -- two v_readlane instead of a ballot search
    static_assert((1 << STAGE_BSH) % COLS_PER_TEST == 0, "a bucket's upper bound is a whole number of column groups");
    const uint32_t ltl = (uint32_t)__builtin_amdgcn_readlane((int)lt, (int)last);
    const uint32_t tmax = (((ltl ? ltl : 1u) - 1u) | ((1u << STAGE_BSH) - 1u)) + 1u;
    const bool wide = __ballot(fast && (vary & 0x60u)) != 0ull; // six-plane rounds run as seven (register budget)
    const uint32_t lt0 = uniform(lt);
    const uint32_t tmin = (((lt0 ? lt0 : 1u) - 1u) & ~((1u << STAGE_BSH) - 1u)) + 1u;
    if (MEASURE == LEVENSHTEIN) {
        uint32_t code;
        if (wide) code = stage_lev_code<7, LUT>(lut, wt, la, wp, lb, tmin, tmax, stripped);
        else code = stage_lev_code<5, LUT>(lut, wt, la, wp, lb, tmin, tmax, stripped);
        if (fast) STRSIM_CHECK_INDEX(K_STAGE, 42, ~0ull, idx, STAGE_ROWS);
        if (fast) s_code[idx] = (uint16_t)code;
    } else if (MEASURE == ALL_MEASURES) {
        unsigned long long pk;
        if (wide) pk = stage_all_ints<7, LUT>(lut, wt, la, wp, lb, tmin, tmax);
        else pk = stage_all_ints<5, LUT>(lut, wt, la, wp, lb, tmin, tmax);
        if (fast) STRSIM_CHECK_INDEX(K_STAGE, 43, ~0ull, idx, STAGE_ROWS);
        if (fast) reinterpret_cast<unsigned long long *>(s_val)[idx] = pk;
    } else {
        // the measure's integers, 32 bits per row; its f64 epilogue runs in the store phase (stage_epilogue)
        uint32_t pk;
        if (wide) pk = stage_ints<MEASURE, 7, LUT>(lut, wt, la, wp, lb, tmin, tmax);
        else pk = stage_ints<MEASURE, 5, LUT>(lut, wt, la, wp, lb, tmin, tmax);
        if (fast) STRSIM_CHECK_INDEX(K_STAGE, 44, ~0ull, idx, STAGE_ROWS);
        if (fast) s_word[idx] = pk;
    }
}

template <int MEASURE, bool LUT, bool LONG = false>
__device__ __forceinline__ void
lane_stage_body(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, uint64_t rowsA,
             const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB, uint64_t rowsB, OutPtrs outs,
             uint64_t n, unsigned long long *__restrict__ slowmask, DevStatus *__restrict__ status,
             const double *__restrict__ qtab, uint32_t *__restrict__ sched, DevStatus *__restrict__ publish, uint32_t ticket)
{
    constexpr bool LEV = MEASURE == LEVENSHTEIN;
    constexpr bool ALL = MEASURE == ALL_MEASURES; // five outputs: outs.p[measure]; else outs.p[0]
    constexpr int B = STAGE_ROWS, RPT = STAGE_RPT, NBK = STAGE_NBK;
    constexpr int STAGE_CAP = StageGeom<LUT, LONG>::CAP, STAGE_COL = StageGeom<LUT, LONG>::COL;
    constexpr int STAGE_DMA_ITERS = StageGeom<LUT, LONG>::DMA_ITERS;
    constexpr uint32_t COLB = STAGE_COL, LIT = 2u * STAGE_COL; // s_bytes: column a | column b | the literals' windows
    // Match-mask tables (strsim_lane_lut.h), for the measures that use them: per wave 3 KB at a 4 KB boundary -- entry e of
    // lane l at e * 256 + l * 4, and the boundary makes (table base >> 8) | e the address byte.
    // The 1 KB behind each wave's tables holds 128 of the block's row descriptors (without tables: an array of their own).
    static_assert(!LUT || (B <= 128 * STAGE_WAVES && LUT_WAVE_BYTES == 3072), "descriptor slices: 128 rows behind each wave's tables");
    __shared__ __attribute__((aligned(4096))) uint8_t s_lut[LUT ? 4096 * STAGE_WAVES : 16];
    __shared__ uint2 s_desc_own[LUT ? 1 : B];
    // (+ 16: a window is read as NINE dwords from the aligned address below it -- the ninth only feeds the alignment shift -- and the
    //  second literal's window starts 32 bytes before the end of the 64: its ninth dword lay 4 bytes past the array.  [r5] Found by the
    //  bounds-checked lab build (csrc/strsim_bounds.h) on a one-row x one-row call; the value was never used, the read was out of the array.)
    __shared__ __attribute__((aligned(4096))) uint8_t s_bytes[2 * STAGE_COL + 64 + 16];
    auto desc_at = [&](uint32_t p) -> uint2 * { // descriptor of position p of the length order
        if (LUT) return reinterpret_cast<uint2 *>(s_lut + ((p >> 7) << 12) + LUT_WAVE_BYTES + ((p & 127u) << 3));
        return s_desc_own + p;
    };
    __shared__ __attribute__((aligned(16))) uint32_t s_off[2][B + 4]; // offsets of the rows from the next block's start on
    __shared__ uint32_t s_cnt[32];
    __shared__ uint32_t s_left;                // rows of this workgroup's blocks that stay in the mask
    __shared__ uint32_t s_sched[2];            // the next range of 64-row chunks of this workgroup: first chunk, chunks
    // results wait in LDS for the coalesced store phase as the integers their f64 epilogue needs (all-ones = the row was not
    // computed here): Levenshtein a 16-bit table index, the other measures 32 bits, the five-output pass 64
    __shared__ uint16_t s_code[LEV ? B : 1];
    __shared__ uint32_t s_word[(!LEV && !ALL) ? B : 1];
    __shared__ double s_val[ALL ? B : 1];

    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    // the call's status block (counters of the kernels that follow in the stream) is cleared here, not by a memset node
    if (blockIdx.x == 0u && tid < (uint32_t)(sizeof(DevStatus) / sizeof(uint32_t))) reinterpret_cast<uint32_t *>(status)[tid] = 0u;
    if (tid < 32u) s_cnt[tid] = 0u;
    if (tid == 0u) s_left = 0u;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const uint32_t i = (uint32_t)q * STAGE_BLOCK + tid;
        if (LEV) s_code[i] = 0xFFFFu;
        else if (ALL) reinterpret_cast<unsigned long long *>(&s_val[i])[0] = ~0ull;
        else s_word[i] = 0xFFFFFFFFu;
    }

    const uint32_t totalA = load_invariant(offA + rowsA), totalB = load_invariant(offB + rowsB);
#if STRSIM_BOUNDS_ON
    const uint32_t firstA = load_invariant(offA), firstB = load_invariant(offB); // (lab: where the columns' bytes begin)
#endif
    const bool bcastA = rowsA == 1, bcastB = rowsB == 1; // a one-row side is the literal (strsim.rs:48-52, :61-66)
    const uint32_t litA0 = bcastA ? load_invariant(offA) : 0u, litB0 = bcastB ? load_invariant(offB) : 0u;
    const uint32_t litAlen = bcastA ? totalA - litA0 : 0u, litBlen = bcastB ? totalB - litB0 : 0u;
    if (wv == 0u) {
        // the literals' 32-byte windows behind the two staging areas, padded with copies of the literal's first byte: the
        // bytes behind a string are never compared, but they take part in the test of which bit-planes tell a round's bytes
        // apart, and zeros there would ask for seven planes in every round
        auto literal_window = [&](const uint8_t *val, uint32_t at, uint32_t total, uint32_t len, uint32_t *dst) {
            uint32_t w[8];
            load_window32(val, at, total, w);
            const uint32_t rep = (w[0] & 0xFFu) * 0x01010101u;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t v = len > 4u * q ? (len - 4u * q < 4u ? len - 4u * q : 4u) : 0u; // bytes of this dword that are text
                const uint32_t keep = v >= 4u ? 0xFFFFFFFFu : ((1u << (8u * v)) - 1u);
                dst[q] = (w[q] & keep) | (rep & ~keep);
            }
        };
        if (bcastA) literal_window(valA, litA0, totalA, litAlen, reinterpret_cast<uint32_t *>(s_bytes + LIT));
        if (bcastB) literal_window(valB, litB0, totalB, litBlen, reinterpret_cast<uint32_t *>(s_bytes + LIT + 32u));
    }
    // Work distribution: ranges of 64-row chunks handed out by a device-wide counter (sched[0]; sched[1] counts the
    // workgroups that have finished, the last one clears both for the next launch).  A static split would do if all
    // workgroups ran equally fast, but the SIMD serves its OLDEST wave first, so of the four workgroups resident on a CU
    // the first finishes its share at 57 % of the kernel's time and the last one runs alone at the end (measured: ends at
    // 1.15 / 1.40 / 1.70 / 2.01 ms of a 2.05 ms launch).  The size of a range shrinks with the work that is left
    // (remaining / (4 x workgroups), between one block and 64 chunks), so all workgroups end within about a block.
    const uint64_t nchunks = (n + 63u) >> 6;
    const uint32_t nchunks32 = (uint32_t)nchunks; // (at most 2^32 - 1 rows per call)
    auto grab = [&](uint32_t seen, uint32_t &lo, uint32_t &sz) { // thread 0 only
        // (64 * 4 * workgroups chunks left or more: the cap; no division on this path)
        const uint32_t left = nchunks32 > seen ? nchunks32 - seen : 0u;
        uint32_t want = (uint32_t)STRSIM_STAGE_RANGE_MAX;
        if (left < (uint32_t)(STRSIM_STAGE_RANGE_MAX * STRSIM_STAGE_RANGE_DIV) * gridDim.x) want = left / ((uint32_t)STRSIM_STAGE_RANGE_DIV * gridDim.x);
        sz = want < (uint32_t)(STRSIM_STAGE_RANGE_MIN_BLOCKS * (B / 64)) ? (uint32_t)(STRSIM_STAGE_RANGE_MIN_BLOCKS * (B / 64)) : want;
        lo = __hip_atomic_fetch_add(&sched[0], sz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t grab_lo = 0u, grab_sz = 0u; // thread 0: the range after the next one, on its way
    if (tid == 0u) {
        grab(0u, grab_lo, grab_sz);
        s_sched[0] = grab_lo;
        s_sched[1] = grab_sz;
    }
    lds_barrier();
    uint64_t row_end; // first row behind the current range
    uint64_t row0;    // first row of the block about to be processed
    {
        const uint64_t lo = uniform(s_sched[0]), hi = lo + uniform(s_sched[1]); // (uniform: row counters live in SGPRs)
        row0 = lo * 64u < n ? lo * 64u : n;
        row_end = hi * 64u < n ? hi * 64u : n;
    }
    lds_barrier();
    if (tid == 0u) grab(grab_lo, grab_lo, grab_sz); // (written to s_sched at the top of the first block)
    bool grab_pending = true;

    EqLut lut; // this wave's match-mask tables
    lut.lane4 = lane * 4u;
    lut.krep = ((STRSIM_LDS_ADDR(&s_lut[0]) >> 8) + 16u * wv) * 0x01010101u;
    const uint32_t ldsOffA = STRSIM_LDS_ADDR(&s_off[0][0]), ldsOffB = STRSIM_LDS_ADDR(&s_off[1][0]);
    const uint32_t ldsBytes = STRSIM_LDS_ADDR(&s_bytes[0]);
    const uint32_t wvu = uniform(wv);
    // offsets DMA: the offsets of rows row .. row + min(B, row_end - row) (one more than rows) of each column side that
    // is not a literal -> s_off[side][0 ..]
    const uint32_t tid4 = tid * 4u, tid16 = tid * 16u;
    auto dma_offsets = [&](uint64_t row) {
        const uint32_t cnt = (uint32_t)(row_end - row < (uint64_t)B ? row_end - row : (uint64_t)B); // rows available
        const uint32_t *const pa = offA + row, *const pb = offB + row;
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            if (tid + (uint32_t)it * STAGE_BLOCK <= cnt) {
                // (lab: offsets[row + it * 256 + tid] of an array of rows + 1 entries -> s_off[side][it * 256 + tid] of B + 4)
                if (!bcastA) STRSIM_CHECK_INDEX(K_STAGE, 1, row, row + (uint64_t)it * STAGE_BLOCK + tid, rowsA + 1u);
                if (!bcastB) STRSIM_CHECK_INDEX(K_STAGE, 2, row, row + (uint64_t)it * STAGE_BLOCK + tid, rowsB + 1u);
                STRSIM_CHECK_INDEX(K_STAGE, 3, row, (uint32_t)it * STAGE_BLOCK + tid, B + 4);
                if (!bcastA) lds_dma_b32(pa + it * STAGE_BLOCK, tid4, ldsOffA + 4u * ((uint32_t)it * STAGE_BLOCK + wvu * 64u));
                if (!bcastB) lds_dma_b32(pb + it * STAGE_BLOCK, tid4, ldsOffB + 4u * ((uint32_t)it * STAGE_BLOCK + wvu * 64u));
            }
        }
        if (tid == 0u && cnt == (uint32_t)B) { // the offset behind the last row of a full block
            if (!bcastA) STRSIM_CHECK_INDEX(K_STAGE, 4, row, row + (uint64_t)B, rowsA + 1u);
            if (!bcastB) STRSIM_CHECK_INDEX(K_STAGE, 5, row, row + (uint64_t)B, rowsB + 1u);
            if (!bcastA) lds_dma_b32(pa + B, tid4, ldsOffA + 4u * (uint32_t)B);
            if (!bcastB) lds_dma_b32(pb + B, tid4, ldsOffB + 4u * (uint32_t)B);
        }
    };

    uint64_t prev_row0 = 0;           // the block whose results are waiting in LDS
    uint32_t prev_rows = 0;
    if (row0 < row_end) dma_offsets(row0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    // store(j-1): staged results -> global, coalesced; rows nobody computed go into the mask word of their chunk.
    // Three straight-line passes -- staged integers out of LDS, every table load of every row of the thread, then the
    // arithmetic and the stores: left to itself hipcc sinks each load into the branch that stores its row and waits for it
    // there (s_waitcnt vmcnt(0) per row: the L2 round trips of a thread's rows one after the other, each behind the previous
    // row's store).  `pin` keeps a loaded value outside those branches.
    auto pin = [](double &x) { asm volatile("" : "+v"(x)); };
    auto store_block = [&](uint64_t r0, uint32_t rows) {
        unsigned long long *__restrict__ const maskb = slowmask + (r0 >> 6);
        constexpr int M1 = (MEASURE == ALL_MEASURES || MEASURE == LEVENSHTEIN) ? JARO : MEASURE;
        constexpr bool JARO_LIKE = ALL || M1 == JARO || M1 == JARO_WINKLER;
        unsigned long long pk[RPT]; // (rows of the block behind `rows` hold all-ones: nobody computed them)
        double t0[RPT], t1[RPT], t2[RPT], t3[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = (uint32_t)q * STAGE_BLOCK + tid;
            if (ALL) {
                pk[q] = reinterpret_cast<const unsigned long long *>(s_val)[i];
                reinterpret_cast<unsigned long long *>(s_val)[i] = ~0ull;
            } else if (LEV) {
                pk[q] = s_code[i];
                s_code[i] = 0xFFFFu;
            } else {
                pk[q] = s_word[i];
                s_word[i] = 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            // the table entries the row's epilogue needs; an all-ones (not computed) row reads valid entries too
            if (LEV) {
                // 1.0 - dist / den (strsim.rs:160) with the quotient from the context's table of integer quotients: the
                // same IEEE division, done once per context on the host instead of once per row (the table is 34 KB,
                // L2-resident; it used to be 8.5 KB of LDS per workgroup = one workgroup per CU less)
                STRSIM_CHECK_INDEX(K_STAGE, 12, r0, (uint32_t)pk[q] == 0xFFFFu ? 0u : (uint32_t)pk[q], QTAB_N * QTAB_N);
                t0[q] = qtab[(uint32_t)pk[q] == 0xFFFFu ? 0u : (uint32_t)pk[q]];
            } else if (JARO_LIKE) {
                const uint32_t lo = (uint32_t)pk[q], hi = (uint32_t)(pk[q] >> 32);
                const uint32_t m = (lo >> (ALL ? 6 : 0)) & 63u, t = (lo >> (ALL ? 12 : 6)) & 63u;
                const uint32_t la = ALL ? (((lo >> 27) & 31u) | ((hi & 1u) << 5)) : ((lo >> 12) & 63u);
                const uint32_t lb = ALL ? ((hi >> 1) & 63u) : ((lo >> 18) & 63u);
                const uint32_t h = t >> 1;
                STRSIM_CHECK_INDEX(K_STAGE, 13, r0, m * (uint32_t)QTAB_N + la, QTAB_N * QTAB_N);
                STRSIM_CHECK_INDEX(K_STAGE, 14, r0, m * (uint32_t)QTAB_N + lb, QTAB_N * QTAB_N);
                t0[q] = qtab[m * (uint32_t)QTAB_N + la];
                t1[q] = qtab[m * (uint32_t)QTAB_N + lb];
                t2[q] = qtab[(m > h ? m - h : 0u) * (uint32_t)QTAB_N + m];
                if (ALL) {
                    const uint32_t dist = lo & 63u;
                    t3[q] = qtab[dist * (uint32_t)QTAB_N + (la > lb ? la : lb)];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (LEV || JARO_LIKE) pin(t0[q]);
            if (JARO_LIKE) { pin(t1[q]); pin(t2[q]); }
            if (ALL) pin(t3[q]);
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = (uint32_t)q * STAGE_BLOCK + tid;
            if ((uint32_t)q * STAGE_BLOCK < rows) { // (uniform)
                bool undone;
                double v = 0.0, v5[5];
                if (ALL) {
                    undone = pk[q] == ~0ull;
                    if (!undone) stage_all_epilogues(pk[q], t0[q], t1[q], t2[q], t3[q], v5);
                } else if (LEV) {
                    undone = (uint32_t)pk[q] == 0xFFFFu;
                    v = 1.0 - t0[q];
                } else {
                    undone = (uint32_t)pk[q] == 0xFFFFFFFFu;
                    if (!undone) v = stage_epilogue<M1>((uint32_t)pk[q], t0[q], t1[q], t2[q]);
                }
                const bool valid = i < rows;
                const unsigned long long left = __ballot(undone && valid);
                if (valid) STRSIM_CHECK_INDEX(K_STAGE, 10, r0 + i, r0 + i, n);                  // the result column
                if (lane == 0u && valid) STRSIM_CHECK_INDEX(K_STAGE, 11, r0 + i, (r0 >> 6) + (i >> 6), nchunks); // the mask word
                if (valid && !undone) {
                    if (ALL) {
#pragma unroll
                        for (int o = 0; o < 5; ++o) outs.p[o][r0 + i] = v5[o];
                    } else {
                        outs.p[0][r0 + i] = v;
                    }
                }
                if (lane == 0u && valid) {
                    maskb[i >> 6] = left;
                    if (publish && left) atomicAdd(&s_left, (uint32_t)__builtin_popcountll(left));
                }
            }
        }
    };

#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_last = st_t0;
#endif
    while (row0 < row_end) {
        if (grab_pending) { // the range grabbed when this one was entered: publish it (read behind at least one barrier)
            if (tid == 0u) {
                s_sched[0] = grab_lo;
                s_sched[1] = grab_sz;
            }
            grab_pending = false;
        }
        // ---- B: store(j-1).  Before the bytes DMA of this block is issued: a wave's vector-memory results return in order,
        //         so the table loads of the store phase would otherwise sit behind its share of the DMA (HBM latency).
        if (prev_rows) store_block(prev_row0, prev_rows);
        STAGE_STAMP(1);
        // ---- the block: as many of the next 64-row chunks (at most B / 64) as have their bytes inside the staging areas
        const uint32_t avail = (uint32_t)(row_end - row0 < (uint64_t)B ? row_end - row0 : (uint64_t)B); // rows whose offsets are in s_off
        const uint32_t baseA = bcastA ? litA0 : uniform(s_off[0][0]), baseB = bcastB ? litB0 : uniform(s_off[1][0]);
        const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(valA + baseA) & 15u);
        const uint32_t misB = (uint32_t)(reinterpret_cast<uintptr_t>(valB + baseB) & 15u);
        uint32_t rows;
        {
            // lane c < B / 64: does the block still fit if it ends behind chunk c?
            const uint32_t e = (lane + 1u) * 64u < avail ? (lane + 1u) * 64u : avail; // rows if the block ends behind chunk `lane`
            const bool exists = lane * 64u < avail && lane < (uint32_t)(B / 64);
            const uint32_t endA = bcastA ? baseA : s_off[0][exists ? e : 0u], endB = bcastB ? baseB : s_off[1][exists ? e : 0u];
            const bool fits = exists && endA - baseA + misA <= (uint32_t)STAGE_CAP && endB - baseB + misB <= (uint32_t)STAGE_CAP;
            const unsigned long long okm = __ballot(fits);
            // chunks 0 .. k-1 fit: k = number of trailing ones; at least one chunk is taken even if it overflows
            uint32_t k = (uint32_t)__builtin_ctzll(~okm);
            if (k == 0u) k = 1u;
            rows = uniform(k * 64u < avail ? k * 64u : avail);
        }
        const uint32_t endA = bcastA ? baseA : uniform(s_off[0][rows]), endB = bcastB ? baseB : uniform(s_off[1][rows]);
        // ---- A: bytes DMA(j).  Column side X: the 16-byte chunks from the aligned address below the block's first byte
        //         up to the block's last byte, at most STAGE_CAP bytes -> s_bytes[X]
        const uint32_t spanA = endA - baseA + misA, spanB = endB - baseB + misB;
        // the strings of the block must lie inside the first stagedX bytes; the copy itself takes the 32 bytes behind them
        // along when the column has them (the window of the block's last rows reaches there), and 32 zeros follow the copy --
        // a window must never show what an earlier block left in LDS: one stale byte with its high bit set and the row would
        // be taken for non-ASCII and left to the slow kernels
        const uint32_t stagedA = bcastA ? 0u : (((spanA < (uint32_t)STAGE_CAP ? spanA : (uint32_t)STAGE_CAP) + 15u) & ~15u);
        const uint32_t stagedB = bcastB ? 0u : (((spanB < (uint32_t)STAGE_CAP ? spanB : (uint32_t)STAGE_CAP) + 15u) & ~15u);
        const uint32_t leftA = totalA - baseA + misA, leftB = totalB - baseB + misB; // column bytes from the aligned start on
        const uint32_t chunksA = bcastA ? 0u : ((stagedA + 32u < leftA ? stagedA + 32u : leftA) + 15u) >> 4;
        const uint32_t chunksB = bcastB ? 0u : ((stagedB + 32u < leftB ? stagedB + 32u : leftB) + 15u) >> 4;
        if (tid < 2u) STRSIM_CHECK_RANGE(K_STAGE, 20, row0, (chunksA << 4) + tid * 16u + 16u, 0, COLB);
        else if (tid < 4u) STRSIM_CHECK_RANGE(K_STAGE, 21, row0, COLB + (chunksB << 4) + (tid - 2u) * 16u + 16u, 0, sizeof(s_bytes));
        if (tid < 2u) *reinterpret_cast<uint4 *>(s_bytes + (chunksA << 4) + tid * 16u) = make_uint4(0u, 0u, 0u, 0u);
        else if (tid < 4u) *reinterpret_cast<uint4 *>(s_bytes + COLB + (chunksB << 4) + (tid - 2u) * 16u) = make_uint4(0u, 0u, 0u, 0u);
        {
            const uint8_t *__restrict__ const gA = valA + baseA - misA, *__restrict__ const gB = valB + baseB - misB;
#pragma unroll
            for (int it = 0; it < STAGE_DMA_ITERS; ++it) {
                if (tid + (uint32_t)it * STAGE_BLOCK < chunksA) {
                    // (lab: 16 bytes of the column by the over-read contract -> 16 bytes of column a's staging area)
                    STRSIM_CHECK_COLUMN(K_STAGE, 22, row0, gA + 16 * it * STAGE_BLOCK + tid16, 16, valA, firstA, totalA);
                    STRSIM_CHECK_RANGE(K_STAGE, 23, row0, 16u * ((uint32_t)it * STAGE_BLOCK + tid) + 16u, 0, COLB);
                    lds_dma_b128(gA + 16 * it * STAGE_BLOCK, tid16, ldsBytes + 16u * ((uint32_t)it * STAGE_BLOCK + wvu * 64u));
                }
                if (tid + (uint32_t)it * STAGE_BLOCK < chunksB) {
                    STRSIM_CHECK_COLUMN(K_STAGE, 24, row0, gB + 16 * it * STAGE_BLOCK + tid16, 16, valB, firstB, totalB);
                    STRSIM_CHECK_RANGE(K_STAGE, 25, row0, COLB + 16u * ((uint32_t)it * STAGE_BLOCK + tid) + 16u, 0, sizeof(s_bytes));
                    lds_dma_b128(gB + 16 * it * STAGE_BLOCK, tid16, ldsBytes + COLB + 16u * ((uint32_t)it * STAGE_BLOCK + wvu * 64u));
                }
            }
        }
        STAGE_STAMP(0);
        // ---- C: sortA(j): lengths -> bucket keys (number of DP columns the pair will run), ranks by LDS atomics
        uint32_t skey[RPT], srank[RPT], sd0[RPT], sd1[RPT];
        {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t i = (uint32_t)RPT * tid + (uint32_t)q;
                STRSIM_CHECK_INDEX(K_STAGE, 30, row0, i + 1u, B + 4);
                const bool have = i < rows;
                const uint32_t a0 = bcastA ? 0u : s_off[0][i] - baseA, b0 = bcastB ? 0u : s_off[1][i] - baseB;
                const uint32_t la8 = bcastA ? litAlen : s_off[0][i + 1u] - s_off[0][i];
                const uint32_t lb8 = bcastB ? litBlen : s_off[1][i + 1u] - s_off[1][i];
                // mine: both strings <= 32 bytes and inside what was copied to the staging areas (the literal has its own copy)
                const bool stA = bcastA || misA + a0 + la8 <= stagedA, stB = bcastB || misB + b0 + lb8 <= stagedB;
                bool mine = have && la8 <= 32u && lb8 <= 32u && stA && stB;
                if (LONG && STRSIM_STAGE_LONG_TEXT_MAX < 32 && !bcastA && !bcastB) mine = mine && (la8 < lb8 ? la8 : lb8) <= (uint32_t)STRSIM_STAGE_LONG_TEXT_MAX;
                const uint32_t wa = bcastA ? LIT : misA + a0, wb = bcastB ? LIT + 32u : COLB + misB + b0;
                const bool swap = !bcastA && !bcastB && la8 > lb8; // the columns walk the shorter string (every measure is symmetric: strsim_lane_core.h)
#if STRSIM_STRIP_ON
                // (lab, Levenshtein, two columns) the strings behind their common prefix and in front of their common suffix
                uint32_t sp = 0u, ss = 0u;
                if (LEV && !bcastA && !bcastB && have) {
                    const uint32_t ps = g_strip_ptr[row0 + i];
                    sp = ps & 0xFFu; ss = ps >> 8;
                }
                const uint32_t lt = (swap ? lb8 : la8) - sp - ss, lp = (swap ? la8 : lb8) - sp - ss;
                const uint32_t key = mine ? (((lt ? lt : 1u) - 1u) >> STAGE_BSH) : (uint32_t)(NBK - 1);
                skey[q] = key;
                srank[q] = atomicAdd(&s_cnt[key], 1u);
                sd0[q] = mine ? (swap ? ((wb + sp) | ((wa + sp) << 16)) : ((wa + sp) | ((wb + sp) << 16))) : 0u;
                sd1[q] = mine ? (lt | (lp << 8) | (i << 16) | ((sp + ss) << 25)) : STAGE_DEAD;
#else
                const uint32_t lt = swap ? lb8 : la8, lp = swap ? la8 : lb8;
                const uint32_t key = mine ? (((lt ? lt : 1u) - 1u) >> STAGE_BSH) : (uint32_t)(NBK - 1);
                skey[q] = key;
                srank[q] = atomicAdd(&s_cnt[key], 1u);
                sd0[q] = mine ? (swap ? (wb | (wa << 16)) : (wa | (wb << 16))) : 0u;
                sd1[q] = mine ? (lt | (lp << 8) | (i << 16)) : STAGE_DEAD;
#endif
            }
        }
        STAGE_STAMP(2);
        lds_barrier();
        STAGE_STAMP(3);
        // ---- C: sortB(j): exclusive scan of the bucket counters (every wave for itself), descriptors in length order
        uint32_t nmine;
        {
            const uint32_t c = s_cnt[lane & 31u];
            const uint32_t exc = scan32_inclusive(c) - c;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t base = __shfl(exc, skey[q], 32);
                const uint32_t p = base + srank[q]; // position in length order
                STRSIM_CHECK_INDEX(K_STAGE, 31, row0, p, B);
                *desc_at(p) = make_uint2(sd0[q], sd1[q]);
            }
            nmine = uniform(__shfl(exc, NBK - 1, 32));
        }
        STAGE_STAMP(4);
        // ---- D: the bytes have landed (every wave waits for its own DMA, the barrier makes that workgroup-wide)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // The column's last 16-byte chunk may reach past the column's last byte: what lies there is not the caller's (pad bytes,
        // another allocation) and a high bit in it would send the block's last rows to the slow kernels at random.  The lane
        // that copied that chunk zeroes its tail (its own copy has landed; the barrier publishes the zeros).
        if (!bcastA && leftA < (chunksA << 4) && tid == ((chunksA - 1u) & (uint32_t)(STAGE_BLOCK - 1)))
            for (uint32_t b = leftA; b < (chunksA << 4); ++b) { STRSIM_CHECK_INDEX(K_STAGE, 32, row0, b, COLB); s_bytes[b] = 0; }
        if (!bcastB && leftB < (chunksB << 4) && tid == ((chunksB - 1u) & (uint32_t)(STAGE_BLOCK - 1)))
            for (uint32_t b = leftB; b < (chunksB << 4); ++b) { STRSIM_CHECK_INDEX(K_STAGE, 33, row0, COLB + b, sizeof(s_bytes)); s_bytes[COLB + b] = 0; }
        lds_barrier();
        STAGE_STAMP(5);
        if (tid < 32u) s_cnt[tid] = 0u; // read by sortB above, next used by sortA behind barrier G
        // ---- E: offsets DMA(j+1), in flight during the rounds
        uint64_t next_row0 = row0 + rows;
        if (next_row0 >= row_end) { // this range is done: move to the next one and ask for the one after it
            const uint64_t lo = uniform(s_sched[0]), hi = lo + uniform(s_sched[1]); // (uniform: row counters live in SGPRs)
            next_row0 = lo * 64u < n ? lo * 64u : n;
            row_end = hi * 64u < n ? hi * 64u : n;
            if (tid == 0u) grab((uint32_t)lo, grab_lo, grab_sz);
            grab_pending = true;
        }
        if (next_row0 < row_end) dma_offsets(next_row0);
        STAGE_STAMP(6);
        // ---- F: rounds(j), 64 rows of the length order each
        __builtin_amdgcn_s_setprio(0); // the column loops yield to waves that have copies to issue or results to store
        {
            // the t-th round from the LONG end goes to wave t, 2W-1-t, 2W+t, ... (long + short = balanced also when the block
            // holds fewer than B rows of this kernel: cfg3 5.1 rounds per block, 2.70 -> 2.45 ms), and the rounds are cut from
            // the long end too: the one that is not full holds the block's shortest texts, not its longest
            const uint32_t nrounds = (nmine + 63u) >> 6;
#pragma unroll 1
            for (int k = 0; k < STAGE_RPW; ++k) {
                const uint32_t t = (uint32_t)(k >> 1) * (2u * STAGE_WAVES) + ((k & 1) ? (uint32_t)(2 * STAGE_WAVES - 1) - wv : wv);
                if (t >= nrounds) continue;
                const uint32_t hi = nmine - 64u * t;                // positions of the length order below this round's end
                const uint32_t first = hi >= 64u ? hi - 64u : 0u;   // its first position
                const uint32_t in_round = hi - first;               // (>= 1)
                __builtin_amdgcn_s_setprio(1); // descriptor + window fetch ahead of the waves that grind columns
                STRSIM_CHECK_INDEX(K_STAGE, 41, row0, first + lane, B);
                const uint2 d = *desc_at(first + lane);
                uint32_t wt[8], wp[8];
                stage_window(s_bytes, d.x & 0xFFFFu, wt, (uint32_t)sizeof(s_bytes), STRSIM_BOUNDS_ON ? 1u : 0u);
                stage_window(s_bytes, d.x >> 16, wp, (uint32_t)sizeof(s_bytes), STRSIM_BOUNDS_ON ? 1u : 0u);
    #if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    #endif
                STAGE_STAMP(7);
                __builtin_amdgcn_s_setprio(0);
                stage_compute<MEASURE, LUT>(lut, wt, wp, lane < in_round ? d.y : STAGE_DEAD, in_round - 1u, s_code, s_word, s_val);
                STAGE_STAMP(8);
            }
        }
        // ---- G: the rounds are done: the staging area, the descriptors and the result codes change hands
        __builtin_amdgcn_s_setprio(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        STAGE_STAMP(9);
        prev_row0 = row0;
        prev_rows = rows;
        row0 = next_row0;
    }
    if (prev_rows) store_block(prev_row0, prev_rows);
    if (publish) lds_barrier(); // (uniform) every wave's contribution to s_left is in
    if (tid == 0u) {
        // the last workgroup to leave clears the counters for the next launch on this slot; for an eager (small) call it
        // also tells the host how many rows are left for the kernels behind this one -- usually none, and then the host
        // launches nothing else (one launch instead of five)
        if (publish && s_left) __hip_atomic_fetch_add(&sched[2], s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t done = __hip_atomic_fetch_add(&sched[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1u) {
            if (publish) {
                const uint32_t total = __hip_atomic_load(&sched[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&publish->lane_left, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                // a call that consists of this kernel alone (strsim_capi.cpp: no slow rows expected) is complete with it: its
                // ticket tells the host so (every result store of every workgroup happened before that workgroup's increment
                // of sched[1], which this thread has acquired)
                if (ticket) __hip_atomic_store(&publish->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(&sched[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sched[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sched[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
    if (lane == 0u) {
        const uint32_t w = (blockIdx.x * STAGE_WAVES + wv) & 16383u;
        for (int q = 0; q < 10; ++q) g_stage_stamps[w][q] = st_acc[q];
        g_stage_stamps[w][10] = __builtin_amdgcn_s_memtime() - st_t0;
        const unsigned long long st_r1 = __builtin_amdgcn_s_memrealtime();
        g_stage_stamps[w][11] = st_r1 - st_r0;
        g_stage_stamps[w][12] = st_r0;
        g_stage_stamps[w][13] = st_r1;
        g_stage_stamps[w][14] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        g_stage_stamps[w][15] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    }
#endif
}

// workgroups per CU (= waves per SIMD): 5 by LDS (28 KB) and registers (<= 96) without match-mask tables, 4 with them (40 KB;
// Jaro's two passes want the registers anyway), 3 for the five-output pass (three cores' state at once)
#ifndef STRSIM_STAGE_PLAIN_WAVES
#define STRSIM_STAGE_PLAIN_WAVES 5 // workgroups per CU of the instantiations without tables (6 needs <= 26.6 KB of LDS and 80 registers: measured)
#endif
#ifndef STRSIM_STAGE_ALL_WAVES_PER_EU
#define STRSIM_STAGE_ALL_WAVES_PER_EU 3
#endif
template <int MEASURE, bool TABLES, bool LONG = false> constexpr int stage_waves_per_eu()
{
    constexpr int lim = MEASURE == ALL_MEASURES ? STRSIM_STAGE_ALL_WAVES_PER_EU : ((TABLES || LONG) ? 4 : STRSIM_STAGE_PLAIN_WAVES);
    return STRSIM_STAGE_WAVES_PER_EU < lim ? STRSIM_STAGE_WAVES_PER_EU : lim;
}

// TABLES: match masks from the LDS tables (a measure whose bit is set in STRSIM_STAGE_LUT); LONG: the long-row geometry
// (StageGeom), without tables -- what the kernel does on such frames is mostly staging.
template <int MEASURE, bool TABLES, bool LONG = false>
__global__ __launch_bounds__(STAGE_BLOCK) __attribute__((amdgpu_waves_per_eu(stage_waves_per_eu<MEASURE, TABLES, LONG>()))) void
k_lane_stage(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, uint64_t rowsA,
             const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB, uint64_t rowsB, OutPtrs outs, uint64_t n,
             unsigned long long *__restrict__ slowmask, DevStatus *__restrict__ status, const double *__restrict__ qtab,
             uint32_t *__restrict__ sched, DevStatus *__restrict__ publish, uint32_t ticket)
{
    lane_stage_body<MEASURE, TABLES, LONG>(offA, valA, rowsA, offB, valB, rowsB, outs, n, slowmask, status, qtab, sched, publish, ticket);
}

// The five-output instantiation keeps the matching state of three cores alive at once and stages 64 bits per row.
__global__ __launch_bounds__(STAGE_BLOCK) __attribute__((amdgpu_waves_per_eu(stage_waves_per_eu<ALL_MEASURES, stage_uses_lut<ALL_MEASURES>()>()))) void
k_lane_stage_all(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, uint64_t rowsA,
                 const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB, uint64_t rowsB, OutPtrs outs, uint64_t n,
                 unsigned long long *__restrict__ slowmask, DevStatus *__restrict__ status, const double *__restrict__ qtab,
                 uint32_t *__restrict__ sched, DevStatus *__restrict__ publish, uint32_t ticket)
{
    lane_stage_body<ALL_MEASURES, stage_uses_lut<ALL_MEASURES>()>(offA, valA, rowsA, offB, valB, rowsB, outs, n, slowmask, status, qtab, sched, publish, ticket);
}
