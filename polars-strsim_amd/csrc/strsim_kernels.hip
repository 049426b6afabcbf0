// strsim_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the five pairwise string-similarity
// measures of polars-strsim (reference src/expressions/strsim.rs:109-345).
//
// Device data layout (per column shard): uint32 offsets[rows+1] + packed UTF-8 bytes; output f64[n].
//
// Kernels, chained through one 64-bit "not finished yet" mask per 64 rows:
//
//   k_lane_stage<M>   ONE PAIR PER LANE, strings <= 32 ASCII bytes (the dominant kernel; strsim_lane_stage.h).  A persistent
//                     256-thread workgroup copies the contiguous byte range of a block of up to 512 rows into LDS with
//                     LDS-DMA, buckets the rows by DP column count and runs them as rounds of 64 rows of similar length:
//                     each lane picks its two 32-byte windows out of LDS, transposes one into bit-planes and runs the
//                     bit-parallel cores of strsim_lane_core.h in registers.  k_lane_stage_all: five outputs from one pass.
//   k_lane_lit<M>     a column against a Utf8 literal (strsim_lane_lit.h): the literal is the wave-uniform text.
//   k_lane_wide<M>    ONE PAIR PER LANE, 33..128 ASCII bytes: the same cores with masks of 1..4 words -- as wide as the PATTERN
//                     (strsim_lane_wide.h); the rows of a 16 384-row super sorted by (mask words, columns to run), a round's
//                     windows fetched by the wave together through its LDS rows, where the text stays.
//   k_lane_utf8<M>    ONE PAIR PER LANE, short non-ASCII strings (<= 32 scalar values, BMP): per-lane UTF-8 decode
//                     into 16-bit symbols in LDS, then the same cores on symbols (strsim_lane_sym.h).
//   k_wave_pairs<M>   the rest, up to WAVE_CAP bytes, any UTF-8, taken chunk by chunk from a work list k_lane_utf8 builds.
//                     Levenshtein: up to 16 pairs per wave in runs of lanes, block-parallel Myers (one 64-row block per
//                     lane, DPP hand-off), match masks from a per-lane LDS table, texts in a global arena; on bytes
//                     (ASCII) or on 16-bit scalar values (the reference works on `char`s, strsim.rs:133,189,297).
//                     Other measures: one pair per wave, scalar values in LDS, ballot matching / histograms.
//   k_huge_pairs<M>   strings beyond WAVE_CAP, scratch in global memory, launched from strsim_ctx_synchronize() only
//                     when such rows were counted; Levenshtein: the block step in stripes of 64 blocks, any length.
//
// No MFMA anywhere: this is integer/byte work whose roofline is HBM bytes.
// Built with -ffp-contract=off so the f64 epilogues are bit-identical to the Rust source.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <cstdlib>
#include <stdint.h>
#include <type_traits>

#include "strsim_bounds.h"
#include "strsim_lane_core.h"
#include "strsim_lane_lut.h"
#include "strsim_lane_wide.h"
#include "strsim_lane_sym.h"
#include "strsim_kernels.h"
#include "strsim_lane_common.h"

namespace strsim {

constexpr int ALL_MEASURES = 5; // MEASURE value of the fused five-output instantiation

struct OutPtrs {
    double *p[5]; // indexed by Measure; single-measure kernels use p[0]
};

#include "strsim_lane_stage.h"
#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_PC)
#include "strsim_lane_stage_pc.h" // lab only: the producer / consumer form of k_lane_stage (round 6's experiment)
#endif
#include "strsim_lane_lit.h"

#include "strsim_kernel_wide.h"
#include "strsim_kernel_utf8.h"
#include "strsim_kernel_wave.h"
#include "strsim_kernel_huge.h"

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// k_lane_wide / k_lane_utf8 geometry: spans (WIDE_SPAN / U8_SPAN mask words) per super-span and workgroups.  A super is what one
// workgroup tests for "anything to do" with a single barrier; full-size frames use the largest (8 spans), smaller
// ones shrink it until there are about two supers per RESIDENT workgroup (a.wide_grid = 3 per CU).  The launch
// itself may be much larger than what is resident (a.wide_grid_cap): workgroups that finish early are replaced by
// fresh ones instead of waiting for the longest grid-stride loop (cfg3: 12.4 -> 10.3 ms).
static void wide_geometry(const LaunchArgs &a, int span, uint32_t &sps, unsigned &grid)
{
    const uint64_t nchunks = (a.n + 63u) >> 6;
    const uint64_t nspans = (nchunks + (uint64_t)span - 1) / (uint64_t)span;
    const uint64_t want = 2u * (uint64_t)a.wide_grid;
    uint64_t k = nspans / (want ? want : 1u);
    if (k < 1u) k = 1u;
    if (k > (uint64_t)(WIDE_BLOCK / span)) k = WIDE_BLOCK / span;
    sps = (uint32_t)k;
    const uint64_t nsuper = (nspans + k - 1u) / k;
    grid = (unsigned)(nsuper < (uint64_t)a.wide_grid_cap ? nsuper : (uint64_t)a.wide_grid_cap);
}

// Persistent workgroups of k_lane_stage for a frame of nsb blocks: all that are resident when the frame is large; a frame
// of a few blocks per workgroup runs faster on FEWER workgroups -- the pipeline of a workgroup (offsets and bytes of the next
// block in flight behind the current one) only pays from its second block on, and every workgroup pays the prologue: a
// 1 M-row call takes 0.085 ms on 5 workgroups per CU and 0.051 ms on 2.  So: about four blocks per workgroup, but at least
// one workgroup per CU while there are blocks for them.
static uint64_t stage_launch_size(uint64_t nsb, uint64_t resident, uint64_t cus)
{
    constexpr uint64_t per = 4; // blocks per workgroup (round 3's sweep: 2 / 4 / 8 -> 0.051 / 0.046 / 0.049 ms per 1 M rows)
    uint64_t g = nsb / per;
    const uint64_t floor_ = nsb < cus ? nsb : cus;
    if (g < floor_) g = floor_;
    if (g > resident) g = resident;
    return g ? g : 1u;
}

template <int M>
static void launch_lane_t(const LaunchArgs &a)
{
#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STRIP)
    if (M == LEVENSHTEIN && a.rowsA != 1 && a.rowsB != 1) { // (lab) the strip lengths of every row, by a kernel of its own IN FRONT of the timed one
        static uint16_t *buf = nullptr;
        static uint64_t cap = 0;
        if (a.n > cap) { if (buf) (void)hipFree(buf); (void)hipMalloc((void **)&buf, a.n * 2 + 64); cap = a.n; }
        hipLaunchKernelGGL(k_strip_oracle, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, a.stream, a.offA, a.valA, a.offB, a.valB, a.n, buf);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_strip_ptr), &buf, sizeof buf, 0, hipMemcpyHostToDevice, a.stream);
    }
#endif
    if (a.ev_lane0) (void)hipEventRecord(a.ev_lane0, a.stream);
    OutPtrs op{};
    op.p[0] = a.out;
    // a column against a literal (strsim_lane_lit.h): the literal is the wave-uniform text, the column is staged -- on either
    // side, for every measure ([r5] Jaro / Jaro-Winkler included: they are symmetric, strsim_lane_core.h)
    const bool lit_a = a.rowsA == 1 && a.rowsB != 1, lit_b = a.rowsB == 1 && a.rowsA != 1;
    const bool lit_path = lit_a || lit_b;
    if (lit_path && !a.no_literal_path) {
        const bool litA = lit_a;
        const uint64_t nlb = (a.n + (LIT_ROWS - 1)) / LIT_ROWS;
        const uint64_t want = (uint64_t)a.stage_grid * 3u; // ~15 workgroups per CU in the launch, 5 resident
        const uint64_t gl = nlb < want ? nlb : want;
        hipLaunchKernelGGL((k_lane_lit<M>), dim3((unsigned)gl), dim3(LIT_BLOCK), 0, a.stream, litA ? a.offB : a.offA,
                           litA ? a.valB : a.valA, litA ? a.offA : a.offB, litA ? a.valA : a.valB, a.out, a.n, a.slowmask,
                           a.status, a.qtab, a.sched);
        if (a.publish_host) hipLaunchKernelGGL(k_publish_lit, dim3(1), dim3(64), 0, a.stream, a.sched, a.publish_host, a.publish_ticket);
    } else {
        // bytes staged through LDS (strsim_lane_stage.h): persistent workgroups, one per resident slot
        const uint64_t nsb = (a.n + (STAGE_ROWS - 1)) / STAGE_ROWS;
        // (a.stage_grid counts STRSIM_STAGE_WAVES_PER_EU workgroups per CU; an instantiation that runs fewer gets its share)
        auto go = [&](auto tables, auto long_rows) {
            constexpr bool T = decltype(tables)::value, L = decltype(long_rows)::value;
            const uint64_t res = (uint64_t)a.stage_grid * (uint64_t)stage_waves_per_eu<M, T, L>() / (uint64_t)STRSIM_STAGE_WAVES_PER_EU;
            const uint64_t gs = stage_launch_size(nsb, res, (uint64_t)a.stage_grid / (uint64_t)STRSIM_STAGE_WAVES_PER_EU);
            hipLaunchKernelGGL((k_lane_stage<M, T, L>), dim3((unsigned)gs), dim3(STAGE_BLOCK), 0, a.stream, a.offA, a.valA, a.rowsA,
                               a.offB, a.valB, a.rowsB, op, a.n, a.slowmask, a.status, a.qtab, a.sched, a.publish_host, a.publish_ticket);
        };
#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_PC)
        if (((STRSIM_STAGE_PC >> M) & 1) != 0 && a.rowsA != 1 && a.rowsB != 1 && !a.long_rows) {
            const uint64_t cus = (uint64_t)a.stage_grid / (uint64_t)STRSIM_STAGE_WAVES_PER_EU;
            const uint64_t gs = stage_launch_size(nsb, cus * (uint64_t)STRSIM_PC_WG_PER_CU, cus);
            hipLaunchKernelGGL((k_lane_stage_pc<M, STRSIM_PC_LUT != 0>), dim3((unsigned)gs), dim3(PC_THREADS), 0, a.stream, a.offA, a.valA, a.rowsA,
                               a.offB, a.valB, a.rowsB, op, a.n, a.slowmask, a.status, a.qtab, a.sched, a.publish_host, a.publish_ticket);
        } else
#endif
        if (a.long_rows) go(std::false_type{}, std::true_type{});
        else if constexpr (stage_uses_lut<M>()) go(std::true_type{}, std::false_type{});
        else go(std::false_type{}, std::false_type{});
    }
    if (a.ev_lane1) (void)hipEventRecord(a.ev_lane1, a.stream);
}

template <int M>
static void launch_slow_t(const LaunchArgs &a)
{
    const uint64_t nchunks = (a.n + 63u) >> 6;
    // waves per CU: Levenshtein by its 8 KB table; Jaro by its 12.4 KB of LDS (12); Jaccard / Dice 16 (a.wave_grid)
    const uint64_t wg = M == LEVENSHTEIN ? (uint64_t)a.wave_grid_lev
                        : (M == JARO || M == JARO_WINKLER) ? (uint64_t)a.wave_grid * 3u / 4u : (uint64_t)a.wave_grid;
    const uint64_t g2 = nchunks < wg ? nchunks : wg;
    {
        uint32_t sps, sps8;
        unsigned g3, g8;
        wide_geometry(a, WIDE_SPAN, sps, g3);
        wide_geometry(a, U8_SPAN, sps8, g8);
        hipLaunchKernelGGL((k_lane_wide<M>), dim3(g3), dim3(WIDE_BLOCK), 0, a.stream, a.offA, a.valA, a.rowsA,
                           a.offB, a.valB, a.rowsB, a.out, a.n, a.slowmask, sps);
        hipLaunchKernelGGL((k_lane_utf8<M>), dim3(g8), dim3(WIDE_BLOCK), 0, a.stream, a.offA, a.valA, a.rowsA,
                           a.offB, a.valB, a.rowsB, a.out, a.n, a.slowmask, sps8, a.worklist, a.status);
    }
    hipLaunchKernelGGL((k_wave_pairs<M>), dim3((unsigned)g2), dim3(64), 0, a.stream, a.offA, a.valA, a.rowsA, a.offB,
                       a.valB, a.rowsB, a.out, a.n, a.slowmask, a.worklist, a.status, a.lev_ws);
    if (a.ev_wave1) (void)hipEventRecord(a.ev_wave1, a.stream);
}


template <int M>
static void launch_pair(const LaunchArgs &a)
{
    launch_lane_t<M>(a);
    launch_slow_t<M>(a);
}

template <int M>
static void launch_huge_t(const LaunchArgs &a, uint32_t *ws, uint32_t cap, int grid)
{
    hipLaunchKernelGGL((k_huge_pairs<M>), dim3((unsigned)grid), dim3(64), 0, a.stream, a.offA, a.valA, a.rowsA, a.offB,
                       a.valB, a.rowsB, a.out, a.n, ws, cap);
}

hipError_t launch_huge(int measure, const LaunchArgs &a, uint32_t *ws, uint32_t cap, int grid)
{
    switch (measure) {
    case LEVENSHTEIN: launch_huge_t<LEVENSHTEIN>(a, ws, cap, grid); break;
    case JARO: launch_huge_t<JARO>(a, ws, cap, grid); break;
    case JARO_WINKLER: launch_huge_t<JARO_WINKLER>(a, ws, cap, grid); break;
    case JACCARD: launch_huge_t<JACCARD>(a, ws, cap, grid); break;
    case SORENSEN_DICE: launch_huge_t<SORENSEN_DICE>(a, ws, cap, grid); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int M>
static void launch_slow_kernels(const LaunchArgs &a, double *out)
{
    const uint64_t nchunks = (a.n + 63u) >> 6;
    uint32_t sps, sps8;
    unsigned g3, g8;
    wide_geometry(a, WIDE_SPAN, sps, g3);
    wide_geometry(a, U8_SPAN, sps8, g8);
    // waves per CU: Levenshtein by its 8 KB table; Jaro by its 12.4 KB of LDS (12); Jaccard / Dice 16 (a.wave_grid)
    const uint64_t wg = M == LEVENSHTEIN ? (uint64_t)a.wave_grid_lev
                        : (M == JARO || M == JARO_WINKLER) ? (uint64_t)a.wave_grid * 3u / 4u : (uint64_t)a.wave_grid;
    const uint64_t g2 = nchunks < wg ? nchunks : wg;
    hipLaunchKernelGGL((k_lane_wide<M>), dim3(g3), dim3(WIDE_BLOCK), 0, a.stream, a.offA, a.valA, a.rowsA, a.offB,
                       a.valB, a.rowsB, out, a.n, a.slowmask, sps);
    hipLaunchKernelGGL((k_lane_utf8<M>), dim3(g8), dim3(WIDE_BLOCK), 0, a.stream, a.offA, a.valA, a.rowsA, a.offB,
                       a.valB, a.rowsB, out, a.n, a.slowmask, sps8, a.worklist, a.status);
    hipLaunchKernelGGL((k_wave_pairs<M>), dim3((unsigned)g2), dim3(64), 0, a.stream, a.offA, a.valA, a.rowsA, a.offB,
                       a.valB, a.rowsB, out, a.n, a.slowmask, a.worklist, a.status, a.lev_ws);
}

// All five measures of one frame: one fused lane kernel (five outputs), then the slow-row kernels per measure,
// each starting from the same mask (k_lane_wide clears what it finishes, so the mask is restored in between).
hipError_t launch_lane_all_only(const LaunchArgs &a, double *const outs[5])
{
    if (a.n == 0) return hipSuccess;
    OutPtrs op{};
    for (int q = 0; q < 5; ++q) op.p[q] = outs[q];
    if (a.ev_lane0) (void)hipEventRecord(a.ev_lane0, a.stream);
    {
        // one staged pass, five outputs (strsim_lane_stage.h, MEASURE = ALL_MEASURES)
        const uint64_t nsb = (a.n + (STAGE_ROWS - 1)) / STAGE_ROWS;
        const uint64_t cap = (uint64_t)a.stage_grid * (uint64_t)stage_waves_per_eu<ALL_MEASURES, stage_uses_lut<ALL_MEASURES>()>() / (uint64_t)STRSIM_STAGE_WAVES_PER_EU; // resident workgroups
        const uint64_t gs = stage_launch_size(nsb, cap, (uint64_t)a.stage_grid / (uint64_t)STRSIM_STAGE_WAVES_PER_EU);
        hipLaunchKernelGGL(k_lane_stage_all, dim3((unsigned)gs), dim3(STAGE_BLOCK), 0, a.stream, a.offA, a.valA,
                           a.rowsA, a.offB, a.valB, a.rowsB, op, a.n, a.slowmask, a.status, a.qtab, a.sched, a.publish_host,
                           a.publish_ticket);
    }
    if (a.ev_lane1) (void)hipEventRecord(a.ev_lane1, a.stream);
    return hipGetLastError();
}

hipError_t launch_slow_all_only(const LaunchArgs &a, double *const outs[5], unsigned long long *mask_backup)
{
    if (a.n == 0) return hipSuccess;
    const uint64_t nchunks = (a.n + 63u) >> 6;
    hipError_t e = hipMemcpyAsync(mask_backup, a.slowmask, nchunks * sizeof(unsigned long long), hipMemcpyDeviceToDevice, a.stream);
    if (e != hipSuccess) return e;
    launch_slow_kernels<LEVENSHTEIN>(a, outs[LEVENSHTEIN]);
    for (int m = 1; m < 5; ++m) {
        e = hipMemcpyAsync(a.slowmask, mask_backup, nchunks * sizeof(unsigned long long), hipMemcpyDeviceToDevice, a.stream);
        if (e != hipSuccess) return e;
        switch (m) {
        case JARO: launch_slow_kernels<JARO>(a, outs[m]); break;
        case JARO_WINKLER: launch_slow_kernels<JARO_WINKLER>(a, outs[m]); break;
        case JACCARD: launch_slow_kernels<JACCARD>(a, outs[m]); break;
        default: launch_slow_kernels<SORENSEN_DICE>(a, outs[m]); break;
        }
    }
    if (a.ev_wave1) (void)hipEventRecord(a.ev_wave1, a.stream);
    return hipGetLastError();
}

hipError_t launch_pairs_all(const LaunchArgs &a, double *const outs[5], unsigned long long *mask_backup)
{
    hipError_t e = launch_lane_all_only(a, outs);
    if (e != hipSuccess) return e;
    return launch_slow_all_only(a, outs, mask_backup);
}

// waves of k_wave_pairs<levenshtein> a CU holds at once (its grid is persistent: a wave that is not resident would start on
// its static share of the work list only when another one leaves); 0 if the runtime cannot tell
int wave_lev_resident_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wave_pairs<LEVENSHTEIN>, 64, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int lane_kernel_launches(int measure, const LaunchArgs &a)
{
    if (measure == 5) return 1;
    const bool lit_a = a.rowsA == 1 && a.rowsB != 1, lit_b = a.rowsB == 1 && a.rowsA != 1;
    const bool lit_path = lit_a || lit_b;
    return (lit_path && !a.no_literal_path) ? 2 : 1; // k_lane_lit + k_publish_lit
}

hipError_t launch_lane_only(int measure, const LaunchArgs &a)
{
    if (a.n == 0) return hipSuccess;
    switch (measure) {
    case LEVENSHTEIN: launch_lane_t<LEVENSHTEIN>(a); break;
    case JARO: launch_lane_t<JARO>(a); break;
    case JARO_WINKLER: launch_lane_t<JARO_WINKLER>(a); break;
    case JACCARD: launch_lane_t<JACCARD>(a); break;
    case SORENSEN_DICE: launch_lane_t<SORENSEN_DICE>(a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_slow_only(int measure, const LaunchArgs &a)
{
    if (a.n == 0) return hipSuccess;
    switch (measure) {
    case LEVENSHTEIN: launch_slow_t<LEVENSHTEIN>(a); break;
    case JARO: launch_slow_t<JARO>(a); break;
    case JARO_WINKLER: launch_slow_t<JARO_WINKLER>(a); break;
    case JACCARD: launch_slow_t<JACCARD>(a); break;
    case SORENSEN_DICE: launch_slow_t<SORENSEN_DICE>(a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pairs(int measure, const LaunchArgs &a)
{
    if (a.n == 0) return hipSuccess;
    switch (measure) {
    case LEVENSHTEIN: launch_pair<LEVENSHTEIN>(a); break;
    case JARO: launch_pair<JARO>(a); break;
    case JARO_WINKLER: launch_pair<JARO_WINKLER>(a); break;
    case JACCARD: launch_pair<JACCARD>(a); break;
    case SORENSEN_DICE: launch_pair<SORENSEN_DICE>(a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// The call's status block, copied to pinned host memory by a one-wave kernel: an SDMA/blit copy behind the last kernel costs
// a queue hand-over of 10-16 us per call, this costs ~2 us (visible to the host once the stream has been synchronised).
__global__ void k_publish_status(const DevStatus *__restrict__ src, DevStatus *__restrict__ dst, uint32_t ticket)
{
    const unsigned *s = reinterpret_cast<const unsigned *>(src);
    unsigned *d = reinterpret_cast<unsigned *>(dst);
    // (lane_left belongs to k_lane_stage's last workgroup, which writes the host copy itself; the ticket goes last)
    constexpr unsigned TICKET_WORD = offsetof(DevStatus, ticket) / 4, LEFT_WORD = offsetof(DevStatus, lane_left) / 4;
    if (threadIdx.x < sizeof(DevStatus) / 4 && threadIdx.x != TICKET_WORD && threadIdx.x != LEFT_WORD)
        __builtin_nontemporal_store(s[threadIdx.x], d + threadIdx.x);
    __threadfence_system();
    // the ticket goes last: a host that sees it sees the block (one wave: the stores above were issued before this one)
    if (threadIdx.x == 0u) __hip_atomic_store(&dst->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_publish_status(const DevStatus *src, DevStatus *dst_mapped, uint32_t ticket, hipStream_t stream)
{
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(64), 0, stream, src, dst_mapped, ticket);
    return hipGetLastError();
}

} // namespace strsim


#if STRSIM_BOUNDS_ON
// lab build only (EXTRA="-DSTRSIM_LAB -DSTRSIM_BOUNDS"): the address-check record of this translation unit's kernels -> dst[0..5]
// (hits, kernel << 32 | site, row, value, lo, hi), and cleared.  Call after the stream has been synchronised.
extern "C" __attribute__((visibility("default"))) int strsim_debug_bounds_kernels(unsigned long long *dst)
{
    strsim::bounds::Record r{};
    hipError_t e = hipMemcpyFromSymbol(&r, HIP_SYMBOL(strsim::bounds::g_rec), sizeof(r), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    dst[0] = r.hits; dst[1] = r.site; dst[2] = r.row; dst[3] = r.value; dst[4] = r.lo; dst[5] = r.hi;
    const strsim::bounds::Record z{};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(strsim::bounds::g_rec), &z, sizeof(z), 0, hipMemcpyHostToDevice);
}
#endif

#if defined(STRSIM_LAB) && defined(STRSIM_WIDE_STAMPS)
// lab build only: copies the per-wave phase cycle sums of the last k_lane_wide launch to the host
extern "C" __attribute__((visibility("default"))) int strsim_debug_wide_stamps(unsigned long long *dst, size_t waves)
{
    if (waves > 16384) waves = 16384;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(strsim::g_wide_stamps), waves * 16 * sizeof(unsigned long long), 0,
                                    hipMemcpyDeviceToHost);
}
#endif

#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
// lab build only: copies the per-wave phase cycle sums of the last k_lane_stage launch to the host
extern "C" __attribute__((visibility("default"))) int strsim_debug_stage_stamps(unsigned long long *dst, size_t waves)
{
    if (waves > 16384) waves = 16384;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(strsim::g_stage_stamps), waves * 16 * sizeof(unsigned long long), 0,
                                    hipMemcpyDeviceToHost);
}
#endif
