// strsim_bins.h -- the BINNED path for rows of 33..128 ASCII bytes: geometry, the device-side bin table, and the three small
// kernels that lay the bins out before k_lane_stage runs.  Included by strsim_kernels.hip inside namespace strsim.
//
// Why: a frame of long strings (BASELINE config 3: Zipf 4..128 bytes, a third of the rows beyond 32 bytes) used to have
// k_lane_wide find those rows again through the mask: a second pass over the offsets, per-lane scattered window loads out of the
// columns (4.6x the rows' bytes fetched), a sort of whatever 16 384 consecutive rows hold -- rounds that use 91 % of the
// column-words they run -- and one set of registers / LDS for every mask width.  Here k_lane_stage, which has the bytes of every
// row in LDS anyway, hands the rows it does not finish over in the form the wide kernels want:
//
//   bin      = (text columns in steps of 4) x (pattern length in steps of 16 bytes): every row of a bin runs the same number of
//              column groups at the same mask width with the same slot sizes -- no sort, no idle columns beyond the rounding;
//   page     = 64 rows of one bin = one round of one wave of k_wide_bins: 64 RECORDS of 16 bytes (row, where its text starts,
//              where its pattern starts, the two lengths within the bin's ranges) -- 1 KB, one coalesced load per round.  The
//              strings themselves stay where they are: the reader's lanes fetch their own rows' pieces from the columns while
//              the round in front computes.  (Pages that carried the rows' BYTES -- slots filled out of k_lane_stage's staging
//              area -- were built first and measured, profiles/r4_cfg3_fullcopy_bins_summary.txt: the reader's side was all one
//              could wish for, but 36 M rows x 9 pieces are 313 M scattered 16-byte stores on the writer's side: 3.1e9 busy
//              cycles of the address units per launch, six times what the kernel spends otherwise, 10.8 GB written for 4.9 GB
//              of pages, k_lane_stage 2.65 -> 6.0 ms.  A GPU gathers well and scatters badly.)
//   position = a row's place in its bin is known before k_lane_stage starts: k_bin_hist counts the rows of every bin per GROUP of
//              2 048 rows (lengths only: one pass over the offsets), k_bin_top / k_bin_base turn the counts into exclusive
//              prefixes, and a workgroup of k_lane_stage that takes a group gets the group's first position in every bin -- no
//              device-wide atomics, no over-allocation, every record of a bin accounted for (a row k_wide_bins finds non-ASCII
//              gets its mask bit back from there -- the code-point kernels run behind it).
//
// Reference semantics are untouched: which rows are binned depends on byte lengths only, and a binned row's result comes from
// the same W-word cores (strsim_lane_wide.h; strsim.rs:141-160, :200-237, :297-305, :333-341).
#pragma once

#include <stdint.h>

#include "strsim_lane_core.h"

namespace strsim {

constexpr int BIN_GROUP_ROWS = 2048;            // rows per group: the unit k_lane_stage takes from its range counter in binned mode
constexpr int BIN_GROUP_CHUNKS = BIN_GROUP_ROWS / 64;
constexpr int BIN_COUNT = 256;                  // (text bucket 0..31) * 8 + (pattern class 0..7)
constexpr int BIN_SEG_GROUPS = 64;              // groups per segment of the two-level prefix sum
constexpr int BIN_CLASSES = 3;                  // mask widths 2 / 3 / 4 words
constexpr uint32_t BIN_PAGE16 = 64;             // a page in 16-byte units: 64 records
constexpr uint32_t BIN_DEAD_ROW = 0xFFFFFFFFu;  // record word 0 of a row that was not handed over after all (it stays in the mask)
constexpr uint32_t BIN_REC_TEXT_IN_B = 0x80u;   // record word 3: the text is the row's string of column b (symmetric measures walk the shorter one)
static_assert(BIN_SEG_GROUPS % 16 == 0, "k_bin_base walks a segment sixteen groups at a time");

// A row is a CANDIDATE for the bins by its byte lengths alone (what k_lane_wide takes: the longer side 33..128, neither empty).
STRSIM_HD bool bin_candidate(uint32_t la8, uint32_t lb8)
{
    const uint32_t mx = la8 > lb8 ? la8 : lb8, mn = la8 < lb8 ? la8 : lb8;
    return mx > 32u && mx <= 128u && mn >= 1u;
}
// lt = bytes of the text (the string whose characters are the DP columns), lp = bytes of the pattern (the bit-planes): 1..128
STRSIM_HD uint32_t bin_of(uint32_t lt, uint32_t lp) { return (((lt - 1u) >> 2) << 3) | ((lp - 1u) >> 4); }
STRSIM_HD uint32_t bin_text_bucket(uint32_t bin) { return bin >> 3; }                 // the text has 4 tb + 1 .. 4 tb + 4 bytes
STRSIM_HD uint32_t bin_pat_class(uint32_t bin) { return bin & 7u; }                   // the pattern has 16 pc + 1 .. 16 pc + 16 bytes
STRSIM_HD uint32_t bin_text_pieces(uint32_t bin) { return (bin >> 5) + 1u; }          // 16-byte pieces that cover the text (1..8)
STRSIM_HD uint32_t bin_pat_pieces(uint32_t bin) { return (bin & 7u) + 1u; }           // ... the pattern
// mask words of the bin's rows: by the PATTERN (the masks are over pattern positions), at least two
STRSIM_HD uint32_t bin_words(uint32_t bin) { const uint32_t w = ((bin & 7u) >> 1) + 1u; return w < 2u ? 2u : w; }
STRSIM_HD uint32_t bin_class(uint32_t bin) { return bin_words(bin) - 2u; }
// the length byte of a record: lt = 4 tb + 1 + (b & 3), lp = 16 pc + 1 + ((b >> 2) & 15)
STRSIM_HD uint32_t bin_len_byte(uint32_t lt, uint32_t lp) { return ((lt - 1u) & 3u) | (((lp - 1u) & 15u) << 2); }

// What the three layout kernels leave for k_lane_stage and k_wide_bins (device memory, one per context).
struct BinTable {
    uint32_t count[BIN_COUNT];                    // candidate rows of the bin
    uint32_t base16[BIN_COUNT];                   // first page of the bin, 16-byte units from the start of the bins buffer
    uint32_t cls_bin[BIN_CLASSES][BIN_COUNT];     // per mask width: its bins, longest texts first
    uint32_t cls_cum[BIN_CLASSES][BIN_COUNT + 1]; // ... and the rounds (pages) in front of each
    uint32_t cls_nbins[BIN_CLASSES];
    uint32_t cls_rounds[BIN_CLASSES];
    uint32_t cls_next[BIN_CLASSES];               // k_wide_bins: next round to hand out (zeroed by k_bin_top)
    uint32_t enabled;                             // 0: the buffer cannot hold this frame's bins: nothing is binned in this call
    uint32_t total16;                             // what the frame's bins take, 16-byte units (saturating)
    uint32_t rows;                                // candidate rows of the frame
};

#if defined(__HIPCC__)

// ---- k_bin_hist: candidate rows per (group, bin); one workgroup per SEGMENT of BIN_SEG_GROUPS groups, which also leaves the
//      segment's column sums.  hist[g * 256 + b], seg[s * 256 + b].
template <bool SYMMETRIC>
__global__ __launch_bounds__(256) void k_bin_hist(const uint32_t *__restrict__ offA, const uint32_t *__restrict__ offB, uint64_t n,
                                                  uint32_t *__restrict__ hist, uint32_t *__restrict__ seg)
{
    __shared__ uint32_t s_h[BIN_COUNT];
    const uint32_t tid = threadIdx.x;
    const uint64_t ngroups = (n + BIN_GROUP_ROWS - 1) / BIN_GROUP_ROWS;
    const uint64_t g0 = (uint64_t)blockIdx.x * BIN_SEG_GROUPS;
    uint32_t acc = 0u;
    s_h[tid] = 0u;
    __syncthreads();
    static_assert(BIN_GROUP_ROWS == 8 * 256, "k_bin_hist: eight consecutive rows per thread");
    for (uint64_t g = g0; g < g0 + BIN_SEG_GROUPS && g < ngroups; ++g) {
        // thread t: rows 8 t .. 8 t + 7 of the group -- nine consecutive offsets per column, two 16-byte loads and one dword
        const uint64_t r0 = g * BIN_GROUP_ROWS + 8u * tid;
        if (r0 + 8u <= n && ((reinterpret_cast<uintptr_t>(offA + r0) | reinterpret_cast<uintptr_t>(offB + r0)) & 15u) == 0u) {
            const uint4 a0 = *reinterpret_cast<const uint4 *>(offA + r0), a1 = *reinterpret_cast<const uint4 *>(offA + r0 + 4);
            const uint4 b0 = *reinterpret_cast<const uint4 *>(offB + r0), b1 = *reinterpret_cast<const uint4 *>(offB + r0 + 4);
            const uint32_t oa[9] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, offA[r0 + 8]};
            const uint32_t ob[9] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, offB[r0 + 8]};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t la8 = oa[k + 1] - oa[k], lb8 = ob[k + 1] - ob[k];
                if (bin_candidate(la8, lb8)) {
                    const bool swap = SYMMETRIC && la8 > lb8; // symmetric measures walk the shorter string
                    atomicAdd(&s_h[bin_of(swap ? lb8 : la8, swap ? la8 : lb8)], 1u);
                }
            }
        } else {
            for (uint64_t row = r0; row < r0 + 8u && row < n; ++row) {
                const uint32_t la8 = offA[row + 1] - offA[row], lb8 = offB[row + 1] - offB[row];
                if (bin_candidate(la8, lb8)) {
                    const bool swap = SYMMETRIC && la8 > lb8;
                    atomicAdd(&s_h[bin_of(swap ? lb8 : la8, swap ? la8 : lb8)], 1u);
                }
            }
        }
        __syncthreads();
        const uint32_t v = s_h[tid];
        hist[g * BIN_COUNT + tid] = v;
        acc += v;
        s_h[tid] = 0u;
        __syncthreads();
    }
    seg[(uint64_t)blockIdx.x * BIN_COUNT + tid] = acc;
}

// ---- k_bin_top: one workgroup.  Segment sums -> exclusive prefixes (in place), bin totals -> the table: where each bin's pages
//      start, which bins each mask width runs and in which order (longest texts first: the work counter hands rounds out in that
//      order, so the long ones do not end up alone at the tail), and whether the buffer holds it all.
__global__ __launch_bounds__(256) void k_bin_top(uint32_t *__restrict__ seg, uint32_t nseg, BinTable *__restrict__ table,
                                                 uint32_t capacity16, uint32_t *__restrict__ status_total16)
{
    __shared__ uint32_t s_cnt[BIN_COUNT];
    const uint32_t b = threadIdx.x;
    uint32_t run = 0u;
    for (uint32_t s0 = 0; s0 < nseg; s0 += 16u) { // (sixteen independent loads in flight: the loop is all latency)
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = s0 + (uint32_t)j < nseg ? seg[(uint64_t)(s0 + (uint32_t)j) * BIN_COUNT + b] : 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (s0 + (uint32_t)j < nseg) seg[(uint64_t)(s0 + (uint32_t)j) * BIN_COUNT + b] = run;
            run += v[j];
        }
    }
    s_cnt[b] = run;
    table->count[b] = run;
    __syncthreads();
    if (b == 0u) {
        unsigned long long at16 = 0ull, rows = 0ull;
        for (uint32_t q = 0; q < (uint32_t)BIN_COUNT; ++q) {
            table->base16[q] = at16 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)at16;
            at16 += (unsigned long long)((s_cnt[q] + 63u) >> 6) * BIN_PAGE16;
            rows += s_cnt[q];
        }
        const uint32_t total16 = at16 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)at16;
        table->total16 = total16;
        table->rows = (uint32_t)rows;
        table->enabled = (at16 <= (unsigned long long)capacity16) ? 1u : 0u;
        if (status_total16) *status_total16 = total16;
        for (uint32_t c = 0; c < (uint32_t)BIN_CLASSES; ++c) {
            uint32_t k = 0u, cum = 0u;
            for (int tb = 31; tb >= 0; --tb) {
                for (uint32_t pc = 0; pc < 8u; ++pc) {
                    const uint32_t q = ((uint32_t)tb << 3) | pc;
                    if (bin_class(q) != c || s_cnt[q] == 0u) continue;
                    table->cls_bin[c][k] = q;
                    table->cls_cum[c][k] = cum;
                    cum += (s_cnt[q] + 63u) >> 6;
                    ++k;
                }
            }
            table->cls_cum[c][k] = cum;
            table->cls_nbins[c] = k;
            table->cls_rounds[c] = cum;
            table->cls_next[c] = 0u;
        }
    }
}

// ---- k_bin_base: hist -> each group's first position in every bin (exclusive prefix over the groups, in place); one workgroup
//      per segment, starting from the segment's prefix.
__global__ __launch_bounds__(256) void k_bin_base(uint32_t *__restrict__ hist, const uint32_t *__restrict__ seg, uint64_t ngroups)
{
    const uint32_t b = threadIdx.x;
    const uint64_t g0 = (uint64_t)blockIdx.x * BIN_SEG_GROUPS;
    uint32_t run = seg[(uint64_t)blockIdx.x * BIN_COUNT + b];
    for (uint64_t s0 = g0; s0 < g0 + BIN_SEG_GROUPS && s0 < ngroups; s0 += 16u) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = s0 + (uint64_t)j < ngroups ? hist[(s0 + (uint64_t)j) * BIN_COUNT + b] : 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (s0 + (uint64_t)j < ngroups) hist[(s0 + (uint64_t)j) * BIN_COUNT + b] = run;
            run += v[j];
        }
    }
}

// What k_lane_stage gets in binned mode.
struct BinArgs {
    const uint32_t *base;   // [group][bin]: the group's first position in the bin
    const BinTable *table;
    uint4 *buf;             // the bins buffer (pages of records)
};

#endif // __HIPCC__

} // namespace strsim
