// strsim_kernels.h -- launch interface between the C ABI (strsim_capi.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "strsim_lane_core.h" // QTAB_N

namespace strsim {

constexpr int WAVE_CAP = 1024; // wave-per-pair kernels: max bytes (hence scalar values) per string

// k_wave_pairs<LEVENSHTEIN>: up to LEV_JOBS pairs are advanced together by one wave.  Its per-wave global scratch: the
// three scalar-value arrays of the fallback, then the text arena of a BYTES batch (ASCII texts as bytes), the text
// arena of a SYMBOLS batch (16-bit scalar values) and the SYMBOLS patterns (32 values per lane); a staged text has
// TXT_PAD units in front of and behind it.
#ifndef STRSIM_LEV_JOBS
#define STRSIM_LEV_JOBS 16
#endif
constexpr int LEV_JOBS = STRSIM_LEV_JOBS;
#ifndef STRSIM_LEV_BYTES_ROWS
#define STRSIM_LEV_BYTES_ROWS 64 // pattern rows per lane of an ASCII batch of k_wave_pairs<levenshtein>: 64 (two mask words) or 32
#endif
constexpr int LEV_BYTES_ROWS = STRSIM_LEV_BYTES_ROWS;
#ifndef STRSIM_LEV_POOL
#define STRSIM_LEV_POOL 384 // rows ranked together before they are dealt into batches, at most (a multiple of 64; 16 bits of LDS per row)
#endif
constexpr int LEV_POOL = STRSIM_LEV_POOL;
constexpr int TXT_PAD = 48;
constexpr int ARENA0_BYTES = 8192;
constexpr int ARENA1_BYTES = 12288;
constexpr int PAT1_BYTES = 64 * 32 * 2;
constexpr int LEV_WS_WORDS = 3 * (WAVE_CAP + 64) + (ARENA0_BYTES + ARENA1_BYTES + PAT1_BYTES) / 4; // per wave

// k_huge_pairs: words of global workspace per wave for strings of up to `cap` bytes: a front pad, the two arrays of
// scalar values, and an auxiliary array four times as long (hash table of the multiset intersection: 2 (cap + 64)
// 64-bit entries; Jaro flags; Levenshtein hand-off bytes)
#define HUGE_WS_WORDS(cap) (6u * ((uint64_t)(cap) + 64u) + 64u)

struct DevStatus {
    unsigned int wave_rows; // rows finished by k_wave_pairs
    unsigned int huge_rows; // rows longer than WAVE_CAP (left for the long-string pass)
    unsigned int max_len;   // longest such string, bytes
    unsigned int lane_left; // rows the one-pair-per-lane kernel left to the kernels behind it (written for eager calls only)
    // per measure (the fused five-measure call runs the slow-row kernels once per measure):
    unsigned int list_count[5]; // k_lane_utf8: chunks of 64 rows that still hold unfinished rows ...
    unsigned int list_rows[5];  // ... and how many rows that is
    unsigned int next_entry[5]; // k_wave_pairs: work-list entries handed out beyond the first static round
    unsigned int pad1[12];
    unsigned int ticket;    // host-mapped copy only: the call's ticket, written LAST (release, system scope) by whoever publishes
                            // the block -- strsim_ctx_retire_oldest() refuses a slot whose ticket has not arrived
};
static_assert(sizeof(DevStatus) == 128, "DevStatus is 128 bytes");

struct LaunchArgs {
    const uint32_t *offA; const uint8_t *valA; uint64_t rowsA;
    const uint32_t *offB; const uint8_t *valB; uint64_t rowsB;
    double *out; uint64_t n;
    unsigned long long *slowmask; // one 64-bit mask per 64-row chunk
    uint32_t *worklist;           // ceil(n/64) words: the non-empty chunks, compacted by k_lane_utf8
    const double *qtab;           // QTAB_N x QTAB_N integer quotients a / b for the epilogues of the one-pair-per-lane kernels (device)
    DevStatus *status;            // cleared by the first kernel of the call
    uint32_t *sched;              // k_lane_stage: four zeroed words (range counter, finished workgroups, rows left, -); left zeroed by the kernel
    DevStatus *publish_host;      // the last workgroup of k_lane_stage writes lane_left there (host-mapped), else nullptr
    uint32_t publish_ticket;      // != 0: ... and then this ticket: the call consists of that kernel alone
    hipStream_t stream;
    int wide_grid, wave_grid;     // resident workgroups of k_lane_wide / k_lane_utf8, waves of k_wave_pairs
    bool no_literal_path;         // A/B runs: a literal call takes k_lane_stage like any other
    bool long_rows;               // the context's last call left many rows behind k_lane_stage: take its instantiation without tables
    int stage_grid;               // k_lane_stage: persistent workgroups at STRSIM_STAGE_WAVES_PER_EU per CU
    int wide_grid_cap;            // k_lane_wide / k_lane_utf8: launch size limit (wide_grid = what is resident)
    int wave_grid_lev;            // k_wave_pairs<LEVENSHTEIN> (LDS-light: more waves per CU)
    uint32_t *lev_ws;             // its global scratch: wave_grid_lev * LEV_WS_WORDS words
    hipEvent_t ev_lane0, ev_lane1, ev_wave1; // optional (nullptr = no timing)
};

hipError_t launch_pairs(int measure, const LaunchArgs &a);
// the two halves of launch_pairs for a call that looks at lane_left in between (small calls: usually nothing is left)
hipError_t launch_lane_only(int measure, const LaunchArgs &a);
hipError_t launch_slow_only(int measure, const LaunchArgs &a);
// the same two halves of launch_pairs_all
hipError_t launch_lane_all_only(const LaunchArgs &a, double *const outs[5]);
hipError_t launch_slow_all_only(const LaunchArgs &a, double *const outs[5], unsigned long long *mask_backup);
// (every first kernel of a call -- k_lane_stage, k_lane_stage_all, k_lane_lit + k_publish_lit -- reports lane_left and the ticket)
// launches the first kernel of such a call takes: 1, or 2 when a literal takes k_lane_lit (+ k_publish_lit behind it)
int lane_kernel_launches(int measure, const LaunchArgs &a);
int wave_lev_resident_per_cu();

// All five measures in one go (a.out unused): outs[] indexed by measure id; mask_backup = ceil(n/64) words of scratch.
hipError_t launch_pairs_all(const LaunchArgs &a, double *const outs[5], unsigned long long *mask_backup);

// Second pass for rows with a string longer than WAVE_CAP bytes: `grid` waves, each with HUGE_WS_WORDS(cap) words of `ws`.
hipError_t launch_huge(int measure, const LaunchArgs &a, uint32_t *ws, uint32_t cap, int grid);
// copies one DevStatus to host-mapped pinned memory from the device side (no copy-engine hand-over)
hipError_t launch_publish_status(const DevStatus *src, DevStatus *dst_mapped, uint32_t ticket, hipStream_t stream);

} // namespace strsim
