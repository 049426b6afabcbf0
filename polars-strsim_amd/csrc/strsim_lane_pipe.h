// strsim_lane_pipe.h -- k_lane_pipe<M>: the software-pipelined form of the one-pair-per-lane kernel (strings of
// <= 32 ASCII bytes).  Included by strsim_kernels.hip inside namespace strsim, after its helpers.
//
// Same per-pair arithmetic as k_lane_pairs (strsim_lane_core.h; reference strsim.rs:125-162, :180-245, :257-272,
// :286-308, :322-345); what changes is WHEN memory is touched.  k_lane_pairs exposes three memory latencies per
// block of rows (offsets, first window, second window) and only hides them behind other workgroups; here a
// persistent workgroup walks its blocks as a three-stage pipeline and every load is issued at least one 64-pair
// round before its data is needed:
//
//   body j of the loop (blocks are PIPE_ROWS consecutive rows; block j of this workgroup is being computed):
//     [X]  store(j-1)   results of block j-1, staged in LDS by the rounds, leave as coalesced 8-byte stores
//          sortB(j+1)   bucket bases (wave scan of the counters) -> descriptors of block j+1 in length order
//     [Y]  loads(j+2)   the offsets of block j+2 into registers (consumed by sortA at the end of the body)
//          rounds(j)    each wave: PIPE_RPW rounds of 64 pairs of similar length; round k first issues the window
//                       loads of round k+1 (the last round of the block: round 0 of block j+1) into a second register
//                       set, then runs the bit-parallel cores on its own windows, which were loaded a round ago
//          sortA(j+2)   lengths -> bucket keys, ranks by LDS atomics
//     [X]
//   Two workgroup barriers per block, both LDS-only, neither with a load in front of it that was issued less than
//   a round earlier.
//
// LDS per workgroup (PIPE_ROWS = 1024): descriptors 2 x 8 KB, Levenshtein result codes 2 x 2 KB (or f64 results
// 2 x 8 KB), the 33 x 33 table of 1 - d/m (8.5 KB): 29 KB -> 5 workgroups per CU, the same 5 waves per SIMD the
// register budget (<= 96 VGPRs with the second window set) admits.
#pragma once

#ifndef STRSIM_PIPE_ROWS
#define STRSIM_PIPE_ROWS 1024
#endif
#ifndef STRSIM_PIPE_WAVES_PER_EU
#define STRSIM_PIPE_WAVES_PER_EU 5
#endif
#ifndef STRSIM_PIPE_BUCKET_SHIFT
#define STRSIM_PIPE_BUCKET_SHIFT 1
#endif

#ifdef STRSIM_PIPE_STAMPS
// diagnostic build only (never in the product library): per-wave cycle sums of the phases of k_lane_pipe, read back
// with strsim_debug_pipe_stamps().  [0] store+sortB [1] barrier Y [2] offset loads [3] window-load issue
// [4] cores [5] window wait + moves [6] sortA [7] barrier X [8] all (s_memtime) [9] all (s_memrealtime, 100 MHz)
__device__ unsigned long long g_pipe_stamps[16384][10];
#define PIPE_STAMP(cat) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                             __builtin_amdgcn_sched_barrier(0); st_acc[cat] += t_ - st_last; st_last = t_; } while (0)
#else
#define PIPE_STAMP(cat) do { } while (0)
#endif

constexpr int PIPE_BLOCK = 256;                     // threads per workgroup
constexpr int PIPE_WAVES = PIPE_BLOCK / 64;         // 4
constexpr int PIPE_ROWS = STRSIM_PIPE_ROWS;         // rows per block
constexpr int PIPE_RPT = PIPE_ROWS / PIPE_BLOCK;    // rows per thread in the coalesced phases (2 or 4)
constexpr int PIPE_NR = PIPE_ROWS / 64;             // rounds per block
constexpr int PIPE_RPW = PIPE_NR / PIPE_WAVES;      // rounds per wave and block (even: two register sets alternate)
constexpr int PIPE_BSH = STRSIM_PIPE_BUCKET_SHIFT;  // buckets of 2^BSH column counts
constexpr int PIPE_NBK = (32 >> PIPE_BSH) + 1;      // + one for the rows this kernel leaves to the later ones
static_assert(PIPE_RPT == 2 || PIPE_RPT == 4, "PIPE_ROWS is 512 or 1024");
static_assert(PIPE_RPW >= 1, "at least one round per wave and block");
static_assert(PIPE_NBK <= 32, "the bucket scan runs on 32 lanes");

typedef uint32_t u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));

// N + 1 consecutive offsets p[e .. e + N], indices clamped to `last` (the column's final offset when the block reaches the
// end of the column: rows past the end get a length of 0).  The loads are unconditional -- a load under a branch
// would make the compiler's s_waitcnt for every older load in flight conservative (vmcnt counts in order).  p and last
// are uniform, so the address is a scalar base + a 32-bit lane offset.
template <int N>
__device__ __forceinline__ void pipe_load_offsets(const uint32_t *__restrict__ p, uint32_t e, uint32_t last, uint32_t (&o)[N + 1])
{
#pragma unroll
    for (int q = 0; q <= N; ++q) {
        const uint32_t r = e + (uint32_t)q;
        o[q] = p[r < last ? r : last];
    }
}

// 32 bytes at p (any alignment)
__device__ __forceinline__ void pipe_load32(const uint8_t *__restrict__ p, uint32_t (&w)[8])
{
    const u32x4_unaligned lo = *reinterpret_cast<const u32x4_unaligned *>(p);
    const u32x4_unaligned hi = *reinterpret_cast<const u32x4_unaligned *>(p + 16);
    w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w;
    w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
}

// What a round needs to know about its block (uniform over the workgroup): where the block's bytes start in each
// column, and how many of its rows are this kernel's.
struct PipeBlock {
    const uint8_t *pA, *pB; // values + offset of the block's first row
    uint32_t nmine;         // rows of the block this kernel computes (they come first in the descriptor order)
};

// Descriptor of a row, 8 bytes, written in length order by sortB:
//   x = text offset | pattern offset << 16   (bytes from the block's first row in the respective column, < 65536)
//   y = text length | pattern length << 8 | row index within the block << 16 | roles swapped << 30 | not mine << 31
// "text" is the string the columns of the bit-parallel cores walk: a, or the shorter one for the symmetric measures
// when both sides are columns.  A row is only "mine" if both of its 32-byte windows lie inside their columns, so the
// loads below need no bounds (the one to three rows at the very end of a column are left to k_wave_pairs, as are
// longer and non-ASCII rows).  A one-row side (the literal, strsim.rs:61-66) is read from a 32-byte copy in LDS.
template <bool SYMMETRIC>
__device__ __forceinline__ void pipe_prefetch(const uint2 d, const PipeBlock &b, uint32_t (&wt)[8], uint32_t (&wp)[8])
{
    // Unconditional loads (see pipe_load_offsets): rows that are not ours have offset 0 = the block's first byte, which is
    // readable (sortA points a block without 32 readable bytes, and a literal side, at a scratch area instead).
    const uint32_t t0 = d.x & 0xFFFFu, p0 = d.x >> 16;
    const bool swap = SYMMETRIC && ((d.y >> 30) & 1u);
    const uint8_t *cT = swap ? b.pB : b.pA, *cP = swap ? b.pA : b.pB;
    pipe_load32(cT + t0, wt);
    pipe_load32(cP + p0, wp);
}

// Levenshtein result as an index into the 33 x 33 table of 1 - dist/den (dist * 33 + den); 0xFFFF is never produced
template <int NP>
__device__ __forceinline__ uint32_t pipe_lev_code(const uint32_t (&wa)[8], uint32_t la, const uint32_t (&wb)[8], uint32_t lb,
                                                  uint32_t tmax)
{
    uint32_t P[NP];
    build_planes<NP>(wb, P);
    const bool live = la != 0u && lb != 0u;
    const uint32_t la1 = live ? la : 1u, lb1 = live ? lb : 1u;
    const uint32_t s = 32u - lb1;
#pragma unroll
    for (int k = 0; k < NP; ++k) P[k] <<= s;
    const uint32_t dist = lev_myers32<NP>(wa, la1, tmax, P, lb1);
    uint32_t code = dist * 33u + (la1 > lb1 ? la1 : lb1);
    // both empty: 1.0 = entry (0, 1); one side empty: 0.0 = entry (1, 1)   (strsim.rs:128, :160)
    if (!live) code = (la == 0u && lb == 0u) ? 1u : 34u;
    return code;
}

template <int MEASURE, bool LIT>
__device__ __forceinline__ void pipe_compute(const uint32_t (&wt_in)[8], const uint32_t (&wp_in)[8], uint32_t meta, uint16_t *s_code,
                                             double *s_val, const double *__restrict__ qtab, bool litA, bool litB,
                                             const uint32_t *s_lit)
{
    // a literal side (no role swap in such a call: text = a, pattern = b) comes from its 32-byte copy in LDS
    uint32_t wt[8], wp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        wt[q] = (LIT && litA) ? s_lit[q] : wt_in[q];
        wp[q] = (LIT && litB) ? s_lit[8 + q] : wp_in[q];
    }
    bool fast = (meta >> 31) == 0u;
    const uint32_t lt = meta & 0xFFu, lp = (meta >> 8) & 0xFFu, idx = (meta >> 16) & 0x3FFFu;
    // conservative tests on the whole 32-byte windows (bytes past a string belong to its neighbours): any high bit leaves
    // the row to the code-point kernels; the varying low bits decide how many bit-planes the match masks need
    uint32_t any;
    const uint32_t vary = window_vary(wt, wp, any);
    if (any & 0x80u) fast = false;
    if (__ballot(fast) == 0ull) return;
    const uint32_t la = fast ? lt : 0u, lb = fast ? lp : 0u;
    const uint32_t tmax = wave_max_rounded(la);
    const bool wide = __ballot(fast && (vary & 0x60u)) != 0ull; // six-plane rounds run as seven (register budget)
    if (MEASURE == LEVENSHTEIN) {
        uint32_t code;
        if (wide) code = pipe_lev_code<7>(wt, la, wp, lb, tmax);
        else code = pipe_lev_code<5>(wt, la, wp, lb, tmax);
        if (fast) s_code[idx] = (uint16_t)code;
    } else {
        double res;
        if (wide) res = lane_pair_result<MEASURE, 7>(wt, la, wp, lb, tmax, nullptr, qtab);
        else res = lane_pair_result<MEASURE, 5>(wt, la, wp, lb, tmax, nullptr, qtab);
        if (fast) s_val[idx] = res;
    }
}

// LIT: the instantiation for calls with a one-row side (the Utf8 literal, strsim.rs:48-52,61-66)
template <int MEASURE, bool LIT>
__global__ __launch_bounds__(PIPE_BLOCK) __attribute__((amdgpu_waves_per_eu(STRSIM_PIPE_WAVES_PER_EU))) void
k_lane_pipe(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, uint64_t rowsA,
            const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB, uint64_t rowsB, double *__restrict__ out,
            uint64_t n, unsigned long long *__restrict__ slowmask, DevStatus *__restrict__ status,
            const double *__restrict__ qtab)
{
    constexpr bool LEV = MEASURE == LEVENSHTEIN;
    constexpr bool SYMMETRIC = MEASURE == LEVENSHTEIN || MEASURE == JACCARD || MEASURE == SORENSEN_DICE;
    constexpr int B = PIPE_ROWS, RPT = PIPE_RPT, NBK = PIPE_NBK;
    __shared__ uint32_t s_cnt[2][32];
    __shared__ uint2 s_desc[2][B];
    __shared__ uint16_t s_code[LEV ? 2 : 1][LEV ? B : 1];   // Levenshtein: table index per row, 0xFFFF = not computed here
    __shared__ double s_val[LEV ? 1 : 2][LEV ? 1 : B];      // other measures: the f64 result, all-ones = not computed here
    __shared__ double s_levtab[LEV ? 33 * 33 : 1];
    __shared__ uint32_t s_lit[16];                         // 32-byte windows of the literals (one-row sides), zero-padded

    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    // the call's status block (counters of the kernels that follow in the stream) is cleared here, not by a memset node
    if (blockIdx.x == 0u && tid < (uint32_t)(sizeof(DevStatus) / sizeof(uint32_t))) reinterpret_cast<uint32_t *>(status)[tid] = 0u;
    if (LEV) {
        // 1.0 - dist/den for every (dist, den) a <= 32-byte pair can produce: the epilogue's own IEEE division
        // (strsim.rs:160), done once per workgroup instead of once per pair
        for (uint32_t i = tid; i < 33u * 33u; i += PIPE_BLOCK) {
            const uint32_t d = i / 33u, m = i % 33u;
            s_levtab[i] = m ? epilogue_levenshtein(d, m, m) : 0.0;
        }
    }
    if (tid < 64u) s_cnt[tid >> 5][tid & 31u] = 0u;
#pragma unroll
    for (int q = 0; q < 2 * RPT; ++q) {
        const uint32_t i = (uint32_t)q * PIPE_BLOCK + tid; // 0 .. 2 B - 1
        if (LEV) s_code[i / B][i % B] = 0xFFFFu;
        else reinterpret_cast<unsigned long long *>(&s_val[i / B][i % B])[0] = ~0ull;
    }

    const uint32_t totalA = offA[rowsA], totalB = offB[rowsB];
    const bool bcastA = LIT && rowsA == 1, bcastB = LIT && rowsB == 1;
    const uint32_t litA0 = bcastA ? offA[0] : 0u, litB0 = bcastB ? offB[0] : 0u; // a one-row side is the literal (strsim.rs:61-66)
    if (wv == 0u) {
        uint32_t w[8];
        if (bcastA) {
            load_window32(valA, litA0, totalA, w);
#pragma unroll
            for (int q = 0; q < 8; ++q) s_lit[q] = w[q];
        }
        if (bcastB) {
            load_window32(valB, litB0, totalB, w);
#pragma unroll
            for (int q = 0; q < 8; ++q) s_lit[8 + q] = w[q];
        }
    }
    const uint64_t nblocks = (n + (uint64_t)(B - 1)) / (uint64_t)B;
    const uint64_t G = gridDim.x;
    const long nb = blockIdx.x < nblocks ? (long)((nblocks - 1u - blockIdx.x) / G) + 1 : 0; // blocks of this workgroup

    // rows of a block are dealt RPT consecutive rows per thread in the sort phases (vector loads of the offsets) and
    // q * 256 + tid in the store phase (coalesced 8-byte stores, one ballot = the mask word of 64 consecutive rows)
    uint32_t oa[RPT + 1], ob[RPT + 1];           // offsets of block j+2 (in flight across the rounds of block j)
    uint32_t skey[RPT], srank[RPT], sd0[RPT], sd1[RPT]; // sortA -> sortB
#pragma unroll
    for (int q = 0; q <= RPT; ++q) { oa[q] = 0u; ob[q] = 0u; }
#pragma unroll
    for (int q = 0; q < RPT; ++q) { skey[q] = 0u; srank[q] = 0u; sd0[q] = 0u; sd1[q] = 0u; }
    const uint8_t *const safe = reinterpret_cast<const uint8_t *>(qtab); // >= 32 readable bytes, always
    PipeBlock blk0{safe, safe, 0u}, blk1 = blk0; // block j, block j+1
    uint32_t baseA2 = 0u, baseB2 = 0u;                          // block j+2
    uint32_t wt[8], wp[8];                                      // the windows of the round about to be computed
    uint32_t meta = 0x80000000u;
#pragma unroll
    for (int d = 0; d < 8; ++d) { wt[d] = 0u; wp[d] = 0u; }
    lds_barrier();
#ifdef STRSIM_PIPE_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_last = st_t0;
#endif

    for (long j = -2; j <= nb; ++j) {
        const uint32_t par = (uint32_t)j & 1u; // parity of block j
        // ---- store(j-1): staged results -> global, coalesced; rows nobody computed go into the mask word of their chunk
        if (j >= 1) {
            const uint64_t row0 = (blockIdx.x + (uint64_t)(j - 1) * G) * (uint64_t)B;
            const uint32_t left_rows = (uint32_t)(n - row0 < (uint64_t)B ? n - row0 : (uint64_t)B); // rows of the block that exist
            double *__restrict__ const outb = out + row0;
            unsigned long long *__restrict__ const maskb = slowmask + (row0 >> 6);
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t i = (uint32_t)q * PIPE_BLOCK + tid;
                bool undone;
                double v;
                if (LEV) {
                    const uint32_t code = s_code[par ^ 1u][i];
                    s_code[par ^ 1u][i] = 0xFFFFu;
                    undone = code == 0xFFFFu;
                    v = s_levtab[undone ? 0u : code];
                } else {
                    v = s_val[par ^ 1u][i];
                    reinterpret_cast<unsigned long long *>(&s_val[par ^ 1u][i])[0] = ~0ull;
                    undone = (uint32_t)(__double_as_longlong(v) >> 32) == 0xFFFFFFFFu;
                }
                const bool valid = i < left_rows;
                const unsigned long long left = __ballot(undone && valid);
                if (valid && !undone) outb[i] = v;
                if (lane == 0u && valid) maskb[i >> 6] = left;
            }
        }
        // ---- sortB(j+1): exclusive scan of the bucket counters (every wave for itself), descriptors in length order
        if (j + 1 >= 0 && j + 1 < nb) {
            uint32_t c = s_cnt[par ^ 1u][lane & 31u];
            uint32_t inc = c;
#pragma unroll
            for (int sft = 1; sft < 32; sft <<= 1) {
                const uint32_t up = __shfl_up(inc, sft, 32);
                if ((lane & 31u) >= (uint32_t)sft) inc += up;
            }
            const uint32_t exc = inc - c;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t base = __shfl(exc, skey[q], 32);
                s_desc[par ^ 1u][base + srank[q]] = make_uint2(sd0[q], sd1[q]);
            }
            blk1.nmine = uniform(__shfl(exc, NBK - 1, 32));
        }
        PIPE_STAMP(0);
        lds_barrier(); // Y
        PIPE_STAMP(1);
        if (tid < 32u) s_cnt[par ^ 1u][tid] = 0u; // read by sortB above; next used by sortA two bodies from now
        // ---- loads(j+2): offsets of block j+2 into registers (unconditional, clamped: see pipe_load_offsets)
        const bool have2 = j + 2 < nb;
        const uint64_t row0_2u = (blockIdx.x + (uint64_t)(j + 2) * G) * (uint64_t)B;
        const uint64_t row0_2 = row0_2u < n ? row0_2u : n;                                   // a block that does not exist: row n
        const uint32_t rows2 = (uint32_t)(n - row0_2 < (uint64_t)B ? n - row0_2 : (uint64_t)B); // rows of the block that exist
        {
            const uint32_t *__restrict__ const pa = offA + (bcastA ? 0u : row0_2), *__restrict__ const pb = offB + (bcastB ? 0u : row0_2);
            baseA2 = pa[0];
            baseB2 = pb[0];
            pipe_load_offsets<RPT>(pa, (uint32_t)RPT * tid, bcastA ? 1u : rows2, oa);
            pipe_load_offsets<RPT>(pb, (uint32_t)RPT * tid, bcastB ? 1u : rows2, ob);
        }
        PIPE_STAMP(2);
        // ---- rounds(j): wave w runs rounds w, 2W-1-w, 2W+w, 4W-1-w, ... of the length order (short + long = balanced).
        // Each round first issues the window loads of the NEXT round (the last one: of round 0 of block j+1) into the
        // staging registers, computes on the current windows, and only then moves the staged windows over -- by then the
        // loads have had a whole round to land.  (Two alternating register sets instead of the 17 moves were tried: the
        // compiler then copies a set at the loop edges while its loads are still in flight.)
        if (j >= -1 && j < nb) {
            uint16_t *const codes = LEV ? &s_code[LEV ? par : 0][0] : nullptr;
            double *const vals = LEV ? nullptr : &s_val[LEV ? 0 : par][0];
#pragma unroll 1
            for (int k = (j < 0 ? PIPE_RPW - 1 : 0); k < PIPE_RPW; ++k) {
                const bool nextblk = k + 1 == PIPE_RPW;
                const int kn = nextblk ? 0 : k + 1;
                const uint32_t rn = (uint32_t)(kn >> 1) * (2u * PIPE_WAVES) + ((kn & 1) ? (uint32_t)(2 * PIPE_WAVES - 1) - wv : wv);
                uint2 d = s_desc[nextblk ? (par ^ 1u) : par][rn * 64u + lane];
                if (nextblk && j + 1 >= nb) d = make_uint2(0u, 0x80000000u); // no next block: stale descriptors
                const PipeBlock bsel{nextblk ? blk1.pA : blk0.pA, nextblk ? blk1.pB : blk0.pB, 0u};
                uint32_t nt[8], np[8];
                pipe_prefetch<SYMMETRIC>(d, bsel, nt, np);
                PIPE_STAMP(3);
                if (j >= 0) pipe_compute<MEASURE, LIT>(wt, wp, meta, codes, vals, qtab, bcastA, bcastB, s_lit);
                PIPE_STAMP(4);
#pragma unroll
                for (int q = 0; q < 8; ++q) { wt[q] = nt[q]; wp[q] = np[q]; }
                meta = d.y;
#ifdef STRSIM_PIPE_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                PIPE_STAMP(5);
            }
        }
        // ---- sortA(j+2): lengths -> bucket keys (number of DP columns the pair will run), ranks by LDS atomics
        PipeBlock blk2{safe, safe, 0u};
#pragma unroll
        for (int q = 0; q < RPT; ++q) { skey[q] = 0u; srank[q] = 0u; sd0[q] = 0u; sd1[q] = 0u; }
        if (have2) {
            const uint32_t availA = totalA - baseA2, availB = totalB - baseB2; // column bytes from the block's first row on
            // where the block's rows are read from; without 32 readable bytes there (then no row of the block is ours), and
            // for a literal side (read from LDS instead), the unconditional window loads go to a scratch area
            blk2.pA = (!bcastA && availA >= 32u) ? valA + baseA2 : safe;
            blk2.pB = (!bcastB && availB >= 32u) ? valB + baseB2 : safe;
            const uint32_t litAlen = bcastA ? offA[1] - litA0 : 0u, litBlen = bcastB ? offB[1] - litB0 : 0u;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t i = (uint32_t)RPT * tid + (uint32_t)q;
                const uint32_t a0 = bcastA ? 0u : oa[q] - baseA2, b0 = bcastB ? 0u : ob[q] - baseB2;
                const uint32_t la8 = bcastA ? litAlen : oa[q + 1] - oa[q], lb8 = bcastB ? litBlen : ob[q + 1] - ob[q];
                // mine: both strings <= 32 bytes, both 32-byte windows inside their columns, offsets that fit the descriptor
                const bool inA = bcastA || (a0 < 65536u && a0 + 32u <= availA), inB = bcastB || (b0 < 65536u && b0 + 32u <= availB);
                const bool mine = i < rows2 && la8 <= 32u && lb8 <= 32u && inA && inB;
                const bool swap = SYMMETRIC && !bcastA && !bcastB && la8 > lb8; // symmetric measures walk the shorter string
                const uint32_t lt = swap ? lb8 : la8, lp = swap ? la8 : lb8;
                const uint32_t t0 = swap ? b0 : a0, p0 = swap ? a0 : b0;
                const uint32_t key = mine ? (((lt ? lt : 1u) - 1u) >> PIPE_BSH) : (uint32_t)(NBK - 1);
                skey[q] = key;
                srank[q] = atomicAdd(&s_cnt[par][key], 1u);
                sd0[q] = mine ? (t0 | (p0 << 16)) : 0u;
                sd1[q] = mine ? (lt | (lp << 8) | (i << 16) | (swap ? 0x40000000u : 0u)) : 0x80000000u;
            }
        }
        PIPE_STAMP(6);
        lds_barrier(); // X
        PIPE_STAMP(7);
        blk0 = blk1;
        blk1 = blk2;
    }
#ifdef STRSIM_PIPE_STAMPS
    if (lane == 0u) {
        const uint32_t w = (blockIdx.x * PIPE_WAVES + wv) & 16383u;
        for (int q = 0; q < 8; ++q) g_pipe_stamps[w][q] = st_acc[q];
        g_pipe_stamps[w][8] = __builtin_amdgcn_s_memtime() - st_t0;
        g_pipe_stamps[w][9] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
}
