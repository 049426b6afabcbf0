// strsim_lane_stage_pc.h -- LAB ONLY (make EXTRA="-DSTRSIM_LAB -DSTRSIM_STAGE_PC=<measures bit mask>"; never in the product library).
//
// Round 6's one structural experiment on the headline kernel (LAB_NOTES 3.1a "a producer wave
// per workgroup"): k_lane_stage with its per-block phase chain taken off the critical path by ROLES.  A 512-thread workgroup =
// four PRODUCER waves (store of block j-1; cut, bytes DMA, sort of block j+1) beside four CONSUMER waves (the rounds of block j),
// staging areas, descriptors and result codes double-buffered, two workgroup barriers per block:
//
//   epoch e:   producers   store(e-1) | cut(e+1), bytes DMA(e+1), sortA(e+1)  | X1 | sortB(e+1), offsets DMA(e+2), DMA wait | X2
//              consumers   round w of block e                                  | X1 | round 7-w of block e                   | X2
//
// What it is for: the match-mask tables of strsim_lane_lut.h take 16 % of the vector instructions out of the Levenshtein kernel, but
// their 12 KB of LDS cost the fifth workgroup per CU, and at four the uniform kernel is bound by the latency of its own chain
// (LAB 3.0 [r3]).  Here the chain runs beside the rounds instead of between them.  Columns only (no literal side), one measure.
#pragma once

#ifndef STRSIM_PC_CAP
#define STRSIM_PC_CAP 10240 // staged bytes per column and block
#endif
#ifndef STRSIM_PC_WG_PER_CU
#define STRSIM_PC_WG_PER_CU 2
#endif
#ifndef STRSIM_PC_LUT
#define STRSIM_PC_LUT 1 // the consumers take their match masks from the LDS tables
#endif

constexpr int PC_THREADS = 512, PC_PROD = 256, PC_CW = 4; // threads; producer threads (waves 0..3); consumer waves (4..7)

template <int MEASURE, bool LUT>
__device__ __forceinline__ void
lane_stage_pc_body(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, uint64_t rowsA,
                   const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB, uint64_t rowsB, OutPtrs outs,
                   uint64_t n, unsigned long long *__restrict__ slowmask, DevStatus *__restrict__ status,
                   const double *__restrict__ qtab, uint32_t *__restrict__ sched, DevStatus *__restrict__ publish, uint32_t ticket)
{
    static_assert(MEASURE != ALL_MEASURES, "single measures only");
    constexpr bool LEV = MEASURE == LEVENSHTEIN;
    constexpr int B = STAGE_ROWS, RPT = B / PC_PROD, NBK = STAGE_NBK;
    constexpr int CAP = STRSIM_PC_CAP, COL = CAP + 96, AREA = 2 * COL + 64 + 16;
    constexpr int DMA_ITERS = (CAP + 48 + 16 * PC_PROD - 1) / (16 * PC_PROD);
    constexpr uint32_t COLB = COL;
    static_assert(B == 512 && RPT == 2 && B / 64 == 2 * PC_CW, "eight rounds per block, two per consumer wave");
    static_assert(CAP % 16 == 0 && AREA <= 65536 && AREA % 16 == 0, "staging area");

    __shared__ __attribute__((aligned(4096))) uint8_t s_lut[LUT ? 4096 * PC_CW : 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_bytes[2][AREA];
    __shared__ uint2 s_desc[2][B];
    __shared__ __attribute__((aligned(16))) uint32_t s_off[2][B + 4];
    __shared__ uint32_t s_cnt[2][32];
    __shared__ uint32_t s_rows[2], s_nmine[2]; // per buffer: rows of the block (0: there is none), rows of it that are this kernel's
    __shared__ uint32_t s_left, s_sched[2];
    __shared__ uint16_t s_code[2][LEV ? B : 1];
    __shared__ uint32_t s_word[2][LEV ? 1 : B];
    __shared__ double s_none[1];

    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const bool producer = uniform(wv) < 4u;
    const uint32_t ptid = tid & (uint32_t)(PC_PROD - 1); // thread of the producer half
    const uint32_t pwv = uniform(wv & 3u);                // wave within its role
    if (blockIdx.x == 0u && tid < (uint32_t)(sizeof(DevStatus) / sizeof(uint32_t))) reinterpret_cast<uint32_t *>(status)[tid] = 0u;
    if (tid < 64u) s_cnt[tid >> 5][tid & 31u] = 0u;
    if (tid == 0u) { s_left = 0u; s_rows[0] = s_rows[1] = 0u; s_nmine[0] = s_nmine[1] = 0u; }
    if (producer) {
#pragma unroll
        for (int q = 0; q < 2 * RPT; ++q) {
            const uint32_t i = (uint32_t)q * PC_PROD + ptid; // 0 .. 2 B - 1
            if (LEV) s_code[i / B][i % B] = 0xFFFFu;
            else s_word[i / B][i % B] = 0xFFFFFFFFu;
        }
    }
    const uint32_t totalA = load_invariant(offA + rowsA), totalB = load_invariant(offB + rowsB);

    const uint64_t nchunks = (n + 63u) >> 6;
    const uint32_t nchunks32 = (uint32_t)nchunks;
    auto grab = [&](uint32_t seen, uint32_t &lo, uint32_t &sz) { // thread 0 only (k_lane_stage's rule)
        const uint32_t left = nchunks32 > seen ? nchunks32 - seen : 0u;
        uint32_t want = (uint32_t)STRSIM_STAGE_RANGE_MAX;
        if (left < (uint32_t)(STRSIM_STAGE_RANGE_MAX * STRSIM_STAGE_RANGE_DIV) * gridDim.x) want = left / ((uint32_t)STRSIM_STAGE_RANGE_DIV * gridDim.x);
        sz = want < (uint32_t)(STRSIM_STAGE_RANGE_MIN_BLOCKS * (B / 64)) ? (uint32_t)(STRSIM_STAGE_RANGE_MIN_BLOCKS * (B / 64)) : want;
        lo = __hip_atomic_fetch_add(&sched[0], sz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t grab_lo = 0u, grab_sz = 0u;
    if (tid == 0u) {
        grab(0u, grab_lo, grab_sz);
        s_sched[0] = grab_lo;
        s_sched[1] = grab_sz;
    }
    lds_barrier();
    uint64_t row_end, row0; // producers: first row behind the current range / first row of the block about to be prepared
    {
        const uint64_t lo = uniform(s_sched[0]), hi = lo + uniform(s_sched[1]);
        row0 = lo * 64u < n ? lo * 64u : n;
        row_end = hi * 64u < n ? hi * 64u : n;
    }
    lds_barrier();
    if (tid == 0u) grab(grab_lo, grab_lo, grab_sz);
    bool grab_pending = true;

    EqLut lut; // a consumer wave's match-mask tables
    lut.lane4 = lane * 4u;
    lut.krep = ((STRSIM_LDS_ADDR(&s_lut[0]) >> 8) + 16u * pwv) * 0x01010101u;
    const uint32_t ldsOffA = STRSIM_LDS_ADDR(&s_off[0][0]), ldsOffB = STRSIM_LDS_ADDR(&s_off[1][0]);
    const uint32_t ptid4 = ptid * 4u, ptid16 = ptid * 16u;
    auto dma_offsets = [&](uint64_t row) { // producers
        const uint32_t cnt = (uint32_t)(row_end - row < (uint64_t)B ? row_end - row : (uint64_t)B);
        const uint32_t *const pa = offA + row, *const pb = offB + row;
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            if (ptid + (uint32_t)it * PC_PROD <= cnt) {
                lds_dma_b32(pa + it * PC_PROD, ptid4, ldsOffA + 4u * ((uint32_t)it * PC_PROD + pwv * 64u));
                lds_dma_b32(pb + it * PC_PROD, ptid4, ldsOffB + 4u * ((uint32_t)it * PC_PROD + pwv * 64u));
            }
        }
        if (ptid == 0u && cnt == (uint32_t)B) {
            lds_dma_b32(pa + B, ptid4, ldsOffA + 4u * (uint32_t)B);
            lds_dma_b32(pb + B, ptid4, ldsOffB + 4u * (uint32_t)B);
        }
    };
    if (producer && row0 < row_end) dma_offsets(row0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    auto pin = [](double &x) { asm volatile("" : "+v"(x)); };
    // store(j): staged integers of buffer `buf` -> global, coalesced (producers; k_lane_stage's store_block for one measure)
    auto store_block = [&](uint64_t r0, uint32_t rows, uint32_t buf) {
        unsigned long long *__restrict__ const maskb = slowmask + (r0 >> 6);
        constexpr int M1 = LEV ? JARO : MEASURE;
        constexpr bool JARO_LIKE = M1 == JARO || M1 == JARO_WINKLER;
        uint32_t pk[RPT];
        double t0[RPT], t1[RPT], t2[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = (uint32_t)q * PC_PROD + ptid;
            if (LEV) { pk[q] = s_code[buf][i]; s_code[buf][i] = 0xFFFFu; }
            else { pk[q] = s_word[buf][i]; s_word[buf][i] = 0xFFFFFFFFu; }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (LEV) {
                t0[q] = qtab[pk[q] == 0xFFFFu ? 0u : pk[q]];
            } else if (JARO_LIKE) {
                const uint32_t lo = pk[q];
                const uint32_t m = lo & 63u, t = (lo >> 6) & 63u, la = (lo >> 12) & 63u, lb = (lo >> 18) & 63u, h = t >> 1;
                t0[q] = qtab[m * (uint32_t)QTAB_N + la];
                t1[q] = qtab[m * (uint32_t)QTAB_N + lb];
                t2[q] = qtab[(m > h ? m - h : 0u) * (uint32_t)QTAB_N + m];
            }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (LEV || JARO_LIKE) pin(t0[q]);
            if (JARO_LIKE) { pin(t1[q]); pin(t2[q]); }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = (uint32_t)q * PC_PROD + ptid;
            if ((uint32_t)q * PC_PROD < rows) {
                bool undone;
                double v = 0.0;
                if (LEV) {
                    undone = pk[q] == 0xFFFFu;
                    v = 1.0 - t0[q];
                } else {
                    undone = pk[q] == 0xFFFFFFFFu;
                    if (!undone) v = stage_epilogue<M1>(pk[q], t0[q], t1[q], t2[q]);
                }
                const bool valid = i < rows;
                const unsigned long long left = __ballot(undone && valid);
                if (valid && !undone) outs.p[0][r0 + i] = v;
                if (lane == 0u && valid) {
                    maskb[i >> 6] = left;
                    if (publish && left) atomicAdd(&s_left, (uint32_t)__builtin_popcountll(left));
                }
            }
        }
    };

#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
    // per-wave cycle sums (g_stage_stamps, read by bench_support/stage_pc_stamps.py).  Producers: [0] store [1] cut + bytes DMA issue +
    // sortA [2] wait at X1 [3] sortB + offsets DMA issue [4] DMA wait [5] wait at X2.  Consumers: [6] descriptor + windows [7] cores
    // [8] wait at X1 [9] wait at X2.  [10] all [11] all (100 MHz) [14] role (1 = producer)
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_last = st_t0;
#endif
    // producers' registers: the block being stored next (its results are written by the consumers one epoch before)
    uint64_t st_row0_0 = 0, st_row0_1 = 0; // by buffer (scalars, not an indexed array: registers)
    uint32_t st_rows_0 = 0, st_rows_1 = 0;
    bool more = true; // producers: blocks may still come
    int e = -1;
    uint32_t cur_rows = 0u; // rows of block e (0: none)
    for (;;) {
        const uint32_t p = (uint32_t)e & 1u, q = p ^ 1u; // block e lives in buffer p; block e + 1 is prepared into q
        if (producer) {
            __builtin_amdgcn_s_setprio(1);
            if (grab_pending) {
                if (tid == 0u) { s_sched[0] = grab_lo; s_sched[1] = grab_sz; }
                grab_pending = false;
            }
            // ---- store(e-1): the codes the consumers left in buffer q an epoch ago (before this epoch's DMA is issued: in-order returns)
            if (q ? st_rows_1 : st_rows_0) {
                store_block(q ? st_row0_1 : st_row0_0, q ? st_rows_1 : st_rows_0, q);
                if (q) st_rows_1 = 0u; else st_rows_0 = 0u;
            }
            STAGE_STAMP(0);
            uint32_t rows = 0u, skey[RPT], srank[RPT], sd0[RPT], sd1[RPT];
            uint32_t chunksA = 0u, chunksB = 0u, leftA = 0u, leftB = 0u;
            const bool prep = more && row0 < row_end;
            if (prep) {
                // ---- the block: as many of the next 64-row chunks as have their bytes inside the staging areas
                const uint32_t avail = (uint32_t)(row_end - row0 < (uint64_t)B ? row_end - row0 : (uint64_t)B);
                const uint32_t baseA = uniform(s_off[0][0]), baseB = uniform(s_off[1][0]);
                const uint32_t misA = (uint32_t)(reinterpret_cast<uintptr_t>(valA + baseA) & 15u);
                const uint32_t misB = (uint32_t)(reinterpret_cast<uintptr_t>(valB + baseB) & 15u);
                {
                    const uint32_t en = (lane + 1u) * 64u < avail ? (lane + 1u) * 64u : avail;
                    const bool exists = lane * 64u < avail && lane < (uint32_t)(B / 64);
                    const uint32_t eA = s_off[0][exists ? en : 0u], eB = s_off[1][exists ? en : 0u];
                    const bool fits = exists && eA - baseA + misA <= (uint32_t)CAP && eB - baseB + misB <= (uint32_t)CAP;
                    const unsigned long long okm = __ballot(fits);
                    uint32_t k = (uint32_t)__builtin_ctzll(~okm);
                    if (k == 0u) k = 1u;
                    rows = uniform(k * 64u < avail ? k * 64u : avail);
                }
                const uint32_t endA = uniform(s_off[0][rows]), endB = uniform(s_off[1][rows]);
                const uint32_t spanA = endA - baseA + misA, spanB = endB - baseB + misB;
                const uint32_t stagedA = ((spanA < (uint32_t)CAP ? spanA : (uint32_t)CAP) + 15u) & ~15u;
                const uint32_t stagedB = ((spanB < (uint32_t)CAP ? spanB : (uint32_t)CAP) + 15u) & ~15u;
                leftA = totalA - baseA + misA; leftB = totalB - baseB + misB;
                chunksA = ((stagedA + 32u < leftA ? stagedA + 32u : leftA) + 15u) >> 4;
                chunksB = ((stagedB + 32u < leftB ? stagedB + 32u : leftB) + 15u) >> 4;
                uint8_t *const area = s_bytes[q];
                const uint32_t ldsArea = STRSIM_LDS_ADDR(area);
                if (ptid < 2u) *reinterpret_cast<uint4 *>(area + (chunksA << 4) + ptid * 16u) = make_uint4(0u, 0u, 0u, 0u);
                else if (ptid < 4u) *reinterpret_cast<uint4 *>(area + COLB + (chunksB << 4) + (ptid - 2u) * 16u) = make_uint4(0u, 0u, 0u, 0u);
                {
                    const uint8_t *__restrict__ const gA = valA + baseA - misA, *__restrict__ const gB = valB + baseB - misB;
#pragma unroll
                    for (int it = 0; it < DMA_ITERS; ++it) {
                        if (ptid + (uint32_t)it * PC_PROD < chunksA)
                            lds_dma_b128(gA + 16 * it * PC_PROD, ptid16, ldsArea + 16u * ((uint32_t)it * PC_PROD + pwv * 64u));
                        if (ptid + (uint32_t)it * PC_PROD < chunksB)
                            lds_dma_b128(gB + 16 * it * PC_PROD, ptid16, ldsArea + COLB + 16u * ((uint32_t)it * PC_PROD + pwv * 64u));
                    }
                }
                // ---- sortA: lengths -> bucket keys, ranks by LDS atomics
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const uint32_t i = (uint32_t)RPT * ptid + (uint32_t)r;
                    const bool have = i < rows;
                    const uint32_t a0 = s_off[0][i] - baseA, b0 = s_off[1][i] - baseB;
                    const uint32_t la8 = s_off[0][i + 1u] - s_off[0][i], lb8 = s_off[1][i + 1u] - s_off[1][i];
                    const bool stA = misA + a0 + la8 <= stagedA, stB = misB + b0 + lb8 <= stagedB;
                    const bool mine = have && la8 <= 32u && lb8 <= 32u && stA && stB;
                    const uint32_t wa = misA + a0, wb = COLB + misB + b0;
                    const bool swap = la8 > lb8;
                    const uint32_t lt = swap ? lb8 : la8, lp = swap ? la8 : lb8;
                    const uint32_t key = mine ? (((lt ? lt : 1u) - 1u) >> STAGE_BSH) : (uint32_t)(NBK - 1);
                    skey[r] = key;
                    srank[r] = atomicAdd(&s_cnt[q][key], 1u);
                    sd0[r] = mine ? (swap ? (wb | (wa << 16)) : (wa | (wb << 16))) : 0u;
                    sd1[r] = mine ? (lt | (lp << 8) | (i << 16)) : STAGE_DEAD;
                }
            }
            STAGE_STAMP(1);
            lds_barrier(); // ---- X1
            STAGE_STAMP(2);
            uint32_t nmine = 0u;
            if (prep) {
                // ---- sortB: exclusive scan of the bucket counters, descriptors in length order
                const uint32_t c = s_cnt[q][lane & 31u];
                const uint32_t exc = scan32_inclusive(c) - c;
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const uint32_t base = __shfl(exc, skey[r], 32);
                    s_desc[q][base + srank[r]] = make_uint2(sd0[r], sd1[r]);
                }
                nmine = uniform(__shfl(exc, NBK - 1, 32));
            }
            if (ptid < 32u) s_cnt[p][ptid] = 0u; // (the other epoch's counters: last read before the previous X2, next used after this X2)
            // ---- the block after: offsets DMA (s_off was last read in front of X1)
            uint64_t next_row0 = row0 + rows;
            if (prep) {
                if (q) { st_row0_1 = row0; st_rows_1 = rows; } else { st_row0_0 = row0; st_rows_0 = rows; }
                if (next_row0 >= row_end) {
                    const uint64_t lo = uniform(s_sched[0]), hi = lo + uniform(s_sched[1]);
                    next_row0 = lo * 64u < n ? lo * 64u : n;
                    row_end = hi * 64u < n ? hi * 64u : n;
                    if (tid == 0u) grab((uint32_t)lo, grab_lo, grab_sz);
                    grab_pending = true;
                }
                if (next_row0 < row_end) dma_offsets(next_row0);
                row0 = next_row0;
            } else {
                more = false;
            }
            STAGE_STAMP(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the bytes of block e + 1 and the offsets of e + 2 have landed (this wave's)
            STAGE_STAMP(4);
            if (prep) {
                uint8_t *const area = s_bytes[q];
                if (leftA < (chunksA << 4) && ptid == ((chunksA - 1u) & (uint32_t)(PC_PROD - 1)))
                    for (uint32_t b = leftA; b < (chunksA << 4); ++b) area[b] = 0;
                if (leftB < (chunksB << 4) && ptid == ((chunksB - 1u) & (uint32_t)(PC_PROD - 1)))
                    for (uint32_t b = leftB; b < (chunksB << 4); ++b) area[COLB + b] = 0;
            }
            if (tid == 0u) { s_rows[q] = rows; s_nmine[q] = nmine; }
            lds_barrier(); // ---- X2
            STAGE_STAMP(5);
        } else {
            // ---- consumers: the two rounds of this wave, one in front of X1, one behind it
            const uint32_t nmine = cur_rows ? uniform(s_nmine[p]) : 0u;
            const uint32_t nrounds = (nmine + 63u) >> 6;
            uint8_t *const area = s_bytes[p];
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                const uint32_t t = k ? (uint32_t)(2 * PC_CW - 1) - pwv : pwv;
                if (t < nrounds) {
                    const uint32_t hi = nmine - 64u * t;
                    const uint32_t first = hi >= 64u ? hi - 64u : 0u;
                    const uint32_t in_round = hi - first;
                    __builtin_amdgcn_s_setprio(1);
                    const uint2 d = s_desc[p][first + lane];
                    uint32_t wt[8], wp[8];
                    stage_window(area, d.x & 0xFFFFu, wt);
                    stage_window(area, d.x >> 16, wp);
#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    STAGE_STAMP(6);
                    __builtin_amdgcn_s_setprio(0);
                    stage_compute<MEASURE, LUT>(lut, wt, wp, lane < in_round ? d.y : STAGE_DEAD, in_round - 1u, s_code[p], s_word[p], s_none);
                    STAGE_STAMP(7);
                }
                __builtin_amdgcn_s_setprio(1);
                lds_barrier(); // ---- X1, X2
                if (k == 0) STAGE_STAMP(8); else STAGE_STAMP(9);
            }
        }
        if (e >= 0 && cur_rows == 0u) break; // (uniform over the workgroup: block e did not exist; store(e-1) has run in this epoch)
        ++e;
        cur_rows = uniform(s_rows[(uint32_t)e & 1u]);
    }
#if defined(STRSIM_LAB) && defined(STRSIM_STAGE_STAMPS)
    if (lane == 0u) {
        const uint32_t w = (blockIdx.x * 8u + wv) & 16383u;
        for (int c = 0; c < 10; ++c) g_stage_stamps[w][c] = st_acc[c];
        g_stage_stamps[w][10] = __builtin_amdgcn_s_memtime() - st_t0;
        const unsigned long long st_r1 = __builtin_amdgcn_s_memrealtime();
        g_stage_stamps[w][11] = st_r1 - st_r0;
        g_stage_stamps[w][12] = st_r0;
        g_stage_stamps[w][13] = st_r1;
        g_stage_stamps[w][14] = producer ? 1ull : 0ull;
        g_stage_stamps[w][15] = (unsigned long long)e;
    }
#endif
    if (publish) lds_barrier();
    if (tid == 0u) {
        if (publish && s_left) __hip_atomic_fetch_add(&sched[2], s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t done = __hip_atomic_fetch_add(&sched[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1u) {
            if (publish) {
                const uint32_t total = __hip_atomic_load(&sched[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&publish->lane_left, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ticket) __hip_atomic_store(&publish->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(&sched[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sched[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sched[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MEASURE, bool LUT>
__global__ __launch_bounds__(PC_THREADS) __attribute__((amdgpu_waves_per_eu(2 * STRSIM_PC_WG_PER_CU, 2 * STRSIM_PC_WG_PER_CU))) void
k_lane_stage_pc(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA, uint64_t rowsA,
                const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB, uint64_t rowsB, OutPtrs outs, uint64_t n,
                unsigned long long *__restrict__ slowmask, DevStatus *__restrict__ status, const double *__restrict__ qtab,
                uint32_t *__restrict__ sched, DevStatus *__restrict__ publish, uint32_t ticket)
{
    lane_stage_pc_body<MEASURE, LUT>(offA, valA, rowsA, offB, valB, rowsB, outs, n, slowmask, status, qtab, sched, publish, ticket);
}
