// strsim_lane_lit.h -- k_lane_lit<M>: a COLUMN against ONE LITERAL (strsim.rs:48-52, :61-66, :85-92: either side of the
// expression may be a Utf8 literal that is broadcast over the rows), one pair per lane, for literals and column strings of
// <= 32 ASCII bytes.  Included by strsim_kernels.hip inside namespace strsim, after strsim_lane_stage.h.
// Every measure is symmetric ([r5] the reference's Jaro included: strsim_lane_core.h), so the literal may be on either side.
//
// What a literal buys, in the terms of the issue-cost table of bench_support/micro/op_cost.hip:
//   * the literal is the TEXT the recurrence walks: every lane runs exactly len(literal) columns -- no sort by column
//     count, no descriptors, no copies of (Pv, Mv) at different columns, rows stay in their natural order;
//   * the text is wave-uniform, so the per-column bit masks of its characters (five v_bfe_i32 = 20 of a column's 59
//     cycles) are s_bfe_i32 on the scalar unit: a column costs 39 cycles of vector issue instead of 59;
//   * only ONE column is staged through LDS (double-buffered: the bytes of block j+1 and the offsets of block j+2 are in
//     flight while block j computes) and there is one workgroup barrier per block.
// The column string of a row is the pattern (its bit-planes are built per row, as in k_lane_stage).  The other measures run
// lane_cores32 (strsim_lane_core.h) on the same uniform text; their integers wait in LDS as 32-bit words and the f64
// epilogue runs in the store phase, as in k_lane_stage.  Rows this kernel cannot take (longer,
// non-ASCII, a single chunk overflowing the staging area) and every row of a call whose literal is longer than 32 bytes
// or non-ASCII stay in the mask for the later kernels.
#pragma once

#ifndef STRSIM_LIT_WAVES_PER_EU
#define STRSIM_LIT_WAVES_PER_EU 5
#endif

constexpr int LIT_BLOCK = 256;
constexpr int LIT_WAVES = LIT_BLOCK / 64;
constexpr int LIT_ROWS = 512;                 // rows per block (at most)
constexpr int LIT_RPT = LIT_ROWS / LIT_BLOCK;
constexpr int LIT_CAP = 10240;                // staged bytes per block
constexpr int LIT_COL = LIT_CAP + 96;                // + 32 bytes behind the block, a chunk's rounding, 32 zeros
constexpr int LIT_DMA_ITERS = (LIT_CAP + 48 + 16 * LIT_BLOCK - 1) / (16 * LIT_BLOCK);

// Levenshtein column loop with a wave-uniform text of exactly lt characters (lt >= 1): lt columns for every lane, the
// distance is lt + (vertical +1 deltas) - (vertical -1 deltas) over the lp pattern rows of the last column.
template <int NP>
__device__ __forceinline__ uint32_t lit_lev_uniform_text(const uint32_t (&wt)[8], uint32_t lt, const uint32_t (&P)[NP], uint32_t lp)
{
    uint32_t Pv = 0xFFFFFFFFu, Mv = 0u, Pl = 0u, Ml = 0u;
#pragma unroll
    for (int g = 0; g < 32 / 2; ++g) {
        if ((uint32_t)(2 * g) >= lt) break;
        uint32_t Pc[2], Mc[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * g + jj;
            const uint32_t Eq = eq_mask<NP>(P, 0xFFFFFFFFu, wt[j >> 2], j & 3); // wt is uniform: the bit fills are scalar
            const uint32_t D0 = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq) | Mv;
            const uint32_t nHP = bitop3<0x0E>(Mv, D0, Pv);
            const uint32_t nX = twice(nHP);
            const uint32_t HN2 = twice(D0 & Pv);
            Pv = bitop3<0xF2>(HN2, D0, nX);
            Mv = bitop3<0x50>(D0, D0, nX);
            Pc[jj] = Pv;
            Mc[jj] = Mv;
        }
        const bool odd_end = (uint32_t)(2 * g + 1) == lt; // (uniform) the text ends behind the first column of this pair
        Pl = odd_end ? Pc[0] : Pc[1];
        Ml = odd_end ? Mc[0] : Mc[1];
    }
    const uint32_t rows = low_ones(lp);
    return lt + popc32(Pl & rows) - popc32(Ml & rows);
}

// offC / valC: the column; offL / valL: the literal (one row), whichever side of the expression it stood on.
template <int MEASURE>
__global__ __launch_bounds__(LIT_BLOCK) __attribute__((amdgpu_waves_per_eu(STRSIM_LIT_WAVES_PER_EU))) void
k_lane_lit(const uint32_t *__restrict__ offC, const uint8_t *__restrict__ valC, const uint32_t *__restrict__ offL,
           const uint8_t *__restrict__ valL, double *__restrict__ out, uint64_t n, unsigned long long *__restrict__ slowmask,
           DevStatus *__restrict__ status, const double *__restrict__ qtab, uint32_t *__restrict__ sched)
{
    constexpr bool LEV = MEASURE == LEVENSHTEIN;
    constexpr bool JARO_LIKE = MEASURE == JARO || MEASURE == JARO_WINKLER;
    constexpr int B = LIT_ROWS, RPT = LIT_RPT;
    __shared__ __attribute__((aligned(16))) uint8_t s_bytes[2][LIT_COL];
    __shared__ __attribute__((aligned(16))) uint32_t s_off[3][B + 4];
    __shared__ uint16_t s_code[LEV ? 2 : 1][LEV ? B : 1];  // Levenshtein: index into the quotient table
    __shared__ uint32_t s_word[LEV ? 1 : 2][LEV ? 1 : B];  // the other measures: their integers (stage_ints' packing)
    __shared__ uint32_t s_lit[8];
    __shared__ uint32_t s_left; // rows of this workgroup's blocks that stay in the mask (reported like k_lane_stage does)

    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    if (tid == 0u) s_left = 0u;
    if (blockIdx.x == 0u && tid < (uint32_t)(sizeof(DevStatus) / sizeof(uint32_t))) reinterpret_cast<uint32_t *>(status)[tid] = 0u;
#pragma unroll
    for (int q = 0; q < 2 * RPT; ++q) { // both buffers
        if (LEV) (&s_code[0][0])[(uint32_t)q * LIT_BLOCK + tid] = 0xFFFFu;
        else (&s_word[0][0])[(uint32_t)q * LIT_BLOCK + tid] = 0xFFFFFFFFu;
    }
    const uint32_t totalC = load_invariant(offC + n);
#if STRSIM_BOUNDS_ON
    const uint32_t firstC = load_invariant(offC); // (lab: where the column's bytes begin)
#endif
    const uint32_t lit0 = load_invariant(offL), litl = load_invariant(offL + 1) - lit0;
    if (wv == 0u) {
        uint32_t w[8];
        load_window32(valL, lit0, lit0 + litl, w); // zeros behind the literal ...
        // ... replaced by copies of its first byte: the bytes behind the text are never walked, but they take part in the
        // test of which bit-planes tell the round's bytes apart (zeros would always ask for seven)
        const uint32_t rep = (w[0] & 0xFFu) * 0x01010101u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t v = litl > 4u * q ? (litl - 4u * q < 4u ? litl - 4u * q : 4u) : 0u; // bytes of this dword that are text
            const uint32_t keep = v >= 4u ? 0xFFFFFFFFu : ((1u << (8u * v)) - 1u);
            s_lit[q] = (w[q] & keep) | (rep & ~keep);
        }
    }
    lds_barrier();
    uint32_t wt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wt[q] = uniform(s_lit[q]);
    const uint32_t lit_or = wt[0] | wt[1] | wt[2] | wt[3] | wt[4] | wt[5] | wt[6] | wt[7];
    const bool lit_ok = litl <= 32u && (lit_or & 0x80808080u) == 0u; // else: nothing here is ours

    // this workgroup's contiguous range of 64-row chunks (many more workgroups than fit: finished ones are replaced)
    const uint64_t nchunks = (n + 63u) >> 6;
    const uint64_t per = (nchunks + gridDim.x - 1u) / gridDim.x;
    const uint64_t chunk_lo = (uint64_t)blockIdx.x * per < nchunks ? (uint64_t)blockIdx.x * per : nchunks;
    const uint64_t chunk_hi = chunk_lo + per < nchunks ? chunk_lo + per : nchunks;
    const uint64_t row_end = chunk_hi * 64u < n ? chunk_hi * 64u : n;
    if (chunk_lo * 64u < row_end) { // (uniform; a workgroup without rows still reports below)

    const uint32_t ldsBytes = STRSIM_LDS_ADDR(&s_bytes[0][0]), ldsOff = STRSIM_LDS_ADDR(&s_off[0][0]);
    const uint32_t wvu = uniform(wv), tid4 = tid * 4u, tid16 = tid * 16u;
    auto dma_offsets = [&](uint64_t row, uint32_t buf) { // offsets of rows row .. row + min(B, row_end - row) -> s_off[buf]
        const uint32_t cnt = (uint32_t)(row_end - row < (uint64_t)B ? row_end - row : (uint64_t)B);
        const uint32_t *const p = offC + row;
        const uint32_t dst = ldsOff + buf * (uint32_t)((B + 4) * 4);
#pragma unroll
        for (int it = 0; it < RPT; ++it)
            if (tid + (uint32_t)it * LIT_BLOCK <= cnt) {
                STRSIM_CHECK_INDEX(K_LIT, 1, row, row + (uint64_t)it * LIT_BLOCK + tid, n + 1u); // offsets[..] of n + 1
                STRSIM_CHECK_INDEX(K_LIT, 2, row, (uint32_t)it * LIT_BLOCK + tid, B + 4);        // -> s_off[buf][..] of B + 4
                STRSIM_CHECK_INDEX(K_LIT, 3, row, buf, 3);
                lds_dma_b32(p + it * LIT_BLOCK, tid4, dst + 4u * ((uint32_t)it * LIT_BLOCK + wvu * 64u));
            }
        if (tid == 0u && cnt == (uint32_t)B) STRSIM_CHECK_INDEX(K_LIT, 4, row, row + (uint64_t)B, n + 1u);
        if (tid == 0u && cnt == (uint32_t)B) lds_dma_b32(p + B, tid4, dst + 4u * (uint32_t)B);
    };
    // the block that starts at `row` with its offsets in s_off[buf]: as many of the next chunks as fit the staging area
    auto cut = [&](uint64_t row, uint32_t buf, uint32_t &base, uint32_t &mis) -> uint32_t {
        const uint32_t avail = (uint32_t)(row_end - row < (uint64_t)B ? row_end - row : (uint64_t)B);
        base = uniform(s_off[buf][0]);
        mis = (uint32_t)(reinterpret_cast<uintptr_t>(valC + base) & 15u);
        const uint32_t e = (lane + 1u) * 64u < avail ? (lane + 1u) * 64u : avail;
        const bool exists = lane * 64u < avail && lane < (uint32_t)(B / 64);
        const uint32_t end = s_off[buf][exists ? e : 0u];
        const unsigned long long okm = __ballot(exists && end - base + mis <= (uint32_t)LIT_CAP);
        uint32_t k = (uint32_t)__builtin_ctzll(~okm);
        if (k == 0u) k = 1u;
        return uniform(k * 64u < avail ? k * 64u : avail);
    };
    // (tail_lo .. tail_hi: the part of the copy that lies past the column's last byte -- zeroed once the copy has landed, by the
    //  lane that copied the column's last chunk; see k_lane_stage)
    auto dma_bytes = [&](uint32_t base, uint32_t mis, uint32_t end, uint32_t buf, uint32_t &tail_lo, uint32_t &tail_hi) -> uint32_t { // -> bytes the strings may use
        const uint32_t span = end - base + mis;
        const uint32_t staged = ((span < (uint32_t)LIT_CAP ? span : (uint32_t)LIT_CAP) + 15u) & ~15u;
        // + the 32 bytes behind the block when the column has them, then 32 zeros (see k_lane_stage: no stale LDS in a window)
        const uint32_t left = totalC - base + mis;
        const uint32_t chunks = ((staged + 32u < left ? staged + 32u : left) + 15u) >> 4;
        const uint8_t *const g = valC + base - mis;
#pragma unroll
        for (int it = 0; it < LIT_DMA_ITERS; ++it)
            if (tid + (uint32_t)it * LIT_BLOCK < chunks) {
                STRSIM_CHECK_COLUMN(K_LIT, 10, base, g + 16 * it * LIT_BLOCK + tid16, 16, valC, firstC, totalC); // 16 bytes of the column
                STRSIM_CHECK_RANGE(K_LIT, 11, base, 16u * ((uint32_t)it * LIT_BLOCK + tid) + 16u, 0, LIT_COL);   // -> this buffer
                lds_dma_b128(g + 16 * it * LIT_BLOCK, tid16, ldsBytes + buf * (uint32_t)LIT_COL + 16u * ((uint32_t)it * LIT_BLOCK + wvu * 64u));
            }
        if (tid < 2u) STRSIM_CHECK_RANGE(K_LIT, 12, base, (chunks << 4) + tid * 16u + 16u, 0, LIT_COL);
        if (tid < 2u) *reinterpret_cast<uint4 *>(&s_bytes[buf][(chunks << 4) + tid * 16u]) = make_uint4(0u, 0u, 0u, 0u);
        tail_lo = left;
        tail_hi = chunks << 4;
        return staged;
    };
    // results of a block, coalesced; straight-line passes as in k_lane_stage (all table loads of a thread before the first wait)
    auto pin = [](double &x) { asm volatile("" : "+v"(x)); };
    auto store_block = [&](uint64_t r0, uint32_t rows, uint32_t buf) {
        double *__restrict__ const outb = out + r0;
        unsigned long long *__restrict__ const maskb = slowmask + (r0 >> 6);
        uint32_t pk[RPT];
        double t0[RPT], t1[RPT], t2[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = (uint32_t)q * LIT_BLOCK + tid;
            if (LEV) { pk[q] = s_code[LEV ? buf : 0u][LEV ? i : 0u]; s_code[LEV ? buf : 0u][LEV ? i : 0u] = 0xFFFFu; }
            else { pk[q] = s_word[LEV ? 0u : buf][LEV ? 0u : i]; s_word[LEV ? 0u : buf][LEV ? 0u : i] = 0xFFFFFFFFu; }
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (LEV) {
                t0[q] = qtab[pk[q] == 0xFFFFu ? 0u : pk[q]]; // 1.0 - dist / den (strsim.rs:160) with the context's quotient table
            } else if (JARO_LIKE) {
                const uint32_t m = pk[q] & 63u, t = (pk[q] >> 6) & 63u, la = (pk[q] >> 12) & 63u, lb = (pk[q] >> 18) & 63u, h = t >> 1;
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
            const uint32_t i = (uint32_t)q * LIT_BLOCK + tid;
            if ((uint32_t)q * LIT_BLOCK < rows) { // (uniform)
                const bool undone = pk[q] == (LEV ? 0xFFFFu : 0xFFFFFFFFu), valid = i < rows;
                constexpr int M1 = LEV ? JARO : MEASURE;
                double v = 0.0;
                if (LEV) v = 1.0 - t0[q];
                else if (!undone) v = stage_epilogue<M1>(pk[q], t0[q], t1[q], t2[q]);
                const unsigned long long left = __ballot(undone && valid);
                if (valid) STRSIM_CHECK_INDEX(K_LIT, 20, r0 + i, r0 + i, n);
                if (valid && lane == 0u) STRSIM_CHECK_INDEX(K_LIT, 21, r0 + i, (r0 >> 6) + (i >> 6), nchunks);
                if (valid && !undone) outb[i] = v;
                if (lane == 0u && valid) {
                    maskb[i >> 6] = left;
                    if (left) atomicAdd(&s_left, (uint32_t)__builtin_popcountll(left));
                }
            }
        }
    };

    // ---- prologue: offsets(0) -> cut(0) -> bytes(0) and offsets(1) in flight
    uint64_t row0 = chunk_lo * 64u;
    dma_offsets(row0, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    uint32_t base, mis;
    uint32_t rows = cut(row0, 0u, base, mis);
    uint32_t tail_lo = 0u, tail_hi = 0u;
    uint32_t staged = dma_bytes(base, mis, uniform(s_off[0][rows]), 0u, tail_lo, tail_hi);
    if (row0 + rows < row_end) dma_offsets(row0 + rows, 1u);
    uint64_t prev_row0 = 0;
    uint32_t prev_rows = 0, prev_bb = 0;

    for (uint32_t j = 0;; ++j) {
        const uint32_t ob = j % 3u, bb = j & 1u;
        // ---- bytes(j) and offsets(j+1) have landed; everybody is done with block j-1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tail_lo < tail_hi && tid == (((tail_hi >> 4) - 1u) & (uint32_t)(LIT_BLOCK - 1)))
            for (uint32_t b = tail_lo; b < tail_hi; ++b) { STRSIM_CHECK_INDEX(K_LIT, 13, row0, b, LIT_COL); s_bytes[bb][b] = 0; }
        lds_barrier();
        // ---- block j+1: cut it, start its bytes and the offsets of block j+2
        const uint64_t next_row0 = row0 + rows;
        uint32_t nbase = 0u, nmis = 0u, nrows = 0u, nstaged = 0u, ntail_lo = 0u, ntail_hi = 0u;
        if (next_row0 < row_end) {
            const uint32_t ob1 = (j + 1u) % 3u;
            nrows = cut(next_row0, ob1, nbase, nmis);
            nstaged = dma_bytes(nbase, nmis, uniform(s_off[ob1][nrows]), bb ^ 1u, ntail_lo, ntail_hi);
            if (next_row0 + nrows < row_end) dma_offsets(next_row0 + nrows, (j + 2u) % 3u);
        }
        // ---- results of block j-1
        if (prev_rows) store_block(prev_row0, prev_rows, bb ^ 1u);
        // ---- rounds(j): rows in their natural order, round r on wave r % 4
        __builtin_amdgcn_s_setprio(0); // the column loops yield to waves that have copies to issue or results to store
        for (uint32_t r = wv; r * 64u < rows; r += LIT_WAVES) {
            const uint32_t i = r * 64u + lane;
            const bool have = i < rows;
            STRSIM_CHECK_INDEX(K_LIT, 30, row0, have ? i + 1u : 0u, B + 4);
            const uint32_t o0 = s_off[ob][have ? i : 0u], o1 = s_off[ob][have ? i + 1u : 0u];
            const uint32_t lp = o1 - o0, p0 = mis + (o0 - base);
            // mine: a column string of <= 32 bytes inside what was staged (the window behind it may hold anything)
            bool fast = have && lit_ok && lp <= 32u && p0 + lp <= staged;
            uint32_t wp[8];
            stage_window(&s_bytes[bb][0], fast ? p0 : 0u, wp, (uint32_t)LIT_COL, STRSIM_BOUNDS_ON ? 2u : 0u);
            uint32_t any;
            const uint32_t vary = window_vary(wt, wp, any);
            if (any & 0x80u) fast = false;
            if (__ballot(fast) == 0ull) continue;
            const bool wide = __ballot(fast && (vary & 0x60u)) != 0ull;
            if (LEV) {
                uint32_t code;
                if (litl == 0u) { // an empty literal: 1.0 against an empty string, else 0.0 (strsim.rs:128, :160)
                    code = lp == 0u ? 1u : (uint32_t)QTAB_N + 1u;
                } else {
                    const uint32_t lp1 = lp ? lp : 1u;
                    uint32_t dist;
                    if (wide) {
                        uint32_t P[7];
                        build_planes<7>(wp, P);
                        dist = lit_lev_uniform_text<7>(wt, litl, P, lp1);
                    } else {
                        uint32_t P[5];
                        build_planes<5>(wp, P);
                        dist = lit_lev_uniform_text<5>(wt, litl, P, lp1);
                    }
                    code = lp ? dist * (uint32_t)QTAB_N + (litl > lp ? litl : lp) : (uint32_t)QTAB_N + 1u; // an empty column string: 0.0
                }
                if (fast) STRSIM_CHECK_INDEX(K_LIT, 31, row0, i, B);
                if (fast) s_code[LEV ? bb : 0u][LEV ? i : 0u] = (uint16_t)code;
            } else {
                // text = the literal, pattern = the column string: exactly litl columns for every
                // lane, the bit fills of the uniform text on the scalar unit.  The integers as stage_ints packs them; an empty
                // side is caught by the epilogue's early-outs (strsim.rs:182-186, :288-292, :324-328).
                uint32_t dist = 0u, m = 0u, t = 0u, isect = 0u;
                if (litl != 0u) { // (uniform)
                    const uint32_t lp1 = lp ? lp : 1u;
                    const uint32_t tmax = (litl + (uint32_t)COLS_PER_TEST - 1u) / (uint32_t)COLS_PER_TEST * (uint32_t)COLS_PER_TEST;
                    if (wide) {
                        uint32_t P[7];
                        build_planes<7>(wp, P);
                        lane_cores32<7, false, JARO_LIKE, !JARO_LIKE>(wt, litl, litl, tmax, lp1, P, dist, m, t, isect);
                    } else {
                        uint32_t P[5];
                        build_planes<5>(wp, P);
                        lane_cores32<5, false, JARO_LIKE, !JARO_LIKE>(wt, litl, litl, tmax, lp1, P, dist, m, t, isect);
                    }
                }
                uint32_t pk;
                if (JARO_LIKE) {
                    const uint32_t pre = MEASURE == JARO_WINKLER && litl != 0u && lp != 0u ? common_prefix4(wt[0], litl, wp[0], lp) : 0u;
                    pk = m | (t << 6) | (litl << 12) | (lp << 18) | (pre << 24);
                } else {
                    pk = isect | (litl << 6) | (lp << 12);
                }
                if (fast) s_word[LEV ? 0u : bb][LEV ? 0u : i] = pk;
            }
        }
        __builtin_amdgcn_s_setprio(1);
        prev_row0 = row0;
        prev_rows = rows;
        prev_bb = bb;
        if (next_row0 >= row_end) break;
        row0 = next_row0; rows = nrows; base = nbase; mis = nmis; staged = nstaged; tail_lo = ntail_lo; tail_hi = ntail_hi;
    }
    lds_barrier();
    store_block(prev_row0, prev_rows, prev_bb);
    } // (this workgroup's rows)
    // Rows this workgroup left in the mask -> sched[2] (relaxed: k_publish_lit reads the sum behind this kernel, in stream order).
    // No "last workgroup" protocol here, unlike k_lane_stage: this kernel's many short-lived workgroups would each have to drain
    // their result stores before signalling (a release per workgroup: measured +7 % on the kernel); the end of the kernel does that
    // for free.
    lds_barrier();
    if (tid == 0u && s_left) __hip_atomic_fetch_add(&sched[2], s_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// behind k_lane_lit: lane_left (and, for a call that consists of that kernel alone, the ticket) to the host-mapped status block
__global__ void k_publish_lit(uint32_t *__restrict__ sched, DevStatus *__restrict__ publish, uint32_t ticket)
{
    if (threadIdx.x != 0u) return;
    const uint32_t total = __hip_atomic_load(&sched[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sched[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&publish->lane_left, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (ticket) __hip_atomic_store(&publish->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
