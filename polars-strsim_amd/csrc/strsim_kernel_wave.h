// strsim_kernel_wave.h -- k_wave_pairs<M>: everything else up to WAVE_CAP bytes -- Levenshtein as block-parallel Myers with DPP hand-off (up to 16 pairs per
// wave), the other measures one pair per wave -- and the stripe form of the block step that k_huge_pairs uses.
// Included by strsim_kernels.hip inside namespace strsim, after the kernels in front of it ([r5] split out of strsim_kernels.hip
// along its seams, VERDICT r4 item 8: no behaviour change -- the translation unit's ISA is byte-identical before and after).
// Reference semantics: /root/reference/src/expressions/strsim.rs:125-345 (the cores cite their lines).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_wave_pairs: one pair per wave, any UTF-8, strings up to WAVE_CAP bytes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ unsigned long long lanemask_lt(uint32_t lane) { return (1ull << lane) - 1ull; }

// number of scalar values of a valid UTF-8 string = its bytes that are not continuation bytes
__device__ __forceinline__ uint32_t wave_count_chars(const uint8_t *__restrict__ p, uint32_t len8)
{
    const uint32_t lane = lane_id();
    uint32_t cnt = 0;
    for (uint32_t c0 = 0; c0 < len8; c0 += 64u) {
        const uint32_t i = c0 + lane;
        cnt += (uint32_t)__popcll(__ballot(i < len8 && (p[i] & 0xC0u) != 0x80u));
    }
    return cnt;
}

// Levenshtein: a row with a string beyond WAVE_CAP BYTES still runs in k_wave_pairs' SYMBOLS batches when both strings
// have at most WAVE_CAP scalar values (e.g. 600 Cyrillic letters = 1 200 bytes); k_huge_pairs applies the same test
// and leaves such rows alone.
constexpr uint32_t LEV_SYMBOL_ROUTE_MAX_BYTES = 4u * WAVE_CAP;
__device__ __forceinline__ bool lev_fits_symbols(const uint8_t *__restrict__ pa, uint32_t la8, const uint8_t *__restrict__ pb,
                                                 uint32_t lb8)
{
    if (la8 > LEV_SYMBOL_ROUTE_MAX_BYTES || lb8 > LEV_SYMBOL_ROUTE_MAX_BYTES) return false;
    return wave_count_chars(pa, la8) <= (uint32_t)WAVE_CAP && wave_count_chars(pb, lb8) <= (uint32_t)WAVE_CAP;
}

// `str::chars()` across a wave: lane i looks at byte i, lead bytes decode their scalar value and
// compact it to dst[rank].  Valid UTF-8 only (the Rust &str / Arrow Utf8 contract).
__device__ __forceinline__ uint32_t wave_decode(const uint8_t *__restrict__ p, uint32_t len8, uint32_t *dst,
                                                bool &nonascii)
{
    // Four byte positions per lane and trip (256 bytes of the string per trip, two loads per lane, both issued before anything
    // waits), every position decoded without a branch: its byte and the three behind it lined up in one register, a continuation
    // byte taken only if the lead byte asks for it AND it lies inside the string (a sequence cut off by the end of the string
    // yields the bits it has, as the byte-per-lane loop of rounds 1-3 did -- which took a global round trip per 64 bytes and
    // another for the continuation bytes: long non-ASCII rows cost 34 us each).  A value goes to dst[number of lead bytes in
    // front of it].
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    const uint32_t lane = lane_id();
    uint32_t base = 0;
    bool high = false;
    for (uint32_t c0 = 0; c0 < len8; c0 += 256u) {
        const uint32_t i = c0 + 4u * lane;
        uint32_t d0 = 0u, d1 = 0u; // bytes i .. i + 7 (zeros behind the string)
        if (i + 8u <= len8) {
            d0 = *reinterpret_cast<const u32_unaligned *>(p + i);
            d1 = *reinterpret_cast<const u32_unaligned *>(p + i + 4u);
        } else if (i < len8) { // the string's last bytes, one by one: nothing is read behind a string
            for (uint32_t k = 0; k < 8u && i + k < len8; ++k) {
                const uint32_t by = p[i + k];
                if (k < 4u) d0 |= by << (8u * k);
                else d1 |= by << (8u * (k - 4u));
            }
        }
        uint32_t cp[4];
        bool lead[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // w: byte q of the lane's window in byte 0, the three bytes behind it in bytes 1..3
            const uint32_t w = q == 0 ? d0 : (uint32_t)__builtin_amdgcn_alignbyte(d1, d0, q);
            const uint32_t pos = i + (uint32_t)q;
            const uint32_t b0 = w & 0xFFu;
            const bool in = pos < len8;
            lead[q] = in && ((b0 & 0xC0u) != 0x80u);
            high = high || (in && b0 >= 0x80u);
            const uint32_t need = b0 < 0xC0u ? 0u : (b0 < 0xE0u ? 1u : (b0 < 0xF0u ? 2u : 3u)); // (0x80..0xBF are not leads)
            uint32_t v = b0 < 0x80u ? b0 : (b0 < 0xE0u ? (b0 & 0x1Fu) : (b0 < 0xF0u ? (b0 & 0x0Fu) : (b0 & 0x07u)));
#pragma unroll
            for (int k = 1; k <= 3; ++k)
                if ((uint32_t)k <= need && pos + (uint32_t)k < len8) v = (v << 6) | ((w >> (8 * k)) & 0x3Fu);
            cp[q] = v;
        }
        uint32_t before = base; // lead bytes in front of this lane's first position
        unsigned long long bal[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bal[q] = __ballot(lead[q]);
            before += (uint32_t)__popcll(bal[q] & lanemask_lt(lane));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (lead[q]) dst[before] = cp[q];
            before += lead[q] ? 1u : 0u;
            base += (uint32_t)__popcll(bal[q]);
        }
    }
    if (__ballot(high) != 0ull) nonascii = true;
    __syncthreads();
    return base;
}

// Levenshtein distance, anti-diagonal wavefront: lanes own 64 consecutive rows (chars of a) of a
// strip, the strip sweeps over the columns (chars of b) with lane l one step behind lane l-1;
// `up`/`diag` arrive from the lane below through a one-lane shift, lane 0 is fed from the previous
// strip's bottom row kept in LDS (brow), lane 63 writes the new bottom row in place.
__device__ __forceinline__ uint32_t wave_levenshtein(const uint32_t *sA, uint32_t la, const uint32_t *sB, uint32_t lb,
                                                     uint32_t *brow)
{
    const uint32_t lane = lane_id();
    for (uint32_t j = lane; j <= lb; j += 64u) brow[j] = j;
    __syncthreads();
    uint32_t result = 0;
    for (uint32_t s0 = 0; s0 < la; s0 += 64u) {
        const uint32_t nrows = (la - s0) < 64u ? (la - s0) : 64u;
        const uint32_t r = s0 + lane + 1u;
        const uint32_t pc = (r <= la) ? sA[r - 1u] : 0xFFFFFFFFu;
        uint32_t cur = r;                      // D[r][0]
        uint32_t upprev = s0, tcprev = 0u;     // lane 0's first diag is D[s0][0] = s0; other lanes get theirs by shift
        uint32_t qb = 0u, qt = 0u;             // conveyors feeding lane 0: previous bottom row / text
        const uint32_t nsteps = lb + nrows - 1u;
        for (uint32_t t = 0; t < nsteps; ++t) {
            const uint32_t k = t & 63u;
            if (k == 0u) { // refill lane 0's feed for the next 64 steps
                const uint32_t jb = t + 1u + lane, jt = t + lane;
                qb = (jb <= lb) ? brow[jb] : 0u;
                qt = (jt < lb) ? sB[jt] : 0xFFFFFFFEu;
            }
            uint32_t up = __shfl_up(cur, 1);
            uint32_t tc = __shfl_up(tcprev, 1);
            const uint32_t fb = (uint32_t)__builtin_amdgcn_readlane((int)qb, (int)k);
            const uint32_t ft = (uint32_t)__builtin_amdgcn_readlane((int)qt, (int)k);
            if (lane == 0u) {
                up = fb;
                tc = ft;
            }
            const uint32_t j = t + 1u - lane; // column (1-based); wraps for t+1 < lane
            if (j - 1u < lb) {
                const uint32_t sub = upprev + (pc != tc ? 1u : 0u);
                uint32_t nv = up + 1u;
                nv = nv < sub ? nv : sub;
                const uint32_t lf = cur + 1u;
                nv = nv < lf ? nv : lf;
                cur = nv;
                if (lane == 63u) brow[j] = nv;
            }
            upprev = up;
            tcprev = tc;
        }
        __syncthreads();
        if (s0 + 64u >= la) result = (uint32_t)__shfl((int)cur, (int)((la - 1u) & 63u));
    }
    return result;
}

// Levenshtein distance of ASCII strings, block-based bit-parallel DP (Myers 1999 / Hyyro 2003, the published
// "advanced block" step), up to LEV_JOBS pairs per wave, each in its own run of lanes (sum of runs <= 64).  Within
// a job, lane k owns rows 32k .. 32k+31 of the SHORTER string (<= 32 blocks = 1024 bytes) as bit-planes in registers,
// the LONGER string (bytes in LDS) supplies the columns -- that orientation needs the fewest lane-steps,
// ceil(m/32) * (n + ceil(m/32) - 1); block k works on column t-k at step t and hands its
// bottom-row delta (+1/0/-1) to block k+1 through a one-lane DPP shift.  The pattern is left-aligned to the
// top of its last block (window that ENDS at the end of the string), so every hand-off and the score row are
// bit 31.  max(n + B - 1) steps of ~31 VALU (five planes) for 32*B*n cells per job.
__device__ __forceinline__ uint32_t bfe_u32(uint32_t w, uint32_t off, uint32_t width)
{
    return (uint32_t)__builtin_amdgcn_ubfe(w, off, width); // width 0 -> 0
}

// One pair of a batch.  Two kinds of batches: BYTES (both strings ASCII: the pattern is read straight from its column,
// the text is staged as bytes) and SYMBOLS (any UTF-8 inside the Basic Multilingual Plane: both strings are decoded
// by the wave and staged as 16-bit scalar values).
struct BlockJob {
    uint32_t p0;      // BYTES: byte offset of the pattern (the SHORTER string) in its column
    uint32_t row;     // row of the frame
    uint16_t m, n;    // pattern / text length (bytes = scalar values for BYTES, scalar values for SYMBOLS)
    uint16_t txt;     // the staged text starts at unit 4 * txt of the batch's text arena (front pad included)
    uint8_t seg;      // first lane of the job's run of ceil(m / 32) lanes
    uint8_t in_a;     // BYTES: the pattern lives in column A
};

struct BlockCols { // the two value columns (BYTES batches read their patterns from them)
    const uint8_t *valA, *valB;
    uint32_t totalA, totalB;
};

// Match masks: the masks of the 32 possible values of the low five bits of a symbol are tabulated per lane in LDS when
// the batch starts (tab[code][lane], 8 KB per wave, each lane writes and reads its own column only), and a step
// reads its mask instead of computing it; the higher bits are compared with bit-planes on top of that, as many as
// vary inside the batch: none for a-z / A-Z / digits (NP = 5), two for general ASCII or one script block (NP = 7),
// up to bit 10 or bit 15 for mixed scripts / CJK (NP = 11, 16): 22 + 2 * (NP - 5) VALU per step.
template <int NP, bool SYMBOLS>
__device__ __forceinline__ void wave_lev_blocks(const BlockJob *jobs, uint32_t njobs, uint32_t T, const BlockCols &cols,
                                                const uint8_t *arena, const uint16_t *pats, uint32_t *tab,
                                                double *__restrict__ out)
{
    static_assert(NP == 5 || NP == 7 || NP == 11 || NP == 16, "five tabulated planes plus 0, 2, 6 or 11 computed ones");
    constexpr int UNIT = SYMBOLS ? 2 : 1; // bytes per text column
    const uint32_t lane = lane_id();
    uint32_t jdx = 0;
    for (uint32_t q = 1; q < njobs; ++q) jdx += lane >= (uint32_t)jobs[q].seg ? 1u : 0u;
    const uint32_t m = jobs[jdx].m, n = jobs[jdx].n;
    const uint32_t blk = lane - jobs[jdx].seg;
    const uint32_t B = (m + 31u) >> 5;
    const bool mine = blk < B; // lanes past the last job's run fall into it with blk >= B
    const uint32_t s = 32u * B - m; // fictitious shared-prefix rows at the bottom of block 0 (0..31)
    uint32_t P[NP];
    if (SYMBOLS) {
        // the 32 scalar values of this block were staged end-aligned (zeros in front of the string) at pats[lane * 32]
        uint32_t w[16];
        const uint4 *src = reinterpret_cast<const uint4 *>(pats + lane * 32u);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint4 v = mine ? src[d] : make_uint4(0u, 0u, 0u, 0u);
            w[4 * d] = v.x; w[4 * d + 1] = v.y; w[4 * d + 2] = v.z; w[4 * d + 3] = v.w;
        }
        build_planes_sym<NP>([&](int k) { return (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu; }, P);
    } else {
        const bool in_a = jobs[jdx].in_a != 0;
        uint32_t w[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) w[d] = 0u;
        if (mine)
            load_window_any<8>(in_a ? cols.valA : cols.valB, (int64_t)jobs[jdx].p0 + (int64_t)m - 32 * (int64_t)(B - blk),
                               in_a ? cols.totalA : cols.totalB, w);
        build_planes<NP>(w, P);
    }
    const uint32_t valid = blk == 0u ? ~low_ones(s) : 0xFFFFFFFFu;
    uint32_t Pv = valid, Mv = ~valid;
    // Hand-off of a block's bottom row to the block above it in the job (the next lane): the -1 as a bit (it joins the match
    // mask at bit 0), the +1 as the block's whole Ph word -- the neighbour takes its top bit with the v_alignbit that shifts its
    // own Ph.  The first block of a job (and lane 0, which the DPP shift fills with 0) ORs the top bit in: the row above block 0
    // grows by one per column, whatever the last block of the job in front left in its word.
    const uint32_t first_top = blk == 0u ? 0x80000000u : 0u;
    const uint32_t pubw = blk + 1u == B ? 0u : 1u;  // field width of the published -1 bit (the LAST block of a job publishes 0)
    uint32_t houtP = 0u, houtN = 0u;
    // Column j = t - blk of the text at step t.  A staged text has TXT_PAD units in front and behind (never used as
    // columns: a block is idle for blk <= 31 steps before its own columns), so a lane just walks on: four columns
    // per four steps, fetched two trips ahead with one unaligned load; once past its text (the batch runs for the
    // longest job) the fetch position stops at the rear pad.
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    typedef uint64_t u64_unaligned __attribute__((aligned(2)));
    const uint8_t *const col0 = arena + 4u * (uint32_t)jobs[jdx].txt + UNIT * TXT_PAD;
    const uint32_t ncol = mine ? n : 0u;
    const int32_t jlast = (int32_t)n + TXT_PAD - 4; // last position a four-column fetch may start at
    int32_t j = -(int32_t)blk;
    using word_t = typename std::conditional<SYMBOLS, uint64_t, uint32_t>::type; // four columns
    auto fetch = [&](int32_t at) -> word_t {
        at = at < jlast ? at : jlast;
        if constexpr (SYMBOLS) return *reinterpret_cast<const u64_unaligned *>(col0 + 2 * at);
        else return *reinterpret_cast<const u32_unaligned *>(col0 + at);
    };
    auto step = [&](uint32_t Eq0) {
        uint32_t hinP = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)houtP, 0x138 /* wave_shr:1 */, 0xF, 0xF, true) | first_top;
        asm volatile("" : "+v"(hinP)); // one v_or_b32_dpp (the compiler would move the OR behind the shift that uses it)
        const uint32_t hinN = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)houtN, 0x138, 0xF, 0xF, true);
        if ((uint32_t)j < ncol) {
            const uint32_t Xv = Eq0 | Mv;
            const uint32_t Eq = Eq0 | hinN;
            const uint32_t Xh = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq); // (sum ^ Pv) | Eq
            const uint32_t Ph = bitop3<0xF1>(Mv, Xh, Pv);             // Mv | ~(Xh | Pv)
            const uint32_t Mh = Pv & Xh;
            houtP = Ph;
            houtN = bfe_u32(Mh, 31u, pubw);
            const uint32_t PhS = (uint32_t)__builtin_amdgcn_alignbit(Ph, hinP, 31), MhS = (Mh << 1) | hinN;
            Pv = bitop3<0xF1>(MhS, Xv, PhS);                          // MhS | ~(Xv | PhS)
            Mv = PhS & Xv;
        }
        ++j;
    };
    word_t w0 = fetch(j), w1 = fetch(j + 4);
    uint32_t t = 0;
    // the split tables of strsim_lane_lut.h (12 entries, 3 KB per wave): Eq(c) = L[c & 7] & M[(c >> 3) & 3]
    EqLut tl;
    tl.lane4 = lane * 4u;
    tl.krep = (STRSIM_LDS_ADDR(tab) >> 8) * 0x01010101u;
    {
        uint32_t P5[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) P5[k] = P[k];
        lut_build<5>(tl, P5, valid);
    }
    // column q (0..3) of the fetched word: its 32-bit half and the bit its symbol starts at
    auto half = [&](word_t w, int q) { return SYMBOLS ? (uint32_t)((uint64_t)w >> (32 * (q >> 1))) : (uint32_t)w; };
    auto bit0 = [&](int q) { return SYMBOLS ? 16 * (q & 1) : 8 * q; };
    // mask of column q of w: the two table entries of its low five bits, then the higher planes
    auto high_planes = [&](uint32_t e, word_t w, int q) {
#pragma unroll
        for (int k = 5; k < NP; ++k) e = bitop3<0x90>(e, P[k], (uint32_t)__builtin_amdgcn_sbfe((int)half(w, q), bit0(q) + k, 1u));
        return e;
    };
    auto lookup = [&](const LutIndex &ix, int q) { // ix: the table coordinates of the bytes of half(w, q)
        const int byte = SYMBOLS ? 2 * (q & 1) : q;
        return lut_read(tl, ix.l, byte) & lut_read(tl, ix.m, byte);
    };
    auto trip = [&](word_t w) { // four steps on the four columns of w
        uint32_t e[4];
        const LutIndex ix0 = lut_index(tl, half(w, 0));
        const LutIndex ix1 = SYMBOLS ? lut_index(tl, half(w, 2)) : ix0;
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = lookup(q < 2 ? ix0 : ix1, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) step(high_planes(e[q], w, q));
    };
    __builtin_amdgcn_s_setprio(0); // the step loop yields to waves that are staging their next batch (loads to issue)
    // three words in flight, refilled in turn (no register is copied, so no load is waited for before its turn)
    word_t w2 = fetch(j + 8);
    for (; t + 12u <= T; t += 12u) {
        trip(w0); w0 = fetch(j + 8);
        trip(w1); w1 = fetch(j + 8);
        trip(w2); w2 = fetch(j + 8);
    }
    if (t + 4u <= T) {
        trip(w0); t += 4u; w0 = w1; w1 = w2;
        if (t + 4u <= T) { trip(w0); t += 4u; w0 = w1; }
    }
    for (; t < T; ++t, w0 >>= 8 * UNIT) step(high_planes(lookup(lut_index(tl, (uint32_t)w0), 0), w0, 0));
    __builtin_amdgcn_s_setprio(1);
    // No running score: every block stops updating after its last column, so once all are done the column-n vertical
    // deltas are in Pv/Mv.  The row above block 0 sits at s + n (it starts at s because the s fictitious rows below it
    // start at -1 each), hence  D[m][n] = s + n + sum over the job's blocks of popc(Pv) - popc(Mv).
    int pre = mine ? (int)popc32(Pv) - (int)popc32(Mv) : 0; // -> inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(pre, d);
        if (lane >= (uint32_t)d) pre += up;
    }
    // lane q finishes job q
    const uint32_t q = lane < njobs ? lane : 0u;
    const uint32_t qseg = jobs[q].seg, qm = jobs[q].m, qn = jobs[q].n;
    const uint32_t qB = (qm + 31u) >> 5;
    const int hi_sum = __shfl(pre, (int)(qseg + qB - 1u));
    const int lo_sum = __shfl(pre, (int)(qseg ? qseg - 1u : 0u));
    if (lane < njobs) {
        const int sum = hi_sum - (qseg ? lo_sum : 0);
        const uint32_t dist = (uint32_t)((int)(32u * qB - qm + qn) + sum);
        out[jobs[q].row] = epilogue_levenshtein(dist, qm, qn);
    }
}

// wave_lev_blocks for BYTES batches with SIXTY-FOUR pattern rows per lane (two mask words): what a step spends per lane
// whatever the word holds -- the hand-off through DPP, its decoding and publishing, the column test, the table addresses --
// is paid once per 64 cells instead of once per 32 (about 12 of the 23 instructions of the one-word step; the two-word
// step is 30 + 3.5 for the table addresses and the text fetch: 0.7 of two one-word steps).  A job takes ceil(m / 64)
// lanes.  The match masks come from the split tables of strsim_lane_lut.h (L[c & 7] & M[(c >> 3) & 3], 12 entries per
// word): the 32-entry table of the one-word form would be 16 KB per wave.  Low words at tab (4 KB-aligned), high words
// 3 KB above; one address (a v_perm_b32) serves both.  Same recurrences, hand-off and distance formula as above.
template <int NP>
__device__ __forceinline__ void wave_lev_blocks64(const BlockJob *jobs, uint32_t njobs, uint32_t T, const BlockCols &cols,
                                                  const uint8_t *arena, uint32_t *tab, double *__restrict__ out)
{
    static_assert(NP == 5 || NP == 7, "ASCII: five tabulated planes plus 0 or 2 computed ones");
    const uint32_t lane = lane_id();
    uint32_t jdx = 0;
    for (uint32_t q = 1; q < njobs; ++q) jdx += lane >= (uint32_t)jobs[q].seg ? 1u : 0u;
    const uint32_t m = jobs[jdx].m, n = jobs[jdx].n;
    const uint32_t blk = lane - jobs[jdx].seg;
    const uint32_t B = (m + 63u) >> 6;
    const bool mine = blk < B;
    const uint32_t s = 64u * B - m; // fictitious shared-prefix rows at the bottom of block 0 (0..63)
    uint32_t Plo[NP], Phi[NP];
    {
        const bool in_a = jobs[jdx].in_a != 0;
        uint32_t w[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) w[d] = 0u;
        if (mine)
            load_window_any<16>(in_a ? cols.valA : cols.valB, (int64_t)jobs[jdx].p0 + (int64_t)m - 64 * (int64_t)(B - blk),
                                in_a ? cols.totalA : cols.totalB, w);
        uint32_t h[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) h[d] = w[d];
        build_planes<NP>(h, Plo);
#pragma unroll
        for (int d = 0; d < 8; ++d) h[d] = w[8 + d];
        build_planes<NP>(h, Phi);
    }
    const uint64_t valid = blk == 0u ? ~((1ull << s) - 1ull) : ~0ull;
    uint64_t Pv = valid, Mv = ~valid;
    const uint32_t first = blk == 0u ? 1u : 0u;
    const uint32_t pubw = blk + 1u == B ? 0u : 1u;
    // Hand-off: the +1 and the -1 leaving a block's bottom row travel in two registers, so neither side packs or unpacks
    // anything.  The +1 is not even extracted: the neighbour fetches the block's whole top Ph word (one v_or_b32_dpp that also
    // sets the top bit for the first block of a job -- whatever the last block of the job in front of it left there) and
    // shifts its top bit into its own Ph with the v_alignbit that shifts Ph anyway.  The -1 is needed at bit 0 as well (it joins
    // the match mask), so it is published as a bit.
    uint32_t houtP = 0u, houtN = 0u;
    const uint32_t first_top = first << 31;
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    const uint32_t txt0 = 4u * (uint32_t)jobs[jdx].txt + TXT_PAD; // this job's column 0 in the arena (uniform base + 32-bit offset)
    const uint32_t ncol = mine ? n : 0u;
    const uint32_t end = mine ? blk + n : 0u;   // first step at which this block has no column left
    const int32_t jlast = (int32_t)n + TXT_PAD - 4;
    uint32_t tt = 0u;                            // the step (uniform); this block's column is tt - blk
    const uint32_t fetch0 = txt0 - blk, fetch_last = txt0 + (uint32_t)jlast; // (blk <= 15 < TXT_PAD <= txt0)
    auto fetch = [&](uint32_t ahead) -> uint32_t { // columns tt + ahead - blk .. + 3 of this lane's text, not past its rear pad
        const uint32_t at = tt + ahead + fetch0;
        return *reinterpret_cast<const u32_unaligned *>(arena + (at < fetch_last ? at : fetch_last));
    };
    // the tables: this lane's column of the low-word table at tab, of the high-word table 3 KB above
    EqLut tl, th;
    tl.lane4 = th.lane4 = lane * 4u;
    tl.krep = (STRSIM_LDS_ADDR(tab) >> 8) * 0x01010101u;
    th.krep = ((STRSIM_LDS_ADDR(tab) >> 8) + (uint32_t)LUT_ENTRIES) * 0x01010101u;
    lut_build<NP>(tl, Plo, (uint32_t)valid);
    lut_build<NP>(th, Phi, (uint32_t)(valid >> 32));
    struct Col { uint32_t llo, lhi, mlo, mhi; };
    auto lookup = [&](const LutIndex &ix, uint32_t w, int q) {
        const uint32_t al = __builtin_amdgcn_perm(ix.l, tl.lane4, 0x0C0C0000u | ((4u + (uint32_t)q) << 8));
        const uint32_t am = __builtin_amdgcn_perm(ix.m, tl.lane4, 0x0C0C0000u | ((4u + (uint32_t)q) << 8));
        Col c;
#if defined(__HIP_DEVICE_COMPILE__)
        c.llo = lut_lds_read(al); c.lhi = lut_lds_read(al + (uint32_t)LUT_WAVE_BYTES);
        c.mlo = lut_lds_read(am); c.mhi = lut_lds_read(am + (uint32_t)LUT_WAVE_BYTES);
#else
        c.llo = c.lhi = al; c.mlo = c.mhi = am; // (host pass of hipcc: never run)
#endif
#pragma unroll
        for (int k = 5; k < NP; ++k) { // planes 5.. on top of M: m & ~(P_k ^ bit k of the column's byte)
            const uint32_t fill = bit_fill(w, 8 * q + k);
            c.mlo = bitop3<0x90>(c.mlo, Plo[k < NP ? k : 0], fill);
            c.mhi = bitop3<0x90>(c.mhi, Phi[k < NP ? k : 0], fill);
        }
        return c;
    };
    // RAMP: the first steps, while blocks are still starting (block k runs columns from step k on); after step 16 every
    // block has started and "has a column left" is one compare against the uniform step.
    auto step = [&](const Col &c, auto ramp) {
        uint32_t hinP = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)houtP, 0x138 /* wave_shr:1 */, 0xF, 0xF, true) | first_top;
        asm volatile("" : "+v"(hinP)); // keeps the OR next to the move (one v_or_b32_dpp) instead of behind the shift that uses it
        const uint32_t hinN = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)houtN, 0x138, 0xF, 0xF, true);
        const bool on = decltype(ramp)::value ? (tt - blk) < ncol : tt < end;
        if (on) {
            const uint32_t Pvl = (uint32_t)Pv, Pvh = (uint32_t)(Pv >> 32), Mvl = (uint32_t)Mv, Mvh = (uint32_t)(Mv >> 32);
            const uint32_t Xvl = bitop3<0xEA>(c.llo, c.mlo, Mvl), Xvh = bitop3<0xEA>(c.lhi, c.mhi, Mvh); // Eq0 | Mv
            const uint32_t Eql = bitop3<0xEA>(c.llo, c.mlo, hinN), Eqh = c.lhi & c.mhi;                    // Eq0 | hinN
            uint64_t masked = ((uint64_t)(Eqh & Pvh) << 32) | (Eql & Pvl);
            asm volatile("" : "+v"(masked)); // one 64-bit add on the register pair (left alone the compiler adds the halves apart: + 1)
            const uint64_t sum = masked + Pv;
            const uint32_t Xhl = bitop3<0xBE>((uint32_t)sum, Pvl, Eql), Xhh = bitop3<0xBE>((uint32_t)(sum >> 32), Pvh, Eqh);
            const uint32_t Phl = bitop3<0xF1>(Mvl, Xhl, Pvl), Phh = bitop3<0xF1>(Mvh, Xhh, Pvh);
            const uint32_t Mhl = Pvl & Xhl, Mhh = Pvh & Xhh;
            houtP = Phh;
            houtN = bfe_u32(Mhh, 31u, pubw);
            const uint32_t PhSl = (uint32_t)__builtin_amdgcn_alignbit(Phl, hinP, 31), PhSh = (uint32_t)__builtin_amdgcn_alignbit(Phh, Phl, 31);
            const uint32_t MhSl = (Mhl << 1) | hinN, MhSh = (uint32_t)__builtin_amdgcn_alignbit(Mhh, Mhl, 31);
            Pv = ((uint64_t)bitop3<0xF1>(MhSh, Xvh, PhSh) << 32) | bitop3<0xF1>(MhSl, Xvl, PhSl);
            Mv = ((uint64_t)(PhSh & Xvh) << 32) | (PhSl & Xvl);
        }
        ++tt;
    };
    uint32_t w0 = fetch(0u), w1 = fetch(4u);
    auto trip = [&](uint32_t w, auto ramp) { // four steps on the four columns of w
        const LutIndex ix = lut_index(tl, w);
        Col c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = lookup(ix, w, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) step(c[q], ramp);
    };
    __builtin_amdgcn_s_setprio(0);
    uint32_t w2 = fetch(8u);
    // three words in flight, refilled in turn (before a trip, the word two trips ahead of it is the one fetched: + 8 columns)
    const std::true_type ramp_on{};
    const std::false_type ramp_off{};
    for (; tt < 24u && tt + 12u <= T;) {
        trip(w0, ramp_on); w0 = fetch(8u);
        trip(w1, ramp_on); w1 = fetch(8u);
        trip(w2, ramp_on); w2 = fetch(8u);
    }
    for (; tt + 12u <= T;) {
        trip(w0, ramp_off); w0 = fetch(8u);
        trip(w1, ramp_off); w1 = fetch(8u);
        trip(w2, ramp_off); w2 = fetch(8u);
    }
    if (tt + 4u <= T) {
        trip(w0, ramp_on); w0 = w1; w1 = w2;
        if (tt + 4u <= T) { trip(w0, ramp_on); w0 = w1; }
    }
    for (; tt < T; w0 >>= 8) step(lookup(lut_index(tl, w0), w0, 0), ramp_on);
    __builtin_amdgcn_s_setprio(1);
    int pre = mine ? (int)popc32((uint32_t)Pv) + (int)popc32((uint32_t)(Pv >> 32)) - (int)popc32((uint32_t)Mv) -
                         (int)popc32((uint32_t)(Mv >> 32))
                   : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(pre, d);
        if (lane >= (uint32_t)d) pre += up;
    }
    const uint32_t q = lane < njobs ? lane : 0u;
    const uint32_t qseg = jobs[q].seg, qm = jobs[q].m, qn = jobs[q].n;
    const uint32_t qB = (qm + 63u) >> 6;
    const int hi_sum = __shfl(pre, (int)(qseg + qB - 1u));
    const int lo_sum = __shfl(pre, (int)(qseg ? qseg - 1u : 0u));
    if (lane < njobs) {
        const int sum = hi_sum - (qseg ? lo_sum : 0);
        const uint32_t dist = (uint32_t)((int)(64u * qB - qm + qn) + sum);
        out[jobs[q].row] = epilogue_levenshtein(dist, qm, qn);
    }
}

// The same block step for ONE pair of any length (k_huge_pairs): both strings decoded to 32-bit scalar values in the
// global workspace, the pattern (the one with fewer values) cut into stripes of 64 blocks = 2048 rows that run one
// after the other over all the columns.  The bottom-row deltas of a stripe go through a byte array hb[column]
// (global, written by the stripe's last block 63 steps after block 0 of the same stripe consumed the entry the
// previous stripe left there); the distance is s + n + the vertical deltas of every block at its last column.
// txt[-TXT_PAD .. n + TXT_PAD) and hb[-TXT_PAD .. n + TXT_PAD) must be readable.  Scalar values <= 0xFFFF.
template <int NP>
__device__ __forceinline__ uint32_t wave_lev_stripes(const uint32_t *pat, uint32_t m, const uint32_t *txt, uint32_t n,
                                                     uint8_t *hb, uint32_t *tab)
{
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    struct u32x4 { uint32_t v[4]; };
    const uint32_t lane = lane_id();
    const uint32_t Btot = (m + 31u) >> 5;
    const uint32_t s = 32u * Btot - m;
    int total = 0;
    uint32_t *const trow = tab + lane;
    const int32_t jlast = (int32_t)n + TXT_PAD - 4;
    for (uint32_t g0 = 0; g0 < Btot; g0 += 64u) {
        const uint32_t nb = Btot - g0 < 64u ? Btot - g0 : 64u;
        const bool mine = lane < nb;
        const uint32_t g = g0 + lane;
        const bool later = g0 != 0u;            // block 0 of this stripe is fed from hb[]
        const bool more = g0 + 64u < Btot;      // the last block of this stripe feeds hb[]
        uint32_t w[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int64_t idx = (int64_t)g * 32 - (int64_t)s + k;
            w[k] = (mine && idx >= 0) ? pat[idx] : 0u;
        }
        uint32_t P[NP];
        build_planes_sym<NP>([&](int k) { return w[k] & 0xFFFFu; }, P);
        const uint32_t valid = g == 0u ? ~low_ones(s) : 0xFFFFFFFFu;
        uint32_t Pv = valid, Mv = ~valid;
        const uint32_t first = g == 0u ? 1u : 0u;
        const uint32_t pubw = g + 1u == Btot ? 0u : 1u;
        const uint32_t pubn = g + 1u == Btot ? 0u : 2u;
        uint32_t hout = 0u;
        {
            uint32_t P5[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) P5[k] = P[k];
#pragma unroll
            for (int code = 0; code < 32; ++code) trow[code * 64] = eq_mask<5>(P5, valid, (uint32_t)code, 0);
        }
        const uint32_t ncol = mine ? n : 0u;
        const bool feeder = more && lane + 1u == nb;
        int32_t j = -(int32_t)lane;
        auto clampj = [&](int32_t at) { return at < jlast ? at : jlast; };
        auto fetch_t = [&](int32_t at) { return *reinterpret_cast<const u32x4 *>(txt + clampj(at)); };
        auto fetch_h = [&](int32_t at) { return *reinterpret_cast<const u32_unaligned *>(hb + clampj(at)); };
        const uint32_t T = n + nb - 1u;
        auto trip = [&](const u32x4 &wt, uint32_t wh, uint32_t t) { // four steps on the four columns of wt
            uint32_t e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = trow[(wt.v[q] & 31u) * 64u];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t hin = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hout, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
                if (later && lane == 0u) hin = (wh >> (8 * q)) & 3u;
                if (t + (uint32_t)q < T && (uint32_t)j < ncol) {
                    uint32_t Eq0 = e[q];
#pragma unroll
                    for (int k = 5; k < NP; ++k) Eq0 = bitop3<0x90>(Eq0, P[k], bit_fill(wt.v[q], k));
                    const uint32_t hinP = (hin & 1u) | first, hinN = hin >> 1;
                    const uint32_t Xv = Eq0 | Mv;
                    const uint32_t Eq = Eq0 | hinN;
                    const uint32_t Xh = bitop3<0xBE>((Eq & Pv) + Pv, Pv, Eq);
                    const uint32_t Ph = bitop3<0xF1>(Mv, Xh, Pv);
                    const uint32_t Mh = Pv & Xh;
                    hout = bfe_u32(Ph, 31u, pubw) | ((Mh >> 30) & pubn);
                    const uint32_t PhS = (Ph << 1) | hinP, MhS = (Mh << 1) | hinN;
                    Pv = bitop3<0xF1>(MhS, Xv, PhS);
                    Mv = PhS & Xv;
                    if (feeder) hb[j] = (uint8_t)hout;
                }
                ++j;
            }
        };
        // two words in flight, refilled in turn one trip ahead of their use
        u32x4 ta = fetch_t(j), tb = fetch_t(j + 4);
        uint32_t ha = later ? fetch_h(j) : 0u, hc = later ? fetch_h(j + 4) : 0u; // only lane 0 uses them
        for (uint32_t t = 0; t < T; t += 8u) {
            trip(ta, ha, t);
            ta = fetch_t(j + 4);
            if (later) ha = fetch_h(j + 4);
            if (t + 4u < T) {
                trip(tb, hc, t + 4u);
                tb = fetch_t(j + 4);
                if (later) hc = fetch_h(j + 4);
            }
        }
        int v = mine ? (int)popc32(Pv) - (int)popc32(Mv) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        total += v;
        __syncthreads(); // hb[] written by this stripe is read by the next one (same wave; see flush_bytes)
    }
    return (uint32_t)((int)(s + n) + total);
}

// Copy the ASCII string p[0, len) into LDS bytes and/or just test it: returns true when every byte is < 0x80.
// or6/and6 accumulate, wave-uniformly, whether bits 5 and 6 are set in any / in every byte (plane-count choice).
// One pass over the two strings of a pair: are all their bytes ASCII, which of bits 5 / 6 vary over them (or6 / and6
// accumulate over the jobs of a batch), and the text copied to its arena slot `dst` (4-aligned).
__device__ __forceinline__ bool wave_ascii_stage(const uint8_t *__restrict__ pat, uint32_t mlen, const uint8_t *__restrict__ txt,
                                                 uint32_t nlen, uint8_t *dst, uint32_t &or6, uint32_t &and6)
{
    // Four bytes per lane and trip (a string of 1 024 bytes: 4 trips; both strings in the same trip, so their loads are in
    // flight together), OR / AND of the bytes kept per lane and reduced over the wave once at the end.  Until late in round 3:
    // a pass per string, a byte per lane and five ballots per trip -- a tenth of k_wave_pairs<levenshtein>'s instructions on
    // cfg5 and, one row after the other, a latency chain per row (207 -> 241 M pairs/s).  Staged, the last dword of a text
    // reaches up to 3 bytes into its rear pad.
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    const uint32_t lane = lane_id();
    uint32_t o = 0u, a = 0xFFFFFFFFu;
    auto dword = [&](const uint8_t *p, uint32_t len, uint32_t i) {
        const uint32_t rem = len - i;
        uint32_t w;
        if (rem >= 4u) {
            w = *reinterpret_cast<const u32_unaligned *>(p + i);
        } else { // the string's last one to three bytes, one by one: nothing is read behind a string
            w = p[i];
            if (rem > 1u) w |= (uint32_t)p[i + 1u] << 8;
            if (rem > 2u) w |= (uint32_t)p[i + 2u] << 16;
        }
        const uint32_t keep = rem >= 4u ? 0xFFFFFFFFu : ((1u << (8u * rem)) - 1u);
        o |= w & keep;
        a &= w | ~keep;
        return w;
    };
    const uint32_t longest = mlen > nlen ? mlen : nlen;
    for (uint32_t c0 = 0; c0 < longest; c0 += 256u) {
        const uint32_t i = c0 + 4u * lane;
        if (i < nlen) *reinterpret_cast<uint32_t *>(dst + i) = dword(txt, nlen, i);
        if (i < mlen) (void)dword(pat, mlen, i);
    }
    if (__ballot((o & 0x20202020u) != 0u) != 0ull) or6 |= 0x20u;
    if (__ballot((o & 0x40404040u) != 0u) != 0ull) or6 |= 0x40u;
    if (__ballot((a & 0x20202020u) != 0x20202020u) != 0ull) and6 &= ~0x20u;
    if (__ballot((a & 0x40404040u) != 0x40404040u) != 0ull) and6 &= ~0x40u;
    return __ballot((o & 0x80808080u) != 0u) == 0ull;
}

// In-place compaction of the flagged entries of s[0..len) to the front; returns how many.
__device__ __forceinline__ uint32_t wave_compact(uint32_t *s, const uint8_t *flag, uint32_t len)
{
    const uint32_t lane = lane_id();
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < len; c0 += 64u) {
        const uint32_t i = c0 + lane;
        const bool f = i < len && flag[i] != 0;
        const uint32_t v = f ? s[i] : 0u;
        const unsigned long long bal = __ballot(f);
        __syncthreads();
        if (f) s[base + (uint32_t)__popcll(bal & lanemask_lt(lane))] = v;
        base += (uint32_t)__popcll(bal);
        __syncthreads();
    }
    return base;
}

// Jaro matching on scalar values (strsim.rs:200-237): a is walked sequentially, the window of b is
// scanned 64 positions at a time and the lowest hit wins (ballot + ctz).  Destroys sA/sB.
// In-place compaction of the entries of s[0..len) whose bit is set in the bit array flagw[] (bit i of word i / 32).
__device__ __forceinline__ uint32_t wave_compact_bits(uint32_t *s, const uint32_t *flagw, uint32_t len)
{
    const uint32_t lane = lane_id();
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < len; c0 += 64u) {
        const uint32_t i = c0 + lane;
        const bool f = i < len && ((flagw[i >> 5] >> (i & 31u)) & 1u) != 0u;
        const uint32_t v = f ? s[i] : 0u;
        const unsigned long long bal = __ballot(f);
        __syncthreads();
        if (f) s[base + (uint32_t)__popcll(bal & lanemask_lt(lane))] = v;
        base += (uint32_t)__popcll(bal);
        __syncthreads();
    }
    return base;
}

// Jaro matching of one pair per wave, bit-parallel across the lanes (strsim.rs:200-237): lane k keeps positions
// 32k .. 32k+31 of b -- its flags, the two window masks and the bit-planes of its 32 scalar values -- in registers
// (lb <= 2048, values <= 0xFFFF, NP > highest bit that varies in the pair).  The characters of a are walked in order;
// a character is wave-uniform, so its plane bits are scalar (s_bfe) and the match mask costs one v_bitop3 per plane;
// the window masks slide by one position per step with a one-lane DPP carry; the first unflagged equal position in
// the window is the lowest set bit of the lowest lane with a candidate (ballot + s_ff1).  Flags of a are collected
// 32 per scalar word.  Transpositions as in the reference: the flagged values of both sides, compacted, compared
// in order.  aw: >= la / 32 + 64 + 64 words of scratch.
template <int NP>
__device__ __forceinline__ void wave_jaro_bits(uint32_t *sA, uint32_t la, uint32_t *sB, uint32_t lb, uint32_t *aw,
                                               uint32_t &m_out, uint32_t &t_out)
{
    const uint32_t lane = lane_id();
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u;
    uint32_t *faw = aw;                          // flags of a, one bit per position
    uint32_t *fbw = aw + ((la + 31u) >> 5) + 1u; // flags of b (64 words), written at the end
    // planes of this lane's 32 values of b
    uint32_t P[NP];
    {
        uint32_t w[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const uint32_t j = lane * 32u + (uint32_t)k;
            w[k] = j < lb ? sB[j] : 0u;
        }
        build_planes_sym<NP>([&](int k) { return w[k] & 0xFFFFu; }, P);
    }
    const uint32_t base = lane * 32u;
    const uint32_t lbmask = lb <= base ? 0u : low_ones(lb - base);   // positions of b in this block
    uint32_t hw = bound + 1u < lb ? bound + 1u : lb;                // himask: ones at [0, min(i + bound, lb - 1)]
    uint32_t himask = hw <= base ? 0u : low_ones(hw - base);
    uint32_t lomask = 0u;                                           // ones below max(0, i - bound)
    uint32_t fb = 0u;
    const uint32_t first = lane == 0u ? 1u : 0u;
    uint32_t m = 0, fa_acc = 0;
    const uint32_t ni = la < lb + bound ? la : lb + bound; // `.take(b.len() + bound)` (:208)
    uint32_t c_next = ni ? sA[0] : 0u;
    for (uint32_t i = 0; i < ni; ++i) {
        const uint32_t c = uniform(c_next);
        c_next = sA[i + 1u < la ? i + 1u : i];
        uint32_t Eq = lbmask;
#pragma unroll
        for (int k = 0; k < NP; ++k) Eq = bitop3<0x90>(Eq, P[k], 0u - ((c >> k) & 1u)); // scalar plane bit
        const uint32_t cand = bitop3<0x20>(Eq & himask, lomask | fb, 0xFFFFFFFFu);      // a & ~b
        const unsigned long long bal = __ballot(cand != 0u);
        if (bal != 0ull) {
            const uint32_t wl = (uint32_t)__builtin_ctzll(bal);
            if (lane == wl) fb |= cand & (0u - cand);
            ++m;
            fa_acc |= 1u << (i & 31u);
        }
        if ((i & 31u) == 31u) {
            if (lane == 0u) faw[i >> 5] = fa_acc;
            fa_acc = 0u;
        }
        // slide the window: both masks shift up by one position, ones enter at the bottom
        {
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)himask, 0x138, 0xF, 0xF, true);
            himask = (__builtin_amdgcn_alignbit(himask, prev, 31) | first) & lbmask;
        }
        if (i >= bound) {
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lomask, 0x138, 0xF, 0xF, true);
            lomask = __builtin_amdgcn_alignbit(lomask, prev, 31) | first;
        }
    }
    // flush the flags (the tail of a beyond ni is unflagged)
    {
        const uint32_t nw = (la + 31u) >> 5;
        const uint32_t done = ni >> 5; // whole words already written
        if (lane == 0u && (ni & 31u) != 0u) faw[done] = fa_acc;
        const uint32_t from = done + ((ni & 31u) != 0u ? 1u : 0u);
        for (uint32_t w = from + lane; w < nw; w += 64u) faw[w] = 0u;
        fbw[lane] = fb;
    }
    __syncthreads();
    wave_compact_bits(sA, faw, la);
    wave_compact_bits(sB, fbw, lb);
    uint32_t t = 0;
    for (uint32_t k0 = 0; k0 < m; k0 += 64u) {
        const uint32_t k = k0 + lane;
        t += (uint32_t)__popcll(__ballot(k < m && sA[k] != sB[k]));
    }
    m_out = m;
    t_out = t;
}

__device__ __forceinline__ void wave_jaro(uint32_t *sA, uint32_t la, uint32_t *sB, uint32_t lb, uint8_t *fa, uint8_t *fb,
                                          uint32_t &m_out, uint32_t &t_out)
{
    const uint32_t lane = lane_id();
    const uint32_t mx = la > lb ? la : lb;
    const uint32_t half = mx >> 1;
    const uint32_t bound = (half ? half : 1u) - 1u;
    for (uint32_t i = lane; i < la; i += 64u) fa[i] = 0;
    for (uint32_t j = lane; j < lb; j += 64u) fb[j] = 0;
    __syncthreads();
    uint32_t m = 0;
    const uint32_t ni = la < lb + bound ? la : lb + bound; // `.take(b.len() + bound)` (:208)
    for (uint32_t i = 0; i < ni; ++i) {
        const uint32_t lo = i > bound ? i - bound : 0u;
        const uint32_t hi = (i + bound) < (lb - 1u) ? (i + bound) : (lb - 1u);
        if (lo > hi) continue;
        const uint32_t ai = sA[i];
        for (uint32_t cb = lo & ~63u; cb <= hi; cb += 64u) {
            const uint32_t j = cb + lane;
            const bool ok = j >= lo && j <= hi && sB[j] == ai && fb[j] == 0;
            const unsigned long long bal = __ballot(ok);
            if (bal != 0ull) {
                const uint32_t jm = cb + (uint32_t)__builtin_ctzll(bal);
                if (lane == 0u) { fb[jm] = 1; fa[i] = 1; }
                ++m;
                break;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    wave_compact(sA, fa, la);
    wave_compact(sB, fb, lb);
    uint32_t t = 0;
    for (uint32_t k0 = 0; k0 < m; k0 += 64u) {
        const uint32_t k = k0 + lane;
        t += (uint32_t)__popcll(__ballot(k < m && sA[k] != sB[k]));
    }
    m_out = m;
    t_out = t;
}

// Multiset intersection size.  ASCII: two 128-bin LDS histograms.  Otherwise rank counting: the
// k-th occurrence (0-based) of a scalar value in a is matched iff k < its count in b.
// Multiset intersection of two strings of scalar values with an open-addressing hash table (linear probing) of the
// SHORTER string's values: entry = value << CB | count.  The other string then walks the table and every value that
// finds a count left takes one unit (CAS), which is exactly sum_c min(countA[c], countB[c]).  E = u32 with CB = 11
// (counts <= 1024: k_wave_pairs) or u64 with CB = 32 (any length: k_huge_pairs).  `size` (a power of two) entries at
// `tab`; returns false if an insertion found the table full (the caller falls back to counting).
template <class E, int CB>
__device__ __forceinline__ bool wave_isect_hash(const uint32_t *S, uint32_t ns, const uint32_t *L, uint32_t nl, E *tab,
                                                uint32_t bits, uint32_t &isect)
{
    const uint32_t lane = lane_id();
    const uint32_t size = 1u << bits, mask = size - 1u;
    const E EMPTY = ~(E)0; // value field all ones: not a scalar value
    for (uint32_t c = lane; c < size; c += 64u) tab[c] = EMPTY;
    __syncthreads();
    bool failed = false;
    for (uint32_t i = lane; i < ns; i += 64u) {
        const uint32_t sym = S[i];
        uint32_t h = (sym * 2654435761u) >> (32u - bits);
        uint32_t probe = 0;
        for (; probe < size; ++probe, h = (h + 1u) & mask) {
            E cur = *reinterpret_cast<volatile E *>(tab + h);
            if (cur == EMPTY) cur = atomicCAS(tab + h, EMPTY, ((E)sym << CB) | (E)1);
            if (cur == EMPTY) break;                                             // claimed the slot, count 1
            if ((uint32_t)(cur >> CB) == sym) { atomicAdd(tab + h, (E)1); break; } // this value's slot
        }
        failed = failed || probe == size;
    }
    __syncthreads();
    if (__ballot(failed) != 0ull) return false;
    uint32_t acc = 0;
    for (uint32_t j = lane; j < nl; j += 64u) {
        const uint32_t sym = L[j];
        uint32_t h = (sym * 2654435761u) >> (32u - bits);
        for (uint32_t probe = 0; probe < size; ++probe, h = (h + 1u) & mask) {
            E cur = *reinterpret_cast<volatile E *>(tab + h);
            if (cur == EMPTY) break; // not in the shorter string
            if ((uint32_t)(cur >> CB) != sym) continue;
            while ((cur & (((E)1 << CB) - (E)1)) != (E)0) { // take one unit if any is left
                const E old = atomicCAS(tab + h, cur, cur - (E)1);
                if (old == cur) { ++acc; break; }
                cur = old;
            }
            break;
        }
    }
    isect = wave_sum(acc);
    __syncthreads();
    return true;
}

// auxwords = words available at hist (>= 256)
__device__ __forceinline__ uint32_t wave_multiset_isect(const uint32_t *sA, uint32_t la, const uint32_t *sB, uint32_t lb,
                                                        uint32_t *hist, uint32_t auxwords, bool nonascii)
{
    const uint32_t lane = lane_id();
    uint32_t acc = 0;
    if (!nonascii) {
        for (uint32_t c = lane; c < 256u; c += 64u) hist[c] = 0u;
        __syncthreads();
        for (uint32_t i = lane; i < la; i += 64u) atomicAdd(&hist[sA[i]], 1u);
        for (uint32_t j = lane; j < lb; j += 64u) atomicAdd(&hist[128u + sB[j]], 1u);
        __syncthreads();
        for (uint32_t c = lane; c < 128u; c += 64u) {
            const uint32_t x = hist[c], y = hist[128u + c];
            acc += x < y ? x : y;
        }
    } else {
        const bool a_short = la <= lb;
        const uint32_t *S = a_short ? sA : sB, *L = a_short ? sB : sA;
        const uint32_t ns = a_short ? la : lb, nl = a_short ? lb : la;
        uint32_t isect = 0;
        bool ok;
        // at least two entries per value of the shorter string (load <= 1/2), no more than fit: a small table is
        // cleared faster
        uint32_t want = 32u - (uint32_t)__builtin_clz(2u * ns - 1u);
        want = want < 6u ? 6u : want;
        if (ns <= 1024u) { // counts fit 11 bits: one word per entry
            uint32_t bits = 31u - (uint32_t)__builtin_clz(auxwords);
            bits = bits < want ? bits : want;
            ok = wave_isect_hash<unsigned int, 11>(S, ns, L, nl, hist, bits, isect);
        } else {
            uint32_t bits = 31u - (uint32_t)__builtin_clz(auxwords >> 1);
            bits = bits < want ? bits : want;
            ok = wave_isect_hash<unsigned long long, 32>(S, ns, L, nl, reinterpret_cast<unsigned long long *>(hist), bits, isect);
        }
        if (ok) return isect;
        // table full (more distinct values than entries): count, a chunk of `a` at a time
        for (uint32_t i0 = 0; i0 < la; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const bool in = i < la;
            const uint32_t c = in ? sA[i] : 0u;
            uint32_t k = 0, cb = 0;
            const uint32_t lim = (i0 + 64u) < la ? (i0 + 64u) : la;
            for (uint32_t i2 = 0; i2 < lim; ++i2) k += (sA[i2] == c && i2 < i) ? 1u : 0u;
            for (uint32_t j = 0; j < lb; ++j) cb += (sB[j] == c) ? 1u : 0u;
            acc += (in && k < cb) ? 1u : 0u;
        }
    }
    return wave_sum(acc);
}

// One row on one wave: decode both strings to scalar values in sA/sB (capacity >= their byte lengths), run the
// measure.  aux: >= max(len)+64 words of scratch (DP boundary row / Jaro flags / histograms).  The scratch may
// be LDS (k_wave_pairs) or global memory (k_huge_pairs): the code is address-space agnostic after inlining.
template <int MEASURE>
__device__ __forceinline__ double wave_row(const uint8_t *__restrict__ valA, uint32_t a0, uint32_t la8, uint32_t totalA,
                                           const uint8_t *__restrict__ valB, uint32_t b0, uint32_t lb8, uint32_t totalB,
                                           uint32_t *sA, uint32_t *sB, uint32_t *aux, uint32_t cap, uint32_t auxwords)
{
    const uint32_t lane = lane_id();
    if (la8 == 0u && lb8 == 0u) return 1.0;
    if (la8 == 0u || lb8 == 0u) return 0.0; // also Levenshtein: 1 - max/max
    bool nonascii = false;
    __syncthreads();
    uint32_t la = wave_decode(valA + a0, la8, sA, nonascii);
    uint32_t lb = wave_decode(valB + b0, lb8, sB, nonascii);
    double r;
    if (MEASURE == LEVENSHTEIN) {
        // (k_wave_pairs<LEVENSHTEIN> only gets here for an empty side; everything else runs in wave_lev_blocks)
        const uint32_t dist = wave_levenshtein(sA, la, sB, lb, aux);
        r = epilogue_levenshtein(dist, la, lb);
    } else if (MEASURE == JARO || MEASURE == JARO_WINKLER) {
        uint32_t prefix = 0;
        if (MEASURE == JARO_WINKLER) {
            const uint32_t lim = la < lb ? (la < 4u ? la : 4u) : (lb < 4u ? lb : 4u);
            const unsigned long long ne = __ballot(lane < lim && sA[lane < lim ? lane : 0u] != sB[lane < lim ? lane : 0u]);
            prefix = ne ? (uint32_t)__builtin_ctzll(ne) : lim;
        }
        uint32_t m, t;
        // [r5] the matching loop runs once per character of "a": let that be the shorter string (Jaro is symmetric, strsim_lane_core.h)
        if (la > lb) {
            uint32_t *const ts = sA; sA = sB; sB = ts;
            const uint32_t tl = la; la = lb; lb = tl;
        }
        // bits that vary over both strings decide how many planes the match masks need
        uint32_t o = 0u, n_ = 0xFFFFFFFFu;
        for (uint32_t i = lane; i < la; i += 64u) { o |= sA[i]; n_ &= sA[i]; }
        for (uint32_t i = lane; i < lb; i += 64u) { o |= sB[i]; n_ &= sB[i]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            o |= (uint32_t)__shfl_xor((int)o, d);
            n_ &= (uint32_t)__shfl_xor((int)n_, d);
        }
        o = uniform(o);
        const uint32_t vary = uniform(o ^ n_);
        if (lb <= 2048u && o <= 0xFFFFu) {
            if (vary >> 11) wave_jaro_bits<16>(sA, la, sB, lb, aux, m, t);
            else if (vary >> 7) wave_jaro_bits<11>(sA, la, sB, lb, aux, m, t);
            else wave_jaro_bits<7>(sA, la, sB, lb, aux, m, t);
        } else {
            uint8_t *fl = reinterpret_cast<uint8_t *>(aux);
            wave_jaro(sA, la, sB, lb, fl, fl + cap, m, t);
        }
        r = epilogue_jaro(m, t, la, lb);
        if (MEASURE == JARO_WINKLER) r = epilogue_jaro_winkler(r, prefix);
    } else {
        const uint32_t isect = wave_multiset_isect(sA, la, sB, lb, aux, auxwords, nonascii);
        r = MEASURE == JACCARD ? epilogue_jaccard(isect, la, lb) : epilogue_sorensen_dice(isect, la, lb);
    }
    return r;
}

// Five waves per SIMD: the Levenshtein step loop is one dependent chain per wave and bound by what the SIMD issues (cfg5:
// 59.4 -> 56.0 ms against four; the compiler takes 101 VGPRs on its own, 96 with six spilled outside the step loop; six
// waves per SIMD: 53.3 against 52.8).  The other measures fit 96 anyway and are bound by their LDS.
#ifndef STRSIM_WAVE_WAVES_PER_EU
#define STRSIM_WAVE_WAVES_PER_EU 5
#endif
#if STRSIM_WAVE_WAVES_PER_EU > 0
#define STRSIM_WAVE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(STRSIM_WAVE_WAVES_PER_EU)))
#else
#define STRSIM_WAVE_OCCUPANCY
#endif
#ifndef STRSIM_LEV_SHARE_DIV
#define STRSIM_LEV_SHARE_DIV 1 // a grab of k_wave_pairs<levenshtein> = what is left on the list / (waves x this), at most a pool
#endif
template <int MEASURE>
__global__ __launch_bounds__(64) STRSIM_WAVE_OCCUPANCY void k_wave_pairs(const uint32_t *__restrict__ offA, const uint8_t *__restrict__ valA,
                                                   uint64_t rowsA, const uint32_t *__restrict__ offB,
                                                   const uint8_t *__restrict__ valB, uint64_t rowsB,
                                                   double *__restrict__ out, uint64_t n,
                                                   const unsigned long long *__restrict__ slowmask,
                                                   const uint32_t *__restrict__ worklist,
                                                   DevStatus *__restrict__ status, uint32_t *__restrict__ lev_ws)
{
    // Levenshtein runs in wave_lev_blocks, whose step loop needs occupancy more than anything: LDS holds only its
    // match table (8 KB), everything else -- staged texts, the scalar-value arrays of the fallback -- lives in a per-wave
    // global workspace (lev_ws, LEV_WS_WORDS words per wave).
    constexpr bool LEV = MEASURE == LEVENSHTEIN;
    if (LEV) __builtin_amdgcn_s_setprio(1); // wave_lev_blocks lowers it for its step loop
    // Jaro: flags; Jaccard / Dice: the hash table of the multiset intersection (two entries per value)
    constexpr int AUXW = (MEASURE == JARO || MEASURE == JARO_WINKLER) ? WAVE_CAP + 64 : 2 * WAVE_CAP + 64;
    __shared__ uint32_t aux_l[LEV ? 1 : AUXW];
    // The scalar-value arrays live in the per-wave global workspace (L2-resident) where LDS would only limit the waves
    // per CU (Levenshtein, Jaccard, Dice: +42 % on 33-128 Cyrillic letters); Jaro reads a[i] once per step of a
    // dependent chain and keeps them in LDS (the global variant lost 35 %).
    constexpr bool JARO_LDS = MEASURE == JARO || MEASURE == JARO_WINKLER;
    __shared__ uint32_t sA_l[JARO_LDS ? WAVE_CAP : 1];
    __shared__ uint32_t sB_l[JARO_LDS ? WAVE_CAP : 1];
    uint32_t *const sA = JARO_LDS ? sA_l : lev_ws + (uint64_t)blockIdx.x * LEV_WS_WORDS;
    uint32_t *const sB = JARO_LDS ? sB_l : sA + (WAVE_CAP + 64);
    uint32_t *const aux = LEV ? sB + (WAVE_CAP + 64) : aux_l;
    // LEV: the arenas of the two kinds of batches (global scratch): staged texts as bytes / as 16-bit scalar values,
    // and the end-aligned 16-bit patterns of a SYMBOLS batch (32 per lane)
    uint8_t *const g_ar0 = reinterpret_cast<uint8_t *>(aux + (WAVE_CAP + 64));
    uint8_t *const g_ar1 = g_ar0 + ARENA0_BYTES;
    uint16_t *const g_pat = reinterpret_cast<uint16_t *>(g_ar1 + ARENA1_BYTES);
    __shared__ BlockJob s_job[2][LEV ? LEV_JOBS : 1];
    // wave_lev_blocks: the split match tables of a lane (3 KB per wave); wave_lev_blocks64: one per mask word (6 KB)
    #ifndef STRSIM_WAVE_TAB_WORDS
#define STRSIM_WAVE_TAB_WORDS (2 * LUT_ENTRIES * 64)
#endif
    __shared__ __attribute__((aligned(4096))) uint32_t s_tab[LEV ? STRSIM_WAVE_TAB_WORDS : 1];
    // Levenshtein: the pooled rows in the order of their ranking -- bits 0..5 row of its chunk, 6..10 slot of the chunk in
    // s_pool_chunk, 11..15 the lanes an ASCII job of the row takes (half the 32-row blocks of its shorter side, rounded up; 0: not
    // a candidate for the block kernel) -- and the ranks still to do.  (The kernel's LDS is counted in 1 280-byte granules:
    // with 7 968 bytes the CU held 18 waves, not 20.)
    __shared__ uint32_t s_pool_chunk[LEV ? 32 : 1];
    __shared__ uint16_t s_pool[LEV ? LEV_POOL : 1];
    __shared__ unsigned long long s_todo[LEV ? LEV_POOL / 64 : 1];
    const uint32_t lane = lane_id();
    const bool bcastA = rowsA == 1, bcastB = rowsB == 1;
    const uint32_t totalA = offA[rowsA], totalB = offB[rowsB];
    uint32_t my_rows = 0, my_huge = 0, my_maxlen = 0;
    // Levenshtein: rows are collected into two pending batches (BYTES: both strings ASCII; SYMBOLS: anything else
    // inside the BMP) until their lane runs, their text arena or their job list is full, then run together
    struct Batch { uint32_t njobs, lanes, T, used; };
    Batch bq0{0u, 0u, 0u, 0u}, bq1{0u, 0u, 0u, 0u};
    uint32_t job_or6 = 0u, job_and6 = 0x60u;   // BYTES: bits 5/6 over the bytes of the pending jobs (wave-uniform)
    uint32_t sym_or = 0u, sym_and = 0xFFFFu;   // SYMBOLS: OR / AND of the scalar values this lane staged
    const BlockCols cols{valA, valB, totalA, totalB};
    auto flush_bytes = [&]() {
        if constexpr (LEV) {
            if (bq0.njobs == 0u) return;
            // the staged texts were written by this wave's own lanes (plain global stores): the workgroup-scope
            // fence of the barrier orders them before the loads below, and the CU's L1 is coherent for its own waves
            __syncthreads();
            // five planes when bits 5 and 6 are constant over every byte of the jobs (a-z), else all seven
            if constexpr (LEV_BYTES_ROWS == 64) {
                if ((job_or6 ^ job_and6) & 0x60u) wave_lev_blocks64<7>(s_job[0], bq0.njobs, bq0.T, cols, g_ar0, s_tab, out);
                else wave_lev_blocks64<5>(s_job[0], bq0.njobs, bq0.T, cols, g_ar0, s_tab, out);
            } else {
                if ((job_or6 ^ job_and6) & 0x60u)
                    wave_lev_blocks<7, false>(s_job[0], bq0.njobs, bq0.T, cols, g_ar0, g_pat, s_tab, out);
                else
                    wave_lev_blocks<5, false>(s_job[0], bq0.njobs, bq0.T, cols, g_ar0, g_pat, s_tab, out);
            }
            job_or6 = 0u; job_and6 = 0x60u;
            __syncthreads();
            bq0 = Batch{0u, 0u, 0u, 0u};
        }
    };
    auto flush_symbols = [&]() {
        if constexpr (LEV) {
            if (bq1.njobs == 0u) return;
            __syncthreads();
            uint32_t o = sym_or, n_ = sym_and;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                o |= (uint32_t)__shfl_xor((int)o, d);
                n_ &= (uint32_t)__shfl_xor((int)n_, d);
            }
            const uint32_t vary = uniform((o ^ n_) & 0xFFFFu); // only the bits that differ somewhere need comparing
            if (vary >> 11) wave_lev_blocks<16, true>(s_job[1], bq1.njobs, bq1.T, cols, g_ar1, g_pat, s_tab, out);
            else if (vary >> 7) wave_lev_blocks<11, true>(s_job[1], bq1.njobs, bq1.T, cols, g_ar1, g_pat, s_tab, out);
            else wave_lev_blocks<7, true>(s_job[1], bq1.njobs, bq1.T, cols, g_ar1, g_pat, s_tab, out);
            sym_or = 0u; sym_and = 0xFFFFu;
            __syncthreads();
            bq1 = Batch{0u, 0u, 0u, 0u};
        }
    };

    // Work distribution: k_lane_utf8 has compacted the chunks that still hold rows into `worklist` (C entries, R rows).
    // A wave takes `grab` entries at a time -- its first grab is static (entry blockIdx * grab, no atomic, so a call
    // with a handful of slow rows costs none), the following ones come from a counter.  One contended atomic costs
    // ~12 ns, so their number is budgeted against the work: at most max(8192, R / 64) grabs, i.e. chunk-by-chunk
    // balance (tail of one chunk) when the column is full of long rows, bigger grabs when the slow rows are sparse.
    const uint32_t C = status->list_count[MEASURE], R = status->list_rows[MEASURE];
    const uint32_t budget = (R >> 6) > 8192u ? (R >> 6) : 8192u;
    const uint32_t grab = C / budget >= 64u ? 64u : (C / budget ? C / budget : 1u);
    // Levenshtein pools the rows of up to LEV_POOL / 64 chunks: a wave takes that many entries at a time while the list is long,
    // and fewer as it runs out -- what is left (it reads the counter first) divided by the waves, at least `grab` -- so that most
    // rows are ranked in full pools and the waves still finish together (a pool of dense chunks of cfg5 is a third of a wave's
    // whole share; deciding by the wave's own previous grab, one round stale, dealt everything in full pools: + 10 %).
    // A full pool is a third of a wave's share of the list at most (cfg5 at 10 M rows has 15 chunks per wave: pools of five; at
    // 7 M rows, 10.7 per wave, pools of six lost 4 % to the waves' last grabs and pools of three lost nothing).
    uint32_t big = grab;
    if (LEV && grab < (uint32_t)(LEV_POOL / 64)) {
        big = C / (gridDim.x * 3u);
        big = big > (uint32_t)(LEV_POOL / 64) ? (uint32_t)(LEV_POOL / 64) : (big > grab ? big : grab);
    }
    const uint32_t g0 = big; // the static first grab
    for (uint32_t round = 0;; ++round) {
        uint32_t got = blockIdx.x * g0, take = g0;
        if (round != 0u) {
            if (lane == 0u) {
                take = grab;
                if (big != grab) {
                    const uint32_t at = __hip_atomic_load(&status->next_entry[MEASURE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + gridDim.x * g0;
                    const uint32_t share = C > at ? (C - at) / (gridDim.x * (uint32_t)STRSIM_LEV_SHARE_DIV) : 0u;
                    take = share >= big ? big : (share > grab ? share : grab);
                }
                got = atomicAdd(&status->next_entry[MEASURE], take);
            }
            take = uniform(take);
            got = uniform(got) + gridDim.x * g0;
        }
        if (got >= C) break;
        const bool have = lane < take && got + lane < C;
        if (have) STRSIM_CHECK_INDEX(K_WAVE, 1, got, got + lane, (n + 63u) >> 6); // (lab: the work list has one entry per chunk at most)
        const uint32_t entry = have ? worklist[got + lane] : 0u;
        if (have) STRSIM_CHECK_INDEX(K_WAVE, 2, got, entry, (n + 63u) >> 6);
        const unsigned long long mword = have ? slowmask[entry] : 0ull;
        unsigned long long pending = __ballot(mword != 0ull);
        while (pending != 0ull) {
            auto next_chunk = [&](uint64_t &chunk, unsigned long long &mask) {
                const uint32_t src = (uint32_t)__builtin_ctzll(pending);
                pending &= pending - 1ull;
                chunk = (uint32_t)__builtin_amdgcn_readlane((int)entry, (int)src);
                mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mword >> 32), (int)src) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mword, (int)src);
            };
            uint64_t chunk1 = 0;            // the other measures: one chunk at a time, its rows in row order
            unsigned long long mask1 = 0ull;
            uint32_t npool = 0u;
            if constexpr (LEV) {
                // Levenshtein runs several rows at a time for max(steps) of them, so the rows of a batch should be alike: the
                // rows of up to LEV_POOL / 64 chunks are pooled and ranked in descending order of their step count (longer
                // length + blocks of the shorter - 1), and each batch is filled first-fit from that order (rows whose lane
                // run does not fit any more are skipped and start or join a later batch).  One chunk's 64 rows left a tenth
                // of the lane-steps to rows shorter than their batch; 256 rows leave 4 % (cfg5).
                __syncthreads();
                uint32_t nslot = 0u;
                while (pending != 0ull && npool + 64u <= (uint32_t)LEV_POOL && nslot < 32u) {
                    uint64_t chunk;
                    unsigned long long mask;
                    next_chunk(chunk, mask);
                    if (lane == 0u) s_pool_chunk[nslot] = (uint32_t)chunk;
                    if ((mask >> lane) & 1ull)
                        s_pool[npool + (uint32_t)__popcll(mask & lanemask_lt(lane))] = (uint16_t)(lane | (nslot << 6));
                    npool += (uint32_t)__popcll(mask);
                    ++nslot;
                }
                __syncthreads();
                constexpr int PG = LEV_POOL / 64;
                uint32_t key[PG], code[PG], rank[PG];
#pragma unroll
                for (int k = 0; k < PG; ++k) {
                    key[k] = 0u; code[k] = 0u; rank[k] = 0u;
                    const uint32_t pidx = lane + 64u * (uint32_t)k;
                    if (pidx < npool) {
                        code[k] = s_pool[pidx];
                        const uint64_t rw = (uint64_t)s_pool_chunk[code[k] >> 6] * 64u + (code[k] & 63u);
                        const uint64_t ra = bcastA ? 0 : rw, rb = bcastB ? 0 : rw;
                        STRSIM_CHECK_INDEX(K_WAVE, 3, rw, rw, n);
                        STRSIM_CHECK_INDEX(K_WAVE, 4, rw, ra + 1u, rowsA + 1u);
                        STRSIM_CHECK_INDEX(K_WAVE, 5, rw, rb + 1u, rowsB + 1u);
                        const uint32_t x = offA[ra + 1] - offA[ra], y = offB[rb + 1] - offB[rb];
                        const uint32_t mx = x > y ? x : y, mn = x < y ? x : y;
                        if (mn != 0u && mx <= (uint32_t)WAVE_CAP) {
                            const uint32_t nblk = (mn + 31u) >> 5;
                            key[k] = mx + nblk;
                            code[k] |= ((nblk + 1u) >> 1) << 11;
                        }
                    }
                }
                // rank = rows of the pool in front of this one: a larger key, or the same key and a smaller pool index
#pragma unroll
                for (int k2 = 0; k2 < PG; ++k2) {
                    const uint32_t cnt = npool > 64u * (uint32_t)k2 ? (npool - 64u * (uint32_t)k2 < 64u ? npool - 64u * (uint32_t)k2 : 64u) : 0u;
                    for (uint32_t jl = 0; jl < cnt; ++jl) {
                        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key[k2], (int)jl);
#pragma unroll
                        for (int k = 0; k < PG; ++k)
                            rank[k] += (k2 < k ? kj >= key[k] : (k2 > k ? kj > key[k] : (kj > key[k] || (kj == key[k] && jl < lane)))) ? 1u : 0u;
                    }
                }
                __syncthreads(); // every lane has read its rows: the pool is rewritten in ranking order
#pragma unroll
                for (int k = 0; k < PG; ++k)
                    if (lane + 64u * (uint32_t)k < npool) s_pool[rank[k]] = (uint16_t)code[k];
                if (lane < (uint32_t)PG) {
                    const uint32_t cnt = npool > 64u * lane ? npool - 64u * lane : 0u;
                    s_todo[lane] = cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull); // bit = position in the ranking
                }
                __syncthreads();
            } else {
                next_chunk(chunk1, mask1);
            }
            const uint32_t nwords = LEV ? (npool + 63u) >> 6 : 1u;
            for (;;) {
              bool left = false;
#pragma unroll 1
              for (uint32_t wsel = 0; wsel < nwords; ++wsel) {
              unsigned long long cur = mask1; // rows (other measures) / ranks (Levenshtein) of this word still to do
              if constexpr (LEV) {
                  const unsigned long long tw = s_todo[wsel];
                  cur = ((unsigned long long)uniform((uint32_t)(tw >> 32)) << 32) | uniform((uint32_t)tw);
              }
              for (unsigned long long scan = cur; scan != 0ull; scan &= scan - 1ull) {
                const uint32_t vb = (uint32_t)__builtin_ctzll(scan);
                uint64_t row = chunk1 * 64u + vb;
                if constexpr (LEV) {
                    const uint32_t code = uniform((uint32_t)s_pool[64u * wsel + vb]);
                    // no room in this pass: an ASCII job takes a lane per 64 rows of its shorter side, any other one per 32 (for
                    // those the code holds an upper bound)
                    const uint32_t need0 = LEV_BYTES_ROWS == 64 ? code >> 11 : 2u * (code >> 11);
                    if (bq0.lanes + need0 > 64u || bq1.lanes + 2u * (code >> 11) > 64u) continue;
                    row = (uint64_t)uniform(s_pool_chunk[(code >> 6) & 31u]) * 64u + (code & 63u);
                }
                cur &= ~(1ull << vb);
                const uint64_t ra = bcastA ? 0 : row, rb = bcastB ? 0 : row;
                STRSIM_CHECK_INDEX(K_WAVE, 6, row, row, n);
                STRSIM_CHECK_INDEX(K_WAVE, 7, row, ra + 1u, rowsA + 1u);
                STRSIM_CHECK_INDEX(K_WAVE, 8, row, rb + 1u, rowsB + 1u);
                const uint32_t a0 = uniform(offA[ra]), a1 = uniform(offA[ra + 1]);
                const uint32_t b0 = uniform(offB[rb]), b1 = uniform(offB[rb + 1]);
                const uint32_t la8 = a1 - a0, lb8 = b1 - b0;
                ++my_rows;
                bool long_bytes = la8 > (uint32_t)WAVE_CAP || lb8 > (uint32_t)WAVE_CAP;
                if (long_bytes && !(LEV && la8 != 0u && lb8 != 0u && lev_fits_symbols(valA + a0, la8, valB + b0, lb8))) {
                    ++my_huge; // finished by k_huge_pairs, launched from strsim_ctx_synchronize()
                    const uint32_t ml = la8 > lb8 ? la8 : lb8;
                    my_maxlen = my_maxlen > ml ? my_maxlen : ml;
                    continue;
                }
                if constexpr (LEV) {
                  if (la8 != 0u && lb8 != 0u) {
                    // ---- BYTES: shorter string = DP rows (read from its column), longer = columns (staged bytes)
                    const bool a_short = la8 <= lb8;
                    const uint32_t ms = a_short ? la8 : lb8, nl = a_short ? lb8 : la8;
                    const uint32_t Bn = (ms + (uint32_t)LEV_BYTES_ROWS - 1u) / (uint32_t)LEV_BYTES_ROWS;
                    if (!long_bytes) {
                        // (the first-fit test above left room for the lanes; the job list is flushed when it fills up: only the
                        // text arena can be short here -- a row that then turns out not to be ASCII has flushed for nothing)
                        const uint32_t slot = (2u * TXT_PAD + nl + 3u) & ~3u;
                        if (bq0.njobs == (uint32_t)LEV_JOBS || bq0.lanes + Bn > 64u || bq0.used + slot > (uint32_t)ARENA0_BYTES) flush_bytes();
                        uint32_t o6 = job_or6, n6 = job_and6;
                        STRSIM_CHECK_RANGE(K_WAVE, 10, row, bq0.used + slot, 0, ARENA0_BYTES);   // (lab: the staged text inside its arena)
                        STRSIM_CHECK_RANGE(K_WAVE, 11, row, (uint64_t)a0 + la8, 0, totalA);       // ... the strings inside their columns
                        STRSIM_CHECK_RANGE(K_WAVE, 12, row, (uint64_t)b0 + lb8, 0, totalB);
                        if (wave_ascii_stage(a_short ? valA + a0 : valB + b0, ms, a_short ? valB + b0 : valA + a0, nl,
                                             g_ar0 + bq0.used + TXT_PAD, o6, n6)) {
                            job_or6 = o6; job_and6 = n6;
                            if (lane == 0u)
                                s_job[0][bq0.njobs] = BlockJob{a_short ? a0 : b0, (uint32_t)row, (uint16_t)ms, (uint16_t)nl,
                                                               (uint16_t)(bq0.used >> 2), (uint8_t)bq0.lanes, (uint8_t)(a_short ? 1 : 0)};
                            ++bq0.njobs;
                            bq0.lanes += Bn;
                            bq0.used += slot;
                            const uint32_t Tj = nl + Bn - 1u;
                            bq0.T = bq0.T > Tj ? bq0.T : Tj;
                            if (bq0.njobs == (uint32_t)LEV_JOBS || bq0.lanes == 64u) flush_bytes();
                            continue;
                        }
                    }
                    // ---- SYMBOLS: decode both strings; the one with fewer scalar values is the pattern
                    bool nonascii = false;
                    __syncthreads();
                    const uint32_t la = wave_decode(valA + a0, la8, sA, nonascii);
                    const uint32_t lb = wave_decode(valB + b0, lb8, sB, nonascii);
                    const bool a_pat = la <= lb;
                    const uint32_t *pat = a_pat ? sA : sB, *txt = a_pat ? sB : sA;
                    const uint32_t mp = a_pat ? la : lb, nt = a_pat ? lb : la;
                    const uint32_t Bp = (mp + 31u) >> 5;
                    const uint32_t slot = (2u * (2u * TXT_PAD + nt) + 3u) & ~3u;
                    if (bq1.njobs == (uint32_t)LEV_JOBS || bq1.lanes + Bp > 64u || bq1.used + slot > (uint32_t)ARENA1_BYTES)
                        flush_symbols();
                    bool big = false;
                    uint32_t so = sym_or, sn = sym_and;
                    STRSIM_CHECK_RANGE(K_WAVE, 13, row, bq1.used + slot, 0, ARENA1_BYTES);
                    STRSIM_CHECK_RANGE(K_WAVE, 14, row, (bq1.lanes + Bp) * 32u, 0, 64u * 32u); // the end-aligned patterns: 32 per lane
                    uint16_t *tdst = reinterpret_cast<uint16_t *>(g_ar1 + bq1.used) + TXT_PAD;
                    for (uint32_t i = lane; i < nt; i += 64u) {
                        const uint32_t cp = txt[i];
                        big = big || cp > 0xFFFFu;
                        so |= cp; sn &= cp;
                        tdst[i] = (uint16_t)cp;
                    }
                    // the pattern, end-aligned to its last block: zeros stand for the positions in front of it
                    const uint32_t lead = 32u * Bp - mp;
                    uint16_t *pdst = g_pat + bq1.lanes * 32u;
                    for (uint32_t i = lane; i < 32u * Bp; i += 64u) {
                        uint32_t cp = 0u;
                        if (i >= lead) {
                            cp = pat[i - lead];
                            big = big || cp > 0xFFFFu;
                            so |= cp; sn &= cp;
                        }
                        pdst[i] = (uint16_t)cp;
                    }
                    if (__ballot(big) == 0ull) {
                        sym_or = so & 0xFFFFu; sym_and = sn & 0xFFFFu;
                        if (lane == 0u)
                            s_job[1][bq1.njobs] = BlockJob{0u, (uint32_t)row, (uint16_t)mp, (uint16_t)nt, (uint16_t)(bq1.used >> 2),
                                                           (uint8_t)bq1.lanes, (uint8_t)0};
                        ++bq1.njobs;
                        bq1.lanes += Bp;
                        bq1.used += slot;
                        const uint32_t Tj = nt + Bp - 1u;
                        bq1.T = bq1.T > Tj ? bq1.T : Tj;
                        if (bq1.njobs == (uint32_t)LEV_JOBS || bq1.lanes == 64u) flush_symbols();
                    } else {
                        // scalar values beyond the BMP: the anti-diagonal DP on the decoded arrays
                        const uint32_t dist = wave_levenshtein(sA, la, sB, lb, aux);
                        if (lane == 0u) out[row] = epilogue_levenshtein(dist, la, lb);
                    }
                    continue;
                  }
                }
                STRSIM_CHECK_RANGE(K_WAVE, 15, row, (uint64_t)a0 + la8, 0, totalA);
                STRSIM_CHECK_RANGE(K_WAVE, 16, row, (uint64_t)b0 + lb8, 0, totalB);
                const double r = wave_row<MEASURE>(valA, a0, la8, totalA, valB, b0, lb8, totalB, sA, sB, aux, WAVE_CAP, LEV ? WAVE_CAP + 64 : AUXW);
                if (lane == 0u) out[row] = r;
              }
              if constexpr (LEV) {
                  if (lane == 0u) s_todo[wsel] = cur;
              } else {
                  mask1 = cur;
              }
              left = left || cur != 0ull;
              }
              if (!left) break;
              // rows are left that did not fit: run what has been collected
              __syncthreads();
              flush_bytes();
              flush_symbols();
            }
        }
    }
    flush_bytes();
    flush_symbols();
    if (lane == 0u && my_rows != 0u) {
        atomicAdd(&status->wave_rows, my_rows);
        if (my_huge != 0u) {
            atomicAdd(&status->huge_rows, my_huge);
            atomicMax(&status->max_len, my_maxlen);
        }
    }
}

// Levenshtein similarity of one pair of any length (k_huge_pairs): decode, then the striped block kernel with as many
// bit-planes as the pair's scalar values need; values beyond the BMP fall back to the anti-diagonal DP.
__device__ __forceinline__ double huge_levenshtein(const uint8_t *__restrict__ pa, uint32_t la8, const uint8_t *__restrict__ pb,
                                                   uint32_t lb8, uint32_t *sA, uint32_t *sB, uint32_t *aux, uint32_t *tab)
{
    const uint32_t lane = lane_id();
    if (la8 == 0u && lb8 == 0u) return 1.0;
    if (la8 == 0u || lb8 == 0u) return 0.0;
    bool nonascii = false;
    __syncthreads();
    const uint32_t la = wave_decode(pa, la8, sA, nonascii);
    const uint32_t lb = wave_decode(pb, lb8, sB, nonascii);
    uint32_t o = 0u, a_ = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < la; i += 64u) { o |= sA[i]; a_ &= sA[i]; }
    for (uint32_t i = lane; i < lb; i += 64u) { o |= sB[i]; a_ &= sB[i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        o |= (uint32_t)__shfl_xor((int)o, d);
        a_ &= (uint32_t)__shfl_xor((int)a_, d);
    }
    o = uniform(o);
    const uint32_t vary = uniform(o ^ a_);
    uint32_t dist;
    if (o > 0xFFFFu) {
        dist = wave_levenshtein(sA, la, sB, lb, aux);
    } else {
        const bool a_pat = la <= lb;
        const uint32_t *pat = a_pat ? sA : sB, *txt = a_pat ? sB : sA;
        const uint32_t m = a_pat ? la : lb, n = a_pat ? lb : la;
        uint8_t *hb = reinterpret_cast<uint8_t *>(aux) + 64;
        if (vary >> 11) dist = wave_lev_stripes<16>(pat, m, txt, n, hb, tab);
        else if (vary >> 7) dist = wave_lev_stripes<11>(pat, m, txt, n, hb, tab);
        else dist = wave_lev_stripes<7>(pat, m, txt, n, hb, tab);
    }
    return epilogue_levenshtein(dist, la, lb);
}
