// strsim_lane_bins.h -- k_wide_bins<M, W>: one pair per lane for the rows of 33..128 ASCII bytes that k_lane_stage handed over in
// BINS (strsim_bins.h).  Included by strsim_kernels.hip inside namespace strsim, behind k_lane_wide (whose LDS text column and
// cores -- strsim_lane_wide.h; reference strsim.rs:141-160, :200-237, :297-305, :333-341 -- it shares).
//
// A round is one PAGE: 64 rows of one bin, i.e. 64 pairs that run the same number of column groups at the same mask width W.
// A wave (one workgroup = one wave: no barriers, its LDS is nobody else's) walks the pages of its width class -- longest texts
// first, dealt out in turn -- and keeps TWO rounds in flight: while it computes round i, the strings of round i + 1 are on their
// way into registers (piece q of the page's 64 rows is 1 KB contiguous: one coalesced load per piece, every byte fetched is a
// byte of this round) and the table look-ups of round i + 2 are on theirs; so neither the page's latency nor the look-ups' is
// ever waited for.  The text then goes into an LDS column per lane, the pattern stays in registers, and the W-word cores run.
// Results go to out[row]; rows whose record is dead (non-ASCII, not staged) are somebody else's.
#pragma once

template <int MEASURE, int W, bool LUT>
struct WideBinsGeom {
    static constexpr bool JARO_LIKE = MEASURE == JARO || MEASURE == JARO_WINKLER;
    // pieces of 16 bytes: the pattern of a width class has at most 32 W bytes; the text of a symmetric measure is the shorter
    // string, Jaro's text is a (up to 128 bytes whatever b is)
    static constexpr int PS_MAX = 2 * W;
    static constexpr int TS_MAX = JARO_LIKE ? 8 : 2 * W;
    static constexpr int TROWS = 4 * TS_MAX;              // dword rows of the text column
    // LUT: match masks from per-lane tables (strsim_lane_wide_lut.h), first in the wave's LDS (their address form wants them
    // at a multiple of 16 << ES bytes), the text columns behind them
    static constexpr int TAB_BYTES = LUT ? WideLut<W>::BYTES : 0;
    static constexpr int TAB_ALIGN = LUT ? (16 << WideLut<W>::ES) : 16;
    // 512 VGPRs per SIMD lane: the next round's strings wait in registers beside the cores' state
    static constexpr int WAVES_PER_EU = W == 2 ? 3 : 2;
};

struct BinRound {
    uint32_t bin, pg16, rows; // the page's bin, its place in the buffer (16-byte units), its rows (0: no such round)
};

template <int MEASURE, int W, bool LUT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WideBinsGeom<MEASURE, W, LUT>::WAVES_PER_EU))) void
k_wide_bins(const BinTable *__restrict__ table, const uint8_t *__restrict__ buf, double *__restrict__ out,
            unsigned long long *__restrict__ slowmask)
{
    using G = WideBinsGeom<MEASURE, W, LUT>;
    constexpr uint32_t cls = (uint32_t)W - 2u;
    __shared__ __attribute__((aligned(G::TAB_ALIGN))) uint8_t s_lds[G::TAB_BYTES + G::TROWS * 256];
    const uint32_t lane = lane_id();
    const uint32_t nrounds = load_invariant(&table->cls_rounds[cls]), nbins = load_invariant(&table->cls_nbins[cls]);
    if (load_invariant(&table->enabled) == 0u || nrounds == 0u) return;
    // the rounds in front of each bin's END, four bins per lane: a round's bin = how many ends are at or below it
    uint32_t ends[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = (uint32_t)j * 64u + lane;
        ends[j] = k < nbins ? table->cls_cum[cls][k + 1u] : 0xFFFFFFFFu;
    }
    auto lookup = [&](uint32_t r) -> BinRound {
        BinRound x{0u, 0u, 0u};
        if (r < nrounds) { // (uniform)
            uint32_t k = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) k += (uint32_t)__builtin_popcountll(__ballot(ends[j] <= r));
            x.bin = load_invariant(&table->cls_bin[cls][k]);
            const uint32_t page = r - load_invariant(&table->cls_cum[cls][k]);
            const uint32_t left = load_invariant(&table->count[x.bin]) - page * 64u;
            x.pg16 = load_invariant(&table->base16[x.bin]) + page * bin_page16(x.bin);
            x.rows = left < 64u ? left : 64u;
        }
        return x;
    };
    // a round's strings into registers: header, text pieces, pattern pieces (all coalesced: lane l takes row l of the page)
    uint32_t nrow = 0u, nlens = 0u;
    uint4 nt[G::TS_MAX], np[G::PS_MAX];
    auto fetch = [&](const BinRound &x) {
        if (x.rows == 0u) return; // (uniform)
        const uint8_t *__restrict__ const pg = buf + (size_t)x.pg16 * 16u;
        const uint32_t ts = bin_text_slot16(x.bin), ps = bin_pat_slot16(x.bin);
        nrow = reinterpret_cast<const uint32_t *>(pg)[lane];
        nlens = pg[256u + lane];
        const uint4 *__restrict__ const pieces = reinterpret_cast<const uint4 *>(pg + 16u * BIN_PAGE_HEAD16) + lane;
#pragma unroll
        for (int q = 0; q < G::TS_MAX; ++q)
            if ((uint32_t)q < ts) nt[q] = pieces[q * 64];
#pragma unroll
        for (int q = 0; q < G::PS_MAX; ++q)
            if ((uint32_t)q < ps) np[q] = pieces[(ts + (uint32_t)q) * 64u];
    };
    const uint32_t stride = gridDim.x;
    uint32_t r = blockIdx.x;
    BinRound cur = lookup(r), nxt = lookup(r + stride);
    fetch(cur);
    uint32_t *const txt_col = reinterpret_cast<uint32_t *>(s_lds + G::TAB_BYTES) + lane;
    WideLut<W> lut = wide_lut_at<W>(STRSIM_LDS_ADDR(&s_lds[0]), lane);
    const LdsTxt txt{txt_col};
    const LdsSa sa{STRSIM_LDS_ADDR(txt_col)};
    __builtin_amdgcn_s_setprio(1);
    while (cur.rows != 0u) {
        const uint32_t bin = cur.bin, tb = bin_text_bucket(bin), pc = bin_pat_class(bin), ts = bin_text_slot16(bin), ps = bin_pat_slot16(bin);
        // ---- this round's strings have arrived: text -> the lane's LDS column, pattern -> wp, header -> row / lengths
        bool has = lane < cur.rows && nrow != BIN_DEAD_ROW;
        const uint32_t row = nrow, lens = nlens;
        uint32_t wp[8 * W];
        const uint32_t a0w = nt[0].x;
        // OR and AND of the pair's pieces, whole pieces (what follows a string's end inside its last piece belongs to a neighbour
        // or is padding: conservative, as in k_lane_stage): any high bit sends the row back to the mask, for the code-point
        // kernels behind this one; the bits that vary decide between five planes and seven
        uint32_t o = 0u, n = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < G::TS_MAX; ++q) {
            if ((uint32_t)q < ts) {
                txt_col[(4 * q) * 64] = nt[q].x; txt_col[(4 * q + 1) * 64] = nt[q].y;
                txt_col[(4 * q + 2) * 64] = nt[q].z; txt_col[(4 * q + 3) * 64] = nt[q].w;
                o = bitop3<0xFE>(bitop3<0xFE>(o, nt[q].x, nt[q].y), nt[q].z, nt[q].w);
                n = bitop3<0x80>(bitop3<0x80>(n, nt[q].x, nt[q].y), nt[q].z, nt[q].w);
            }
        }
#pragma unroll
        for (int q = 0; q < G::PS_MAX; ++q) {
            const bool on = (uint32_t)q < ps;
            wp[4 * q] = on ? np[q].x : 0u; wp[4 * q + 1] = on ? np[q].y : 0u;
            wp[4 * q + 2] = on ? np[q].z : 0u; wp[4 * q + 3] = on ? np[q].w : 0u;
            if (on) {
                o = bitop3<0xFE>(bitop3<0xFE>(o, np[q].x, np[q].y), np[q].z, np[q].w);
                n = bitop3<0x80>(bitop3<0x80>(n, np[q].x, np[q].y), np[q].z, np[q].w);
            }
        }
        uint32_t o8 = o | (o >> 16); o8 |= o8 >> 8;
        uint32_t n8 = n & (n >> 16); n8 &= n8 >> 8;
        if (has && (o8 & 0x80u)) {
            atomicOr(&slowmask[row >> 6], 1ull << (row & 63u));
            has = false;
        }
        // ---- the next round's strings start their way; the round after it is looked up
        fetch(nxt);
        r += stride;
        const BinRound nn = lookup(r + stride);
        // ---- the cores: every lane of the round runs tb + 1 groups of four columns; in the first tb of them no lane's text ends
        const uint32_t lt = has ? 4u * tb + 1u + (lens & 3u) : 1u, lp = has ? 16u * pc + 1u + ((lens >> 2) & 15u) : 1u;
        const bool seven = __ballot(has && ((o8 ^ n8) & 0x60u)) != 0ull;
        const uint32_t nb4 = G::JARO_LIKE ? 4u * (pc + 1u) : 0u; // Jaro's second pass walks the pattern's dwords
        double res;
        __builtin_amdgcn_s_setprio(0);
        if (LUT) {
            if (seven) res = lane_wide_result_lut<MEASURE, 7, W>(lut, txt, lt, tb, tb + 1u, wp, lp, nb4, a0w, wp[0], sa);
            else res = lane_wide_result_lut<MEASURE, 5, W>(lut, txt, lt, tb, tb + 1u, wp, lp, nb4, a0w, wp[0], sa);
        } else {
            if (seven) res = lane_wide_result<MEASURE, 7, W>(txt, lt, tb, tb + 1u, wp, lp, nb4, a0w, wp[0], sa);
            else res = lane_wide_result<MEASURE, 5, W>(txt, lt, tb, tb + 1u, wp, lp, nb4, a0w, wp[0], sa);
        }
        __builtin_amdgcn_s_setprio(1);
        if (has) out[row] = res;
        cur = nxt;
        nxt = nn;
    }
}
