// strsim_lane_bins.h -- k_wide_bins<M, W, LUT>: one pair per lane for the rows of 33..128 ASCII bytes that k_lane_stage handed over
// in BINS (strsim_bins.h).  Included by strsim_kernels.hip inside namespace strsim, behind k_lane_wide (whose LDS text column and
// cores -- strsim_lane_wide.h; reference strsim.rs:141-160, :200-237, :297-305, :333-341 -- it shares).
//
// A round is one PAGE: the records of 64 rows of one bin, i.e. 64 pairs that run the same number of column groups at the same mask
// width W -- nothing to sort, no idle columns beyond the bin's rounding.  A wave (one workgroup = one wave: no barriers, its LDS is
// nobody else's) walks the pages of its width class -- longest texts first, dealt out in turn -- three rounds deep: while it
// computes round i, the strings of round i + 1 are on their way from the columns into registers (each lane its own row's 16-byte
// pieces) and the records of round i + 2 are on theirs; so neither latency is ever waited for.  The text then goes into an LDS
// column per lane, the pattern stays in registers, and the W-word cores run -- with match masks from bit fills or from per-lane
// LDS tables (LUT; strsim_lane_wide_lut.h).  Results go to out[row]; a row with a high bit in its pieces gets its mask bit back
// (the code-point kernels run behind this one).
#pragma once

template <int MEASURE, int W, bool LUT>
struct WideBinsGeom {
    static constexpr bool JARO_LIKE = MEASURE == JARO || MEASURE == JARO_WINKLER;
    // pieces of 16 bytes: the pattern of a width class has at most 32 W bytes; the text of a symmetric measure is the shorter
    // string, Jaro's text is a (up to 128 bytes whatever b is)
    static constexpr int PS_MAX = 2 * W;
    static constexpr int TS_MAX = JARO_LIKE ? 8 : 2 * W;
    static constexpr int TROWS = 4 * TS_MAX;              // dword rows of the text column
    // LUT: the tables first in the wave's LDS (their address form wants them at a multiple of 16 << ES bytes), the text columns
    // behind them
    static constexpr int TAB_BYTES = LUT ? WideLut<W>::BYTES : 0;
    static constexpr int TAB_ALIGN = LUT ? (16 << WideLut<W>::ES) : 16;
    // 512 VGPRs per SIMD lane: the next round's strings wait in registers beside the cores' state
    static constexpr int WAVES_PER_EU = W == 2 ? 3 : 2;
};

// 16 bytes at byte offset `off` of a column of `total` bytes, any alignment; a piece that reaches past the column's end (the last
// row's) is read byte by byte, zeros behind the end
__device__ __forceinline__ uint4 load_piece16(const uint8_t *__restrict__ vals, uint32_t off, uint32_t total)
{
    if (off + 16u <= total) {
        const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned *>(vals + off);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (uint32_t k = 0; k < 16u; ++k)
        if (off + k < total) w[k >> 2] |= (uint32_t)vals[off + k] << (8u * (k & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int MEASURE, int W, bool LUT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WideBinsGeom<MEASURE, W, LUT>::WAVES_PER_EU))) void
k_wide_bins(const BinTable *__restrict__ table, const uint4 *__restrict__ recs, const uint32_t *__restrict__ offA,
            const uint8_t *__restrict__ valA, uint64_t rowsA, const uint32_t *__restrict__ offB, const uint8_t *__restrict__ valB,
            uint64_t rowsB, double *__restrict__ out, unsigned long long *__restrict__ slowmask)
{
    using G = WideBinsGeom<MEASURE, W, LUT>;
    constexpr uint32_t cls = (uint32_t)W - 2u;
    __shared__ __attribute__((aligned(G::TAB_ALIGN))) uint8_t s_lds[G::TAB_BYTES + G::TROWS * 256];
    const uint32_t lane = lane_id();
    const uint32_t nrounds = load_invariant(&table->cls_rounds[cls]), nbins = load_invariant(&table->cls_nbins[cls]);
    if (load_invariant(&table->enabled) == 0u || nrounds == 0u) return;
    const uint32_t totalA = load_invariant(offA + rowsA), totalB = load_invariant(offB + rowsB);
    // the rounds in front of each bin's END, four bins per lane: a round's bin = how many ends are at or below it
    uint32_t ends[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = (uint32_t)j * 64u + lane;
        ends[j] = k < nbins ? table->cls_cum[cls][k + 1u] : 0xFFFFFFFFu;
    }
    struct Round {
        uint32_t bin, rows; // the page's bin, its rows (0: no such round)
        uint4 rec;          // this lane's record
    };
    // a round's page: which bin, how many rows, this lane's record (coalesced: 1 KB per page)
    auto page = [&](uint32_t r) -> Round {
        Round x{0u, 0u, make_uint4(BIN_DEAD_ROW, 0u, 0u, 0u)};
        if (r < nrounds) { // (uniform)
            uint32_t k = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) k += (uint32_t)__builtin_popcountll(__ballot(ends[j] <= r));
            x.bin = load_invariant(&table->cls_bin[cls][k]);
            const uint32_t pg = r - load_invariant(&table->cls_cum[cls][k]);
            const uint32_t left = load_invariant(&table->count[x.bin]) - pg * 64u;
            x.rows = left < 64u ? left : 64u;
            x.rec = recs[(size_t)load_invariant(&table->base16[x.bin]) + pg * BIN_PAGE16 + lane];
        }
        return x;
    };
    // a round's strings into registers: each lane its own row's pieces, straight from the columns
    uint4 nt[G::TS_MAX], np[G::PS_MAX];
    auto fetch = [&](const Round &x) {
        if (x.rows == 0u) return; // (uniform)
        const uint32_t ts = bin_text_pieces(x.bin), ps = bin_pat_pieces(x.bin);
        const bool on = lane < x.rows && x.rec.x != BIN_DEAD_ROW;
        const bool t_in_b = (x.rec.w & BIN_REC_TEXT_IN_B) != 0u;
        const uint8_t *__restrict__ const vt = t_in_b ? valB : valA, *__restrict__ const vp = t_in_b ? valA : valB;
        const uint32_t tt = t_in_b ? totalB : totalA, tp = t_in_b ? totalA : totalB;
        const uint32_t toff = on ? x.rec.y : 0u, poff = on ? x.rec.z : 0u;
#pragma unroll
        for (int q = 0; q < G::TS_MAX; ++q)
            if ((uint32_t)q < ts) nt[q] = load_piece16(vt, toff + 16u * (uint32_t)q, tt);
#pragma unroll
        for (int q = 0; q < G::PS_MAX; ++q)
            if ((uint32_t)q < ps) np[q] = load_piece16(vp, poff + 16u * (uint32_t)q, tp);
    };
    const uint32_t stride = gridDim.x;
    uint32_t r = blockIdx.x;
    Round cur = page(r), nxt = page(r + stride);
    fetch(cur);
    uint32_t *const txt_col = reinterpret_cast<uint32_t *>(s_lds + G::TAB_BYTES) + lane;
    WideLut<W> lut = wide_lut_at<W>(STRSIM_LDS_ADDR(&s_lds[0]), lane);
    const LdsTxt txt{txt_col};
    const LdsSa sa{STRSIM_LDS_ADDR(txt_col)};
    __builtin_amdgcn_s_setprio(1);
    while (cur.rows != 0u) {
        const uint32_t bin = cur.bin, tb = bin_text_bucket(bin), pc = bin_pat_class(bin), ts = bin_text_pieces(bin), ps = bin_pat_pieces(bin);
        // ---- this round's strings have arrived: text -> the lane's LDS column, pattern -> wp
        const uint32_t row = cur.rec.x, lens = cur.rec.w;
        bool has = lane < cur.rows && row != BIN_DEAD_ROW;
        uint32_t wp[8 * W];
        const uint32_t a0w = nt[0].x;
        // OR and AND of the pair's pieces, whole pieces (what follows a string's end inside its last piece belongs to its
        // neighbour in the column: conservative, as in k_lane_stage): any high bit sends the row back to the mask, for the
        // code-point kernels behind this one; the bits that vary decide between five planes and seven
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
        // ---- the next round's strings start their way (its records have arrived); the records of the round after it start theirs
        fetch(nxt);
        r += stride;
        const Round nn = page(r + stride);
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
