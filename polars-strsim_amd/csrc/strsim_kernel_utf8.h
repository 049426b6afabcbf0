// strsim_kernel_utf8.h -- k_lane_utf8<M>: one pair per lane for short non-ASCII strings (per-lane UTF-8 decode into 16-bit symbols: strsim_lane_sym.h);
// the last per-lane kernel, so it also builds k_wave_pairs' work list.
// Included by strsim_kernels.hip inside namespace strsim, after the kernels in front of it ([r5] split out of strsim_kernels.hip
// along its seams, VERDICT r4 item 8: no behaviour change -- the translation unit's ISA is byte-identical before and after).
// Reference semantics: /root/reference/src/expressions/strsim.rs:125-345 (the cores cite their lines).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_lane_utf8: one pair per lane for short NON-ASCII strings -- both strings <= 128 bytes and, once decoded,
// <= 32 Unicode scalar values, all in the BMP (names and words in any script).  Each lane decodes its two strings
// from registers into 16-bit symbols in LDS columns and runs the up-to-16-plane cores of strsim_lane_sym.h.
// Same span / collect / 64-row-round structure as k_lane_wide; rows it cannot take (longer, astral, empty side)
// stay in the mask for k_wave_pairs.  Without it every non-ASCII row costs a whole wave (~500x slower).
// ------------------------------------------------------------------------------------------------
constexpr int U8_DW = 32; // 128-byte window per string

struct LdsSym {
    const uint16_t *col; // &s_sym[wave][0][lane]; symbol k at col[k * 64]
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return col[(k & 31u) * 64u]; }
};
struct LdsSymEmit {
    uint16_t *col;
    // (utf8_decode_lane emits every byte position, a value's index possibly several times, and a don't-care at index `count`:
    // with 32 values that is index 32 -- dropped, like everything behind it: a longer string is not this kernel's anyway)
    __device__ __forceinline__ void operator()(uint32_t k, uint32_t cp) const { if (k < 32u) col[k * 64u] = (uint16_t)cp; }
};

// load one string's window (4*NDW bytes) and decode it into the lane's LDS symbol column; returns its scalar-value count
template <int NDW>
__device__ __forceinline__ uint32_t decode_column(const uint8_t *__restrict__ vals, uint32_t off, uint32_t len8, uint32_t total,
                                                  bool has, uint32_t maxlen, uint16_t *col, bool &big, uint32_t &ov, uint32_t &av)
{
    uint32_t w[NDW];
#pragma unroll
    for (int d = 0; d < NDW; ++d) w[d] = 0u;
    if (has) load_window_any<NDW>(vals, (int64_t)off, total, w);
    return utf8_decode_lane<NDW>(w, len8, (maxlen + 3u) >> 2, LdsSymEmit{col}, big, ov, av);
}

template <int MEASURE>
__global__ __launch_bounds__(WIDE_BLOCK) void k_lane_utf8(const uint32_t *__restrict__ offA,
                                                          const uint8_t *__restrict__ valA, uint64_t rowsA,
                                                          const uint32_t *__restrict__ offB,
                                                          const uint8_t *__restrict__ valB, uint64_t rowsB,
                                                          double *__restrict__ out, uint64_t n,
                                                          unsigned long long *__restrict__ slowmask, uint32_t sps,
                                                          uint32_t *__restrict__ worklist, DevStatus *__restrict__ status)
{
    __shared__ unsigned long long s_mask[U8_SPAN];
    __shared__ uint32_t s_cnt[8];
    __shared__ uint16_t s_list[U8_ROWS];
    __shared__ uint16_t s_symA[WIDE_WAVES][32][64];
    __shared__ uint16_t s_symB[WIDE_WAVES][32][64];

    constexpr int RPT = U8_ROWS / WIDE_BLOCK;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    __builtin_amdgcn_s_setprio(1);
    const uint32_t totalA = offA[rowsA], totalB = offB[rowsB];
    const bool bcastA = rowsA == 1, bcastB = rowsB == 1;
    const uint64_t nchunks = (n + 63u) >> 6;
    const uint64_t nspans = (nchunks + U8_SPAN - 1) / U8_SPAN;
    // sps spans per super (1 .. WIDE_BLOCK / U8_SPAN, chosen by the launcher so that a mid-size frame still makes
    // enough workgroups to fill the chip)
    const uint64_t nsuper = (nspans + sps - 1) / sps;
    const uint32_t super_words = sps * (uint32_t)U8_SPAN;

    for (uint64_t sup = blockIdx.x; sup < nsuper; sup += gridDim.x) {
      const uint64_t cw = sup * super_words + tid;
      const unsigned long long myword = (tid < super_words && cw < nchunks) ? slowmask[cw] : 0ull;
      if (!__syncthreads_or(myword != 0ull)) continue;
      for (uint64_t span = sup * sps; span < (sup + 1) * sps && span < nspans; ++span) {
        const uint64_t c0 = span * U8_SPAN;
        if (tid < (uint32_t)U8_SPAN) s_mask[tid] = (c0 + tid < nchunks) ? slowmask[c0 + tid] : 0ull;
        if (tid < 8u) s_cnt[tid] = 0u;
        __syncthreads();
        const bool any = __ballot(s_mask[lane & (U8_SPAN - 1)] != 0ull) != 0ull; // same answer in every wave
        if (any) {
            uint32_t key[RPT], rank[RPT], len_a[RPT], len_b[RPT];
            // (the lengths of all RPT rows in flight at once, whole lines; looked at afterwards -- as in k_lane_wide [r4])
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const uint64_t r = c0 * 64u + (uint32_t)k * WIDE_BLOCK + tid, row = r < n ? r : n - 1u;
                const uint64_t ra = bcastA ? 0 : row, rb = bcastB ? 0 : row;
                STRSIM_CHECK_INDEX(K_UTF8, 1, row, ra + 1u, rowsA + 1u);
                STRSIM_CHECK_INDEX(K_UTF8, 2, row, rb + 1u, rowsB + 1u);
                len_a[k] = offA[ra + 1] - offA[ra];
                len_b[k] = offB[rb + 1] - offB[rb];
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const uint32_t i = k * WIDE_BLOCK + tid;
                key[k] = 0xFFFFFFFFu;
                if ((s_mask[i >> 6] >> (i & 63u)) & 1ull) {
                    const uint32_t la8 = len_a[k], lb8 = len_b[k];
                    const uint32_t mx = la8 > lb8 ? la8 : lb8, mn = la8 < lb8 ? la8 : lb8;
                    if (mx <= 128u && mx >= 1u) { // an empty side is fine here: the result is 0.0 without any DP
                        (void)mn;
                        key[k] = (mx - 1u) >> 4; // similar byte lengths together
                        rank[k] = atomicAdd(&s_cnt[key[k]], 1u);
                    }
                }
            }
            __syncthreads();
            uint32_t total = 0;
            {
                uint32_t c[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { c[q] = s_cnt[q]; total += c[q]; }
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    if (key[k] == 0xFFFFFFFFu) continue;
                    uint32_t base = 0;
#pragma unroll
                    for (int q = 0; q < 7; ++q) base += ((uint32_t)q < key[k]) ? c[q] : 0u;
                    STRSIM_CHECK_INDEX(K_UTF8, 3, c0 * 64u, base + rank[k], U8_ROWS);
                    s_list[base + rank[k]] = (uint16_t)(k * WIDE_BLOCK + tid);
                }
            }
            __syncthreads();
            const uint32_t nrounds = (total + 63u) >> 6;
            for (uint32_t rr = wv; rr < nrounds; rr += WIDE_WAVES) {
                // (cut from the long end: the round that is not full holds the list's cheapest rows, not its dearest)
                const uint32_t hi = total - 64u * rr, first = hi >= 64u ? hi - 64u : 0u;
                const uint32_t li = first + lane;
                const bool has = li < hi;
                if (has) STRSIM_CHECK_INDEX(K_UTF8, 4, c0 * 64u, li, U8_ROWS);
                const uint32_t i = has ? s_list[li] : 0u;
                const uint64_t row = c0 * 64u + i;
                uint32_t a0 = 0, la8 = 0, b0 = 0, lb8 = 0;
                if (has) {
                    const uint64_t ra = bcastA ? 0 : row, rb = bcastB ? 0 : row;
                    STRSIM_CHECK_INDEX(K_UTF8, 5, row, row, n);
                    STRSIM_CHECK_INDEX(K_UTF8, 6, row, ra + 1u, rowsA + 1u);
                    STRSIM_CHECK_INDEX(K_UTF8, 7, row, rb + 1u, rowsB + 1u);
                    a0 = offA[ra]; la8 = offA[ra + 1] - a0;
                    b0 = offB[rb]; lb8 = offB[rb + 1] - b0;
                }
                uint16_t *colA = &s_symA[wv][0][lane], *colB = &s_symB[wv][0][lane];
                bool big = false;
                uint32_t ov = 0u, av = 0xFFFFFFFFu;
                uint32_t la, lb;
                const uint32_t maxa = wave_max_u8(la8), maxb = wave_max_u8(lb8);
                if (maxa <= 64u && maxb <= 64u) { // the common case: 64-byte windows
                    la = decode_column<U8_DW / 2>(valA, a0, la8, totalA, has, maxa, colA, big, ov, av);
                    lb = decode_column<U8_DW / 2>(valB, b0, lb8, totalB, has, maxb, colB, big, ov, av);
                } else {
                    la = decode_column<U8_DW>(valA, a0, la8, totalA, has, maxa, colA, big, ov, av);
                    lb = decode_column<U8_DW>(valB, b0, lb8, totalB, has, maxb, colB, big, ov, av);
                }
                const bool empty = has && (la8 == 0u || lb8 == 0u); // exactly one side empty (both-empty rows never get here)
                const bool ok = has && !empty && !big && la <= 32u && lb <= 32u;
                if (empty) {
                    out[row] = 0.0; // strsim.rs:184-186, :290-292, :326-328; Levenshtein: 1 - max/max (:160)
                    atomicAnd(&s_mask[i >> 6], ~(1ull << (i & 63u)));
                }
                if (__ballot(ok) == 0ull) continue;
                const bool swap = la > lb; // the columns walk the shorter string (every measure is symmetric: strsim_lane_core.h)
                const uint16_t *tcol = swap ? colB : colA, *pcol = swap ? colA : colB;
                const uint32_t lt = ok ? (swap ? lb : la) : 1u, lp = ok ? (swap ? la : lb) : 1u;
                const uint32_t steps = wave_max_u8(ok ? lt : 0u);
                const uint32_t vary = (ov ^ av) & 0xFFFFu;
                const bool need16 = __ballot(ok && (vary >> 11)) != 0ull;
                const bool need11 = __ballot(ok && (vary >> 8)) != 0ull;
                double res;
                __builtin_amdgcn_s_setprio(0); // the column loops yield to waves that have loads to issue
                if (need16) res = lane_sym_result<MEASURE, 16>(LdsSym{tcol}, lt, steps, LdsSym{pcol}, lp);
                else if (need11) res = lane_sym_result<MEASURE, 11>(LdsSym{tcol}, lt, steps, LdsSym{pcol}, lp);
                else res = lane_sym_result<MEASURE, 8>(LdsSym{tcol}, lt, steps, LdsSym{pcol}, lp);
                __builtin_amdgcn_s_setprio(1);
                if (ok) {
                    STRSIM_CHECK_INDEX(K_UTF8, 10, row, row, n);
                    STRSIM_CHECK_INDEX(K_UTF8, 11, row, i >> 6, U8_SPAN);
                    out[row] = res;
                    atomicAnd(&s_mask[i >> 6], ~(1ull << (i & 63u)));
                }
            }
            __syncthreads();
            if (tid < (uint32_t)U8_SPAN && c0 + tid < nchunks) slowmask[c0 + tid] = s_mask[tid];
        }
        __syncthreads();
      }
      // This is the last per-lane kernel: the chunks that still hold rows go onto k_wave_pairs' work list
      // (order irrelevant), with their count and their rows for its work distribution.
      const unsigned long long fin = (tid < super_words && cw < nchunks) ? *reinterpret_cast<const volatile unsigned long long *>(slowmask + cw) : 0ull;
      const unsigned long long bal = __ballot(fin != 0ull);
      if (bal != 0ull) {
          uint32_t rows = (uint32_t)__popcll(fin);
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) rows += (uint32_t)__shfl_xor((int)rows, d);
          uint32_t pos = 0u;
          if (lane == 0u) {
              pos = atomicAdd(&status->list_count[MEASURE], (unsigned int)__popcll(bal));
              atomicAdd(&status->list_rows[MEASURE], rows);
          }
          pos = uniform(pos);
          if (fin != 0ull) STRSIM_CHECK_INDEX(K_UTF8, 12, cw, pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)), nchunks);
          if (fin != 0ull) worklist[pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)cw;
      }
    }
}
