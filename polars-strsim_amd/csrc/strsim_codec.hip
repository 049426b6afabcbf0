// strsim_codec.hip -- lossless 16-bit transport codec for result columns.
//
// Why: the one collective of the path is the gather of the f64 result shards to rank 0 over xGMI, and xGMI is
// point-to-point: every peer owns ONE link into the root (~50-70 GB/s per direction).  A 100 M-row shard is
// 800 MB of f64 -- 12-16 ms on a link, against ~2.3 ms of kernel time.  But a similarity of two strings of at
// most L characters can only take a small set of values (L = 32: 325 for Levenshtein, 631 for Jaccard/Dice,
// 22 856 for Jaro, 57 359 for Jaro-Winkler), because it is a fixed IEEE expression of a few small integers.
// The codec enumerates that set with the library's own epilogues (same operations, same order => same bits),
// sorts it, and ships the 16-bit rank of each value instead of the value: 4x fewer bytes per link, bit-exact.
// A value that is not in the table (a row with a longer string) gets code 0xFFFF and travels as an explicit
// (row, value) exception.
//
//   encode: f64[n] -> u16[n] (+ exceptions)    hash lookup keyed by the f64 bit pattern (open addressing, L2-resident)
//   decode: u16[n] -> f64[n]                   table[code]; 0xFFFF rows are left for the exception patch
#include <hip/hip_runtime.h>
#include "strsim_bounds.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "strsim_amd.h"
#include "strsim_internal.h"
#include "strsim_lane_core.h"

using namespace strsim;

struct strsim_codec {
    int measure = 0;
    uint32_t max_chars = 0;
    uint32_t entries = 0;
    uint32_t hash_mask = 0;
    double *d_table = nullptr;            // sorted distinct values, entries long
    unsigned long long *d_keys = nullptr; // hash table: value bit patterns (EMPTY = all ones)
    uint16_t *d_vals = nullptr;           // hash table: code of the key in the same slot
};

namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull; // a NaN pattern: never a similarity

__host__ __device__ inline uint32_t hash_bits(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xFF51AFD7ED558CCDull;
    k ^= k >> 33;
    return (uint32_t)k;
}

inline unsigned long long bits_of(double v)
{
    unsigned long long b;
    memcpy(&b, &v, 8);
    return b;
}

// every value `measure` can take on two strings of 0..L characters
void enumerate_values(int measure, uint32_t L, std::vector<double> &out)
{
    out.push_back(1.0); // both empty, or equal
    out.push_back(0.0); // one side empty
    switch (measure) {
    case LEVENSHTEIN:
        for (uint32_t den = 1; den <= L; ++den)
            for (uint32_t d = 0; d <= den; ++d) out.push_back(epilogue_levenshtein(d, den, den));
        break;
    case JACCARD:
    case SORENSEN_DICE:
        for (uint32_t la = 1; la <= L; ++la)
            for (uint32_t lb = la; lb <= L; ++lb)
                for (uint32_t i = 0; i <= la; ++i)
                    out.push_back(measure == JACCARD ? epilogue_jaccard(i, la, lb) : epilogue_sorensen_dice(i, la, lb));
        break;
    case JARO:
    case JARO_WINKLER:
        for (uint32_t la = 1; la <= L; ++la)
            for (uint32_t lb = 1; lb <= L; ++lb) {
                const uint32_t mn = la < lb ? la : lb;
                for (uint32_t m = 1; m <= mn; ++m)
                    for (uint32_t t = 0; t <= m; t += 2) { // only t/2 matters
                        const double j = epilogue_jaro(m, t, la, lb);
                        if (measure == JARO) { out.push_back(j); continue; }
                        const uint32_t pmax = mn < 4u ? mn : 4u;
                        for (uint32_t p = 0; p <= pmax; ++p) out.push_back(epilogue_jaro_winkler(j, p));
                    }
            }
        break;
    default: break;
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
}

__global__ void k_encode(const double *__restrict__ vals, uint64_t n, uint16_t *__restrict__ codes,
                         const unsigned long long *__restrict__ keys, const uint16_t *__restrict__ kvals, uint32_t mask,
                         uint32_t *__restrict__ exc_count, uint32_t *__restrict__ exc_rows, double *__restrict__ exc_vals,
                         uint32_t exc_cap)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double v = vals[i];
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        uint32_t h = hash_bits(b) & mask;
        uint16_t code = 0xFFFFu;
        for (;;) {
            const unsigned long long k = keys[h];
            if (k == b) { code = kvals[h]; break; }
            if (k == EMPTY_KEY) break;
            h = (h + 1u) & mask;
        }
        codes[i] = code;
        if (code == 0xFFFFu) {
            const uint32_t slot = atomicAdd(exc_count, 1u);
            if (slot < exc_cap) { exc_rows[slot] = (uint32_t)i; exc_vals[slot] = v; }
        }
    }
}

// four rows per thread: one 8-byte load of codes, four table reads (L2-resident table), two 16-byte stores
__global__ void k_decode(const uint16_t *__restrict__ codes, uint64_t n, double *__restrict__ out,
                         const double *__restrict__ table)
{
    const uint64_t nq = n >> 2;
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < nq; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 c4 = reinterpret_cast<const uint2 *>(codes)[q];
        const uint32_t c0 = c4.x & 0xFFFFu, c1 = c4.x >> 16, c2 = c4.y & 0xFFFFu, c3 = c4.y >> 16;
        double *o = out + 4 * q;
        if ((c0 != 0xFFFFu) & (c1 != 0xFFFFu) & (c2 != 0xFFFFu) & (c3 != 0xFFFFu)) {
            const double2 lo = {table[c0], table[c1]}, hi = {table[c2], table[c3]};
            reinterpret_cast<double2 *>(o)[0] = lo;
            reinterpret_cast<double2 *>(o)[1] = hi;
        } else {
            if (c0 != 0xFFFFu) o[0] = table[c0];
            if (c1 != 0xFFFFu) o[1] = table[c1];
            if (c2 != 0xFFFFu) o[2] = table[c2];
            if (c3 != 0xFFFFu) o[3] = table[c3];
        }
    }
    const uint64_t i = 4 * nq + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; // tail (< 4 rows)
    if (i < n) {
        const uint16_t c = codes[i];
        if (c != 0xFFFFu) out[i] = table[c];
    }
}

// Packed transport: `bits` bits per code (2^bits > entries; the all-ones code is the escape), 64 / bits codes per
// 64-bit word, one word per thread.  Rows past n code as 0.
__global__ void k_encode_packed(const double *__restrict__ vals, uint64_t n, unsigned long long *__restrict__ words,
                                uint64_t nwords, uint32_t bits, const unsigned long long *__restrict__ keys,
                                const uint16_t *__restrict__ kvals, uint32_t mask, uint32_t *__restrict__ exc_count,
                                uint32_t *__restrict__ exc_rows, double *__restrict__ exc_vals, uint32_t exc_cap)
{
    const uint32_t per = 64u / bits;
    const unsigned long long esc = (1ull << bits) - 1ull;
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long acc = 0ull;
        for (uint32_t j = 0; j < per; ++j) {
            const uint64_t i = w * per + j;
            if (i >= n) break;
            const double v = vals[i];
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            uint32_t h = hash_bits(b) & mask;
            unsigned long long code = esc;
            for (;;) {
                const unsigned long long k = keys[h];
                if (k == b) { code = kvals[h]; break; }
                if (k == EMPTY_KEY) break;
                h = (h + 1u) & mask;
            }
            if (code == esc) {
                const uint32_t slot = atomicAdd(exc_count, 1u);
                if (slot < exc_cap) { exc_rows[slot] = (uint32_t)i; exc_vals[slot] = v; }
            }
            acc |= code << (j * bits);
        }
        words[w] = acc;
    }
}

// Packed codes -> f64, one row per thread and trip (coalesced 8-byte stores; the `per` threads of a word share its load): row i
// sits in word i / per at bit (i % per) * bits.  [r5] Rows are walked in TILES of 256 whole words (per x 256 rows) so that the
// division is of a number below 8 192 by a launch constant -- one v_mul_hi_u32 with magic32 = ceil(2^32 / per), exact for
// x * (per - 1) < 2^32 -- where rounds 3-4 divided the 64-bit row index (a 64 x 64 -> 128-bit multiply per row: the root's decode
// of a gathered cfg2 column ran at 2.9 x its byte roofline, profiles/r5_root_rehearsal.txt); and a table of up to
// DECODE_LDS_ENTRIES values (Levenshtein: 325, Jaccard / Dice: 631) is read from LDS instead of through the address unit.
// Escaped rows are left untouched (patched from the exception list).
constexpr uint32_t DECODE_LDS_ENTRIES = 2048;
#ifndef STRSIM_DECODE_ROWS_PER_THREAD
#define STRSIM_DECODE_ROWS_PER_THREAD 2
#endif
constexpr int DECODE_RPT = STRSIM_DECODE_ROWS_PER_THREAD;
static_assert(DECODE_RPT == 2 || DECODE_RPT == 4, "pairs of rows: one or two 16-byte stores per thread");
template <bool LDS_TABLE>
__device__ __forceinline__ void decode_tiles(const unsigned long long *__restrict__ words, uint64_t n, uint32_t per, uint32_t bits,
                                             uint32_t magic32, double *__restrict__ out, const double *__restrict__ table,
                                             const double *s_tab, uint64_t tile0, uint64_t tile_step)
{
    const unsigned long long esc = (1ull << bits) - 1ull;
    const uint32_t tile_rows = per * 256u, tid = threadIdx.x;
    const uint64_t ntiles = (n + tile_rows - 1u) / tile_rows;
    // [r6] two rows per thread and ONE 16-byte store for the pair when neither row escaped (escapes are rare: rows outside the table) --
    // 8-byte-per-lane stores reach ~3.2 TB/s here, 16-byte ones more (the root of a gather at N = 8 writes 700 MB per column and step:
    // profiles/r6_root_rehearsal.txt).  Only when the pairs are 16-byte aligned (tile_rows is even; `out` decides: uniform).
    const bool pairs = (reinterpret_cast<uintptr_t>(out) & 15u) == 0u && (tile_rows & 1u) == 0u;
    for (uint64_t tile = tile0; tile < ntiles; tile += tile_step) {
        const uint64_t row0 = tile * tile_rows;
        const unsigned long long *__restrict__ const wt = words + tile * 256u;
        const uint32_t left = (uint32_t)(n - row0 < (uint64_t)tile_rows ? n - row0 : (uint64_t)tile_rows);
        if (pairs) {
            typedef double double2_a16 __attribute__((ext_vector_type(2), aligned(16)));
            // STRSIM_DECODE_ROWS_PER_THREAD (2 or 4) consecutive rows per thread: one or two 16-byte stores
            for (uint32_t x = (uint32_t)DECODE_RPT * tid; x < left; x += 256u * (uint32_t)DECODE_RPT) {
                uint32_t c[DECODE_RPT];
                double v[DECODE_RPT];
#pragma unroll
                for (int q = 0; q < DECODE_RPT; ++q) {
                    const uint32_t xq = x + (uint32_t)q, wl = __umulhi(xq, magic32), k = xq - wl * per;
                    c[q] = xq < left ? (uint32_t)((wt[wl] >> (k * bits)) & esc) : (uint32_t)esc; // (a row that does not exist: "escaped")
                }
#pragma unroll
                for (int q = 0; q < DECODE_RPT; ++q) v[q] = c[q] != (uint32_t)esc ? (LDS_TABLE ? s_tab[c[q]] : table[c[q]]) : 0.0;
#pragma unroll
                for (int q = 0; q < DECODE_RPT; q += 2) {
                    if (c[q] != (uint32_t)esc && c[q + 1] != (uint32_t)esc) {
                        double2_a16 w;
                        w.x = v[q]; w.y = v[q + 1];
                        *reinterpret_cast<double2_a16 *>(out + row0 + x + (uint32_t)q) = w;
                    } else {
                        if (c[q] != (uint32_t)esc) out[row0 + x + (uint32_t)q] = v[q];
                        if (c[q + 1] != (uint32_t)esc) out[row0 + x + (uint32_t)q + 1u] = v[q + 1];
                    }
                }
            }
            continue;
        }
        for (uint32_t x = tid; x < left; x += 256u) {
            const uint32_t wl = __umulhi(x, magic32); // x / per
            const uint32_t k = x - wl * per;
            const uint32_t code = (uint32_t)((wt[wl] >> (k * bits)) & esc);
            if (code != (uint32_t)esc) out[row0 + x] = LDS_TABLE ? s_tab[code] : table[code];
        }
    }
}

__global__ __launch_bounds__(256) void k_decode_packed(const unsigned long long *__restrict__ words, uint64_t n, uint32_t per, uint32_t bits,
                                                       uint32_t magic32, double *__restrict__ out, const double *__restrict__ table,
                                                       uint32_t entries)
{
    __shared__ double s_tab[DECODE_LDS_ENTRIES];
    if (entries <= DECODE_LDS_ENTRIES) {
        for (uint32_t i = threadIdx.x; i < entries; i += 256u) s_tab[i] = table[i];
        __syncthreads();
        decode_tiles<true>(words, n, per, bits, magic32, out, table, s_tab, blockIdx.x, gridDim.x);
    } else {
        decode_tiles<false>(words, n, per, bits, magic32, out, table, s_tab, blockIdx.x, gridDim.x);
    }
}

__global__ void k_patch(double *__restrict__ out, const uint32_t *__restrict__ rows, const double *__restrict__ vals,
                        uint32_t count, uint64_t row_base)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
        out[row_base + rows[i]] = vals[i];
}

// the same patch with the count read on the device (the shipped exception block of another rank: nothing about it is known
// on the host without a synchronisation); more exceptions than the block could hold -> *overflow is incremented
__global__ void k_patch_indirect(double *__restrict__ out, uint64_t row_base, const uint32_t *__restrict__ count_ptr,
                                 const uint32_t *__restrict__ rows, const double *__restrict__ vals, uint32_t cap,
                                 uint32_t *__restrict__ overflow)
{
    const uint32_t count = *count_ptr;
    if (count > cap && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(overflow, 1u);
    const uint32_t n = count < cap ? count : cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[row_base + rows[i]] = vals[i];
}

// The root of a gather in ONE launch: segment r of the receive buffer (blockIdx.y) holds rank r's codes -- packed (bits per
// code) or 16-bit -- and, code_bytes into the segment, its exception block (count | rows | values, cap entries); rows
// r * chunk .. of `out` are decoded and the exceptions written over the rows the codes escaped (disjoint rows: no order between
// the two is needed).  Replaces one decode + one patch launch per peer and column.
__global__ __launch_bounds__(256) void k_decode_gathered(const uint8_t *__restrict__ buf, uint64_t stride, uint32_t nseg, uint64_t chunk, uint64_t last_rows,
                                  uint32_t packed, uint32_t per, uint32_t bits, uint32_t magic32, uint64_t code_bytes,
                                  uint32_t cap, double *__restrict__ out, const double *__restrict__ table, uint32_t entries,
                                  uint32_t *__restrict__ overflow, uint32_t seg0)
{
    __shared__ double s_tab[DECODE_LDS_ENTRIES];
    const uint32_t seg = seg0 + blockIdx.y; // (seg0: the root's own segment is not coded at all -- strsim_codec_decode_gathered_from)
    const uint8_t *base = buf + (uint64_t)seg * stride;
    const uint64_t n = seg + 1u == nseg ? last_rows : chunk;
    double *o = out + (uint64_t)seg * chunk;
    if (packed) {
        const unsigned long long *words = reinterpret_cast<const unsigned long long *>(base);
        if (entries <= DECODE_LDS_ENTRIES) { // (uniform over the launch)
            for (uint32_t i = threadIdx.x; i < entries; i += 256u) s_tab[i] = table[i];
            __syncthreads();
            decode_tiles<true>(words, n, per, bits, magic32, o, table, s_tab, blockIdx.x, gridDim.x);
        } else {
            decode_tiles<false>(words, n, per, bits, magic32, o, table, s_tab, blockIdx.x, gridDim.x);
        }
    } else {
        const uint16_t *codes = reinterpret_cast<const uint16_t *>(base);
        for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
            const uint16_t c = codes[i];
            if (c != 0xFFFFu) o[i] = table[c];
        }
    }
    const uint32_t count = *reinterpret_cast<const uint32_t *>(base + code_bytes);
    const uint32_t *rows = reinterpret_cast<const uint32_t *>(base + code_bytes + 16u);
    const double *vals = reinterpret_cast<const double *>(base + code_bytes + 16u + 4ull * cap);
    if (count > cap && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(overflow, 1u);
    const uint32_t m = count < cap ? count : cap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) o[rows[i]] = vals[i];
}

// ---- offsets from lengths -------------------------------------------------------------------------------------------
// The host ships a column's string LENGTHS as one byte per row (strings of at most 255 bytes) instead of its u32 offsets --
// 35 instead of 41 bytes per pair over PCIe for cfg2's lengths -- and the offsets are rebuilt here: offsets[0] = 0,
// offsets[i + 1] = offsets[i] + lengths[i].  Two launches: sums of blocks of LEN_ROWS rows, then every workgroup adds up the
// sums in front of its block (at most SCAN_WS_WORDS of them, L2-resident) and scans its own rows.
constexpr int LEN_THREADS = 256, LEN_PER_THREAD = 16, LEN_ROWS = LEN_THREADS * LEN_PER_THREAD;

// the 16 lengths of thread `t` of the block that starts at row r0 (zeros behind the last row), as four dwords
__device__ __forceinline__ uint4 load_lengths16(const uint8_t *__restrict__ len, uint64_t rows, uint64_t r0, uint32_t t)
{
    const uint64_t at = r0 + (uint64_t)t * LEN_PER_THREAD;
    if (at + LEN_PER_THREAD <= rows) return *reinterpret_cast<const uint4 *>(len + at); // (len is 16-byte aligned, r0 a multiple of 4096)
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int k = 0; k < LEN_PER_THREAD; ++k)
        if (at + (uint64_t)k < rows) w[k >> 2] |= (uint32_t)len[at + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t sum_bytes16(uint4 v)
{
    uint32_t s = __builtin_amdgcn_sad_u8(v.x, 0u, 0u);
    s = __builtin_amdgcn_sad_u8(v.y, 0u, s);
    s = __builtin_amdgcn_sad_u8(v.z, 0u, s);
    return __builtin_amdgcn_sad_u8(v.w, 0u, s);
}

// sum over the workgroup of one value per thread (every thread gets it)
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *s_part)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) s_part[wv] = v;
    __syncthreads();
    uint32_t tot = 0;
    for (int w = 0; w < LEN_THREADS / 64; ++w) tot += s_part[w];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(LEN_THREADS) void k_len_block_sums(const uint8_t *__restrict__ len, uint64_t rows, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_part[LEN_THREADS / 64];
    const uint32_t tot = block_sum(sum_bytes16(load_lengths16(len, rows, (uint64_t)blockIdx.x * LEN_ROWS, threadIdx.x)), s_part);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(LEN_THREADS) void k_len_offsets(const uint8_t *__restrict__ len, uint64_t rows, const uint32_t *__restrict__ sums,
                                                            uint32_t *__restrict__ off)
{
    __shared__ uint32_t s_part[LEN_THREADS / 64];
    __shared__ uint32_t s_wave[LEN_THREADS / 64];
    __shared__ uint32_t s_off[LEN_ROWS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * LEN_ROWS;
    // bytes in front of this block
    uint32_t mine = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += LEN_THREADS) mine += sums[b];
    const uint32_t base = block_sum(mine, s_part);
    // this thread's 16 rows: their total, its exclusive prefix over the workgroup, then the 16 running ends
    const uint4 v = load_lengths16(len, rows, r0, tid);
    const uint32_t tot = sum_bytes16(v);
    uint32_t inc = tot;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += up;
    }
    if (lane == 63u) s_wave[wv] = inc;
    __syncthreads();
    uint32_t run = base + inc - tot;
    for (uint32_t w = 0; w < wv; ++w) run += s_wave[w];
    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < LEN_PER_THREAD; ++k) {
        run += (w4[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        s_off[tid * LEN_PER_THREAD + k] = run; // offsets[r0 + 16 tid + k + 1]
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) off[0] = 0u;
    const uint64_t left = rows - r0;
    const uint32_t cnt = left < (uint64_t)LEN_ROWS ? (uint32_t)left : (uint32_t)LEN_ROWS;
    for (uint32_t i = tid; i < cnt; i += LEN_THREADS) off[r0 + i + 1] = s_off[i];
}

// ---- a column from Utf8View slots (strsim_column_from_views) ------------------------------------------------------------
// The engine holds a String column as 16-byte VIEWS (length; the string itself when it has at most 12 bytes, else its first four
// bytes, a buffer index and an offset): the reference iterates them in place (strsim.rs:46-47).  A host that does not want to
// gather the strings ships the views as they lie -- a streaming copy -- plus the bytes of the strings that do not fit their
// view, and the column layout the kernels take (offsets + packed values) is made here: block sums of the lengths, then every
// workgroup scans its rows and each thread writes its own row's bytes (consecutive lanes write consecutive bytes of `values`).
constexpr int VIEW_THREADS = 256, VIEW_STRIPS = 8, VIEW_ROWS = VIEW_THREADS * VIEW_STRIPS;

__global__ __launch_bounds__(VIEW_THREADS) void k_view_block_sums(const uint4 *__restrict__ views, uint64_t rows, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t s_part[VIEW_THREADS / 64];
    const uint64_t r0 = (uint64_t)blockIdx.x * VIEW_ROWS;
    // (sums SATURATE at 2^32 - 1: a malformed slot with an absurd length must not wrap a later row's offset back into range)
    auto sat_add = [](uint32_t a, uint32_t b) { const uint32_t c = a + b; return c < a ? 0xFFFFFFFFu : c; };
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < VIEW_STRIPS; ++k) {
        const uint64_t r = r0 + (uint64_t)k * VIEW_THREADS + threadIdx.x;
        if (r < rows) mine = sat_add(mine, reinterpret_cast<const uint32_t *>(views + r)[0]);
    }
    for (int d = 32; d >= 1; d >>= 1) mine = sat_add(mine, (uint32_t)__shfl_xor((int)mine, d));
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < VIEW_THREADS / 64; ++w) tot = sat_add(tot, s_part[w]);
        sums[blockIdx.x] = tot;
    }
}

// long_bytes / values_bytes: the extents of `longs` and `values` (~0: not stated by the caller); a row whose bytes would come from
// outside `longs` or land outside `values` is NOT copied and counted in *bad (if given): a malformed slot cannot fault the GPU.
__global__ __launch_bounds__(VIEW_THREADS) void k_view_column(const uint4 *__restrict__ views, uint64_t rows, const uint8_t *__restrict__ longs,
                                                             uint64_t long_bytes, const uint32_t *__restrict__ sums, uint32_t *__restrict__ off,
                                                             uint8_t *__restrict__ values, uint64_t values_bytes, uint32_t *__restrict__ bad)
{
    typedef uint32_t u32_u __attribute__((aligned(1)));
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * VIEW_ROWS;
    // bytes in front of this block.  The prefix in 64 bits (the 32-bit running offset was unchecked): a malformed slot with an absurd length -- or a
    // block sum that saturated -- puts every later row beyond 32-bit offsets, i.e. out of range below, instead of wrapping it back in.
    __shared__ unsigned long long s_part64[VIEW_THREADS / 64], s_wave64[VIEW_THREADS / 64];
    unsigned long long mine = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += VIEW_THREADS) mine += sums[b];
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if (lane == 0u) s_part64[wv] = mine;
    __syncthreads();
    unsigned long long run = 0;
    for (int w = 0; w < VIEW_THREADS / 64; ++w) run += s_part64[w];
    // (offsets are 32 bits, and the block sums SATURATE at 2^32 - 1: an end offset of 2^32 - 1 or more is out of range whatever the
    //  caller stated -- ADVICE r5: with the extents not stated, a one-byte row behind a saturated sum passed `<= 2^32` and was written
    //  at values + 0xFFFFFFFF)
    const unsigned long long cap = values_bytes < 0xFFFFFFFEull ? values_bytes : 0xFFFFFFFEull;
    if (blockIdx.x == 0 && tid == 0) off[0] = 0u;
#pragma unroll 1
    for (int k = 0; k < VIEW_STRIPS; ++k) {
        const uint64_t r = r0 + (uint64_t)k * VIEW_THREADS + tid;
        const uint4 v = r < rows ? views[r] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t len = v.x;
        unsigned long long inc = len;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += up;
        }
        if (lane == 63u) s_wave64[wv] = inc;
        __syncthreads();
        unsigned long long at = run + inc - len, strip = 0;
        for (uint32_t w = 0; w < (uint32_t)(VIEW_THREADS / 64); ++w) {
            if (w < wv) at += s_wave64[w];
            strip += s_wave64[w];
        }
        __syncthreads();
        run += strip;
        if (r < rows) {
            STRSIM_CHECK_INDEX(K_VIEWS, 1, r, r + 1u, rows + 1u); // (lab: offsets[rows + 1])
            off[r + 1] = (uint32_t)(at + len);
            const bool in_values = at + len <= cap;
            const bool in_longs = len <= 12u || (longs != nullptr && (uint64_t)v.w + len <= long_bytes);
            if (!(in_values && in_longs)) {
                if (bad) atomicAdd(bad, 1u);
                continue;
            }
            uint8_t *__restrict__ const dst = values + at;
            STRSIM_CHECK_RANGE(K_VIEWS, 2, r, at + len, 0, cap);                                      // the packed values
            if (len > 12u) STRSIM_CHECK_RANGE(K_VIEWS, 3, r, (unsigned long long)v.w + len, 0, long_bytes); // the long strings
            if (len <= 12u) { // the string is in the view: whole dwords, then the tail byte by byte
                if (len >= 4u) *reinterpret_cast<u32_u *>(dst) = v.y;
                if (len >= 8u) *reinterpret_cast<u32_u *>(dst + 4) = v.z;
                if (len >= 12u) *reinterpret_cast<u32_u *>(dst + 8) = v.w;
                const uint32_t tail = len & ~3u, word = tail == 0u ? v.y : (tail == 4u ? v.z : v.w);
                for (uint32_t b = 0; b < (len & 3u); ++b) dst[tail + b] = (uint8_t)(word >> (8u * b));
            } else { // ... in the bytes shipped beside the views, at offset v.w (the host has made it so)
                const uint8_t *__restrict__ const src = longs + v.w;
                uint32_t b = 0;
                for (; b + 16u <= len; b += 16u) *reinterpret_cast<u32x4_u *>(dst + b) = *reinterpret_cast<const u32x4_u *>(src + b);
                for (; b < len; ++b) dst[b] = src[b];
            }
        }
    }
}

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);    \
    } while (0)

unsigned grid_for(uint64_t n)
{
    uint64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

} // namespace

// ---- segment compaction (strsim_compact_segments): one workgroup column per segment, 16 bytes per lane and step ----------------
struct CompactArgs {
    uint64_t src_off[STRSIM_COMPACT_MAX_SEGMENTS], dst_off[STRSIM_COMPACT_MAX_SEGMENTS], bytes[STRSIM_COMPACT_MAX_SEGMENTS];
};
__global__ __launch_bounds__(256) void k_compact_segments(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, CompactArgs a)
{
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
    const uint32_t k = blockIdx.y;
    const uint8_t *s = src + a.src_off[k];
    uint8_t *d = dst + a.dst_off[k];
    const uint64_t n = a.bytes[k], n16 = n & ~(uint64_t)15;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u; i < n16; i += (uint64_t)gridDim.x * 256u * 16u)
        *reinterpret_cast<u32x4_u *>(d + i) = *reinterpret_cast<const u32x4_u *>(s + i);
    if (blockIdx.x == 0u && threadIdx.x < (uint32_t)(n - n16)) d[n16 + threadIdx.x] = s[n16 + threadIdx.x];
}

extern "C" {

int strsim_codec_create(strsim_ctx_t *ctx, int measure, uint32_t max_chars, strsim_codec_t **out)
{
    if (!ctx || !out) { set_error("strsim_codec_create: NULL argument"); return STRSIM_ERR_ARG; }
    *out = nullptr;
    if (measure < 0 || measure >= STRSIM_NUM_MEASURES || max_chars == 0 || max_chars > 255) {
        set_error("strsim_codec_create: bad measure/max_chars");
        return STRSIM_ERR_ARG;
    }
    std::vector<double> vals;
    enumerate_values(measure, max_chars, vals);
    if (vals.size() >= 0xFFFFu) {
        set_error("strsim_codec_create: %zu distinct values do not fit 16-bit codes (measure %d, max_chars %u)", vals.size(),
                  measure, max_chars);
        return STRSIM_ERR_ARG;
    }
    uint32_t hsize = 64;
    while (hsize < 4 * vals.size()) hsize <<= 1;
    std::vector<unsigned long long> keys(hsize, EMPTY_KEY);
    std::vector<uint16_t> kvals(hsize, 0);
    for (size_t i = 0; i < vals.size(); ++i) {
        const unsigned long long b = bits_of(vals[i]);
        uint32_t h = hash_bits(b) & (hsize - 1);
        while (keys[h] != EMPTY_KEY) h = (h + 1) & (hsize - 1);
        keys[h] = b;
        kvals[h] = (uint16_t)i;
    }
    strsim_codec *c = new (std::nothrow) strsim_codec();
    if (!c) { set_error("out of host memory"); return STRSIM_ERR_OOM; }
    c->measure = measure; c->max_chars = max_chars; c->entries = (uint32_t)vals.size(); c->hash_mask = hsize - 1;
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    hipError_t e = hipMalloc((void **)&c->d_table, vals.size() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_keys, hsize * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_vals, hsize * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_table, vals.data(), vals.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_keys, keys.data(), hsize * sizeof(unsigned long long), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_vals, kvals.data(), hsize * sizeof(uint16_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { strsim_codec_destroy(c); return hip_fail(e, "codec table upload"); }
    *out = c;
    return STRSIM_OK;
}

void strsim_codec_destroy(strsim_codec_t *c)
{
    if (!c) return;
    if (c->d_table) (void)hipFree(c->d_table);
    if (c->d_keys) (void)hipFree(c->d_keys);
    if (c->d_vals) (void)hipFree(c->d_vals);
    delete c;
}

uint32_t strsim_codec_entries(const strsim_codec_t *c) { return c ? c->entries : 0; }

int strsim_codec_encode(strsim_ctx_t *ctx, const strsim_codec_t *c, const double *vals, uint64_t n, uint16_t *codes,
                        uint32_t *exc_count, uint32_t *exc_rows, double *exc_vals, uint32_t exc_cap)
{
    if (!ctx || !c || (!vals && n) || (!codes && n) || !exc_count) { set_error("strsim_codec_encode: NULL argument"); return STRSIM_ERR_ARG; }
    if (n >> 32) { set_error("strsim_codec_encode: more than 2^32 rows per call"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    HIP_TRY(hipMemsetAsync(exc_count, 0, sizeof(uint32_t), st));
    if (n == 0) return STRSIM_OK;
    hipLaunchKernelGGL(k_encode, dim3(grid_for(n)), dim3(256), 0, st, vals, n, codes, c->d_keys, c->d_vals, c->hash_mask,
                       exc_count, exc_rows, exc_vals, exc_cap);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_codec_decode(strsim_ctx_t *ctx, const strsim_codec_t *c, const uint16_t *codes, uint64_t n, double *out)
{
    if (!ctx || !c || (!codes && n) || (!out && n)) { set_error("strsim_codec_decode: NULL argument"); return STRSIM_ERR_ARG; }
    if (n == 0) return STRSIM_OK;
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    if ((reinterpret_cast<uintptr_t>(codes) & 7u) || (reinterpret_cast<uintptr_t>(out) & 15u)) {
        set_error("strsim_codec_decode: codes must be 8-byte and out 16-byte aligned");
        return STRSIM_ERR_ARG;
    }
    hipLaunchKernelGGL(k_decode, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, codes, n, out, c->d_table);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_codec_patch(strsim_ctx_t *ctx, double *out, uint64_t row_base, const uint32_t *exc_rows, const double *exc_vals,
                       uint32_t count)
{
    if (!ctx || (count && (!out || !exc_rows || !exc_vals))) { set_error("strsim_codec_patch: NULL argument"); return STRSIM_ERR_ARG; }
    if (count == 0) return STRSIM_OK;
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    hipLaunchKernelGGL(k_patch, dim3(grid_for(count)), dim3(256), 0, st, out, exc_rows, exc_vals, count, row_base);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_codec_patch_indirect(strsim_ctx_t *ctx, double *out, uint64_t row_base, const uint32_t *exc_count,
                                const uint32_t *exc_rows, const double *exc_vals, uint32_t exc_cap, uint32_t *overflow)
{
    if (!ctx || !out || !exc_count || !exc_rows || !exc_vals || !overflow) { set_error("strsim_codec_patch_indirect: NULL argument"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    hipLaunchKernelGGL(k_patch_indirect, dim3(32), dim3(256), 0, st, out, row_base, exc_count, exc_rows, exc_vals, exc_cap, overflow);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_compact_segments(strsim_ctx_t *ctx, const uint8_t *src, uint8_t *dst, const uint64_t *src_off,
                                     const uint64_t *dst_off, const uint64_t *bytes, int nseg)
{
    if (!ctx || !src || !dst || nseg < 0 || nseg > STRSIM_COMPACT_MAX_SEGMENTS) { set_error("strsim_compact_segments: NULL buffer or more than %d segments", STRSIM_COMPACT_MAX_SEGMENTS); return STRSIM_ERR_ARG; }
    if (nseg == 0) return STRSIM_OK;
    CompactArgs a{};
    uint64_t mx = 0;
    for (int k = 0; k < nseg; ++k) { a.src_off[k] = src_off[k]; a.dst_off[k] = dst_off[k]; a.bytes[k] = bytes[k]; mx = std::max<uint64_t>(mx, bytes[k]); }
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((mx + 4095) / 4096, 1), 1024);
    hipLaunchKernelGGL(k_compact_segments, dim3(gx, (unsigned)nseg), dim3(256), 0, (hipStream_t)strsim_ctx_stream(ctx), src, dst, a);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_offsets_from_lengths(strsim_ctx_t *ctx, const uint8_t *lengths, uint64_t rows, uint32_t *offsets)
{
    if (!ctx || !offsets || (!lengths && rows)) { set_error("strsim_offsets_from_lengths: NULL argument"); return STRSIM_ERR_ARG; }
    const uint64_t nblk = (rows + LEN_ROWS - 1) / LEN_ROWS;
    // the running sum is 32 bits wide, like the offsets themselves: a call may hold only as many rows as cannot wrap it
    // whatever the lengths are (255 each)
    const uint64_t max_rows = std::min<uint64_t>(strsim::SCAN_WS_WORDS * (uint64_t)LEN_ROWS, (uint64_t)STRSIM_OFFSETS_FROM_LENGTHS_MAX_ROWS);
    if (rows > max_rows) {
        set_error("strsim_offsets_from_lengths: %llu rows in one call; at most %llu (32-bit offsets of strings of up to 255 bytes)",
                  (unsigned long long)rows, (unsigned long long)max_rows);
        return STRSIM_ERR_ARG;
    }
    if (reinterpret_cast<uintptr_t>(lengths) & 15u) { set_error("strsim_offsets_from_lengths: lengths must be 16-byte aligned"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    if (rows == 0) {
        HIP_TRY(hipMemsetAsync(offsets, 0, sizeof(uint32_t), st));
        return STRSIM_OK;
    }
    uint32_t *sums = nullptr;
    const int rc = strsim_internal_scan_workspace(ctx, &sums);
    if (rc) return rc;
    hipLaunchKernelGGL(k_len_block_sums, dim3((unsigned)nblk), dim3(LEN_THREADS), 0, st, lengths, rows, sums);
    hipLaunchKernelGGL(k_len_offsets, dim3((unsigned)nblk), dim3(LEN_THREADS), 0, st, lengths, rows, sums, offsets);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_column_from_views(strsim_ctx_t *ctx, const void *views, uint64_t rows, const uint8_t *long_values, uint32_t *offsets,
                             uint8_t *values)
{
    // (extents not stated: a NULL long_values still makes every slot beyond 12 bytes "out of range" -- skipped, never dereferenced)
    return strsim_column_from_views_bounded(ctx, views, rows, long_values, long_values ? ~0ull : 0ull, offsets, values, ~0ull, nullptr);
}

int strsim_column_from_views_bounded(strsim_ctx_t *ctx, const void *views, uint64_t rows, const uint8_t *long_values, uint64_t long_bytes,
                                     uint32_t *offsets, uint8_t *values, uint64_t values_bytes, uint32_t *malformed)
{
    if (!ctx || !offsets || (rows && (!views || !values))) { set_error("strsim_column_from_views: NULL argument"); return STRSIM_ERR_ARG; }
    if (!long_values) long_bytes = 0;
    const uint64_t nblk = (rows + VIEW_ROWS - 1) / VIEW_ROWS;
    if (nblk > strsim::SCAN_WS_WORDS) {
        set_error("strsim_column_from_views: %llu rows in one call; at most %llu", (unsigned long long)rows,
                  (unsigned long long)(strsim::SCAN_WS_WORDS * (uint64_t)VIEW_ROWS));
        return STRSIM_ERR_ARG;
    }
    if (reinterpret_cast<uintptr_t>(views) & 15u) { set_error("strsim_column_from_views: views must be 16-byte aligned"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    if (rows == 0) {
        HIP_TRY(hipMemsetAsync(offsets, 0, sizeof(uint32_t), st));
        return STRSIM_OK;
    }
    uint32_t *sums = nullptr;
    const int rc = strsim_internal_scan_workspace(ctx, &sums);
    if (rc) return rc;
    const uint4 *v = static_cast<const uint4 *>(views);
    hipLaunchKernelGGL(k_view_block_sums, dim3((unsigned)nblk), dim3(VIEW_THREADS), 0, st, v, rows, sums);
    hipLaunchKernelGGL(k_view_column, dim3((unsigned)nblk), dim3(VIEW_THREADS), 0, st, v, rows, long_values, long_bytes, sums, offsets, values,
                       values_bytes, malformed);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_codec_decode_gathered(strsim_ctx_t *ctx, const strsim_codec_t *c, const void *buf, uint64_t seg_stride_bytes,
                                 uint32_t nseg, uint64_t chunk_rows, uint64_t last_rows, int packed, uint64_t code_bytes,
                                 uint32_t exc_cap, double *out, uint32_t *overflow)
{
    return strsim_codec_decode_gathered_from(ctx, c, buf, seg_stride_bytes, 0u, nseg, chunk_rows, last_rows, packed, code_bytes, exc_cap, out, overflow);
}

int strsim_codec_decode_gathered_from(strsim_ctx_t *ctx, const strsim_codec_t *c, const void *buf, uint64_t seg_stride_bytes,
                                      uint32_t first_seg, uint32_t nseg, uint64_t chunk_rows, uint64_t last_rows, int packed,
                                      uint64_t code_bytes, uint32_t exc_cap, double *out, uint32_t *overflow)
{
    if (first_seg >= nseg) {
        if (first_seg == nseg && ctx && c && buf && out && overflow) return STRSIM_OK; // nothing to decode
        set_error("strsim_codec_decode_gathered: first_seg beyond the segments");
        return STRSIM_ERR_ARG;
    }
    if (!ctx || !c || !buf || !out || !overflow || nseg == 0) { set_error("strsim_codec_decode_gathered: NULL argument"); return STRSIM_ERR_ARG; }
    if ((reinterpret_cast<uintptr_t>(buf) & 7u) || (seg_stride_bytes & 7u) || (code_bytes & 15u)) {
        set_error("strsim_codec_decode_gathered: the buffer and the segment stride must be 8-byte, code_bytes 16-byte aligned");
        return STRSIM_ERR_ARG;
    }
    if (nseg > 65535u || (chunk_rows >> 32) || (last_rows >> 32)) { set_error("strsim_codec_decode_gathered: too many segments / rows"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    const uint32_t bits = strsim_codec_bits(c), per = 64u / bits;
    const uint32_t magic32 = (uint32_t)(((1ull << 32) + per - 1u) / per); // ceil(2^32 / per): x / per for x < 2^32 / per
    const uint64_t mx = std::max<uint64_t>(chunk_rows, last_rows);
    // workgroups per segment: one per tile of per x 256 rows (packed) / 256 rows, at most 8 192 / segments -- a few per CU and segment
    const uint64_t units = packed ? (mx + per * 256ull - 1) / (per * 256ull) : (mx + 255) / 256;
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>(units, 1), std::max<uint64_t>(8192 / nseg, 256));
    hipLaunchKernelGGL(k_decode_gathered, dim3(gx, nseg - first_seg), dim3(256), 0, st, static_cast<const uint8_t *>(buf), seg_stride_bytes, nseg,
                       chunk_rows, last_rows, packed ? 1u : 0u, per, bits, magic32, code_bytes, exc_cap, out, c->d_table, c->entries, overflow,
                       first_seg);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

uint32_t strsim_codec_bits(const strsim_codec_t *c)
{
    if (!c) return 0;
    uint32_t b = 1;
    while ((1ull << b) <= (unsigned long long)c->entries) ++b; // 2^b > entries: the all-ones code stays free
    return b;
}

uint64_t strsim_codec_packed_words(const strsim_codec_t *c, uint64_t n)
{
    if (!c) return 0;
    const uint64_t per = 64u / strsim_codec_bits(c);
    return (n + per - 1) / per;
}

int strsim_codec_encode_packed(strsim_ctx_t *ctx, const strsim_codec_t *c, const double *vals, uint64_t n, uint64_t *words,
                               uint32_t *exc_count, uint32_t *exc_rows, double *exc_vals, uint32_t exc_cap)
{
    if (!ctx || !c || (!vals && n) || (!words && n) || !exc_count) { set_error("strsim_codec_encode_packed: NULL argument"); return STRSIM_ERR_ARG; }
    if (n >> 32) { set_error("strsim_codec_encode_packed: more than 2^32 rows per call"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    HIP_TRY(hipMemsetAsync(exc_count, 0, sizeof(uint32_t), st));
    if (n == 0) return STRSIM_OK;
    const uint64_t nwords = strsim_codec_packed_words(c, n);
    hipLaunchKernelGGL(k_encode_packed, dim3(grid_for(nwords)), dim3(256), 0, st, vals, n,
                       reinterpret_cast<unsigned long long *>(words), nwords, strsim_codec_bits(c), c->d_keys, c->d_vals,
                       c->hash_mask, exc_count, exc_rows, exc_vals, exc_cap);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

int strsim_codec_decode_packed(strsim_ctx_t *ctx, const strsim_codec_t *c, const uint64_t *words, uint64_t n, double *out)
{
    if (!ctx || !c || (!words && n) || (!out && n)) { set_error("strsim_codec_decode_packed: NULL argument"); return STRSIM_ERR_ARG; }
    if (n == 0) return STRSIM_OK;
    if (reinterpret_cast<uintptr_t>(words) & 7u) { set_error("strsim_codec_decode_packed: words must be 8-byte aligned"); return STRSIM_ERR_ARG; }
    hipStream_t st = (hipStream_t)strsim_ctx_stream(ctx);
    const uint32_t bits = strsim_codec_bits(c), per = 64u / bits;
    const uint32_t magic32 = (uint32_t)(((1ull << 32) + per - 1u) / per); // ceil(2^32 / per)
    const uint64_t tiles = (n + per * 256ull - 1) / (per * 256ull);
    hipLaunchKernelGGL(k_decode_packed, dim3((unsigned)std::min<uint64_t>(std::max<uint64_t>(tiles, 1), 8192)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned long long *>(words), n, per, bits, magic32, out, c->d_table, c->entries);
    HIP_TRY(hipGetLastError());
    return STRSIM_OK;
}

#if STRSIM_BOUNDS_ON
// lab build only: the address-check record of this translation unit (the view kernels), read and cleared like strsim_debug_bounds_kernels
__attribute__((visibility("default"))) int strsim_debug_bounds_codec(unsigned long long *dst)
{
    strsim::bounds::Record r{};
    hipError_t e = hipMemcpyFromSymbol(&r, HIP_SYMBOL(strsim::bounds::g_rec), sizeof(r), 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    dst[0] = r.hits; dst[1] = r.site; dst[2] = r.row; dst[3] = r.value; dst[4] = r.lo; dst[5] = r.hi;
    const strsim::bounds::Record z{};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(strsim::bounds::g_rec), &z, sizeof(z), 0, hipMemcpyHostToDevice);
}
#endif

} // extern "C"
