// polars_plugin.cpp -- the Polars expression-plugin ABI (include/polars_plugin_abi.h) on top of the
// gfx950 kernels: the C++ counterpart of the reference's macro glue (reference src/expressions/mod.rs:8-31)
// and of `strsim::parallel_apply` (reference src/expressions/strsim.rs:41-107).
//
// What happens per call (one Polars expression evaluation):
//   1. both input Series (any chunking; Arrow "vu" string views as Polars >= 0.20 sends them, or "u"/"U"
//      offset layouts) are described in place; shape rule and literal broadcast exactly as strsim.rs:48-52,61-66;
//   2. rows are cut into slices of 2 M rows; each slice is packed by helper threads (none when the engine says it
//      is already parallel, strsim.rs:53) into the device layout -- uint32 offsets + packed UTF-8 values -- in pinned
//      staging memory;
//   3. a two-slot software pipeline overlaps packing slice k+1 with H2D + kernels of slice k (strsim_pairs_device);
//   4. the f64 column comes back as one Arrow "g" chunk whose validity is the AND of the inputs'
//      validities (null in -> null out, README.md:69-70); values under null slots are computed like the
//      reference's arity helpers do and are never observable.
// Nothing here computes similarities on the CPU; without a GPU the call fails with the plugin's error message.
//
// Documented divergences from the reference: a length-1 literal on the LEFT is broadcast over
// every row of the right column in both engine modes (the reference's rayon branch splits by a.len() == 1,
// strsim.rs:73, and silently returns one row); a NULL literal yields an all-null column instead of the
// reference's `unwrap()` panic (strsim.rs:62,65,87,90).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <emmintrin.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include "polars_plugin_abi.h"
#include "strsim_amd.h"
#include "strsim_internal.h"

namespace {

thread_local std::string g_plugin_error;

struct PluginError {
    std::string msg;
};

[[noreturn]] void fail(const std::string &m) { throw PluginError{m}; }

// ---- one input Series, described (no copy) --------------------------------------------------------
struct Chunk {
    const ArrowArray *a;
    uint64_t row0;         // first row of this chunk within the Series
    const uint8_t *nulls;  // validity bitmap if the chunk has nulls, else nullptr
};

enum Layout { L_VIEW, L_U32, L_U64 };

struct Column {
    Layout layout = L_VIEW;
    std::vector<Chunk> chunks; // non-empty chunks only
    uint64_t rows = 0;
    bool any_null = false;
    std::string name;
};

inline bool bit_at(const uint8_t *bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

struct View { // Arrow BinaryView / Utf8View element
    uint32_t len;
    uint8_t rest[12]; // <= 12 bytes inline, else {prefix[4], buffer_index u32, offset u32}
};

void describe(const SeriesExport &s, Column &c)
{
    if (!s.field || !s.field->format) fail("input series has no schema");
    const std::string fmt = s.field->format;
    if (fmt == "vu") c.layout = L_VIEW;
    else if (fmt == "u") c.layout = L_U32;
    else if (fmt == "U") c.layout = L_U64;
    else fail("invalid series dtype: expected `String`, got Arrow format `" + fmt + "`"); // `.str()?`, strsim.rs:46-47
    c.name = s.field->name ? s.field->name : "";
    for (size_t k = 0; k < s.len; ++k) {
        const ArrowArray *a = s.arrays[k];
        if (!a) fail("null chunk pointer");
        if (a->length < 0 || a->offset < 0) fail("negative length/offset in chunk");
        if (a->length == 0) continue;
        if (c.layout == L_VIEW ? a->n_buffers < 2 : a->n_buffers < 3) fail("string chunk is missing buffers");
        const uint8_t *vb = (a->null_count != 0 && a->n_buffers > 0) ? static_cast<const uint8_t *>(a->buffers[0]) : nullptr;
        if (vb) c.any_null = true;
        c.chunks.push_back(Chunk{a, c.rows, vb});
        c.rows += (uint64_t)a->length;
    }
}

// first chunk containing row r (r < rows)
inline size_t chunk_of(const Column &c, uint64_t r)
{
    size_t lo = 0, hi = c.chunks.size() - 1;
    while (lo < hi) {
        const size_t mid = (lo + hi + 1) / 2;
        if (c.chunks[mid].row0 <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

inline bool row_valid(const Column &c, uint64_t r)
{
    if (!c.any_null) return true;
    const Chunk &k = c.chunks[chunk_of(c, r)];
    return !k.nulls || bit_at(k.nulls, k.a->offset + (int64_t)(r - k.row0));
}

// packed byte count of rows [r0, r1); a null view slot counts as empty.  maxlen (views only): the longest string of the range.
uint64_t range_bytes(const Column &c, uint64_t r0, uint64_t r1, uint32_t *maxlen = nullptr)
{
    uint64_t bytes = 0;
    uint32_t mx = 0;
    for (size_t ci = r0 < r1 ? chunk_of(c, r0) : c.chunks.size(); ci < c.chunks.size() && c.chunks[ci].row0 < r1; ++ci) {
        const Chunk &k = c.chunks[ci];
        const int64_t i0 = (int64_t)(std::max(r0, k.row0) - k.row0);
        const int64_t i1 = (int64_t)(std::min(r1, k.row0 + (uint64_t)k.a->length) - k.row0);
        if (c.layout == L_VIEW) {
            const View *v = static_cast<const View *>(k.a->buffers[1]) + k.a->offset;
            if (!k.nulls) for (int64_t i = i0; i < i1; ++i) { bytes += v[i].len; mx = std::max(mx, v[i].len); }
            else for (int64_t i = i0; i < i1; ++i) if (bit_at(k.nulls, k.a->offset + i)) { bytes += v[i].len; mx = std::max(mx, v[i].len); }
        } else if (c.layout == L_U32) {
            const int32_t *o = static_cast<const int32_t *>(k.a->buffers[1]) + k.a->offset;
            bytes += (uint64_t)(o[i1] - o[i0]);
        } else {
            const int64_t *o = static_cast<const int64_t *>(k.a->buffers[1]) + k.a->offset;
            bytes += (uint64_t)(o[i1] - o[i0]);
        }
    }
    if (maxlen) *maxlen = mx;
    return bytes;
}

// pack rows [r0, r1): off[i - r0 + 1] = end of row i (starting from `base`), bytes appended at val + base;
// `limit` = end of this range's bytes (another thread owns what follows)
// (views only) len8 != nullptr: one length byte per row goes to len8[0 ..] INSTEAD of the offsets (the device rebuilds them,
// strsim_offsets_from_lengths); the caller has checked that no string of the range exceeds 255 bytes
void pack_range(const Column &c, uint64_t r0, uint64_t r1, uint32_t *off, uint64_t base, uint64_t limit, uint8_t *val,
                uint8_t *len8 = nullptr)
{
    uint64_t pos = base;
    uint32_t *o_out = off + 1;
    for (size_t ci = r0 < r1 ? chunk_of(c, r0) : c.chunks.size(); ci < c.chunks.size() && c.chunks[ci].row0 < r1; ++ci) {
        const Chunk &k = c.chunks[ci];
        const ArrowArray *a = k.a;
        const int64_t i0 = (int64_t)(std::max(r0, k.row0) - k.row0);
        const int64_t i1 = (int64_t)(std::min(r1, k.row0 + (uint64_t)a->length) - k.row0);
        if (c.layout == L_VIEW) {
            const View *v = static_cast<const View *>(a->buffers[1]) + a->offset;
            // Arrow C data interface, Utf8View: the last buffer holds the int64 lengths of the variadic data buffers
            const int64_t *sizes = a->n_buffers >= 4 ? static_cast<const int64_t *>(a->buffers[a->n_buffers - 1]) : nullptr;
            const int64_t nvar = sizes ? a->n_buffers - 3 : a->n_buffers - 2; // variadic data buffers
            for (int64_t i = i0; i < i1; ++i) {
                if (!k.nulls || bit_at(k.nulls, a->offset + i)) {
                    const uint32_t len = v[i].len;
                    // Whether a string sits in its view or in a data buffer is a coin flip per row (cfg2: 37 % / 63 %), so the
                    // source pointer is SELECTED, not branched on, and the common case is one fixed 32-byte copy trimmed by
                    // the next row: 32 bytes must be writable, and readable behind the source -- inside the views buffer
                    // (two more views follow) or inside the data buffer (its length is in the trailing sizes buffer).
                    const bool inl = len <= 12;
                    uint32_t bi, bo;
                    memcpy(&bi, v[i].rest + 4, 4);
                    memcpy(&bo, v[i].rest + 8, 4);
                    if (!inl && (int64_t)bi >= nvar) fail("Utf8View buffer index out of range");
                    if (!inl && sizes && (int64_t)bo + (int64_t)len > sizes[bi]) fail("Utf8View string reaches past its data buffer");
                    const uint32_t bsel = inl ? 0u : bi;
                    const uint8_t *data = nvar > 0 ? static_cast<const uint8_t *>(a->buffers[2 + bsel]) : nullptr;
                    const uint8_t *src = inl ? v[i].rest : data + bo;
                    const bool room = inl ? i + 2 < a->length : (sizes != nullptr && (int64_t)bo + 32 <= sizes[bsel]);
                    if (len <= 32 && room && pos + 32 <= limit) memcpy(val + pos, src, 32);
                    else memcpy(val + pos, src, len);
                    pos += len;
                    if (len8) *len8 = (uint8_t)len;
                } else if (len8) {
                    *len8 = 0;
                }
                if (len8) ++len8; else *o_out++ = (uint32_t)pos;
            }
        } else {
            const uint8_t *data = static_cast<const uint8_t *>(a->buffers[2]);
            auto rows = [&](auto *o) {
                const uint64_t b0 = (uint64_t)o[i0], span = (uint64_t)(o[i1] - o[i0]);
                if (span) memcpy(val + pos, data + b0, span);
                for (int64_t i = i0; i < i1; ++i) *o_out++ = (uint32_t)(pos + ((uint64_t)o[i + 1] - b0));
                pos += span;
            };
            if (c.layout == L_U32) rows(static_cast<const int32_t *>(a->buffers[1]) + a->offset);
            else rows(static_cast<const int64_t *>(a->buffers[1]) + a->offset);
        }
    }
}

// The one-pass form for a column of views whose strings fit one length byte: rows [r0, r1) into the thread's OWN segment
// val[base .. limit), one length byte per row into len8 -- no size pass, hence no common prefix between the threads: the gaps
// between the segments are closed on the device (strsim_compact_segments).  Returns the bytes used, or ~0 when the
// segment overflows or a string exceeds 255 bytes (the caller then packs the slice the two-pass way).
// Bytes appended to a destination that nobody reads back on the host (the pinned staging buffer: next stop is the DMA engine):
// they are collected in a cache-resident buffer and leave with non-temporal stores, 16 bytes each, so the destination's lines
// are never fetched for ownership (a third of the packer's memory traffic) and do not push the views and data buffers out of
// the cache.  POLARS_STRSIM_STREAM_STORES=0 / 1 forces plain stores straight into the destination / this path (default: by the
// number of packing threads, see pack_range_onepass).
struct StreamOut {
    static constexpr uint32_t CAP = 4096;
    uint8_t *dst;                     // where buf[0] goes
    uint32_t fill = 0;
    alignas(64) uint8_t buf[CAP + 64]; // + room for the fixed 32-byte copy of the last string
    explicit StreamOut(uint8_t *d) : dst(d) {}
    void drain(bool all)
    {
        uint32_t at = 0;
        const uint32_t head = (uint32_t)((0u - (uintptr_t)dst) & 15u);
        if (head && fill >= head) { memcpy(dst, buf, head); at = head; }
        if (!head || at)
            for (; at + 16u <= fill; at += 16u)
                _mm_stream_si128(reinterpret_cast<__m128i *>(dst + at), _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + at)));
        if (all && at < fill) { memcpy(dst + at, buf + at, fill - at); at = fill; }
        dst += at;
        fill -= at;
        if (fill) memmove(buf, buf + at, fill);
        if (all) _mm_sfence();
    }
};

uint64_t pack_range_onepass(const Column &c, uint64_t r0, uint64_t r1, uint64_t base, uint64_t limit, uint8_t *val, uint8_t *len8,
                            bool many_threads)
{
    // 16 packing threads are bound by the memory system and gain a fifth (10 M rows: 7.9 -> 6.3 ms on the same box); ONE thread
    // (the engine-parallel mode) is not, and loses 9 % to the extra copy: plain stores there.
    static const int knob = [] { const char *e = getenv("POLARS_STRSIM_STREAM_STORES"); return e ? atoi(e) : -1; }();
    const bool stream = knob < 0 ? many_threads : knob != 0;
    uint64_t pos = base;
    StreamOut so(val + base);
    for (size_t ci = r0 < r1 ? chunk_of(c, r0) : c.chunks.size(); ci < c.chunks.size() && c.chunks[ci].row0 < r1; ++ci) {
        const Chunk &k = c.chunks[ci];
        const ArrowArray *a = k.a;
        const int64_t i0 = (int64_t)(std::max(r0, k.row0) - k.row0);
        const int64_t i1 = (int64_t)(std::min(r1, k.row0 + (uint64_t)a->length) - k.row0);
        const View *v = static_cast<const View *>(a->buffers[1]) + a->offset;
        const int64_t *sizes = a->n_buffers >= 4 ? static_cast<const int64_t *>(a->buffers[a->n_buffers - 1]) : nullptr;
        const int64_t nvar = sizes ? a->n_buffers - 3 : a->n_buffers - 2; // variadic data buffers
        for (int64_t i = i0; i < i1; ++i) {
            if (k.nulls && !bit_at(k.nulls, a->offset + i)) { *len8++ = 0; continue; }
            const uint32_t len = v[i].len;
            if (len > 255u || pos + len > limit) return ~0ull;
            const bool inl = len <= 12;
            uint32_t bi, bo;
            memcpy(&bi, v[i].rest + 4, 4);
            memcpy(&bo, v[i].rest + 8, 4);
            if (!inl && (int64_t)bi >= nvar) fail("Utf8View buffer index out of range");
            if (!inl && sizes && (int64_t)bo + (int64_t)len > sizes[bi]) fail("Utf8View string reaches past its data buffer");
            const uint32_t bsel = inl ? 0u : bi;
            const uint8_t *data = nvar > 0 ? static_cast<const uint8_t *>(a->buffers[2 + bsel]) : nullptr;
            const uint8_t *src = inl ? v[i].rest : data + bo;
            // (the fixed 32-byte copy of pack_range: see there)
            const bool room = inl ? i + 2 < a->length : (sizes != nullptr && (int64_t)bo + 32 <= sizes[bsel]);
            if (stream) {
                if (len <= 32 && room) {
                    memcpy(so.buf + so.fill, src, 32);
                    so.fill += len;
                    if (so.fill >= StreamOut::CAP) so.drain(false);
                } else {
                    for (uint32_t done = 0; done < len;) { // (up to 255 bytes: in pieces the buffer takes)
                        const uint32_t n = std::min(len - done, StreamOut::CAP + 64u - so.fill);
                        memcpy(so.buf + so.fill, src + done, n);
                        so.fill += n;
                        done += n;
                        if (so.fill >= StreamOut::CAP) so.drain(false);
                    }
                }
            } else {
                if (len <= 32 && room && pos + 32 <= limit) memcpy(val + pos, src, 32);
                else memcpy(val + pos, src, len);
            }
            pos += len;
            *len8++ = (uint8_t)len;
        }
    }
    if (stream) so.drain(true);
    return pos - base;
}

// VIEW-NATIVE form (opt-in, see run_rows): rows [r0, r1) of a column of views leave as the VIEWS THEMSELVES -- 16 bytes
// per row, a streaming copy; no gather for the strings that sit in their views (<= 12 bytes: three quarters of a column of names)
// -- and the device makes the column layout (strsim_column_from_views).  A string that does not fit its view is appended to the thread's own
// segment lng[base .. limit) and its view's last word becomes its offset there (the buffer index no longer matters); a null
// slot leaves as the empty string.  Returns the packed size (the sum of the lengths) and the segment's fill in `long_end`, or
// ~0 when the segment overflows (the caller sizes the slice's segments exactly and comes again).
uint64_t views_range(const Column &c, uint64_t r0, uint64_t r1, View *vout, uint8_t *lng, uint64_t base, uint64_t limit, uint64_t &long_end,
                     bool stream)
{
    uint64_t pos = base, total = 0;
    for (size_t ci = r0 < r1 ? chunk_of(c, r0) : c.chunks.size(); ci < c.chunks.size() && c.chunks[ci].row0 < r1; ++ci) {
        const Chunk &k = c.chunks[ci];
        const ArrowArray *a = k.a;
        const int64_t i0 = (int64_t)(std::max(r0, k.row0) - k.row0);
        const int64_t i1 = (int64_t)(std::min(r1, k.row0 + (uint64_t)a->length) - k.row0);
        const View *v = static_cast<const View *>(a->buffers[1]) + a->offset;
        const int64_t *sizes = a->n_buffers >= 4 ? static_cast<const int64_t *>(a->buffers[a->n_buffers - 1]) : nullptr;
        const int64_t nvar = sizes ? a->n_buffers - 3 : a->n_buffers - 2; // variadic data buffers
        for (int64_t i = i0; i < i1; ++i) {
            __m128i w = _mm_loadu_si128(reinterpret_cast<const __m128i *>(v + i));
            if (k.nulls && !bit_at(k.nulls, a->offset + i)) {
                w = _mm_setzero_si128();
            } else {
                const uint32_t len = v[i].len;
                total += len;
                if (len > 12u) {
                    uint32_t bi, bo;
                    memcpy(&bi, v[i].rest + 4, 4);
                    memcpy(&bo, v[i].rest + 8, 4);
                    if ((int64_t)bi >= nvar) fail("Utf8View buffer index out of range");
                    if (sizes && (int64_t)bo + (int64_t)len > sizes[bi]) fail("Utf8View string reaches past its data buffer"); // (only the index was checked)
                    if (pos + len > limit) return ~0ull;
                    memcpy(lng + pos, static_cast<const uint8_t *>(a->buffers[2 + bi]) + bo, len);
                    View patched = v[i];
                    const uint32_t at = (uint32_t)pos; // (a segment lies below SLICE_BYTES < 2^32)
                    memcpy(patched.rest + 8, &at, 4);
                    w = _mm_loadu_si128(reinterpret_cast<const __m128i *>(&patched));
                    pos += len;
                }
            }
            if (stream) _mm_stream_si128(reinterpret_cast<__m128i *>(vout), w); // (the pinned staging is 16-byte aligned)
            else _mm_storeu_si128(reinterpret_cast<__m128i *>(vout), w);
            ++vout;
        }
    }
    if (stream) _mm_sfence();
    long_end = pos;
    return total;
}

// bytes of the strings of rows [r0, r1) of a view column that do not fit their views
uint64_t long_bytes(const Column &c, uint64_t r0, uint64_t r1)
{
    uint64_t bytes = 0;
    for (size_t ci = r0 < r1 ? chunk_of(c, r0) : c.chunks.size(); ci < c.chunks.size() && c.chunks[ci].row0 < r1; ++ci) {
        const Chunk &k = c.chunks[ci];
        const int64_t i0 = (int64_t)(std::max(r0, k.row0) - k.row0);
        const int64_t i1 = (int64_t)(std::min(r1, k.row0 + (uint64_t)k.a->length) - k.row0);
        const View *v = static_cast<const View *>(k.a->buffers[1]) + k.a->offset;
        for (int64_t i = i0; i < i1; ++i)
            if (v[i].len > 12u && (!k.nulls || bit_at(k.nulls, k.a->offset + i))) bytes += v[i].len;
    }
    return bytes;
}

// ---- input ownership -------------------------------------------------------------------------------
struct InputGuard { // the callee owns the inputs: release every array, then every SeriesExport, exactly once
    SeriesExport *in;
    size_t n;
    ~InputGuard()
    {
        for (size_t i = 0; i < n; ++i) {
            SeriesExport &s = in[i];
            if (s.arrays)
                for (size_t c = 0; c < s.len; ++c)
                    if (s.arrays[c] && s.arrays[c]->release) s.arrays[c]->release(s.arrays[c]);
            if (s.release) s.release(&s);
        }
    }
};

// ---- output construction ---------------------------------------------------------------------------
// Large result columns are handed to the engine in PINNED host memory from a process-wide pool: the device-to-host copy of
// every slice lands in the column itself, so the host never copies the results (80 MB of memcpy + first-touch page faults per
// 10 M rows -- with the packing that was all the CPU time of a call, and the CPU quota, not PCIe, is what bounds a call).  The
// Arrow release callback returns the block to the pool.  Pinned memory is a limited resource and the engine may keep a
// column for as long as it likes, so: only columns of PINNED_OUT_MIN_BYTES .. PINNED_OUT_MAX_BYTES, at most
// PINNED_OUT_LENT_BYTES lent out at a time (beyond that: malloc + copy, as for small columns), at most
// PINNED_OUT_CACHE_BYTES kept idle.  POLARS_STRSIM_PINNED_OUT=0 switches it off.
constexpr size_t PINNED_OUT_MIN_BYTES = size_t(8) << 20, PINNED_OUT_MAX_BYTES = size_t(1) << 30;
constexpr size_t PINNED_OUT_LENT_BYTES = size_t(4) << 30, PINNED_OUT_CACHE_BYTES = size_t(1) << 30;
class PinnedPool {
  public:
    void *acquire(size_t bytes)
    {
        const char *e = getenv("POLARS_STRSIM_PINNED_OUT"); // (read per call: a large column, one getenv)
        if ((e && atoi(e) == 0) || bytes < PINNED_OUT_MIN_BYTES || bytes > PINNED_OUT_MAX_BYTES) return nullptr;
        std::lock_guard<std::mutex> lk(m_);
        if (lent_ + bytes > PINNED_OUT_LENT_BYTES) return nullptr;
        int best = -1;
        for (size_t i = 0; i < blocks_.size(); ++i)
            if (!blocks_[i].lent && blocks_[i].cap >= bytes && blocks_[i].cap <= 2 * bytes &&
                (best < 0 || blocks_[i].cap < blocks_[(size_t)best].cap))
                best = (int)i;
        if (best < 0) {
            void *p = nullptr;
            const size_t cap = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
            // portable: every device's copy engine may write into it (one call's rows shard over the GPUs)
            if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            blocks_.push_back(Block{p, cap, false});
            best = (int)blocks_.size() - 1;
        } else {
            idle_ -= blocks_[(size_t)best].cap;
        }
        blocks_[(size_t)best].lent = true;
        lent_ += blocks_[(size_t)best].cap;
        return blocks_[(size_t)best].p;
    }
    void release(void *p)
    {
        std::lock_guard<std::mutex> lk(m_);
        for (size_t i = 0; i < blocks_.size(); ++i) {
            if (blocks_[i].p != p) continue;
            lent_ -= blocks_[i].cap;
            if (idle_ + blocks_[i].cap > PINNED_OUT_CACHE_BYTES) {
                (void)hipHostFree(p);
                blocks_.erase(blocks_.begin() + (long)i);
            } else {
                blocks_[i].lent = false;
                idle_ += blocks_[i].cap;
            }
            return;
        }
    }

  private:
    struct Block { void *p; size_t cap; bool lent; };
    std::mutex m_;
    std::vector<Block> blocks_;
    size_t lent_ = 0, idle_ = 0;
};
// (never destroyed: a column may be released after static destructors have begun, and the runtime unmaps pinned memory at exit)
PinnedPool &pinned_pool() { static PinnedPool *p = new PinnedPool; return *p; }

struct ArrayPriv {
    void *data;
    void *validity;
    const void *bufs[2];
    bool data_pinned; // data came from pinned_pool()
};

void release_f64_array(ArrowArray *a)
{
    if (!a || !a->release) return;
    ArrayPriv *p = static_cast<ArrayPriv *>(a->private_data);
    if (p) {
        if (p->data_pinned) pinned_pool().release(p->data); else free(p->data);
        free(p->validity);
        delete p;
    }
    a->release = nullptr;
}

struct SchemaPriv {
    char *name;
};

void release_schema(ArrowSchema *s)
{
    if (!s || !s->release) return;
    SchemaPriv *p = static_cast<SchemaPriv *>(s->private_data);
    if (p) {
        free(p->name);
        delete p;
    }
    s->release = nullptr;
}

void fill_f64_schema(ArrowSchema *s, const char *name)
{
    memset(s, 0, sizeof *s);
    SchemaPriv *p = new SchemaPriv{strdup(name ? name : "")};
    s->format = "g"; // float64
    s->name = p->name;
    s->metadata = nullptr;
    s->flags = ARROW_FLAG_NULLABLE;
    s->release = release_schema;
    s->private_data = p;
}

struct SeriesPriv {
    ArrowSchema *schema;
    ArrowArray **arrays;
    size_t n;
};

void release_series(SeriesExport *e)
{
    if (!e || !e->release) return;
    SeriesPriv *p = static_cast<SeriesPriv *>(e->private_data);
    if (p) {
        // the importer took the arrays by bitwise copy (polars-ffi import_series): free the boxes only
        for (size_t i = 0; i < p->n; ++i) free(p->arrays[i]);
        free(p->arrays);
        if (p->schema) {
            if (p->schema->release) p->schema->release(p->schema);
            free(p->schema);
        }
        delete p;
    }
    e->release = nullptr;
    e->private_data = nullptr;
}

void *alloc64(size_t bytes)
{
    void *p = nullptr;
    const bool big = bytes >= (size_t(2) << 20);
    // large result columns: 2 MiB alignment + transparent huge pages, so first-touch faults do not dominate the
    // final copy (80 MB = 20 000 4-KiB faults otherwise)
    const size_t align = big ? (size_t(2) << 20) : 64;
    const size_t size = bytes ? ((bytes + align - 1) & ~(align - 1)) : 64;
    if (posix_memalign(&p, align, size) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    if (big) (void)madvise(p, size, MADV_HUGEPAGE);
#endif
    return p;
}

// ---- a small persistent fork-join pool for the host-side packing (one per calling thread) ------------------
// A call runs a dozen short jobs back to back (sizes and bytes of every slice), so an idle worker spins on the job counter for
// a moment before it sleeps on the condition variable, and the caller spins for the stragglers the same way: waking 15
// sleeping threads costs 30-50 us per job otherwise -- a millisecond per 10 M-row call.
class ForkJoinPool {
  public:
    ~ForkJoinPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        stop_a_.store(true, std::memory_order_release);
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // run fn(0 .. n-1), fn(0) on the caller; rethrows the first failure as a PluginError
    void run(unsigned n, const std::function<void(unsigned)> &fn)
    {
        if (n <= 1) { fn(0); return; }
        while (th_.size() + 1 < n) {
            const unsigned id = (unsigned)th_.size() + 1;
            th_.emplace_back([this, id] { worker(id); });
        }
        err_.clear();
        pending_.store(n - 1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            njob_ = n;
            ++gen_;
            gen_a_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        call(fn, 0);
        if (!spin_until([this] { return pending_.load(std::memory_order_acquire) == 0; })) {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = nullptr;
        }
        if (!err_.empty()) fail(err_);
    }

  private:
    static constexpr int SPIN_US = 150;
    template <class Pred> static bool spin_until(Pred pred)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if (pred()) return true;
                __builtin_ia32_pause();
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) return pred();
        }
    }
    void call(const std::function<void(unsigned)> &fn, unsigned t)
    {
        try {
            fn(t);
        } catch (const PluginError &e) {
            std::lock_guard<std::mutex> lk(m_);
            if (err_.empty()) err_ = e.msg;
        } catch (...) {
            std::lock_guard<std::mutex> lk(m_);
            if (err_.empty()) err_ = "unexpected failure in a packing thread";
        }
    }
    void worker(unsigned id)
    {
        uint64_t seen = 0;
        for (;;) {
            (void)spin_until([&] { return gen_a_.load(std::memory_order_acquire) != seen || stop_a_.load(std::memory_order_acquire); });
            const std::function<void(unsigned)> *job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (id >= njob_) continue;
                job = job_;
            }
            call(*job, id);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned njob_ = 0;
    std::atomic<unsigned> pending_{0};
    uint64_t gen_ = 0;
    std::atomic<uint64_t> gen_a_{0};
    bool stop_ = false;
    std::atomic<bool> stop_a_{false};
    std::string err_;
};
thread_local ForkJoinPool g_pool;

void fork_join(unsigned nthreads, const std::function<void(unsigned)> &fn) { g_pool.run(nthreads, fn); }

#define HIP_OR_FAIL(expr)                                                                           \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) fail(std::string("HIP error in " #expr ": ") + hipGetErrorString(e__)); \
    } while (0)

// grow-only buffer: pinned host memory or device memory
struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    bool device = false;
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        if (device) HIP_OR_FAIL(hipMalloc(&p, want)); else HIP_OR_FAIL(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    void release()
    {
        if (p) { if (device) (void)hipFree(p); else (void)hipHostFree(p); }
        p = nullptr; cap = 0;
    }
};

// one pipeline slot: a slice of both columns packed in pinned memory + its device mirror + its results
struct Slot {
    Buf h_off[2], h_val[2], h_out, d_off[2], d_val[2], d_out;
    Buf h_len[2], d_len[2]; // one length byte per row, shipped instead of the offsets when lens8[s] (see pack_slice2)
    bool lens8[2] = {false, false};
    // one-pass packing (pack_slice_onepass): nseg[s] > 1: the values of column s sit in nseg[s] segments of h_val[s] (seg_src,
    // seg_bytes), are shipped in one copy into d_land[s] and moved to their final places in d_val[s] (seg_dst) on the device
    int nseg[2] = {0, 0};
    uint64_t seg_src[2][32], seg_dst[2][32], seg_bytes[2][32];
    uint64_t span[2] = {0, 0}; // bytes of h_val[s] to ship (the last segment's end)
    Buf d_land[2];
    // view-native slices (pack_slice_views): the views as they lie + the strings that do not fit them
    bool as_views[2] = {false, false};
    Buf h_views[2], d_views[2], h_long[2], d_long[2];
    uint64_t long_span[2] = {0, 0}; // bytes of h_long[s] to ship
    uint64_t r0 = 0, rows = 0;
    uint64_t bytes[2] = {0, 0};
    bool direct = false; // this slice was computed in place on the pinned staging (see run(): launch)
    hipEvent_t ev_kernels = nullptr, ev_results = nullptr; // behind the slice's kernels (compute stream) / its D2H (copy stream)
    Slot() { for (int i = 0; i < 2; ++i) { d_off[i].device = true; d_val[i].device = true; d_len[i].device = true; d_land[i].device = true; d_views[i].device = true; d_long[i].device = true; } d_out.device = true; }
    void release()
    {
        for (int i = 0; i < 2; ++i) { h_off[i].release(); h_val[i].release(); d_off[i].release(); d_val[i].release(); h_len[i].release(); d_len[i].release(); d_land[i].release();
                                      h_views[i].release(); d_views[i].release(); h_long[i].release(); d_long[i].release(); }
        h_out.release(); d_out.release();
        if (ev_kernels) (void)hipEventDestroy(ev_kernels);
        if (ev_results) (void)hipEventDestroy(ev_results);
        ev_kernels = ev_results = nullptr;
    }
};

// ---- device pipelines per calling thread (Polars may call from several of its threads at once) -----
// One pipeline = one device context + its stream, a copy stream for the results, three slots and the literal's buffers.  A
// calling thread keeps one pipeline per entry of the device list (an ordinal may repeat: two pipelines on one GPU).
struct Pipe {
    strsim_ctx_t *ctx = nullptr;
    int device = 0;
    Slot slot[3];             // two slices in flight on the GPU + the one being packed
    hipStream_t d2h = nullptr; // results travel back on their own stream: D2H of slice k runs beside H2D of slice k+1
    Buf lit_off, lit_val;     // device copy of a literal side
    Buf lit_h_off, lit_h_val; // its pinned host staging
    void close()
    {
        if (!ctx) return;
        (void)hipSetDevice(device);
        if (d2h) (void)hipStreamDestroy(d2h);
        d2h = nullptr;
        for (auto &s : slot) s.release();
        lit_off.release(); lit_val.release();
        lit_h_off.release(); lit_h_val.release();
        strsim_ctx_destroy(ctx);
        ctx = nullptr;
    }
    ~Pipe() { close(); }
    // this pipeline on device `dev` (a pipeline that is asked for another device than last time starts over)
    strsim_ctx_t *open(int dev)
    {
        if (ctx && dev != device) close();
        if (!ctx) {
            device = dev;
            if (strsim_ctx_create(device, nullptr, &ctx) != STRSIM_OK) fail(strsim_last_error_message());
            // one-launch calls (ABI 1.4 opt-in): every slice is retired before its results are handed on, and a slice whose
            // slow rows were finished at retirement is fetched again (strsim_ctx_last_late_rows below)
            if (strsim_ctx_set_stream_ordered(ctx, 0) != STRSIM_OK) fail(strsim_last_error_message());
            lit_off.device = lit_val.device = true;
        }
        HIP_OR_FAIL(hipSetDevice(device));
        if (!d2h) HIP_OR_FAIL(hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking));
        return ctx;
    }
};
struct ThreadPipes {
    std::vector<Pipe *> p;
    Pipe &at(size_t i) { while (p.size() <= i) p.push_back(new Pipe); return *p[i]; }
    ~ThreadPipes() { for (Pipe *q : p) delete q; }
};
thread_local ThreadPipes g_pipes;

// The devices a call uses.  Default: ONE device -- the calling thread's current HIP device (0 unless the host process chose
// another; one process per GPU under a launcher keeps every process on its own).  POLARS_STRSIM_DEVICE = one ordinal.
// POLARS_STRSIM_DEVICES = comma-separated ordinals, or "all": a call of several million rows deals its slices out over these
// devices in turn (an ordinal may repeat: two pipelines on one GPU, which is how this is tested on a one-GPU box).  Opt-in,
// because every pipeline pins staging memory on its device's behalf for as long as the calling thread lives.
std::vector<int> plugin_devices()
{
    std::vector<int> v;
    if (const char *e = getenv("POLARS_STRSIM_DEVICES")) {
        if (strcmp(e, "all") == 0) {
            const int n = strsim_device_count();
            for (int d = 0; d < n; ++d) v.push_back(d);
        } else {
            for (const char *p = e; *p;) {
                char *end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                v.push_back((int)d);
                p = *end == ',' ? end + 1 : end;
            }
        }
    } else if (const char *e1 = getenv("POLARS_STRSIM_DEVICE")) {
        v.push_back(atoi(e1));
    } else {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = 0; }
        v.push_back(cur);
    }
    if (v.empty()) v.push_back(0); // (no device at all: strsim_ctx_create reports it -- there is no CPU path)
    return v;
}
// rows below which a call does not take another device: a device should at least get one full pipeline slice
uint64_t min_rows_per_device()
{
    const char *e = getenv("POLARS_STRSIM_MIN_ROWS_PER_DEVICE");
    return e ? std::max<uint64_t>(1, strtoull(e, nullptr, 10)) : (uint64_t)(2u << 20);
}

// Small calls: up to this many rows (and direct_bytes() packed bytes per column) the kernels read the pinned staging and write
// the pinned result buffer through the device's mapping of host memory.  The bytes cross PCIe from inside the kernels
// instead, but the H2D copies and the D2H copy each cost a hand-over between the copy engine and the compute queue
// (10-16 us apiece), which is most of a small call: 67 -> 50 us at 1..100 rows, 140 -> 75 us at 4 000, 520 -> 300 us at
// 30 000, 364 -> 339 us at 100 000; equal at 200 000..500 000 rows and 10 % slower at 1 M, hence the limits.
uint64_t direct_rows() // read per call: a test (or a user) can switch the path off with POLARS_STRSIM_DIRECT_ROWS=0
{
    const char *e = getenv("POLARS_STRSIM_DIRECT_ROWS");
    return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)131072;
}
uint64_t direct_bytes() { return (uint64_t)2 << 20; }

void *mapped(void *pinned)
{
    void *d = nullptr;
    HIP_OR_FAIL(hipHostGetDevicePointer(&d, pinned, 0));
    return d;
}

constexpr uint64_t SLICE_ROWS = 2u << 20;                      // rows packed / shipped / computed per pipeline step
constexpr uint64_t SLICE_BYTES = (1ull << 32) - (1ull << 24);  // packed values per slice and column (u32 offsets)

// CPUs this process may keep busy: the logical CPUs, or the cgroup v2 / v1 CPU quota when that is lower (a container with a
// 16-CPU quota on a 256-thread host: 32 packing threads there only buy throttling)
unsigned cpu_quota()
{
    unsigned n = std::max<unsigned>(std::thread::hardware_concurrency(), 1u);
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 0;
            fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}

// Packing threads of one call.  A call from a sequential engine context packs on every CPU the process may use (capped at 32).
// CallerContext PARALLEL (reference strsim.rs:53) means the engine is already parallel; the reference then computes on the calling
// thread alone, so as not to oversubscribe the CPUs.  Here such a call may BORROW helper threads from one process-wide budget of
// half the CPU quota: the permits it holds are taken at its entry and given back when it returns (PackGrant), so however the
// engine's calls arrive -- staggered or together -- the helpers running at any moment never exceed that half; a call that finds
// the budget lent out packs on its own thread, as the reference does.  (Round 4 sized the helpers from a one-shot read of a
// counter of calls in flight: sixteen calls arriving staggered got 16, 10, 8, 6, ... helpers each, about 70 in all.)
// A lone call in this mode (a group-by of one partition, a streaming batch) packs on up to half the CPUs -- 10 M rows in 13 ms
// instead of 48-80.  POLARS_STRSIM_PARALLEL_PACK=0 keeps the reference's rule to the letter; POLARS_STRSIM_PACK_THREADS=k caps a
// call's threads in either mode.
std::atomic<int> g_helpers_out{0}; // helper threads lent to engine-parallel calls right now

struct PackGrant {
    unsigned threads = 1; // packing threads of this call, the calling thread included
    int borrowed = 0;
    PackGrant(bool engine_parallel, uint64_t rows)
    {
        if (rows < 32768) return;
        static const unsigned granted = cpu_quota(); // logical CPUs, capped by the cgroup's CPU quota (containers)
        unsigned cap = std::min<unsigned>(granted, 32u);
        if (const char *e = getenv("POLARS_STRSIM_PACK_THREADS")) cap = (unsigned)std::max(1, atoi(e)); // explicit cap, any value
        const unsigned want = (unsigned)std::min<uint64_t>(cap, rows / 16384);
        if (!engine_parallel) { threads = std::max(1u, want); return; }
        static const bool strict = [] { const char *e = getenv("POLARS_STRSIM_PARALLEL_PACK"); return e && atoi(e) == 0; }();
        if (strict || want <= 1u) return;
        const int budget = (int)std::min<unsigned>(granted, 32u) / 2; // the engine's own threads keep the other half
        int out = g_helpers_out.load(std::memory_order_relaxed);
        for (;;) {
            const int take = std::min<int>((int)want - 1, budget - out);
            if (take <= 0) return;
            if (g_helpers_out.compare_exchange_weak(out, out + take, std::memory_order_acq_rel, std::memory_order_relaxed)) {
                borrowed = take;
                threads = 1u + (unsigned)take;
                return;
            }
        }
    }
    ~PackGrant() { if (borrowed) g_helpers_out.fetch_sub(borrowed, std::memory_order_acq_rel); }
    PackGrant(const PackGrant &) = delete;
    PackGrant &operator=(const PackGrant &) = delete;
};

// pack rows [r0, r1) of `c` into pinned staging (u32 offsets rebased to 0); returns the packed byte count
uint64_t pack_slice(const Column &c, uint64_t r0, uint64_t r1, Buf &off, Buf &val, unsigned T)
{
    const uint64_t rows = r1 - r0;
    T = (unsigned)std::min<uint64_t>(T, std::max<uint64_t>(rows / 16384, 1));
    std::vector<uint64_t> part(T + 1, 0);
    auto lo = [&](unsigned t) { return r0 + rows * t / T; };
    fork_join(T, [&](unsigned t) { part[t + 1] = range_bytes(c, lo(t), lo(t + 1)); });
    for (unsigned t = 0; t < T; ++t) part[t + 1] += part[t];
    const uint64_t total = part[T];
    if (total > SLICE_BYTES) return total;
    off.reserve((rows + 1) * sizeof(uint32_t));
    val.reserve(total + 64);
    uint32_t *o = static_cast<uint32_t *>(off.p);
    o[0] = 0;
    fork_join(T, [&](unsigned t) { pack_range(c, lo(t), lo(t + 1), o + (lo(t) - r0), part[t], part[t + 1], static_cast<uint8_t *>(val.p)); });
    return total;
}

// the same for BOTH columns of a slice in two jobs instead of four (sizes of both, then bytes of both): thread t takes rows
// lo(t) .. lo(t+1) of each column.  bytes[s] = packed byte count; false when a column's bytes exceed SLICE_BYTES.
bool pack_slice2(const Column (&col)[2], uint64_t r0, uint64_t r1, Slot &sl, bool allow_lens8, unsigned T)
{
    Buf (&off)[2] = sl.h_off, (&val)[2] = sl.h_val;
    uint64_t (&bytes)[2] = sl.bytes;
    const uint64_t rows = r1 - r0;
    T = (unsigned)std::min<uint64_t>(T, std::max<uint64_t>(rows / 16384, 1));
    std::vector<uint64_t> part[2] = {std::vector<uint64_t>(T + 1, 0), std::vector<uint64_t>(T + 1, 0)};
    std::vector<uint32_t> mx[2] = {std::vector<uint32_t>(T, 0), std::vector<uint32_t>(T, 0)};
    auto lo = [&](unsigned t) { return r0 + rows * t / T; };
    fork_join(T, [&](unsigned t) { for (int s = 0; s < 2; ++s) part[s][t + 1] = range_bytes(col[s], lo(t), lo(t + 1), &mx[s][t]); });
    for (int s = 0; s < 2; ++s) {
        sl.nseg[s] = 0;
        for (unsigned t = 0; t < T; ++t) part[s][t + 1] += part[s][t];
        bytes[s] = part[s][T];
        // a column of views whose strings all fit a byte ships LENGTHS (1 B per row instead of a 4-byte offset over PCIe)
        sl.lens8[s] = allow_lens8 && col[s].layout == L_VIEW && *std::max_element(mx[s].begin(), mx[s].end()) <= 255u;
    }
    if (bytes[0] > SLICE_BYTES || bytes[1] > SLICE_BYTES) return false;
    uint32_t *o[2] = {nullptr, nullptr};
    uint8_t *l8[2] = {nullptr, nullptr};
    for (int s = 0; s < 2; ++s) {
        val[s].reserve(bytes[s] + 64);
        if (sl.lens8[s]) {
            sl.h_len[s].reserve(rows + 16);
            l8[s] = static_cast<uint8_t *>(sl.h_len[s].p);
        } else {
            off[s].reserve((rows + 1) * sizeof(uint32_t));
            o[s] = static_cast<uint32_t *>(off[s].p);
            o[s][0] = 0;
        }
    }
    fork_join(T, [&](unsigned t) {
        for (int s = 0; s < 2; ++s)
            pack_range(col[s], lo(t), lo(t + 1), o[s] ? o[s] + (lo(t) - r0) : nullptr, part[s][t], part[s][t + 1],
                       static_cast<uint8_t *>(val[s].p), l8[s] ? l8[s] + (lo(t) - r0) : nullptr);
    });
    return true;
}

// ONE pass over both view columns of a slice (no size pass): thread t packs rows lo(t) .. lo(t+1) of each column into its own
// segment of the staging area, sized from `bpr256[s]` -- the bytes per row (x 256) the call's previous slice had, plus slack --
// and writes one length byte per row.  True on success (sl.bytes, sl.nseg / seg_*, sl.span and bpr256 updated); false when a
// segment overflowed or a string exceeds 255 bytes: the caller packs the slice the two-pass way (which also re-learns bpr256).
bool pack_slice_onepass(const Column (&col)[2], uint64_t r0, uint64_t r1, Slot &sl, uint64_t (&bpr256)[2], unsigned T)
{
    const uint64_t rows = r1 - r0;
    T = (unsigned)std::min<uint64_t>(std::min<unsigned>(T, 32u), std::max<uint64_t>(rows / 16384, 1));
    auto lo = [&](unsigned t) { return r0 + rows * t / T; };
    uint64_t base[2][33];
    for (int s = 0; s < 2; ++s) {
        base[s][0] = 0;
        for (unsigned t = 0; t < T; ++t) {
            const uint64_t n = lo(t + 1) - lo(t);
            const uint64_t cap = ((n * bpr256[s]) >> 8) + (n >> 4) + 4096; // the estimate + 1/16 + 4 KB of slack
            base[s][t + 1] = base[s][t] + ((cap + 63) & ~(uint64_t)63);
        }
        if (base[s][T] > SLICE_BYTES) return false;
        sl.h_val[s].reserve(base[s][T] + 64);
        sl.h_len[s].reserve(rows + 16);
    }
    uint64_t used[2][32];
    fork_join(T, [&](unsigned t) {
        for (int s = 0; s < 2; ++s)
            used[s][t] = pack_range_onepass(col[s], lo(t), lo(t + 1), base[s][t], base[s][t + 1], static_cast<uint8_t *>(sl.h_val[s].p),
                                            static_cast<uint8_t *>(sl.h_len[s].p) + (lo(t) - r0), T >= 4u);
    });
    for (int s = 0; s < 2; ++s)
        for (unsigned t = 0; t < T; ++t)
            if (used[s][t] == ~0ull) return false;
    for (int s = 0; s < 2; ++s) {
        uint64_t total = 0;
        for (unsigned t = 0; t < T; ++t) {
            sl.seg_src[s][t] = base[s][t];
            sl.seg_dst[s][t] = total;
            sl.seg_bytes[s][t] = used[s][t];
            total += used[s][t];
        }
        sl.nseg[s] = (int)T;
        sl.span[s] = base[s][T - 1] + used[s][T - 1];
        sl.bytes[s] = total;
        sl.lens8[s] = true;
        bpr256[s] = rows ? (total * 256 + rows - 1) / rows : 0;
    }
    return true;
}

// Both view columns of a slice in the view-native form (views_range): thread t takes rows lo(t) .. lo(t+1) of each column, its long
// strings go into its own segment of h_long[s] -- sized from lbpr256[s], the long bytes per row (x 256) the call's previous slice
// had, plus slack; exactly (a pass over the lengths first) when that is not known yet or a segment overflows.  The segments need
// no closing up: the views carry the offsets.  True on success; false when a column's packed values exceed SLICE_BYTES.
bool pack_slice_views(const Column (&col)[2], uint64_t r0, uint64_t r1, Slot &sl, uint64_t (&lbpr256)[2], unsigned T)
{
    const uint64_t rows = r1 - r0;
    T = (unsigned)std::min<uint64_t>(std::min<unsigned>(T, 32u), std::max<uint64_t>(rows / 16384, 1));
    auto lo = [&](unsigned t) { return r0 + rows * t / T; };
    bool exact = lbpr256[0] == ~0ull || lbpr256[1] == ~0ull;
    for (;;) {
        uint64_t base[2][33];
        if (exact) {
            uint64_t need[2][32];
            fork_join(T, [&](unsigned t) { for (int s = 0; s < 2; ++s) need[s][t] = long_bytes(col[s], lo(t), lo(t + 1)); });
            for (int s = 0; s < 2; ++s) {
                base[s][0] = 0;
                for (unsigned t = 0; t < T; ++t) base[s][t + 1] = base[s][t] + ((need[s][t] + 63) & ~(uint64_t)63);
            }
        } else {
            for (int s = 0; s < 2; ++s) {
                base[s][0] = 0;
                for (unsigned t = 0; t < T; ++t) {
                    const uint64_t n = lo(t + 1) - lo(t);
                    const uint64_t cap = ((n * lbpr256[s]) >> 8) + (n >> 3) + 4096; // the estimate + 1/8 + 4 KB of slack
                    base[s][t + 1] = base[s][t] + ((cap + 63) & ~(uint64_t)63);
                }
            }
        }
        for (int s = 0; s < 2; ++s) {
            if (base[s][T] > SLICE_BYTES) return false;
            sl.h_views[s].reserve(rows * sizeof(View) + 64);
            sl.h_long[s].reserve(base[s][T] + 64);
        }
        uint64_t total[2][32], end[2][32];
        fork_join(T, [&](unsigned t) {
            for (int s = 0; s < 2; ++s)
                total[s][t] = views_range(col[s], lo(t), lo(t + 1), static_cast<View *>(sl.h_views[s].p) + (lo(t) - r0),
                                          static_cast<uint8_t *>(sl.h_long[s].p), base[s][t], base[s][t + 1], end[s][t], T >= 4u);
        });
        bool overflow = false;
        for (int s = 0; s < 2; ++s)
            for (unsigned t = 0; t < T; ++t) overflow = overflow || total[s][t] == ~0ull;
        if (overflow) {
            if (exact) fail("internal: a view-native segment overflowed its exact size");
            exact = true; // (once: sized from the lengths themselves)
            continue;
        }
        for (int s = 0; s < 2; ++s) {
            uint64_t bytes = 0, lng = 0;
            for (unsigned t = 0; t < T; ++t) { bytes += total[s][t]; lng += end[s][t] - base[s][t]; }
            if (bytes > SLICE_BYTES) return false;
            sl.bytes[s] = bytes;
            sl.long_span[s] = end[s][T - 1];
            sl.as_views[s] = true;
            sl.lens8[s] = false;
            sl.nseg[s] = 0;
            lbpr256[s] = rows ? (lng * 256 + rows - 1) / rows : 0;
        }
        return true;
    }
}

struct PhaseTimer { // POLARS_STRSIM_TRACE=1: per-phase wall times of one plugin call on stderr
    bool on;
    double t_pack = 0, t_copy = 0;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer() : on(getenv("POLARS_STRSIM_TRACE") != nullptr) {}
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void stop(double &acc) { if (on) acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// a row-count tuning knob from the environment (unset, empty or 0: the default)
uint64_t env_rows(const char *name, uint64_t dflt)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const unsigned long long v = strtoull(e, nullptr, 10);
    return v ? (uint64_t)v : dflt;
}

struct PipeTimes { double t_launch = 0, t_wait = 0, t_d2h = 0; unsigned slices = 0; };

// Rows [0, n) of a call through the pipelines of `devs`: slices are packed one after the other by the calling thread's packing
// pool -- ALL of its threads on every slice -- and dealt out to the pipelines in turn: slice i goes to pipeline i % D, into
// slot (i / D) % 3 of it, is launched there (H2D + kernels on the pipeline's stream, the D2H of its results on the
// pipeline's copy stream, over that device's own PCIe link) and is finished two rounds later, when slice i + 2 D has been
// launched: two slices in flight per device while the host packs the next.  Results go straight into out[] -- by the copy
// engine itself when the output column is pinned memory (out_pinned), else through the slot's pinned result buffer and a
// host copy.  (Reference: the row fan-out of strsim.rs:72-100; here the host has ONE packer, so the devices take turns
// instead of shards -- a device's share of the packing threads would be a fraction of them.)
void run_rows(int measure, const Column (&col)[2], const bool (&lit)[2], uint64_t n, double *out, bool out_pinned, unsigned T,
              bool direct_call, bool engine_parallel, const std::vector<int> &devs, PhaseTimer &tm, std::vector<PipeTimes> &ptimes)
{
    const size_t D = devs.size();
    ptimes.assign(D, PipeTimes{});
    std::vector<Pipe *> pipes(D);
    std::vector<hipStream_t> streams(D);
    for (size_t d = 0; d < D; ++d) {
        pipes[d] = &g_pipes.at(d);
        streams[d] = static_cast<hipStream_t>(strsim_ctx_stream(pipes[d]->open(devs[d])));
    }

    // a literal side is packed once and shipped to every pipeline
    std::vector<const uint32_t *> lit_off_d(D, nullptr);
    std::vector<const uint8_t *> lit_val_d(D, nullptr);
    for (int s = 0; s < 2; ++s) {
        if (!lit[s]) continue;
        for (size_t d = 0; d < D; ++d) {
            Pipe &P = *pipes[d];
            HIP_OR_FAIL(hipSetDevice(P.device));
            Buf &ho = P.lit_h_off, &hv = P.lit_h_val; // persistent pinned staging: the call is synchronous,
            const uint64_t bytes = pack_slice(col[s], 0, 1, ho, hv, 1); // so no earlier copy can still be reading them
            if (bytes > SLICE_BYTES) fail("a single string exceeds the 4 GiB limit");
            if (direct_call && bytes <= direct_bytes()) { // small call: read in place (see direct_rows)
                lit_off_d[d] = static_cast<const uint32_t *>(mapped(ho.p));
                lit_val_d[d] = static_cast<const uint8_t *>(mapped(hv.p));
                continue;
            }
            P.lit_off.reserve(2 * sizeof(uint32_t));
            P.lit_val.reserve(bytes + 64);
            HIP_OR_FAIL(hipMemcpyAsync(P.lit_off.p, ho.p, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, streams[d]));
            if (bytes) HIP_OR_FAIL(hipMemcpyAsync(P.lit_val.p, hv.p, bytes, hipMemcpyHostToDevice, streams[d]));
            lit_off_d[d] = static_cast<const uint32_t *>(P.lit_off.p);
            lit_val_d[d] = static_cast<const uint8_t *>(P.lit_val.p);
        }
    }

    // (slices computed in place read their offsets from the pinned staging; POLARS_STRSIM_LENGTH_BYTES=0: always ship offsets)
    const char *lens8_env = getenv("POLARS_STRSIM_LENGTH_BYTES");
    const bool lens8_ok = !(lens8_env && atoi(lens8_env) == 0) && !direct_call;
    // bytes per row (x 256) of the call's last slice, per column: 0 = not known yet (the first slice is packed the two-pass way)
    uint64_t bpr256[2] = {0, 0};
    const char *onepass_env = getenv("POLARS_STRSIM_ONE_PASS");
    const bool onepass_ok = lens8_ok && !lit[0] && !lit[1] && col[0].layout == L_VIEW && col[1].layout == L_VIEW &&
                            !(onepass_env && atoi(onepass_env) == 0);
    // View-native slices: two view columns, not a small call; POLARS_STRSIM_VIEWS=1 switches them on.  OFF by
    // default, by their own measurement (profiles/r4_f1_views.txt, 10 M rows, same box): a view is 16 bytes whatever the string --
    // more than the packed form of a short string (cfg1: 9 bytes a row and column) -- so the streaming copy writes MORE than the
    // gather it replaces and the link carries more: 9.3 vs 5.8 ms (cfg1) and 13.6 vs 8.6 ms (cfg2) with the packing pool, 61 vs 48
    // and 101 vs 67 ms on the calling thread alone (the engine-parallel mode), with and without non-temporal stores.
    (void)engine_parallel;
    const char *views_env = getenv("POLARS_STRSIM_VIEWS");
    const bool views_ok = !direct_call && !lit[0] && !lit[1] && col[0].layout == L_VIEW && col[1].layout == L_VIEW &&
                          views_env && atoi(views_env) != 0;
    uint64_t lbpr256[2] = {~0ull, ~0ull}; // long bytes per row (x 256) of the call's last slice: not known yet
    auto pack = [&](Slot &sl, uint64_t r0, uint64_t want) -> uint64_t {
        uint64_t rows = std::min<uint64_t>(want, n - r0);
        for (;;) {
            bool fits = true;
            sl.lens8[0] = sl.lens8[1] = false;
            sl.nseg[0] = sl.nseg[1] = 0;
            sl.as_views[0] = sl.as_views[1] = false;
            if (views_ok) {
                fits = pack_slice_views(col, r0, r0 + rows, sl, lbpr256, T);
            } else if (onepass_ok && bpr256[0] && bpr256[1] && pack_slice_onepass(col, r0, r0 + rows, sl, bpr256, T)) {
                // (one pass: lengths + values in per-thread segments)
            } else if (!lit[0] && !lit[1]) {
                fits = pack_slice2(col, r0, r0 + rows, sl, lens8_ok, T);
                for (int s = 0; s < 2 && fits; ++s) bpr256[s] = sl.lens8[s] && rows ? (sl.bytes[s] * 256 + rows - 1) / rows + 1 : 0;
            } else {
                for (int s = 0; s < 2 && fits; ++s) {
                    if (lit[s]) continue;
                    sl.bytes[s] = pack_slice(col[s], r0, r0 + rows, sl.h_off[s], sl.h_val[s], T);
                    fits = sl.bytes[s] <= SLICE_BYTES;
                }
            }
            if (fits) break;
            if (rows == 1) fail("a single string exceeds the 4 GiB limit");
            rows = (rows + 1) / 2; // very long strings: halve the slice until its packed values fit 32-bit offsets
        }
        sl.r0 = r0; sl.rows = rows;
        return rows;
    };
    auto launch = [&](size_t d, Slot &sl) {
        Pipe &P = *pipes[d];
        strsim_ctx_t *ctx = P.ctx;
        hipStream_t stream = streams[d];
        HIP_OR_FAIL(hipSetDevice(P.device));
        const uint32_t *doff[2];
        const uint8_t *dval[2];
        uint64_t drows[2];
        sl.direct = direct_call;
        for (int s = 0; s < 2; ++s)
            if (!lit[s] && sl.bytes[s] > direct_bytes()) sl.direct = false;
        for (int s = 0; s < 2; ++s) {
            if (lit[s]) { doff[s] = lit_off_d[d]; dval[s] = lit_val_d[d]; drows[s] = 1; continue; }
            drows[s] = sl.rows;
            if (sl.direct) {
                doff[s] = static_cast<const uint32_t *>(mapped(sl.h_off[s].p));
                dval[s] = static_cast<const uint8_t *>(mapped(sl.h_val[s].p));
                continue;
            }
            sl.d_off[s].reserve((sl.rows + 1) * sizeof(uint32_t));
            sl.d_val[s].reserve(sl.bytes[s] + 64);
            if (sl.as_views[s]) { // the views as they lie + the long strings over the link, the column made on the device
                sl.d_views[s].reserve(sl.rows * sizeof(View) + 64);
                sl.d_long[s].reserve(sl.long_span[s] + 64);
                HIP_OR_FAIL(hipMemcpyAsync(sl.d_views[s].p, sl.h_views[s].p, sl.rows * sizeof(View), hipMemcpyHostToDevice, stream));
                if (sl.long_span[s]) HIP_OR_FAIL(hipMemcpyAsync(sl.d_long[s].p, sl.h_long[s].p, sl.long_span[s], hipMemcpyHostToDevice, stream));
                if (strsim_column_from_views_bounded(ctx, sl.d_views[s].p, sl.rows, static_cast<const uint8_t *>(sl.d_long[s].p), sl.long_span[s],
                                                     static_cast<uint32_t *>(sl.d_off[s].p), static_cast<uint8_t *>(sl.d_val[s].p),
                                                     sl.bytes[s] + 64, nullptr) != STRSIM_OK)
                    fail(strsim_last_error_message());
                doff[s] = static_cast<const uint32_t *>(sl.d_off[s].p);
                dval[s] = static_cast<const uint8_t *>(sl.d_val[s].p);
                continue;
            }
            if (sl.lens8[s]) { // lengths over the link, offsets rebuilt on the device
                sl.d_len[s].reserve(sl.rows + 16);
                HIP_OR_FAIL(hipMemcpyAsync(sl.d_len[s].p, sl.h_len[s].p, sl.rows, hipMemcpyHostToDevice, stream));
                if (strsim_offsets_from_lengths(ctx, static_cast<const uint8_t *>(sl.d_len[s].p), sl.rows, static_cast<uint32_t *>(sl.d_off[s].p)) != STRSIM_OK)
                    fail(strsim_last_error_message());
            } else {
                HIP_OR_FAIL(hipMemcpyAsync(sl.d_off[s].p, sl.h_off[s].p, (sl.rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            }
            if (sl.nseg[s] > 1) { // one-pass slice: one copy of the segments as they lie, then the gaps are closed on the device
                sl.d_land[s].reserve(sl.span[s] + 64);
                HIP_OR_FAIL(hipMemcpyAsync(sl.d_land[s].p, sl.h_val[s].p, sl.span[s], hipMemcpyHostToDevice, stream));
                if (strsim_compact_segments(ctx, static_cast<const uint8_t *>(sl.d_land[s].p), static_cast<uint8_t *>(sl.d_val[s].p),
                                                     sl.seg_src[s], sl.seg_dst[s], sl.seg_bytes[s], sl.nseg[s]) != STRSIM_OK)
                    fail(strsim_last_error_message());
            } else if (sl.bytes[s]) {
                HIP_OR_FAIL(hipMemcpyAsync(sl.d_val[s].p, sl.h_val[s].p, sl.bytes[s], hipMemcpyHostToDevice, stream));
            }
            doff[s] = static_cast<const uint32_t *>(sl.d_off[s].p);
            dval[s] = static_cast<const uint8_t *>(sl.d_val[s].p);
        }
        const bool via_slot = sl.direct || !out_pinned; // results pass through the slot's pinned buffer
        if (via_slot) sl.h_out.reserve(sl.rows * sizeof(double));
        if (!sl.direct) sl.d_out.reserve(sl.rows * sizeof(double));
        double *res = static_cast<double *>(sl.direct ? mapped(sl.h_out.p) : sl.d_out.p);
        // (a slice computed in place is a small call: one launch when the lane kernel leaves nothing behind)
        if ((sl.direct ? strsim_pairs_device_small : strsim_pairs_device)(ctx, measure, doff[0], dval[0], drows[0], doff[1], dval[1], drows[1], res, sl.rows) != STRSIM_OK)
            fail(strsim_last_error_message());
        if (!sl.direct) {
            // results come back right behind the kernels, on the copy stream: the next slice's H2D does not queue behind them
            if (!sl.ev_kernels) HIP_OR_FAIL(hipEventCreateWithFlags(&sl.ev_kernels, hipEventDisableTiming));
            if (!sl.ev_results) HIP_OR_FAIL(hipEventCreateWithFlags(&sl.ev_results, hipEventDisableTiming));
            HIP_OR_FAIL(hipEventRecord(sl.ev_kernels, stream));
            HIP_OR_FAIL(hipStreamWaitEvent(P.d2h, sl.ev_kernels, 0));
            HIP_OR_FAIL(hipMemcpyAsync(out_pinned ? static_cast<void *>(out + sl.r0) : sl.h_out.p, sl.d_out.p, sl.rows * sizeof(double),
                                       hipMemcpyDeviceToHost, P.d2h));
            HIP_OR_FAIL(hipEventRecord(sl.ev_results, P.d2h));
        }
    };
    // the slice's results have arrived (in the output column, or in the slot's pinned buffer): retire the call; copy if needed
    auto finish = [&](size_t d, Slot &sl) {
        Pipe &P = *pipes[d];
        strsim_ctx_t *ctx = P.ctx;
        PhaseTimer t1 = tm; // (same switch, own clock)
        HIP_OR_FAIL(hipSetDevice(P.device));
        t1.start();
        if (sl.direct) {
            if (strsim_ctx_synchronize(ctx) != STRSIM_OK) fail(strsim_last_error_message());
        } else {
            HIP_OR_FAIL(hipEventSynchronize(sl.ev_results));
            // (the oldest call in flight on this pipeline is this slice's: slices are launched and finished in order)
            if (strsim_ctx_retire_oldest(ctx) != STRSIM_OK) fail(strsim_last_error_message()); // also the deferred slow-row and long-string passes
        }
        t1.stop(ptimes[d].t_wait);
        const bool via_slot = sl.direct || !out_pinned;
        if (strsim_ctx_last_late_rows(ctx) != 0 && !sl.direct) { // rows finished by a pass launched just now: fetch the column again
            t1.start();
            HIP_OR_FAIL(hipMemcpyAsync(via_slot ? sl.h_out.p : static_cast<void *>(out + sl.r0), sl.d_out.p, sl.rows * sizeof(double),
                                       hipMemcpyDeviceToHost, streams[d]));
            HIP_OR_FAIL(hipStreamSynchronize(streams[d]));
            t1.stop(ptimes[d].t_d2h);
        }
        if (!via_slot) return;
        tm.start();
        const double *src = static_cast<const double *>(sl.h_out.p);
        const unsigned Tc = (unsigned)std::min<uint64_t>(T, std::max<uint64_t>(sl.rows / 262144, 1));
        fork_join(Tc, [&](unsigned t) {
            const uint64_t i0 = sl.rows * t / Tc, i1 = sl.rows * (t + 1) / Tc;
            memcpy(out + sl.r0 + i0, src + i0, (i1 - i0) * sizeof(double));
        });
        tm.stop(tm.t_copy);
    };

    // Slices ramp up from RAMP_ROWS and down again at the end, so that neither the first pack nor the last slice's trip is
    // exposed at full size.  PCIe carries 35-41 B in and 8 B out per pair in the two directions at once, the host touches every
    // byte once (pack) -- per 2 M-row slice 1.45 ms of link against 1.3 ms of packing.
    // (knobs read per call: four getenv; tests shrink them to cut a small frame into many slices)
    // (in round 1 cutting a 1 M-row call into four slices lost -- four small packs cost 1.3 ms instead of 0.6 ms; with both columns
    // in one job and a pool that spins between jobs it wins, so only calls up to SINGLE_ROWS stay in one piece)
    const uint64_t RAMP_ROWS = env_rows("POLARS_STRSIM_RAMP_ROWS", 512u << 10);   // first slice (tuning knobs)
    const uint64_t FULL_ROWS = env_rows("POLARS_STRSIM_SLICE_ROWS", SLICE_ROWS);   // steady-state slice
    const uint64_t GROW_PCT = env_rows("POLARS_STRSIM_RAMP_GROW_PCT", 150);         // slice k+1 = slice k x this / 100
    const uint64_t SINGLE_ROWS = env_rows("POLARS_STRSIM_SINGLE_SLICE_ROWS", 300000); // calls up to here are not cut (1 M rows in four slices: 2.39 -> 2.06 ms)
    uint64_t prev_rows = 0;
    auto next_rows = [&](uint64_t r0) -> uint64_t {
        const uint64_t left = n - r0;
        if (direct_call || n <= SINGLE_ROWS) return left;     // small calls: one slice
        const uint64_t ramp = n <= (2u << 20) ? RAMP_ROWS / 2 : RAMP_ROWS; // a mid-size call starts (and stays) smaller
        uint64_t want = prev_rows == 0 ? ramp : std::min<uint64_t>(FULL_ROWS, prev_rows * GROW_PCT / 100);
        want = std::min<uint64_t>(want, SLICE_ROWS);
        if (left < 2 * want) want = std::max<uint64_t>(ramp, ((left / 2 + 65535) >> 16) << 16); // taper
        if (left <= want + ramp / 2) want = left;                   // no crumbs
        prev_rows = want;
        return want;
    };
    // whatever goes wrong below (a failed launch, a string beyond 4 GiB in a later slice): nothing of this call may still be in
    // flight when the error leaves the plugin -- the slots are reused by the next call and the output column is released
    struct Drain {
        std::vector<Pipe *> &pipes; bool armed = true;
        ~Drain()
        {
            if (!armed) return;
            for (Pipe *P : pipes) { (void)hipSetDevice(P->device); (void)strsim_ctx_synchronize(P->ctx); (void)hipStreamSynchronize(P->d2h); }
        }
    } drain{pipes};
    uint64_t r0 = 0, launched = 0, finished = 0; // slices: i -> pipeline i % D, slot (i / D) % 3
    auto slot_of = [&](uint64_t i) -> Slot & { return pipes[i % D]->slot[(i / D) % 3]; };
    while (r0 < n) {
        Slot &sl = slot_of(launched);
        tm.start(); r0 += pack(sl, r0, next_rows(r0)); tm.stop(tm.t_pack); // overlaps the GPU work of the slices in flight
        PhaseTimer t1 = tm;
        t1.start(); launch(launched % D, sl); t1.stop(ptimes[launched % D].t_launch);
        ++ptimes[launched % D].slices;
        ++launched;
        if (launched - finished > 2 * D) { finish(finished % D, slot_of(finished)); ++finished; } // two in flight per pipeline
    }
    for (; finished < launched; ++finished) finish(finished % D, slot_of(finished));
    drain.armed = false;
}

// ---- output validity: AND of the input validities, built word by word on the packing pool ---------------------------
// bits [bit0, bit0 + n) of `src` (LSB-first, Arrow) as 64-bit words of a stream that starts at bit 0: word k = bits
// [64 k, 64 k + 64) of the range; bits past the range read as ones
inline uint64_t bits_word(const uint8_t *src, int64_t bit0, uint64_t n, uint64_t k)
{
    const uint64_t first = 64 * k;
    if (first >= n) return ~0ull;
    const uint64_t take = std::min<uint64_t>(64, n - first);
    const int64_t b = bit0 + (int64_t)first;
    const uint8_t *p = src + (b >> 3);
    const unsigned sh = (unsigned)(b & 7);
    uint64_t w = 0;
    const unsigned nbytes = (unsigned)((sh + take + 7) >> 3); // <= 9
    for (unsigned q = 0; q < nbytes && q < 8; ++q) w |= (uint64_t)p[q] << (8 * q);
    w >>= sh;
    if (nbytes == 9) w |= (uint64_t)p[8] << (64 - sh);
    if (take < 64) w |= ~0ull << take;
    return w;
}

// AND the validity of rows [r0, r1) of `c` into dst, whose bit 0 is row r0 (r0 a multiple of 64): whole words only
void and_validity(const Column &c, uint64_t r0, uint64_t r1, uint64_t *dst)
{
    if (!c.any_null || r0 >= r1) return;
    for (size_t ci = chunk_of(c, r0); ci < c.chunks.size() && c.chunks[ci].row0 < r1; ++ci) {
        const Chunk &k = c.chunks[ci];
        if (!k.nulls) continue;
        const uint64_t lo = std::max(r0, k.row0), hi = std::min(r1, k.row0 + (uint64_t)k.a->length); // rows of this chunk in range
        // destination words that hold rows [lo, hi): the chunk's bits arrive shifted by (lo - r0) & 63
        const uint64_t dbit = lo - r0;
        const int64_t sbit = k.a->offset + (int64_t)(lo - k.row0);
        const uint64_t cnt = hi - lo;
        // head: up to the next destination word boundary, then whole source words, shifted in
        uint64_t done = 0;
        while (done < cnt) {
            const uint64_t db = dbit + done;
            const unsigned dsh = (unsigned)(db & 63);
            const uint64_t take = std::min<uint64_t>(64 - dsh, cnt - done);
            uint64_t w = bits_word(k.nulls, sbit + (int64_t)done, take, 0); // ones beyond `take`
            // place at dsh; ones elsewhere
            const uint64_t placed = (w << dsh) | (dsh ? (~0ull >> (64 - dsh)) : 0ull);
            dst[db >> 6] &= placed;
            done += take;
        }
    }
}

// validity words of rows [0, n) = AND of the non-literal inputs' validities, 64 rows at a time on `T` packing threads (their row
// ranges are cut at multiples of 64, so no two threads share a word); bits past row n are zero.  The value under a null slot is
// set to 0.0 in `out` (never observable; keeps the column deterministic).  Returns the null count.
int64_t build_validity(const Column (&col)[2], const bool (&lit)[2], uint64_t n, bool all_null, unsigned T, uint64_t *vw, double *out)
{
    const uint64_t nwords = (n + 63) / 64;
    if (all_null) {
        memset(vw, 0, nwords * 8);
        return (int64_t)n;
    }
    const unsigned Tv = (unsigned)std::min<uint64_t>(std::max(1u, T), std::max<uint64_t>(nwords / 4096, 1));
    std::vector<int64_t> nulls(Tv, 0);
    fork_join(Tv, [&](unsigned t) {
        const uint64_t w0 = nwords * t / Tv, w1 = nwords * (t + 1) / Tv;
        const uint64_t r0 = w0 * 64, r1 = std::min<uint64_t>(w1 * 64, n);
        for (uint64_t w = w0; w < w1; ++w) vw[w] = ~0ull;
        for (int s = 0; s < 2; ++s)
            if (!lit[s]) and_validity(col[s], r0, r1, vw + w0);
        int64_t cnt = 0;
        for (uint64_t w = w0; w < w1; ++w) {
            uint64_t word = vw[w];
            if (w == nwords - 1 && (n & 63)) word &= ~0ull >> (64 - (n & 63)); // (bits past the column: zero)
            vw[w] = word;
            uint64_t zeros = ~word;
            if (w == nwords - 1 && (n & 63)) zeros &= ~0ull >> (64 - (n & 63));
            cnt += __builtin_popcountll(zeros);
            if (out)
                while (zeros) {
                    out[w * 64 + (uint64_t)__builtin_ctzll(zeros)] = 0.0;
                    zeros &= zeros - 1;
                }
        }
        nulls[t] = cnt;
    });
    int64_t total = 0;
    for (int64_t c : nulls) total += c;
    return total;
}

void run(int measure, SeriesExport *inputs, size_t n_inputs, SeriesExport *ret, bool engine_parallel)
{
    PhaseTimer tm;
    const auto t_begin = std::chrono::steady_clock::now();
    if (n_inputs != 2) fail("expected 2 input series, got " + std::to_string(n_inputs));
    Column col[2];
    describe(inputs[0], col[0]);
    describe(inputs[1], col[1]);
    const Column &a = col[0], &b = col[1];
    // strsim.rs:48-52
    if (a.rows != b.rows && a.rows != 1 && b.rows != 1)
        fail("Inputs must have the same length, or one of them must be a Utf8 literal.");
    const bool lit[2] = {a.rows == 1 && b.rows != 1, b.rows == 1};
    const uint64_t n = lit[0] ? b.rows : a.rows;

    double *out = static_cast<double *>(pinned_pool().acquire(n * sizeof(double))); // large columns (see PinnedPool)
    const bool out_pinned = out != nullptr;
    if (!out) out = static_cast<double *>(alloc64(n * sizeof(double)));
    uint8_t *validity = nullptr;
    int64_t null_count = 0;
    struct Cleanup {
        double *&o; bool pinned; uint8_t *&v; bool armed = true;
        ~Cleanup() { if (armed) { if (pinned) pinned_pool().release(o); else free(o); free(v); } }
    } cleanup{out, out_pinned, validity};

    // a NULL literal: the reference unwrap()s and panics (strsim.rs:62,65,87,90); here every row is null
    const bool all_null = (lit[0] && !row_valid(a, 0)) || (lit[1] && !row_valid(b, 0));

    const PackGrant grant(engine_parallel, n); // (helper threads borrowed for an engine-parallel call go back when it returns)
    const unsigned T = grant.threads;
    std::vector<PipeTimes> ptimes;
    std::vector<int> devs = plugin_devices();
    if (n != 0 && !all_null) {
        const bool direct_call = n <= direct_rows();
        const uint64_t D = std::max<uint64_t>(1, std::min<uint64_t>(devs.size(), n / min_rows_per_device()));
        devs.resize((size_t)D);
        run_rows(measure, col, lit, n, out, out_pinned, T, direct_call, engine_parallel, devs, tm, ptimes);
    }

    // output validity = AND of the input validities (broadcast for a literal; a null literal is the all_null case)
    const bool need_validity = all_null || a.any_null || b.any_null;
    if (need_validity && n != 0) {
        validity = static_cast<uint8_t *>(alloc64((n + 63) / 64 * 8));
        null_count = build_validity(col, lit, n, all_null, T, reinterpret_cast<uint64_t *>(validity), out);
    }

    // one "g" chunk
    ArrayPriv *ap = new ArrayPriv{out, validity, {validity, out}, out_pinned};
    ArrowArray *arr = static_cast<ArrowArray *>(calloc(1, sizeof(ArrowArray)));
    arr->length = (int64_t)n;
    arr->null_count = null_count;
    arr->offset = 0;
    arr->n_buffers = 2;
    arr->n_children = 0;
    arr->buffers = ap->bufs;
    arr->release = release_f64_array;
    arr->private_data = ap;
    cleanup.armed = false;

    ArrowSchema *schema = static_cast<ArrowSchema *>(calloc(1, sizeof(ArrowSchema)));
    fill_f64_schema(schema, a.name.c_str());
    SeriesPriv *sp = new SeriesPriv{schema, static_cast<ArrowArray **>(calloc(1, sizeof(ArrowArray *))), 1};
    sp->arrays[0] = arr;
    ret->field = schema;
    ret->arrays = sp->arrays;
    ret->len = 1;
    ret->release = release_series;
    ret->private_data = sp;
    if (tm.on) {
        double launch = 0, wait = 0, d2h = 0;
        for (const PipeTimes &p : ptimes) { launch += p.t_launch; wait += p.t_wait; d2h += p.t_d2h; }
        fprintf(stderr, "[polars_strsim] rows=%llu total=%.2f ms: pack=%.2f launch=%.2f wait=%.2f d2h=%.2f copy=%.2f (threads %u, pipelines %zu)\n",
                (unsigned long long)n,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), tm.t_pack,
                launch, wait, d2h, tm.t_copy, T, ptimes.size());
        for (size_t d = 0; d < ptimes.size(); ++d)
            fprintf(stderr, "[polars_strsim]   pipeline %zu (device %d): %u slices, launch=%.2f wait=%.2f d2h=%.2f ms\n", d, devs[d],
                    ptimes[d].slices, ptimes[d].t_launch, ptimes[d].t_wait, ptimes[d].t_d2h);
    }
}

void plugin_entry(int measure, SeriesExport *inputs, size_t n_inputs, SeriesExport *ret, const CallerContext *cc)
{
    InputGuard guard{inputs, n_inputs};
    try {
        run(measure, inputs, n_inputs, ret, cc && (cc->bitflags & 1u));
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (const std::bad_alloc &) {
        g_plugin_error = "out of host memory";
    } catch (const std::exception &e) {
        g_plugin_error = std::string("unexpected failure: ") + e.what();
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
}

void field_entry(ArrowSchema *input_fields, size_t n_fields, ArrowSchema *ret)
{
    // output_type=Float64 (mod.rs:8,13,18,23,28): a Float64 field carrying the first input's name
    const char *name = (n_fields > 0 && input_fields && input_fields[0].name) ? input_fields[0].name : "";
    fill_f64_schema(ret, name);
}

} // namespace

extern "C" {

uint32_t _polars_plugin_get_version(void) { return (POLARS_PLUGIN_VERSION_MAJOR << 16) | POLARS_PLUGIN_VERSION_MINOR; }

const char *_polars_plugin_get_last_error_message(void) { return g_plugin_error.c_str(); }

#define POLARS_PLUGIN_DEFINE(name, id)                                                                          \
    void _polars_plugin_##name(SeriesExport *inputs, size_t n_inputs, const uint8_t *, size_t,                  \
                               SeriesExport *return_value, CallerContext *cc)                                   \
    {                                                                                                           \
        plugin_entry(id, inputs, n_inputs, return_value, cc);                                                   \
    }                                                                                                           \
    void _polars_plugin_field_##name(ArrowSchema *input_fields, size_t n_fields, ArrowSchema *return_value)     \
    {                                                                                                           \
        field_entry(input_fields, n_fields, return_value);                                                      \
    }

#ifdef STRSIM_TEST_HOOKS
// Test hooks (no GPU needed), compiled only into tests/cpu_harness/libplugin_testhooks.so (make testhooks) -- the product library
// does not export them.
// (1) the host-side packing of rows [r0, r1) of one Series with `threads` helper threads into caller-provided buffers -- the exact
// describe / range_bytes / pack_range code the plugin entry points use.  Owns and releases the input like a plugin call.
// Returns 0, or -1 with the message in _polars_plugin_get_last_error_message().
POLARS_PLUGIN_API int _strsim_test_pack_series(SeriesExport *series, uint64_t r0, uint64_t r1, uint32_t *off_out,
                                               uint8_t *val_out, uint64_t val_cap, uint64_t *rows_out, uint64_t *bytes_out,
                                               uint8_t *valid_out, unsigned threads)
{
    InputGuard guard{series, 1};
    try {
        Column c;
        describe(*series, c);
        if (rows_out) *rows_out = c.rows;
        if (r1 > c.rows) r1 = c.rows;
        if (r0 > r1) r0 = r1;
        const uint64_t rows = r1 - r0;
        unsigned T = threads ? threads : 1;
        if (T > rows) T = rows ? (unsigned)rows : 1;
        std::vector<uint64_t> part(T + 1, 0);
        auto lo = [&](unsigned t) { return r0 + rows * t / T; };
        fork_join(T, [&](unsigned t) { part[t + 1] = range_bytes(c, lo(t), lo(t + 1)); });
        for (unsigned t = 0; t < T; ++t) part[t + 1] += part[t];
        if (bytes_out) *bytes_out = part[T];
        if (part[T] + 16 > val_cap) fail("test buffer too small");
        off_out[0] = 0;
        fork_join(T, [&](unsigned t) { pack_range(c, lo(t), lo(t + 1), off_out + (lo(t) - r0), part[t], part[t + 1], val_out); });
        if (valid_out)
            for (uint64_t r = r0; r < r1; ++r) valid_out[r - r0] = row_valid(c, r) ? 1 : 0;
        return 0;
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
    return -1;
}

// (1b) the ONE-PASS packer on one view Series used for both columns of a slice: rows [r0, r1) on `threads` threads with a budget of
// bpr256 / 256 bytes per row.  Returns 1 and fills val_out (the segments CLOSED UP the way the device does: seg_dst order),
// len_out (one byte per row) and *bytes_out when the pass succeeds, 0 when it reports "overflow / long string" (the plugin then
// packs two-pass), -1 on error.
POLARS_PLUGIN_API int _strsim_test_pack_onepass(SeriesExport *series, uint64_t r0, uint64_t r1, uint64_t bpr256, unsigned threads,
                                                uint8_t *val_out, uint64_t val_cap, uint8_t *len_out, uint64_t *bytes_out, int *nseg_out)
{
    InputGuard guard{series, 1};
    try {
        Column col[2];
        describe(*series, col[0]);
        describe(*series, col[1]);
        if (col[0].layout != L_VIEW) fail("one-pass packing is for view columns");
        if (r1 > col[0].rows) r1 = col[0].rows;
        if (r0 > r1) r0 = r1;
        Slot sl; // (plain memory stands in for the pinned staging: generous enough that reserve() never allocates -- no GPU here)
        const uint64_t cap = (((r1 - r0) * bpr256) >> 8) + (r1 - r0) + 64 * 4096 + 65536;
        struct Free {
            Slot &s;
            ~Free() { for (int i = 0; i < 2; ++i) { free(s.h_val[i].p); free(s.h_len[i].p); s.h_val[i].p = s.h_len[i].p = nullptr; s.h_val[i].cap = s.h_len[i].cap = 0; } }
        } fr{sl};
        for (int i = 0; i < 2; ++i) {
            sl.h_val[i].p = malloc(cap); sl.h_val[i].cap = cap;
            sl.h_len[i].p = malloc(r1 - r0 + 64); sl.h_len[i].cap = r1 - r0 + 64;
        }
        uint64_t b[2] = {bpr256, bpr256};
        if (!pack_slice_onepass(col, r0, r1, sl, b, threads ? threads : 1)) return 0;
        if (sl.bytes[0] > val_cap) fail("test buffer too small");
        for (int k = 0; k < sl.nseg[0]; ++k)
            memcpy(val_out + sl.seg_dst[0][k], static_cast<const uint8_t *>(sl.h_val[0].p) + sl.seg_src[0][k], sl.seg_bytes[0][k]);
        memcpy(len_out, sl.h_len[0].p, r1 - r0);
        if (bytes_out) *bytes_out = sl.bytes[0];
        if (nseg_out) *nseg_out = sl.nseg[0];
        // both "columns" are the same series: their segments must agree
        if (sl.bytes[1] != sl.bytes[0] || sl.nseg[1] != sl.nseg[0]) fail("the two columns of the slice disagree");
        return 1;
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
    return -1;
}

// (1c) the VIEW-NATIVE form of one view Series used for both columns of a slice (pack_slice_views): rows [r0, r1) on `threads`
// threads; lbpr256: the long bytes per row (x 256) to size the segments by, ~0: size them exactly.  Fills views_out (16 bytes per
// row) and long_out (the segments as they lie, *span_out bytes) and *bytes_out (the packed size).  Returns 0 / -1.
POLARS_PLUGIN_API int _strsim_test_pack_views(SeriesExport *series, uint64_t r0, uint64_t r1, uint64_t lbpr256, unsigned threads,
                                              uint8_t *views_out, uint8_t *long_out, uint64_t long_cap, uint64_t *span_out, uint64_t *bytes_out)
{
    InputGuard guard{series, 1};
    try {
        Column col[2];
        describe(*series, col[0]);
        describe(*series, col[1]);
        if (col[0].layout != L_VIEW) fail("view-native packing is for view columns");
        if (r1 > col[0].rows) r1 = col[0].rows;
        if (r0 > r1) r0 = r1;
        Slot sl; // (plain memory stands in for the pinned staging: generous enough that reserve() never allocates -- no GPU here)
        struct Free {
            Slot &s;
            ~Free() { for (int i = 0; i < 2; ++i) { free(s.h_views[i].p); free(s.h_long[i].p); s.h_views[i].p = s.h_long[i].p = nullptr; s.h_views[i].cap = s.h_long[i].cap = 0; } }
        } fr{sl};
        const uint64_t lcap = range_bytes(col[0], r0, r1) + (r1 - r0) + 64 * 4096 + 65536 + (((r1 - r0) * (lbpr256 == ~0ull ? 0 : lbpr256)) >> 7);
        for (int i = 0; i < 2; ++i) {
            sl.h_views[i].p = malloc((r1 - r0) * 16 + 128); sl.h_views[i].cap = (r1 - r0) * 16 + 128;
            sl.h_long[i].p = malloc(lcap); sl.h_long[i].cap = lcap;
        }
        uint64_t l[2] = {lbpr256, lbpr256};
        if (!pack_slice_views(col, r0, r1, sl, l, threads ? threads : 1)) fail("the slice exceeds the per-slice byte limit");
        if (sl.long_span[0] > long_cap) fail("test buffer too small");
        memcpy(views_out, sl.h_views[0].p, (r1 - r0) * 16);
        memcpy(long_out, sl.h_long[0].p, sl.long_span[0]);
        if (span_out) *span_out = sl.long_span[0];
        if (bytes_out) *bytes_out = sl.bytes[0];
        if (sl.bytes[1] != sl.bytes[0] || memcmp(sl.h_views[0].p, sl.h_views[1].p, (r1 - r0) * 16) != 0) fail("the two columns of the slice disagree");
        return 0;
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
    return -1;
}

// (2) the output validity of a call over two Series (a one-row side is the literal), as the plugin builds it: words[(n + 63) / 64]
// and the null count; vals (optional, n doubles): the slots under nulls are zeroed.  Returns 0 / -1 like (1).
POLARS_PLUGIN_API int _strsim_test_validity(SeriesExport *two_series, uint64_t *words, int64_t *null_count, double *vals,
                                            uint64_t *rows_out, unsigned threads)
{
    InputGuard guard{two_series, 2};
    try {
        Column col[2];
        describe(two_series[0], col[0]);
        describe(two_series[1], col[1]);
        if (col[0].rows != col[1].rows && col[0].rows != 1 && col[1].rows != 1) fail("shape");
        const bool lit[2] = {col[0].rows == 1 && col[1].rows != 1, col[1].rows == 1};
        const uint64_t n = lit[0] ? col[1].rows : col[0].rows;
        if (rows_out) *rows_out = n;
        const bool all_null = (lit[0] && !row_valid(col[0], 0)) || (lit[1] && !row_valid(col[1], 0));
        *null_count = n ? build_validity(col, lit, n, all_null, threads ? threads : 1, words, vals) : 0;
        return 0;
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
    return -1;
}

// (3) the packing threads of `n_calls` engine-parallel calls of `rows` rows each that are ALL in flight together, entered one after
// the other (the staggered arrival of an engine's threads): threads_out[i] = what call i packs on.  *helpers_out = helper threads
// lent out while all of them run; the return value = the same count after all have returned (0).
POLARS_PLUGIN_API int _strsim_test_pack_grants(int engine_parallel, int n_calls, uint64_t rows, unsigned *threads_out, int *helpers_out)
{
    {
        std::vector<std::unique_ptr<PackGrant>> g;
        for (int i = 0; i < n_calls; ++i) {
            g.emplace_back(new PackGrant(engine_parallel != 0, rows));
            threads_out[i] = g.back()->threads;
        }
        *helpers_out = g_helpers_out.load();
    }
    return g_helpers_out.load();
}
#endif // STRSIM_TEST_HOOKS

POLARS_PLUGIN_DEFINE(levenshtein, STRSIM_LEVENSHTEIN)
POLARS_PLUGIN_DEFINE(jaro, STRSIM_JARO)
POLARS_PLUGIN_DEFINE(jaro_winkler, STRSIM_JARO_WINKLER)
POLARS_PLUGIN_DEFINE(jaccard, STRSIM_JACCARD)
POLARS_PLUGIN_DEFINE(sorensen_dice, STRSIM_SORENSEN_DICE)

} // extern "C"
