// plugin_arrow.h -- the Arrow side of the plugin layer: an input Series described without a copy (chunks, layouts, validity), the
// row-range packers that read its buffers (offsets + values, the one-pass form, the view-native form), ownership of the inputs, and
// the f64 result column handed back (pooled pinned memory, release callbacks).
// Included by polars_plugin.cpp inside its anonymous namespace, in this order ([r5] split out of polars_plugin.cpp along its seams,
// no behaviour change -- the object code is identical before and after).
// Reference: parallel_apply, /root/reference/src/expressions/strsim.rs:41-107.
#pragma once

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
