// polars_plugin.cpp -- the Polars expression-plugin ABI (include/polars_plugin_abi.h) on top of the
// gfx950 kernels: the C++ counterpart of the reference's macro glue (reference src/expressions/mod.rs:8-31)
// and of `strsim::parallel_apply` (reference src/expressions/strsim.rs:41-107).
//
// What happens per call (one Polars expression evaluation):
//   1. both input Series (any chunking; Arrow "vu" string views as Polars >= 0.20 sends them, or "u"/"U"
//      offset layouts) are flattened into the device layout: offsets + packed UTF-8 values (+ validity);
//   2. shape rule and literal broadcast exactly as strsim.rs:48-52,61-66;
//   3. rows go to the GPU in batches whose packed values fit 32-bit offsets (strsim_pairs_host);
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
#include <vector>

#include "polars_plugin_abi.h"
#include "strsim_amd.h"
#include "strsim_internal.h"

namespace {

thread_local std::string g_plugin_error;

struct PluginError {
    std::string msg;
};

[[noreturn]] void fail(const std::string &m) { throw PluginError{m}; }

// ---- one input Series, flattened -----------------------------------------------------------------
struct FlatColumn {
    std::vector<uint64_t> offsets; // rows + 1
    std::vector<uint8_t> values;
    std::vector<uint8_t> valid;    // one byte per row; empty = no nulls
    uint64_t rows = 0;
    std::string name;
};

inline bool bit_at(const uint8_t *bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

struct View { // Arrow BinaryView / Utf8View element
    uint32_t len;
    uint8_t rest[12]; // <= 12 bytes inline, else {prefix[4], buffer_index u32, offset u32}
};

void flatten(const SeriesExport &s, FlatColumn &out)
{
    if (!s.field || !s.field->format) fail("input series has no schema");
    const std::string fmt = s.field->format;
    const bool is_view = fmt == "vu", is_u = fmt == "u", is_U = fmt == "U";
    if (!is_view && !is_u && !is_U)
        fail("invalid series dtype: expected `String`, got Arrow format `" + fmt + "`"); // `.str()?`, strsim.rs:46-47
    out.name = s.field->name ? s.field->name : "";
    uint64_t rows = 0, bytes = 0;
    bool any_null = false;
    // pass 1: sizes
    for (size_t c = 0; c < s.len; ++c) {
        const ArrowArray *a = s.arrays[c];
        if (!a) fail("null chunk pointer");
        if (a->length < 0 || a->offset < 0) fail("negative length/offset in chunk");
        const uint8_t *vb = a->n_buffers > 0 ? static_cast<const uint8_t *>(a->buffers[0]) : nullptr;
        if (a->null_count != 0 && vb) any_null = true;
        rows += (uint64_t)a->length;
        if (a->length == 0) continue;
        if (is_view) {
            if (a->n_buffers < 2) fail("Utf8View chunk without a views buffer");
            const View *v = static_cast<const View *>(a->buffers[1]) + a->offset;
            for (int64_t i = 0; i < a->length; ++i)
                if (!vb || bit_at(vb, a->offset + i)) bytes += v[i].len;
        } else if (is_u) {
            if (a->n_buffers < 3) fail("Utf8 chunk without offsets/values buffers");
            const int32_t *o = static_cast<const int32_t *>(a->buffers[1]) + a->offset;
            bytes += (uint64_t)(o[a->length] - o[0]);
        } else {
            if (a->n_buffers < 3) fail("LargeUtf8 chunk without offsets/values buffers");
            const int64_t *o = static_cast<const int64_t *>(a->buffers[1]) + a->offset;
            bytes += (uint64_t)(o[a->length] - o[0]);
        }
    }
    out.rows = rows;
    out.offsets.resize(rows + 1);
    out.values.resize(bytes + 64); // slack so device staging never reads past the vector
    if (any_null) out.valid.assign(rows, 1);
    // pass 2: copy
    uint64_t r = 0, pos = 0;
    out.offsets[0] = 0;
    for (size_t c = 0; c < s.len; ++c) {
        const ArrowArray *a = s.arrays[c];
        if (a->length == 0) continue;
        const uint8_t *vb = (a->null_count != 0 && a->n_buffers > 0) ? static_cast<const uint8_t *>(a->buffers[0]) : nullptr;
        if (is_view) {
            const View *v = static_cast<const View *>(a->buffers[1]) + a->offset;
            for (int64_t i = 0; i < a->length; ++i, ++r) {
                const bool ok = !vb || bit_at(vb, a->offset + i);
                if (ok) {
                    const uint32_t len = v[i].len;
                    const uint8_t *src;
                    if (len <= 12) {
                        src = v[i].rest;
                    } else {
                        uint32_t bi, bo;
                        memcpy(&bi, v[i].rest + 4, 4);
                        memcpy(&bo, v[i].rest + 8, 4);
                        if ((int64_t)bi + 2 >= a->n_buffers) fail("Utf8View buffer index out of range");
                        src = static_cast<const uint8_t *>(a->buffers[2 + bi]) + bo;
                    }
                    memcpy(out.values.data() + pos, src, len);
                    pos += len;
                } else {
                    out.valid[r] = 0; // a null slot contributes an empty string; its result is masked anyway
                }
                out.offsets[r + 1] = pos;
            }
        } else {
            const uint8_t *data = static_cast<const uint8_t *>(a->buffers[2]);
            auto copy_rows = [&](auto *o) {
                const uint64_t base = (uint64_t)o[0], span = (uint64_t)(o[a->length] - o[0]);
                if (span) memcpy(out.values.data() + pos, data + base, span);
                for (int64_t i = 0; i < a->length; ++i, ++r) {
                    out.offsets[r + 1] = pos + ((uint64_t)o[i + 1] - base);
                    if (vb && !bit_at(vb, a->offset + i)) out.valid[r] = 0;
                }
                pos += span;
            };
            if (is_u) copy_rows(static_cast<const int32_t *>(a->buffers[1]) + a->offset);
            else copy_rows(static_cast<const int64_t *>(a->buffers[1]) + a->offset);
        }
    }
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
struct ArrayPriv {
    void *data;
    void *validity;
    const void *bufs[2];
};

void release_f64_array(ArrowArray *a)
{
    if (!a || !a->release) return;
    ArrayPriv *p = static_cast<ArrayPriv *>(a->private_data);
    if (p) {
        free(p->data);
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
    if (posix_memalign(&p, 64, bytes ? ((bytes + 63) & ~size_t(63)) : 64) != 0) throw std::bad_alloc();
    return p;
}

// ---- device context per calling thread (Polars may call from several of its threads at once) -------
struct ThreadCtx {
    strsim_ctx_t *ctx = nullptr;
    ~ThreadCtx() { if (ctx) strsim_ctx_destroy(ctx); }
    strsim_ctx_t *get()
    {
        if (!ctx) {
            int dev = 0;
            if (const char *e = getenv("POLARS_STRSIM_DEVICE")) dev = atoi(e);
            if (strsim_ctx_create(dev, nullptr, &ctx) != STRSIM_OK) fail(strsim_last_error_message());
        }
        return ctx;
    }
};
thread_local ThreadCtx g_ctx;

constexpr uint64_t BATCH_BYTES = (1ull << 32) - (1ull << 20); // packed values per device batch (u32 offsets)
constexpr uint64_t BATCH_ROWS = 1ull << 30;

// rows [r0, r1) of a flattened column as a u32-offset shard
struct Shard {
    std::vector<uint32_t> off;
    const uint8_t *val;
    uint64_t rows;
};

void make_shard(const FlatColumn &c, uint64_t r0, uint64_t r1, Shard &s)
{
    const uint64_t base = c.offsets[r0];
    s.rows = r1 - r0;
    s.off.resize(s.rows + 1);
    for (uint64_t i = 0; i <= s.rows; ++i) s.off[i] = (uint32_t)(c.offsets[r0 + i] - base);
    s.val = c.values.data() + base;
}

void run(int measure, SeriesExport *inputs, size_t n_inputs, SeriesExport *ret)
{
    if (n_inputs != 2) fail("expected 2 input series, got " + std::to_string(n_inputs));
    FlatColumn a, b;
    flatten(inputs[0], a);
    flatten(inputs[1], b);
    // strsim.rs:48-52
    if (a.rows != b.rows && a.rows != 1 && b.rows != 1)
        fail("Inputs must have the same length, or one of them must be a Utf8 literal.");
    const bool lit_a = a.rows == 1 && b.rows != 1, lit_b = b.rows == 1;
    const uint64_t n = lit_a ? b.rows : a.rows;

    double *out = static_cast<double *>(alloc64(n * sizeof(double)));
    uint8_t *validity = nullptr;
    int64_t null_count = 0;
    struct Cleanup {
        double *&o; uint8_t *&v; bool armed = true;
        ~Cleanup() { if (armed) { free(o); free(v); } }
    } cleanup{out, validity};

    // a NULL literal: the reference unwrap()s and panics (strsim.rs:62,65,87,90); here every row is null
    const bool a_null_lit = lit_a && !a.valid.empty() && !a.valid[0];
    const bool b_null_lit = lit_b && !b.valid.empty() && !b.valid[0];
    const bool all_null = a_null_lit || b_null_lit;

    if (n != 0 && !all_null) {
        strsim_ctx_t *ctx = g_ctx.get();
        Shard sa, sb;
        if (lit_a) make_shard(a, 0, 1, sa);
        if (lit_b) make_shard(b, 0, 1, sb);
        uint64_t r0 = 0;
        while (r0 < n) {
            // largest batch whose packed values fit 32-bit offsets on both sides
            uint64_t r1 = n;
            if (r1 - r0 > BATCH_ROWS) r1 = r0 + BATCH_ROWS;
            auto fit = [&](const FlatColumn &c, bool lit) {
                if (lit) return;
                if (c.offsets[r1] - c.offsets[r0] <= BATCH_BYTES) return;
                uint64_t lo = r0 + 1, hi = r1; // first r1 that overflows is > lo
                while (lo < hi) {
                    const uint64_t mid = lo + (hi - lo + 1) / 2;
                    if (c.offsets[mid] - c.offsets[r0] <= BATCH_BYTES) lo = mid; else hi = mid - 1;
                }
                r1 = lo;
            };
            fit(a, lit_a);
            fit(b, lit_b);
            if (r1 == r0) fail("a single string exceeds the 4 GiB batch limit");
            if (!lit_a) make_shard(a, r0, r1, sa);
            if (!lit_b) make_shard(b, r0, r1, sb);
            const int rc = strsim_pairs_host(ctx, measure, sa.off.data(), sa.val, sa.rows, sb.off.data(), sb.val, sb.rows,
                                             out + r0, r1 - r0);
            if (rc != STRSIM_OK) fail(strsim_last_error_message());
            r0 = r1;
        }
    }

    // output validity = AND of the input validities (broadcast for a literal)
    const bool need_validity = all_null || !a.valid.empty() || !b.valid.empty();
    if (need_validity && n != 0) {
        validity = static_cast<uint8_t *>(alloc64((n + 7) / 8));
        memset(validity, 0, (n + 7) / 8);
        for (uint64_t i = 0; i < n; ++i) {
            bool ok = !all_null;
            if (ok && !a.valid.empty()) ok = a.valid[lit_a ? 0 : i] != 0;
            if (ok && !b.valid.empty()) ok = b.valid[lit_b ? 0 : i] != 0;
            if (ok) validity[i >> 3] |= (uint8_t)(1u << (i & 7));
            else { ++null_count; out[i] = 0.0; }
        }
    }

    // one "g" chunk
    ArrayPriv *ap = new ArrayPriv{out, validity, {validity, out}};
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
}

void plugin_entry(int measure, SeriesExport *inputs, size_t n_inputs, SeriesExport *ret)
{
    InputGuard guard{inputs, n_inputs};
    try {
        run(measure, inputs, n_inputs, ret);
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
                               SeriesExport *return_value, CallerContext *)                                     \
    {                                                                                                           \
        plugin_entry(id, inputs, n_inputs, return_value);                                                       \
    }                                                                                                           \
    void _polars_plugin_field_##name(ArrowSchema *input_fields, size_t n_fields, ArrowSchema *return_value)     \
    {                                                                                                           \
        field_entry(input_fields, n_fields, return_value);                                                      \
    }

POLARS_PLUGIN_DEFINE(levenshtein, STRSIM_LEVENSHTEIN)
POLARS_PLUGIN_DEFINE(jaro, STRSIM_JARO)
POLARS_PLUGIN_DEFINE(jaro_winkler, STRSIM_JARO_WINKLER)
POLARS_PLUGIN_DEFINE(jaccard, STRSIM_JACCARD)
POLARS_PLUGIN_DEFINE(sorensen_dice, STRSIM_SORENSEN_DICE)

} // extern "C"
