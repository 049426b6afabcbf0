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

#include "plugin_arrow.h"
#include "plugin_pack.h"
#include "plugin_pipeline.h"

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
        bool combined = false;
        if (direct_call) {
            // a small call: when enough of them are in flight at once they share a launch (plugin_pipeline.h: Combiner)
            const SmallCallGuard in_flight;
            const CoalesceKnobs &ck = coalesce_knobs();
            if (ck.on && D == 1 && !lit[0] && !lit[1] && n <= ck.rows && in_flight.before + 1 >= ck.min_inflight)
                combined = combiner().run(measure, col, n, out, devs[0]);
            if (!combined) {
                combiner().direct_.fetch_add(1, std::memory_order_relaxed);
                run_rows(measure, col, lit, n, out, out_pinned, T, direct_call, engine_parallel, devs, tm, ptimes);
            }
        } else {
            run_rows(measure, col, lit, n, out, out_pinned, T, direct_call, engine_parallel, devs, tm, ptimes);
        }
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

void _polars_plugin_strsim_staging_stats(uint64_t out[8]) { if (out) staging_pool().stats(out); }
void _polars_plugin_strsim_staging_set_budget_mb(uint64_t megabytes) { staging_pool().set_budget(megabytes << 20); }
void _polars_plugin_strsim_coalesce_stats(uint64_t out[4]) { if (out) combiner().stats(out); }

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

// (4) the staging pool under `budget_bytes`: one call's lease of a pipeline set (`need` = its estimate), `grow` bytes of staging
// reserved in it (host memory in this build), held for `hold_us`, given back.  Called from many threads at once by the driver.
// stats8 (optional): _polars_plugin_strsim_staging_stats' eight counters after the call.
POLARS_PLUGIN_API int _strsim_test_staging_lease(uint64_t budget_bytes, uint64_t need, uint64_t grow, unsigned hold_us, uint64_t *stats8)
{
    try {
        staging_pool().set_budget(budget_bytes);
        {
            PipeLease lease(need);
            Pipe &P = lease.set->at(0);
            P.slot[(need >> 3) % 3].h_val[0].reserve(grow);
            memset(P.slot[(need >> 3) % 3].h_val[0].p, 0x5A, grow);
            if (hold_us) std::this_thread::sleep_for(std::chrono::microseconds(hold_us));
        }
        if (stats8) staging_pool().stats(stats8);
        return 0;
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
    return -1;
}

// (5) one small call through the combiner (plugin_pipeline.h), with a CPU stand-in for the combined launch: out[i] = 4096 * (bytes of
// row i of a) + (bytes of row i of b) + (first byte of a's row) / 256.  Owns and releases the two inputs like a plugin call.  Returns 1 when
// the call was combined, 0 when it was not eligible, -1 on failure.
static void test_combined_launch(const uint32_t *oa, const uint8_t *va, const uint32_t *ob, const uint8_t *, uint64_t rows, double *out)
{
    for (uint64_t i = 0; i < rows; ++i) {
        const uint32_t la = oa[i + 1] - oa[i], lb = ob[i + 1] - ob[i];
        out[i] = 4096.0 * la + lb + (la ? va[oa[i]] / 256.0 : 0.0);
    }
}
POLARS_PLUGIN_API int _strsim_test_combine(SeriesExport *two_series, int measure, double *out, uint64_t *rows_out)
{
    InputGuard guard{two_series, 2};
    try {
        Column col[2];
        describe(two_series[0], col[0]);
        describe(two_series[1], col[1]);
        if (col[0].rows != col[1].rows) fail("shape");
        if (rows_out) *rows_out = col[0].rows;
        static const bool stand_in = (combiner().test_launch = test_combined_launch, true); // (once: every caller thread comes through here)
        (void)stand_in;
        const SmallCallGuard in_flight;
        return col[0].rows && combiner().run(measure, col, col[0].rows, out, 0) ? 1 : 0;
    } catch (const PluginError &e) {
        g_plugin_error = e.msg;
    } catch (...) {
        g_plugin_error = "unexpected failure";
    }
    return -1;
}
#endif // STRSIM_TEST_HOOKS

POLARS_PLUGIN_DEFINE(levenshtein, STRSIM_LEVENSHTEIN)
POLARS_PLUGIN_DEFINE(jaro, STRSIM_JARO)
POLARS_PLUGIN_DEFINE(jaro_winkler, STRSIM_JARO_WINKLER)
POLARS_PLUGIN_DEFINE(jaccard, STRSIM_JACCARD)
POLARS_PLUGIN_DEFINE(sorensen_dice, STRSIM_SORENSEN_DICE)

} // extern "C"
