/*
 * polars_plugin_abi.h -- the Polars expression-plugin C ABI exported by libpolars_strsim_amd.so.
 *
 * These are the symbols the reference's five `#[polars_expr(output_type=Float64)]` functions expand to
 * (reference src/expressions/mod.rs:8-31; macro = pyo3-polars-derive 0.11.0, FFI structs = polars-ffi
 * 0.43.1 `version_0`, pins in the reference's Cargo.lock:588-589,855-856,874-875).  Polars dlopen()s the one
 * shared library found in the plugin package directory (`plugin_path=Path(__file__).parent`, reference
 * polars_strsim/__init__.py:11-16), checks the version symbol and calls `_polars_plugin_<name>` with the
 * input Series exported over the Arrow C Data Interface.  Neither crate's source is vendored in the
 * reference tree: the layouts below are restated from the published 0.43.1 / 0.11.0 sources and are
 * "verify on first contact" (SURVEY.md 8b); tests/test_plugin_abi_gpu.py and tests/test_abi_symbols.py drive them with
 * pyarrow as the host (strsim_amd/arrow_host.py).
 *
 * Ownership (polars-ffi `import_series` / `export_series`): the callee owns every input SeriesExport and
 * every ArrowArray in it -- it calls each array's release and then the SeriesExport's release, once.  On
 * success it writes a fully formed SeriesExport into *return_value (the host imports the arrays by
 * bitwise copy and then calls return_value->release).  On failure *return_value is left untouched and
 * the message is available from _polars_plugin_get_last_error_message() on the same thread.
 */
#ifndef POLARS_PLUGIN_ABI_H
#define POLARS_PLUGIN_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define POLARS_PLUGIN_API __attribute__((visibility("default")))
#else
#define POLARS_PLUGIN_API
#endif

/* ---- Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html) ---- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4

struct ArrowSchema {
    const char *format;
    const char *name;
    const char *metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema **children;
    struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};

struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void **buffers;
    struct ArrowArray **children;
    struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif

/* ---- polars-ffi 0.43.1 version_0 ---- */
typedef struct SeriesExport {
    struct ArrowSchema *field;   /* name + dtype of the Series */
    struct ArrowArray **arrays;  /* `len` chunks */
    size_t len;
    void (*release)(struct SeriesExport *);
    void *private_data;
} SeriesExport;

typedef struct CallerContext {
    uint64_t bitflags; /* bit 0 = PARALLEL: the engine is already inside a parallel region (reference strsim.rs:53) */
} CallerContext;

#define POLARS_PLUGIN_VERSION_MAJOR 0u
#define POLARS_PLUGIN_VERSION_MINOR 1u /* minor 1 = call form carrying the CallerContext */

POLARS_PLUGIN_API uint32_t _polars_plugin_get_version(void);                  /* (major << 16) | minor */
POLARS_PLUGIN_API const char *_polars_plugin_get_last_error_message(void);    /* thread-local, NUL-terminated */

#define POLARS_PLUGIN_DECLARE(name)                                                                                  \
    POLARS_PLUGIN_API void _polars_plugin_##name(SeriesExport *inputs, size_t n_inputs, const uint8_t *kwargs,       \
                                                 size_t kwargs_len, SeriesExport *return_value, CallerContext *ctx); \
    POLARS_PLUGIN_API void _polars_plugin_field_##name(struct ArrowSchema *input_fields, size_t n_fields,            \
                                                       struct ArrowSchema *return_value);

/* reference src/expressions/mod.rs:8-11, :13-16, :18-21, :23-26, :28-31 */
POLARS_PLUGIN_DECLARE(levenshtein)
POLARS_PLUGIN_DECLARE(jaro)
POLARS_PLUGIN_DECLARE(jaro_winkler)
POLARS_PLUGIN_DECLARE(jaccard)
POLARS_PLUGIN_DECLARE(sorensen_dice)

/* ---- diagnostics of this implementation (not part of the polars-ffi contract; the engine never calls them) ----
 * The plugin's staging -- pinned host memory and its device mirrors, per pipeline set -- is leased per call from one process-wide
 * pool under POLARS_STRSIM_STAGING_BUDGET_MB (csrc/plugin_pack.h: StagingPool; reference counterpart: the per-call scratch of
 * strsim.rs:78-84, :109-123).  out[0..7] = live pinned bytes, live device bytes, budget bytes (0 = none), pipeline sets, sets in use,
 * sets released so far, calls that had to wait for the budget, peak live bytes seen when a call returned. */
POLARS_PLUGIN_API void _polars_plugin_strsim_staging_stats(uint64_t out[8]);
/* Small calls of concurrent engine threads are combined into one launch when enough of them are in flight (csrc/plugin_pipeline.h:
 * Combiner; POLARS_STRSIM_COALESCE / _COALESCE_MIN_INFLIGHT / _COALESCE_ROWS).  out[0..3] = combined launches, the calls they
 * carried, the most calls in one launch, small calls that took the ordinary path. */
POLARS_PLUGIN_API void _polars_plugin_strsim_coalesce_stats(uint64_t out[4]);
/* Change the budget of a running process (tests; 0 = no budget).  Takes effect with the next call. */
POLARS_PLUGIN_API void _polars_plugin_strsim_staging_set_budget_mb(uint64_t megabytes);

#ifdef __cplusplus
}
#endif
#endif /* POLARS_PLUGIN_ABI_H */
