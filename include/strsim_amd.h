/*
 * strsim_amd.h -- C ABI of the MI355X (gfx950) pairwise string-similarity library.
 *
 * This is the drop-in boundary for the hot path of foxcroftjn/polars-strsim: everything the
 * reference computes inside `strsim::parallel_apply` (reference src/expressions/strsim.rs:41-107) and
 * the five `SimilarityFunction::compute` bodies (:125-162, :180-245, :257-272, :286-308, :322-345).
 *
 * Two layers are exported from the same shared library (libpolars_strsim_amd.so):
 *
 *   1. the thin kernel ABI below (`strsim_*`): Arrow Utf8 offsets+values buffers in, f64 column out.
 *      This is what a Rust host (the reference's `parallel_apply`) would bind through `extern "C"`
 *      after flattening its `StringChunked` -- see INTEGRATION.md for the binding stub.
 *   2. the Polars plugin ABI (`_polars_plugin_*`, include/polars_plugin_abi.h): the symbols the
 *      reference's `#[polars_expr]` macro generates (reference src/expressions/mod.rs:8-31), so the
 *      library can be dropped into the `polars_strsim` package directory unchanged.
 *
 * Plain pointers and sizes only; no C++/torch types.  All functions return a strsim_status_t
 * (0 = OK) unless stated otherwise; on error a message is available from
 * strsim_last_error_message() (thread-local).  Nothing here ever computes on the CPU: without a
 * usable GPU every compute entry point fails with STRSIM_ERR_NO_DEVICE.
 */
#ifndef STRSIM_AMD_H
#define STRSIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STRSIM_ABI_VERSION 0x00010006u /* major<<16 | minor; 1.1: strsim_pairs_device_small, strsim_codec_patch_indirect; 1.2: strsim_ctx_retire_oldest, strsim_offsets_from_lengths; 1.3: one-launch calls (strsim_ctx_set_stream_ordered, strsim_ctx_last_late_rows); 1.4: one-launch calls are OPT-IN -- a new context completes rows in stream order, as in 1.2; 1.5: strsim_column_from_views_bounded, strsim_codec_decode_gathered_from, strsim_gather_*, STRSIM_ERR_EARLIER_CALL; 1.6: strsim_gather_f64_ranges, strsim_gather_comm_count, strsim_ctx_get_stream_ordered */

#if defined(__GNUC__)
#define STRSIM_API __attribute__((visibility("default")))
#else
#define STRSIM_API
#endif

/* Mirrors `enum SimilarityFunctionType` (reference strsim.rs:9-15). */
typedef enum strsim_measure {
    STRSIM_LEVENSHTEIN   = 0, /* normalised Levenshtein similarity, strsim.rs:125-162 */
    STRSIM_JARO          = 1, /* strsim.rs:180-245 */
    STRSIM_JARO_WINKLER  = 2, /* strsim.rs:257-272 */
    STRSIM_JACCARD       = 3, /* character-multiset Jaccard, strsim.rs:286-308 */
    STRSIM_SORENSEN_DICE = 4, /* character-multiset Sorensen-Dice, strsim.rs:322-345 */
    STRSIM_NUM_MEASURES  = 5
} strsim_measure_t;

typedef enum strsim_status {
    STRSIM_OK              = 0,
    STRSIM_ERR_SHAPE       = 1, /* reference ShapeMismatch, strsim.rs:48-52 */
    STRSIM_ERR_ARG         = 2, /* null pointer / bad enum / bad size */
    STRSIM_ERR_NO_DEVICE   = 3, /* no usable HIP device: there is no CPU fallback */
    STRSIM_ERR_HIP         = 4, /* a HIP runtime call failed (message has the detail) */
    STRSIM_ERR_OOM         = 5,
    STRSIM_ERR_DTYPE       = 6, /* plugin ABI: input is not a string column (reference `.str()?`, strsim.rs:46-47) */
    STRSIM_ERR_INTERNAL    = 7,
    STRSIM_ERR_EARLIER_CALL = 8 /* strsim_pairs_device*: retiring EARLIER pending calls at a wrap of the context's ring of 32 failed
                                   (message has the detail); THIS call was not enqueued and none of its buffers was touched */
} strsim_status_t;

typedef struct strsim_ctx strsim_ctx_t; /* one device + one stream + its workspace; not thread-safe: one ctx per thread */

/* (major<<16)|minor of this ABI. */
STRSIM_API uint32_t strsim_abi_version(void);

/* NUL-terminated description of the last error raised on the calling thread ("" if none). */
STRSIM_API const char *strsim_last_error_message(void);

/* Number of usable HIP devices (0 when there is none; never fails). */
STRSIM_API int strsim_device_count(void);

/* Create a context on `device`.  `hip_stream` is a hipStream_t to enqueue on (e.g. the host
 * framework's current stream); NULL makes the context create and own a non-blocking stream. */
STRSIM_API int strsim_ctx_create(int device, void *hip_stream, strsim_ctx_t **out_ctx);
STRSIM_API void strsim_ctx_destroy(strsim_ctx_t *ctx);

/* The stream work is enqueued on (a hipStream_t). */
STRSIM_API void *strsim_ctx_stream(strsim_ctx_t *ctx);

/*
 * Enqueue one pass of `measure` over two DEVICE-RESIDENT Utf8 column shards.
 *
 * Column layout (Arrow "u", rebased): `offsets` = uint32[rows+1] with offsets[0] == 0... (any
 * monotone base is accepted), `values` = the packed UTF-8 bytes [0, offsets[rows]).  A side with
 * rows == 1 is the "Utf8 literal" of strsim.rs:48-52 and is broadcast against every row of the
 * other side (strsim.rs:61-66).  Otherwise a_rows must equal b_rows, else STRSIM_ERR_SHAPE.
 * `out` = double[out_rows] on the same device, out_rows = max(a_rows, b_rows).
 * Nulls are not seen here: like the reference's arity helpers the kernels compute on the bytes under
 * every slot; validity is combined by the caller (the plugin layer does it).
 *
 * Reads beyond the strings: the kernels copy the values of a block of rows in whole 16-byte chunks, from the
 * 16-byte-aligned address at or below the block's first byte (a_values + a_offsets[first row]) up to the chunk that
 * holds the column's last byte (a_values + a_offsets[a_rows]) -- up to 15 bytes in front of and behind the bytes the
 * offsets describe.  Those reads stay inside the 16-byte-aligned chunks the column itself touches (so inside its
 * pages); bytes in front belong to whatever precedes the column in the caller's buffer and only ever sit beside a
 * string in a staging area, bytes behind the column's last byte are replaced by zeros before any row looks at them.
 * Nothing outside [a_offsets[0], a_offsets[a_rows]) can change a result.
 *
 * The call is asynchronous: it returns once the kernels are enqueued on the context's stream.
 * Results are complete after strsim_ctx_synchronize() (or strsim_ctx_retire_oldest() of this call).
 * All buffers of a call must stay valid until then.
 *
 * What the stream alone guarantees: every row of strings <= STRSIM_WAVE_PATH_MAX_BYTES is complete in stream
 * order (work enqueued on strsim_ctx_stream() behind the call sees it); rows with a longer string are always
 * finished by a pass that strsim_ctx_synchronize() / strsim_ctx_retire_oldest() launches
 * (strsim_ctx_last_long_rows() tells).
 *
 * One-launch calls (opt-in since ABI 1.4: strsim_ctx_set_stream_ordered(ctx, 0)): a context whose last retired
 * call had every row finished by the one-pair-per-lane kernel (both strings <= STRSIM_LANE_PATH_MAX_BYTES, ASCII
 * -- the common column) enqueues the NEXT call as that kernel alone: ONE launch instead of five.  If such a call
 * does hold longer or non-ASCII rows, the kernels for them are launched when the call is retired, i.e. AFTER
 * anything the caller enqueued behind the call -- strsim_ctx_last_late_rows() tells, and the calls after it enqueue
 * all their kernels up front again.  For callers that retire every call (strsim_ctx_retire_oldest /
 * strsim_ctx_synchronize) before they consume its results, and look at strsim_ctx_last_late_rows() if they copied
 * results out early: the plugin layer does.  A pending one-launch call owns a "not finished yet" mask (20 bytes per 64 rows)
 * until it is retired, so a caller that keeps k calls in flight holds k of them.  The library never retires a call the
 * caller has not asked it to, except when more than 32 calls are in flight on one context: then everything pending is
 * retired inside strsim_pairs_device (a stream synchronise) and what that finished late is added to the next
 * retirement's strsim_ctx_last_late_rows().  If retiring them fails there, the error is THEIRS: the call returns
 * STRSIM_ERR_EARLIER_CALL and has not been enqueued (ABI 1.5; before, it returned the earlier call's own code).
 */
STRSIM_API int strsim_pairs_device(strsim_ctx_t *ctx, int measure,
                        const uint32_t *a_offsets, const uint8_t *a_values, uint64_t a_rows,
                        const uint32_t *b_offsets, const uint8_t *b_values, uint64_t b_rows,
                        double *out, uint64_t out_rows);

/* strsim_pairs_device for a SMALL call that the caller synchronises right away (strsim_pairs_host's in-place path, the
 * plugin's direct path; reference: one `compute` per row on the calling thread, strsim.rs:53-70).  May block until the
 * one-pair-per-lane kernel has finished: when that kernel leaves no row behind -- short ASCII strings, the common case --
 * the call is complete after one kernel launch and nothing is pending; otherwise the remaining kernels are enqueued and the
 * call completes in strsim_ctx_synchronize() like any other.  Same arguments and errors as strsim_pairs_device. */
STRSIM_API int strsim_pairs_device_small(strsim_ctx_t *ctx, int measure, const uint32_t *a_offsets, const uint8_t *a_values,
                                         uint64_t a_rows, const uint32_t *b_offsets, const uint8_t *b_values, uint64_t b_rows,
                                         double *out, uint64_t out_rows);

/*
 * All five measures of the same two column shards in one call (BASELINE config 4): the rows that fit the
 * lane-per-pair path are read once and produce five outputs from one set of bit-planes; `outs` is indexed by
 * strsim_measure_t, five device buffers of out_rows doubles.  Same asynchronous contract as above.
 */
STRSIM_API int strsim_pairs_device_all(strsim_ctx_t *ctx,
                            const uint32_t *a_offsets, const uint8_t *a_values, uint64_t a_rows,
                            const uint32_t *b_offsets, const uint8_t *b_values, uint64_t b_rows,
                            double *const outs[5], uint64_t out_rows);

/* Wait for everything enqueued through this context and surface any deferred error. */
STRSIM_API int strsim_ctx_synchronize(strsim_ctx_t *ctx);

/* 1 (default): every call enqueues all its kernels up front (results of rows <= STRSIM_WAVE_PATH_MAX_BYTES complete in stream
 * order); 0: calls that are expected to need the first kernel only are one launch (see strsim_pairs_device).  Takes effect
 * with the next call; calls already pending keep the mode they were enqueued in. */
STRSIM_API int strsim_ctx_set_stream_ordered(strsim_ctx_t *ctx, int enable);
/* The mode the NEXT call of the context is enqueued in: 1 = stream-ordered (the default), 0 = one-launch calls (ABI 1.6). */
STRSIM_API int strsim_ctx_get_stream_ordered(strsim_ctx_t *ctx);

/*
 * Same contract with HOST-RESIDENT buffers: stages the shards to the device, runs the kernels and
 * copies the f64 column back; synchronous.  Calls of up to 65 536 rows (and 2 MiB of values per
 * column) are gathered in one pinned block the context owns and computed there in place through the
 * device's mapping of host memory -- no copy engine, ~37 us per small call instead of ~70.
 */
STRSIM_API int strsim_pairs_host(strsim_ctx_t *ctx, int measure,
                      const uint32_t *a_offsets, const uint8_t *a_values, uint64_t a_rows,
                      const uint32_t *b_offsets, const uint8_t *b_values, uint64_t b_rows,
                      double *out, uint64_t out_rows);

/* Row partition used to shard a column over `n` GPUs/ranks: the reference's split_offsets
 * (strsim.rs:21-39).  Writes n (offset,len) pairs into out_offset_len[2*n]. */
STRSIM_API void strsim_split_offsets(uint64_t len, uint64_t n, uint64_t *out_offset_len);

/*
 * Kernel timing for roofline accounting.  When enabled, every strsim_pairs_device() brackets its
 * dominant (lane-per-pair) kernel and its wave-per-pair kernel with hipEvents on the context's stream.
 * strsim_ctx_timing_read() synchronises and returns the accumulated milliseconds and launch counts
 * since the last read, then resets them.
 */
STRSIM_API int strsim_ctx_timing_enable(strsim_ctx_t *ctx, int enable);
STRSIM_API int strsim_ctx_timing_read(strsim_ctx_t *ctx, double *lane_kernel_ms, uint64_t *lane_kernel_launches,
                           double *wave_kernel_ms, uint64_t *wave_kernel_launches);

/* Offsets of a column from its string LENGTHS, on the device: offsets[0] = 0, offsets[i + 1] = offsets[i] + lengths[i]
 * (lengths: `rows` bytes on the device, 16-byte aligned; offsets: rows + 1 words).  Asynchronous on the context's stream.
 * For a host that holds strings as views / (pointer, length) pairs (Polars' Utf8View, the layout the reference iterates at
 * strsim.rs:46-47) and ships one length byte per row over PCIe instead of a u32 offset; strings of at most 255 bytes, at
 * most STRSIM_OFFSETS_FROM_LENGTHS_MAX_ROWS rows per call (= floor((2^32 - 1) / 255): the 32-bit offsets cannot wrap whatever
 * the lengths are; more rows: STRSIM_ERR_ARG).  The result is what strsim_pairs_device() takes as a_offsets / b_offsets. */
#define STRSIM_OFFSETS_FROM_LENGTHS_MAX_ROWS 16843009u
STRSIM_API int strsim_offsets_from_lengths(strsim_ctx_t *ctx, const uint8_t *lengths, uint64_t rows, uint32_t *offsets);

/* A column from Utf8View slots, on the device (ABI 1.4; SURVEY 8 f1: the layout the reference iterates, strsim.rs:46-47).  `views`:
 * rows Arrow Utf8View slots of 16 bytes each, 16-byte aligned, as the engine holds them -- a u32 length, then either the string
 * itself (length <= 12) or its first four bytes, a u32 buffer index and a u32 offset -- with ONE thing changed by the host on the
 * way: a slot whose string does not fit it (length > 12) carries in its last word the string's offset in `long_values`, the bytes
 * of those strings as the host has shipped them (any order, gaps allowed; the buffer index is ignored).  A null slot is shipped as
 * length 0.  Writes offsets[rows + 1] and the packed values (their size, the sum of the lengths, is the host's to know: it has
 * seen every length) -- what strsim_pairs_device() takes.  Asynchronous on the context's stream, two launches; the packed size of
 * one call must fit 32-bit offsets, at most 33 554 432 rows per call.  Reference counterpart: none (the reference reads the views
 * in place); this is what lets a host with few cycles to spare -- the engine-parallel mode packs on the calling thread alone --
 * hand a String column over with a streaming copy instead of a gather. */
STRSIM_API int strsim_column_from_views(strsim_ctx_t *ctx, const void *views, uint64_t rows, const uint8_t *long_values,
                                        uint32_t *offsets, uint8_t *values);
/* The same with the extents stated (ABI 1.5): `long_values` holds long_bytes bytes, `values` has room for values_bytes.  A slot
 * whose string would be read from outside long_values (or whose long_values is NULL), or written outside values -- a malformed
 * column: the slots are the host's to get right -- is NOT copied (its offsets are still written) and counted in *malformed, a
 * u32 on the device that the caller has zeroed (NULL: not counted): a bad slot cannot fault the GPU.  Sums of lengths beyond
 * 2^32 - 1 put every later row out of range instead of wrapping.  strsim_column_from_views() is this call with the extents
 * "not stated" (a NULL long_values still skips every slot beyond 12 bytes). */
STRSIM_API int strsim_column_from_views_bounded(strsim_ctx_t *ctx, const void *views, uint64_t rows, const uint8_t *long_values,
                                                uint64_t long_bytes, uint32_t *offsets, uint8_t *values, uint64_t values_bytes,
                                                uint32_t *malformed);

/* Close the gaps between up to STRSIM_COMPACT_MAX_SEGMENTS byte segments on the device, one launch on the context's stream:
 * dst[dst_off[k] .. + bytes[k]) = src[src_off[k] .. + bytes[k]) for k < nseg; src and dst are distinct device buffers, the three
 * arrays are host memory (read before the call returns).  For a host that packs a column's values with several threads in ONE
 * pass -- each thread into its own segment of a staging buffer, no common prefix, one length byte per row (see
 * strsim_offsets_from_lengths) -- and ships the buffer as it lies: the plugin layer does (reference counterpart: none; the
 * reference iterates the views in place, strsim.rs:46-47). */
#define STRSIM_COMPACT_MAX_SEGMENTS 32
STRSIM_API int strsim_compact_segments(strsim_ctx_t *ctx, const uint8_t *src, uint8_t *dst, const uint64_t *src_off,
                                       const uint64_t *dst_off, const uint64_t *bytes, int nseg);

/* For a caller that keeps several calls in flight on the context's stream and learns of their completion by its own means
 * (an event recorded on strsim_ctx_stream() behind each call): retire the OLDEST pending call only -- what
 * strsim_ctx_synchronize() does for all of them, without waiting for the younger ones.  The caller guarantees that the
 * oldest call's kernels have completed; the library checks it (the call's status block carries a ticket the device writes
 * last) and returns STRSIM_ERR_ARG, leaving the call pending, when they have not -- an event recorded on any OTHER stream
 * than strsim_ctx_stream() proves nothing about this context's kernels.  If that call held strings longer than STRSIM_WAVE_PATH_MAX_BYTES, their second
 * pass is launched and waited for here (strsim_ctx_last_long_rows() tells).  Reference counterpart: none -- the reference's
 * rayon loop (strsim.rs:72-100) has no device queue; this is what lets the plugin overlap H2D of slice k+1, the kernels of
 * slice k and the D2H of slice k-1. */
STRSIM_API int strsim_ctx_retire_oldest(strsim_ctx_t *ctx);

/* Rows the last completed strsim_pairs_device() on this context routed to the wave-per-pair kernel
 * (valid after strsim_ctx_synchronize()). */
STRSIM_API uint64_t strsim_ctx_last_wave_rows(strsim_ctx_t *ctx);

/* Rows of the calls retired by the last strsim_ctx_synchronize() / strsim_ctx_retire_oldest() that held a string longer than
 * STRSIM_WAVE_PATH_MAX_BYTES: their results were written by the second pass that synchronize runs, i.e. AFTER anything
 * the caller enqueued on the stream behind the call (a caller that copies results out early re-copies when > 0). */
STRSIM_API uint64_t strsim_ctx_last_long_rows(strsim_ctx_t *ctx);

/* Rows of the calls retired by the last strsim_ctx_synchronize() / strsim_ctx_retire_oldest() whose results were written by a
 * pass launched from there -- the slow-row kernels of a one-launch call that did hold such rows, and the long-string pass --
 * i.e. AFTER anything the caller enqueued on the stream behind the call (a caller that copied results out early copies again
 * when this is > 0).  >= strsim_ctx_last_long_rows(). */
STRSIM_API uint64_t strsim_ctx_last_late_rows(strsim_ctx_t *ctx);

/* Kernels and device copies this context has enqueued for pair calls since it was created (a call of a column whose rows
 * all fit the one-pair-per-lane kernel adds 1; a call with all kernels up front 5; introspection for tests and benches). */
STRSIM_API uint64_t strsim_ctx_enqueued_ops(strsim_ctx_t *ctx);

/* ---- the gather of the result shards over RCCL (ABI 1.5) --------------------------------------------------------------------
 * Rows are independent, so N processes -- one per GPU -- each run strsim_pairs_device on the shard strsim_split_offsets(rows, N)
 * gives them (the reference's own partition, strsim.rs:21-39) with no data-path collective; the one exchange step of the path is
 * the gather of the f64 result shards onto a root rank (reference counterpart: the threads' chunks collected into one
 * Float64Chunked, strsim.rs:98-104).  These four calls are that step for a host that binds this header (a Rust shim: INTEGRATION.md):
 *   rank 0:      strsim_gather_unique_id(id)            -- and hands the 128 bytes to the other ranks by whatever means it has
 *   every rank:  strsim_gather_create(ctx, id, N, rank, &g)   (collective: returns when all N ranks have called it)
 *   every step:  strsim_gather_f64(g, shard, column, rows, root)   -- enqueued on the context's stream behind the kernels
 * `shard`: this rank's rows of the result column (device memory, split_offsets(rows, N)[rank] of them); `column`: the whole
 * column on the root (device memory, `rows` doubles; ignored elsewhere).  Every peer's shard travels point to point into the root
 * (ncclSend / ncclRecv in one group: xGMI is point-to-point, seven links into the root at N = 8), the root's own shard is a device
 * copy.  RCCL is resolved at first use (a copy the process already holds, else librccl.so.1 from the loader's path) and is not a
 * link-time dependency of the library; without it these calls fail with STRSIM_ERR_NO_DEVICE and nothing else is affected.
 * STRSIM_RCCL_LIB=<path> names the library to use instead (read once per process). */
#define STRSIM_GATHER_ID_BYTES 128
typedef struct strsim_gather strsim_gather_t;
STRSIM_API int strsim_gather_unique_id(uint8_t id[STRSIM_GATHER_ID_BYTES]);
STRSIM_API int strsim_gather_create(strsim_ctx_t *ctx, const uint8_t id[STRSIM_GATHER_ID_BYTES], int world_size, int rank,
                                    strsim_gather_t **out);
STRSIM_API int strsim_gather_f64(strsim_gather_t *g, const double *shard, double *column, uint64_t total_rows, int root);
/* The same over an explicit partition (ABI 1.6): ranges = uint64[2 * N], rank r holds rows [ranges[2r], ranges[2r] + ranges[2r+1])
 * of the column -- for a host whose shards are not split_offsets' (a root that takes a smaller share because it also assembles the
 * column: a named deviation from strsim.rs:21-39).  The same array on every rank; overlapping ranges are refused. */
STRSIM_API int strsim_gather_f64_ranges(strsim_gather_t *g, const double *shard, double *column, const uint64_t *ranges, int root);
/* How many ranks the RCCL communicator of `g` itself holds (ncclCommCount): the answer to "did RCCL form the world I think it did"
 * that does not go through the caller's own bookkeeping (ABI 1.6). */
STRSIM_API int strsim_gather_comm_count(strsim_gather_t *g, int *count);
STRSIM_API void strsim_gather_destroy(strsim_gather_t *g);

/*
 * Lossless 16-bit transport codec for result columns (csrc/strsim_codec.hip).  A similarity of two strings of at
 * most `max_chars` characters takes few distinct values (max_chars = 32: 325 / 22 856 / 57 359 / 631 / 631 for the
 * five measures); the codec ships the 16-bit rank of each value instead of 8 bytes -- 4x less traffic on the
 * point-to-point xGMI link of a gather -- and decodes bit-exactly.  Values outside the table (rows with longer
 * strings) are coded 0xFFFF and reported as (row, value) exceptions.  All buffers are device memory of the
 * context's GPU; calls are asynchronous on the context's stream.
 */
typedef struct strsim_codec strsim_codec_t;
/* Fails with STRSIM_ERR_ARG when the value set does not fit 16 bits (e.g. Jaro with max_chars = 128). */
STRSIM_API int strsim_codec_create(strsim_ctx_t *ctx, int measure, uint32_t max_chars, strsim_codec_t **out);
STRSIM_API void strsim_codec_destroy(strsim_codec_t *codec);
STRSIM_API uint32_t strsim_codec_entries(const strsim_codec_t *codec);
/* vals[n] -> codes[n]; *exc_count (device) = number of exceptions, the first exc_cap of them in exc_rows/exc_vals. */
STRSIM_API int strsim_codec_encode(strsim_ctx_t *ctx, const strsim_codec_t *codec, const double *vals, uint64_t n,
                                   uint16_t *codes, uint32_t *exc_count, uint32_t *exc_rows, double *exc_vals,
                                   uint32_t exc_cap);
/* codes[n] -> out[n]; rows coded 0xFFFF are left untouched. */
STRSIM_API int strsim_codec_decode(strsim_ctx_t *ctx, const strsim_codec_t *codec, const uint16_t *codes, uint64_t n,
                                   double *out);
/* Packed transport of the same codes: strsim_codec_bits() bits per row (the smallest b with 2^b > entries; the
 * all-ones code is the escape), 64 / bits rows per 64-bit word: 9.14 bits per row for Levenshtein (325 values),
 * 10.67 for Jaccard / Dice (631).  words[strsim_codec_packed_words(codec, n)]; exceptions as in strsim_codec_encode. */
STRSIM_API uint32_t strsim_codec_bits(const strsim_codec_t *codec);
STRSIM_API uint64_t strsim_codec_packed_words(const strsim_codec_t *codec, uint64_t n);
STRSIM_API int strsim_codec_encode_packed(strsim_ctx_t *ctx, const strsim_codec_t *codec, const double *vals, uint64_t n,
                                          uint64_t *words, uint32_t *exc_count, uint32_t *exc_rows, double *exc_vals,
                                          uint32_t exc_cap);
STRSIM_API int strsim_codec_decode_packed(strsim_ctx_t *ctx, const strsim_codec_t *codec, const uint64_t *words, uint64_t n,
                                          double *out);
/* out[row_base + exc_rows[i]] = exc_vals[i] for i < count. */
STRSIM_API int strsim_codec_patch(strsim_ctx_t *ctx, double *out, uint64_t row_base, const uint32_t *exc_rows,
                                  const double *exc_vals, uint32_t count);

/* The same with the count read on the device: exc_count / exc_rows / exc_vals are another rank's exception block as it
 * arrived with the gathered codes (strsim_amd/distributed.py).  A count above exc_cap (the sender had more exceptions than
 * the block holds: the column is incomplete) increments *overflow (device memory). */
STRSIM_API int strsim_codec_patch_indirect(strsim_ctx_t *ctx, double *out, uint64_t row_base, const uint32_t *exc_count,
                                           const uint32_t *exc_rows, const double *exc_vals, uint32_t exc_cap,
                                           uint32_t *overflow);

/* The root's side of a gather in ONE launch: `buf` holds nseg segments seg_stride_bytes apart, segment r = rank r's codes (packed:
 * strsim_codec_encode_packed's words; else 16-bit codes) followed, code_bytes into the segment, by its exception block -- count
 * (u32, 16-byte field), exc_cap rows (u32), exc_cap values (f64).  Rows r * chunk_rows .. (last segment: last_rows of them) of `out`
 * are decoded and the exceptions written in; a count above exc_cap increments *overflow (device).  Replaces one
 * strsim_codec_decode(_packed) + one strsim_codec_patch_indirect launch per peer (strsim_amd/distributed.py). */
STRSIM_API int strsim_codec_decode_gathered(strsim_ctx_t *ctx, const strsim_codec_t *codec, const void *buf, uint64_t seg_stride_bytes,
                                            uint32_t nseg, uint64_t chunk_rows, uint64_t last_rows, int packed, uint64_t code_bytes,
                                            uint32_t exc_cap, double *out, uint32_t *overflow);
/* The same over segments first_seg .. nseg - 1 only (ABI 1.5): the root of a gather has no reason to code and decode its OWN shard --
 * it copies its f64 results into `out` and decodes the peers' segments (first_seg = 1: an eighth of the decode and the whole
 * encode off the rank that every step waits for, profiles/r5_root_rehearsal.txt).  Row r * chunk_rows is still segment r's first. */
STRSIM_API int strsim_codec_decode_gathered_from(strsim_ctx_t *ctx, const strsim_codec_t *codec, const void *buf, uint64_t seg_stride_bytes,
                                                 uint32_t first_seg, uint32_t nseg, uint64_t chunk_rows, uint64_t last_rows, int packed,
                                                 uint64_t code_bytes, uint32_t exc_cap, double *out, uint32_t *overflow);

#define STRSIM_LANE_PATH_MAX_BYTES 32u   /* lane-per-pair kernels: both strings <= 32 bytes, ASCII */
#define STRSIM_WAVE_PATH_MAX_BYTES 1024u /* wave-per-pair kernels: both strings <= 1024 bytes, any UTF-8 */

#ifdef __cplusplus
}
#endif
#endif /* STRSIM_AMD_H */
