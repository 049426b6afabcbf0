// plugin_pipeline.h -- one call's rows through the device pipelines (slices packed, shipped, computed and fetched while the next one is packed;
// literals; small calls in place), and the output validity built word-wise on the packing pool.
// Included by polars_plugin.cpp inside its anonymous namespace, in this order ([r5] split out of polars_plugin.cpp along its seams,
// no behaviour change -- the object code is identical before and after).
// Reference: parallel_apply, /root/reference/src/expressions/strsim.rs:41-107.
#pragma once

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
    // this call's pipelines, leased for its duration (StagingPool).  The estimate: three slots of the largest slice the call will cut
    // (the ramp and the taper keep it near 0.4 n below the 2 M-row steady state), ~96 bytes a row in a slot -- pinned: one length byte
    // and ~17 bytes of string per column, the f64 result; device: the same, offsets, and the landing area of one-pass slices; with
    // the slack Buf::reserve adds.  Measured: 411 MB after a 4 M-row call of cfg2's strings (this gives 461), ~0.55 GB after a 10 M-row
    // one (576).  The pool counts a set at its real size as soon as that is larger, so a frame of long strings corrects it.
    const uint64_t need = 3 * 96 * std::min<uint64_t>(std::max<uint64_t>(n * 2 / 5, 1), SLICE_ROWS * D);
    PipeLease lease(need);
    std::vector<Pipe *> pipes(D);
    std::vector<hipStream_t> streams(D);
    for (size_t d = 0; d < D; ++d) {
        pipes[d] = &lease.set->at(d);
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
        if (sl.as_views[0] || sl.as_views[1]) { // the device counts the slots it refuses (a malformed view cannot fault it): see finish()
            sl.h_bad.reserve(64);
            *static_cast<volatile uint32_t *>(sl.h_bad.p) = 0u;
        }
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
                                                     sl.bytes[s] + 64, static_cast<uint32_t *>(mapped(sl.h_bad.p))) != STRSIM_OK)
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
        //  view-native slices: a slot the device skipped would be a row computed on zeroed bytes -- an error, never silence
        if ((sl.as_views[0] || sl.as_views[1]) && *static_cast<volatile uint32_t *>(sl.h_bad.p) != 0u)
            fail("internal: " + std::to_string(*static_cast<volatile uint32_t *>(sl.h_bad.p)) + " string views of rows " + std::to_string(sl.r0) + ".." +
                 std::to_string(sl.r0 + sl.rows) + " point outside the bytes shipped with them");
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

// ---- small calls of CONCURRENT engine threads, combined into one launch ("batching of concurrent small calls") ----
// An elementwise plugin is called per morsel / per group from the engine's own threads (polars_strsim/__init__.py:15), and a small
// call is one kernel launch + one synchronise: 26 us alone, ~100 us each when 16 threads do it at once (160-195 K calls/s in all on a
// box with a 16-CPU quota: bench_support/micro/plugin_small_threads.cpp).  With POLARS_STRSIM_COALESCE=1, when enough small calls are
// in flight (POLARS_STRSIM_COALESCE_MIN_INFLIGHT, default 8) they are combined, group-commit style: a call joins the OPEN batch of its
// measure -- or opens one and becomes its leader -- reserves its rows and bytes there, packs them into the batch's pinned arena (the
// offsets of one batch are ONE column: every member's rows start where the previous member's end), and waits; the leader waits for the
// combined launch in front of it to finish (calls keep joining meanwhile), seals the batch, waits for its members to finish packing,
// launches ONE kernel over all rows, synchronises, and releases the members, who copy their rows out.  Every wait spins for up to
// 200 us before it sleeps (a futex wake-up costs more than the launch: with condition variables alone the combiner ran at HALF the
// ordinary path's rate).  Columns only (a literal side takes the ordinary path), one device.
// OPT-IN, by its own measurement (profiles/r6_small_calls.txt): 6 calls share a launch at 16 threads and the rate is +5..15 % there,
// -34 % at 8 threads, equal at 32 -- what saturates at ~160 K calls/s is the box's CPU quota under threads that spin in the runtime,
// not the launch path, and a combiner cannot give CPUs back that its own members spin on.
// POLARS_STRSIM_COALESCE_ROWS: the largest call that may join (default 8192 rows).
constexpr uint64_t COMBO_ROWS = 65536, COMBO_BYTES = (uint64_t)2 << 20; // a batch's capacity: rows, packed bytes per column

struct ComboBatch {
    int measure = 0;
    Buf h_off[2], h_val[2], h_out; // pinned, fixed capacity (members pack concurrently: nothing may move)
    uint64_t rows = 0, bytes[2] = {0, 0};
    int members = 0, inside = 0;
    std::atomic<int> packed{0};     // (the waits below spin on these before they sleep: a futex wake-up costs more than the launch)
    std::atomic<bool> done{false};
    bool sealed = false, failed = false;
    std::string err;
    void reset(int m)
    {
        measure = m; rows = 0; bytes[0] = bytes[1] = 0; members = inside = 0;
        packed.store(0, std::memory_order_relaxed); done.store(false, std::memory_order_relaxed);
        sealed = failed = false; err.clear();
        for (int s = 0; s < 2; ++s) {
            h_off[s].reserve((COMBO_ROWS + 1) * sizeof(uint32_t));
            h_val[s].reserve(COMBO_BYTES + 4096);
            static_cast<uint32_t *>(h_off[s].p)[0] = 0u;
        }
        h_out.reserve(COMBO_ROWS * sizeof(double));
    }
};

std::atomic<int> g_small_inflight{0}; // small (direct) plugin calls inside the library right now, combined or not
struct SmallCallGuard {
    int before;
    SmallCallGuard() : before(g_small_inflight.fetch_add(1, std::memory_order_relaxed)) {}
    ~SmallCallGuard() { g_small_inflight.fetch_sub(1, std::memory_order_relaxed); }
};

class Combiner {
  public:
    // true: the call was computed as a member of a combined launch (out[0 .. n) is filled); false: not eligible, take the ordinary path
    bool run(int measure, const Column (&col)[2], uint64_t n, double *out, int device)
    {
        const uint64_t need[2] = {range_bytes(col[0], 0, n), range_bytes(col[1], 0, n)};
        if (need[0] > COMBO_BYTES / 4 || need[1] > COMBO_BYTES / 4) return false; // (a call of long strings: its own launch)
        ComboBatch *b;
        bool leader = false;
        uint64_t row0, base[2];
        {
            std::unique_lock<std::mutex> lk(m_);
            b = open_[measure];
            if (!b || b->sealed || b->rows + n > COMBO_ROWS || b->bytes[0] + need[0] > COMBO_BYTES || b->bytes[1] + need[1] > COMBO_BYTES) {
                // (a full batch stays with its leader, who seals it; this call opens the next one)
                if (free_.empty()) {
                    b = new ComboBatch;
                } else {
                    b = free_.back();
                    free_.pop_back();
                }
                b->reset(measure); // (first use: pins the arenas -- under the lock, once per batch object; a handful exist)
                open_[measure] = b;
                leader = true;
            }
            row0 = b->rows; base[0] = b->bytes[0]; base[1] = b->bytes[1];
            b->rows += n; b->bytes[0] += need[0]; b->bytes[1] += need[1];
            b->members++; b->inside++;
        }
        // pack this call's rows into the batch's column: offsets continue the previous member's
        std::string my_err;
        for (int s = 0; s < 2; ++s) {
            uint32_t *off = static_cast<uint32_t *>(b->h_off[s].p) + row0;
            try {
                pack_range(col[s], 0, n, off, base[s], base[s] + need[s], static_cast<uint8_t *>(b->h_val[s].p));
            } catch (const PluginError &e) { // (a malformed view: this call fails -- below -- the batch goes on; keep its offsets monotone)
                if (my_err.empty()) my_err = e.msg;
                for (uint64_t i = 1; i <= n; ++i) off[i] = (uint32_t)(base[s] + need[s]);
            }
        }
        std::string batch_err;
        b->packed.fetch_add(1, std::memory_order_acq_rel);
        {
            std::unique_lock<std::mutex> lk(m_, std::defer_lock);
            if (leader) {
                // group commit: calls keep joining while the launch in front runs
                spin_until([&] { return !launch_in_flight_.load(std::memory_order_acquire); });
                lk.lock();
                cv_.wait(lk, [&] { return !launch_in_flight_.load(std::memory_order_acquire); });
                b->sealed = true;
                if (open_[measure] == b) open_[measure] = nullptr;
                launch_in_flight_.store(true, std::memory_order_release);
                const int members = b->members; // (final: the batch is sealed)
                const uint64_t rows = b->rows;
                lk.unlock();
                if (!spin_until([&] { return b->packed.load(std::memory_order_acquire) == members; })) {
                    lk.lock();
                    cv_.wait(lk, [&] { return b->packed.load(std::memory_order_acquire) == members; });
                    lk.unlock();
                }
                std::string err;
                try {
#ifdef STRSIM_TEST_HOOKS
                    if (test_launch) { // (the test-hooks build: a CPU stand-in for the kernel, so that the protocol runs under the sanitizers)
                        test_launch(static_cast<const uint32_t *>(b->h_off[0].p), static_cast<const uint8_t *>(b->h_val[0].p),
                                    static_cast<const uint32_t *>(b->h_off[1].p), static_cast<const uint8_t *>(b->h_val[1].p), rows,
                                    static_cast<double *>(b->h_out.p));
                        throw 0;
                    }
#endif
                    PipeLease lease(3 * 96 * rows);
                    Pipe &P = lease.set->at(0);
                    strsim_ctx_t *ctx = P.open(device);
                    if (strsim_pairs_device_small(ctx, measure, static_cast<const uint32_t *>(mapped(b->h_off[0].p)), static_cast<const uint8_t *>(mapped(b->h_val[0].p)), rows,
                                                  static_cast<const uint32_t *>(mapped(b->h_off[1].p)), static_cast<const uint8_t *>(mapped(b->h_val[1].p)), rows,
                                                  static_cast<double *>(mapped(b->h_out.p)), rows) != STRSIM_OK ||
                        strsim_ctx_synchronize(ctx) != STRSIM_OK)
                        err = strsim_last_error_message();
                } catch (const PluginError &e) {
                    err = e.msg;
#ifdef STRSIM_TEST_HOOKS
                } catch (int) { // (the stand-in ran)
#endif
                } catch (...) {
                    err = "unexpected failure in a combined launch";
                }
                lk.lock();
                b->failed = !err.empty();
                b->err = err;
                launch_in_flight_.store(false, std::memory_order_release);
                b->done.store(true, std::memory_order_release);
                ++batches_;
                combined_ += (uint64_t)b->members;
                if ((uint64_t)b->members > max_members_) max_members_ = (uint64_t)b->members;
                cv_.notify_all();
            } else {
                if (!spin_until([&] { return b->done.load(std::memory_order_acquire); })) {
                    lk.lock();
                    cv_.notify_all(); // (a leader asleep on `packed` sees this member's count)
                    cv_.wait(lk, [&] { return b->done.load(std::memory_order_acquire); });
                    lk.unlock();
                }
                lk.lock();
            }
            if (b->failed) batch_err = b->err;
        }
        if (batch_err.empty() && my_err.empty()) memcpy(out, static_cast<const double *>(b->h_out.p) + row0, n * sizeof(double));
        {
            std::lock_guard<std::mutex> lk(m_);
            if (--b->inside == 0) free_.push_back(b);
        }
        if (!my_err.empty()) fail(my_err);
        if (!batch_err.empty()) fail(batch_err);
        return true;
    }
    // out[0..3] = combined launches, calls they carried, most calls in one launch, small calls that took the ordinary path
    void stats(uint64_t *out)
    {
        std::lock_guard<std::mutex> lk(m_);
        out[0] = batches_; out[1] = combined_; out[2] = max_members_; out[3] = direct_.load(std::memory_order_relaxed);
    }
    std::atomic<uint64_t> direct_{0};
#ifdef STRSIM_TEST_HOOKS
    void (*test_launch)(const uint32_t *, const uint8_t *, const uint32_t *, const uint8_t *, uint64_t, double *) = nullptr;
#endif

  private:
    std::mutex m_;
    std::condition_variable cv_;
    ComboBatch *open_[STRSIM_NUM_MEASURES] = {};
    std::vector<ComboBatch *> free_;
    std::atomic<bool> launch_in_flight_{false};
    uint64_t batches_ = 0, combined_ = 0, max_members_ = 0;
    // spin on `pred` for up to ~200 us (a combined launch is ~30 us); false: not yet -- the caller sleeps on the condition variable
    template <class Pred> static bool spin_until(Pred pred)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                if (pred()) return true;
                __builtin_ia32_pause();
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) return pred();
        }
    }
};
// (never destroyed, like the staging pool: pinned memory and a static destructor do not mix)
Combiner &combiner() { static Combiner *c = new Combiner; return *c; }

struct CoalesceKnobs {
    bool on;
    int min_inflight;
    uint64_t rows;
    CoalesceKnobs()
    {
        const char *e = getenv("POLARS_STRSIM_COALESCE");
        on = e && atoi(e) != 0; // (opt-in: see above)
        const char *m = getenv("POLARS_STRSIM_COALESCE_MIN_INFLIGHT");
        min_inflight = m && atoi(m) >= 1 ? atoi(m) : 8;
        rows = env_rows("POLARS_STRSIM_COALESCE_ROWS", 8192);
        if (rows > COMBO_ROWS / 2) rows = COMBO_ROWS / 2;
    }
};
const CoalesceKnobs &coalesce_knobs() { static const CoalesceKnobs k; return k; }

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
