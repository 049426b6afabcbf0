// strsim_capi.cpp -- the thin kernel ABI of include/strsim_amd.h: contexts, workspace, launch, timing.
//
// Host-side counterpart of the row loop in the reference's `parallel_apply`
// (reference src/expressions/strsim.rs:41-107): shape rule, literal broadcast, row partition.
// There is no CPU compute path in this library: without a HIP device every compute call fails.
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "strsim_amd.h"
#include "strsim_internal.h"
#include "strsim_kernels.h"

namespace strsim {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error in %s: %s (%d)", what, hipGetErrorString(e), (int)e);
    if (e == hipErrorOutOfMemory) return STRSIM_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return STRSIM_ERR_NO_DEVICE;
    return STRSIM_ERR_HIP;
}

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);    \
    } while (0)

} // namespace strsim

using namespace strsim;

struct strsim_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 0;
    int stage_wg_per_cu = 5;  // resident workgroups of k_lane_stage per CU; STRSIM_STAGE_WG_PER_CU overrides (tuning knob)
    int lev_waves_per_cu = 20; // five per SIMD (96 VGPRs, 7.6 KB of LDS = six 1 280-byte granules); STRSIM_LEV_WAVES_PER_CU overrides (tuning knob)
    // per-call deferred state: a ring of status words + event triples, so calls can be enqueued
    // back to back without a host sync; the ring is drained by strsim_ctx_synchronize()
    static constexpr int RING = 32;
    // workspace (grow-only).  The "not finished yet" masks (+ backup + work list) of a call.  A call whose slow-row kernels may be
    // launched late (a one-launch call, below) needs its mask intact while younger calls run: it OWNS one of the first RING
    // buffers (the lowest free one; allocated on first use, so a caller that keeps k calls in flight touches k of them) until it is
    // retired.  Calls that enqueue all their kernels up front share the last buffer (they use it in stream order).  Nothing is ever
    // retired behind the caller's back to make room: a buffer is always free while a ring slot is.
    static constexpr int MASKBUFS = RING + 1;
    static constexpr int SHARED_MASKBUF = MASKBUFS - 1;
    unsigned long long *slowmask[MASKBUFS] = {};
    size_t slowmask_cap[MASKBUFS] = {}; // bytes
    int maskbuf_owner[MASKBUFS];        // ring slot of the pending one-launch call that owns the buffer, or -1 (set in strsim_ctx_create)
    DevStatus *status = nullptr;      // device, RING entries
    uint32_t *sched = nullptr;        // device, RING x 4 words: work-distribution counters of k_lane_stage (zero between launches)
    DevStatus *status_host = nullptr; // pinned, RING entries
    DevStatus *status_host_dev = nullptr; // the same memory as the device addresses it
    void *pin = nullptr;   // pinned staging of strsim_pairs_host's small calls (kernels work on it in place)
    void *pin_dev = nullptr;
    size_t pin_cap = 0;
    hipEvent_t ev[RING][3] = {};
    bool slot_pending[RING] = {};
    // A call whose lane kernel is EXPECTED to leave nothing behind (the last retired call did not: short ASCII columns) is
    // ONE kernel launch: k_lane_stage's last workgroup publishes lane_left and the ticket itself.  If it did leave rows, the
    // slow-row kernels are launched when the call is retired (strsim_ctx_synchronize / strsim_ctx_retire_oldest) -- like
    // the long-string pass -- and the next calls enqueue the whole chain up front again.
    bool slot_deferred[RING] = {};
    bool expect_slow = false;
    bool long_rows = false;        // ... and left more than 1/16 of its rows: frames of long strings (see k_lane_stage, TABLES)
    bool stream_ordered = true;    // strsim_ctx_set_stream_ordered(ctx, 0) opts in to one-launch calls
    uint64_t last_late_rows = 0;   // rows finished by a pass launched from synchronize / retire (deferred + long-string)
    uint64_t carry_late_rows = 0, carry_long_rows = 0; // the same of calls the library had to retire itself (the ring of status
                                   // slots wrapped): reported with the caller's next strsim_ctx_synchronize / _retire_oldest
    hipEvent_t ev_late[2] = {};    // timing of a deferred slow pass
    uint64_t enqueued_ops = 0;     // kernels + copies this context has put on its stream for pair calls (strsim_ctx_enqueued_ops)
    uint32_t slot_ticket[RING] = {}; // what the device writes into status_host[slot].ticket when the call's status block is out
    uint32_t ticket_seq = 0;
    bool slot_timed[RING] = {};
    LaunchArgs slot_args[RING] = {}; // what each pending call was launched with (for the long-string pass)
    int slot_measure[RING] = {};     // STRSIM_NUM_MEASURES = the fused all-measures call
    double *slot_outs[RING][5] = {};
    uint32_t *lev_ws = nullptr;      // scratch of k_wave_pairs<LEVENSHTEIN>: staged texts + non-ASCII fallback arrays
    size_t lev_ws_cap = 0;
    uint32_t *huge_ws = nullptr;     // workspace of the long-string pass (grow-only)
    size_t huge_ws_cap = 0;
    uint32_t *scan_ws = nullptr;     // block sums of strsim_offsets_from_lengths (SCAN_WS_WORDS, allocated on first use)
    int head = 0;
    double *qtab = nullptr;          // QTAB_N x QTAB_N quotients a / b (strsim_lane_core.h), filled at creation
    uint64_t last_wave_rows = 0;
    uint64_t last_long_rows = 0; // over the slots retired by the last synchronize
    // staging for strsim_pairs_host (grow-only device buffers)
    void *stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t stage_cap[5] = {0, 0, 0, 0, 0};
    // timing
    bool timing = false;
    double lane_ms = 0, wave_ms = 0;
    uint64_t lane_launches = 0, wave_launches = 0;
    // environment knobs, read ONCE in strsim_ctx_create (not on the launch path: a small call is ~23 us, and a host that calls
    // setenv concurrently must not race a getenv of ours per call)
    bool no_literal_path = false;      // STRSIM_NO_LITERAL_PATH (tuning / A-B knob)
    int wide_cap_per_cu = 192;         // STRSIM_WIDE_WG_PER_CU (tuning knob)
    int huge_waves_per_cu = 16;        // STRSIM_HUGE_WAVES_PER_CU (tuning knob); 8 KB of LDS per Levenshtein wave
    uint64_t host_direct_rows = 65536; // STRSIM_HOST_DIRECT_ROWS=0 switches strsim_pairs_host's in-place path off (tests do)
    // fault injection (tests only): the k-th slot retirement of this context fails with STRSIM_ERR_INTERNAL -- the one way to see
    // STRSIM_ERR_EARLIER_CALL and the "a slot is retired whatever fails" rule without breaking the device
    uint64_t fault_retire_at = 0;      // STRSIM_FAULT_RETIRE_AT (0 = never)
    uint64_t retired = 0;
};

static int ctx_set_device(strsim_ctx *c) { HIP_TRY(hipSetDevice(c->device)); return STRSIM_OK; }

static int ctx_reserve(void **p, size_t *cap, size_t bytes);

// strsim_pairs_host computes calls up to this size in place on pinned host memory (see there).  Measured through ctypes
// (bench_support/bench_small_host_calls.py), in place vs copies: 37 vs 70 us at 1..100 rows, 57 vs 95 us at 10 000,
// 92 vs 117 us at 32 768, 150 vs 171 us at 70 000, 246 vs 207 us at 100 000 -- hence the limit.
static constexpr size_t HOST_DIRECT_BYTES = (size_t)2 << 20;

// Rows with a string longer than STRSIM_WAVE_PATH_MAX_BYTES: rerun them with the scratch arrays in global memory.
static int ctx_run_huge(strsim_ctx *c, int slot, const DevStatus &st)
{
    const uint32_t cap = (st.max_len + 63u) & ~63u;
    const size_t per_wave = (size_t)HUGE_WS_WORDS(cap) * sizeof(uint32_t);
    const size_t max_waves = (size_t)c->num_cu * (size_t)c->huge_waves_per_cu;
    size_t waves = st.huge_rows < max_waves ? st.huge_rows : max_waves;
    const size_t budget = (size_t)4 << 30; // keep the workspace under 4 GiB
    if (waves * per_wave > budget) waves = budget / per_wave ? budget / per_wave : 1;
    int rc = ctx_reserve((void **)&c->huge_ws, &c->huge_ws_cap, waves * per_wave);
    if (rc) return rc;
    LaunchArgs a = c->slot_args[slot];
    a.ev_lane0 = a.ev_lane1 = a.ev_wave1 = nullptr;
    if (c->slot_measure[slot] == STRSIM_NUM_MEASURES) {
        for (int m = 0; m < STRSIM_NUM_MEASURES; ++m) {
            a.out = c->slot_outs[slot][m];
            hipError_t e = launch_huge(m, a, c->huge_ws, cap, (int)waves);
            if (e != hipSuccess) return hip_fail(e, "long-string kernel launch");
        }
    } else {
        hipError_t e = launch_huge(c->slot_measure[slot], a, c->huge_ws, cap, (int)waves);
        if (e != hipSuccess) return hip_fail(e, "long-string kernel launch");
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return STRSIM_OK;
}

// Retire one pending slot whose kernels have completed: timing, counters, the long-string pass.  A slot is retired whatever
// fails on the way (a slot left pending would be re-run by a later synchronize with buffers its caller may have freed).
// Has the device published the status block of the call in slot s?  (The ticket is the last word the publisher writes, with
// release semantics at system scope; the acquire fence pairs with it.)
static bool ctx_slot_published(const strsim_ctx *c, int s)
{
    const uint32_t seen = *reinterpret_cast<const volatile uint32_t *>(&c->status_host[s].ticket);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return seen == c->slot_ticket[s];
}

static constexpr uint32_t LANE_LEFT_UNKNOWN = 0xFFFFFFFFu; // the call's first kernel has not reported yet

// The slow-row kernels of a call that was enqueued as its lane kernel alone and left rows behind: launched now, behind
// whatever the stream holds, and waited for.
static int ctx_run_deferred(strsim_ctx *c, int s)
{
    LaunchArgs a = c->slot_args[s];
    a.ev_lane0 = a.ev_lane1 = a.ev_wave1 = nullptr;
    a.publish_host = nullptr;
    a.publish_ticket = 0u;
    const bool timed = c->slot_timed[s];
    if (timed) {
        for (int i = 0; i < 2; ++i)
            if (!c->ev_late[i]) HIP_TRY(hipEventCreateWithFlags(&c->ev_late[i], hipEventDisableSystemFence));
        HIP_TRY(hipEventRecord(c->ev_late[0], c->stream));
    }
    hipError_t e;
    if (c->slot_measure[s] == STRSIM_NUM_MEASURES) {
        const uint64_t nchunks = (a.n + 63u) >> 6;
        e = launch_slow_all_only(a, c->slot_outs[s], a.slowmask + nchunks);
    } else {
        e = launch_slow_only(c->slot_measure[s], a);
    }
    if (e != hipSuccess) return hip_fail(e, "kernel launch (slow rows of a retired call)");
    c->enqueued_ops += (c->slot_measure[s] == STRSIM_NUM_MEASURES ? 20u : 3u) + 1u; // + the status copy
    if (timed) HIP_TRY(hipEventRecord(c->ev_late[1], c->stream));
    if (++c->ticket_seq == 0u) c->ticket_seq = 1u;
    c->slot_ticket[s] = c->ticket_seq;
    c->status_host[s].ticket = 0u;
    HIP_TRY(launch_publish_status(c->status + s, c->status_host_dev + s, c->slot_ticket[s], c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (timed) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_late[0], c->ev_late[1]));
        c->wave_ms += ms;
        c->wave_launches++;
    }
    return STRSIM_OK;
}

static int ctx_retire_slot(strsim_ctx *c, int s)
{
    int rc = STRSIM_OK;
    c->slot_pending[s] = false;
    struct Release { // the slot's mask buffer is free again whichever way this function is left
        strsim_ctx *c; int s;
        ~Release() { for (int b = 0; b < strsim_ctx::MASKBUFS; ++b) if (c->maskbuf_owner[b] == s) c->maskbuf_owner[b] = -1; }
    } release{c, s};
    if (!ctx_slot_published(c, s)) { // (cannot happen behind a stream synchronise; strsim_ctx_retire_oldest checks before it gets here)
        c->slot_timed[s] = c->slot_deferred[s] = false;
        set_error("internal: the status block of a completed call was not published (slot %d)", s);
        return STRSIM_ERR_INTERNAL;
    }
    if (c->fault_retire_at && ++c->retired == c->fault_retire_at) { // (tests only: STRSIM_FAULT_RETIRE_AT)
        c->slot_timed[s] = c->slot_deferred[s] = false;
        set_error("fault injected at retirement %llu of this context (STRSIM_FAULT_RETIRE_AT)", (unsigned long long)c->retired);
        return STRSIM_ERR_INTERNAL;
    }
    const uint32_t left = *reinterpret_cast<const volatile uint32_t *>(&c->status_host[s].lane_left);
    if (left != LANE_LEFT_UNKNOWN) { // what the next call on this context is enqueued for
        c->expect_slow = left != 0u;
        c->long_rows = (uint64_t)left * 16u > c->slot_args[s].n;
    }
    if (c->slot_timed[s]) {
        float a = 0, b = 0;
        // (the ticket says the kernels are done; the event right behind them may be a moment later)
        hipError_t e = hipEventSynchronize(c->ev[s][c->slot_deferred[s] ? 1 : 2]);
        if (e == hipSuccess) e = hipEventElapsedTime(&a, c->ev[s][0], c->ev[s][1]);
        if (e == hipSuccess && !c->slot_deferred[s]) e = hipEventElapsedTime(&b, c->ev[s][1], c->ev[s][2]);
        if (e == hipSuccess) {
            c->lane_ms += a; c->wave_ms += b;
            c->lane_launches++;
            if (!c->slot_deferred[s]) c->wave_launches++; // (a deferred call's slow-row kernels: counted where they are launched)
        } else {
            rc = hip_fail(e, "hipEventElapsedTime");
        }
    }
    if (c->slot_deferred[s] && left != 0u && rc == STRSIM_OK) {
        rc = ctx_run_deferred(c, s);
        if (rc == STRSIM_OK && !ctx_slot_published(c, s)) {
            set_error("internal: the status block of a deferred pass was not published (slot %d)", s);
            rc = STRSIM_ERR_INTERNAL;
        }
        c->last_late_rows += left;
    }
    c->slot_timed[s] = c->slot_deferred[s] = false;
    if (rc != STRSIM_OK) return rc;
    const DevStatus &st = c->status_host[s];
    c->last_wave_rows = st.wave_rows;
    c->last_long_rows += st.huge_rows;
    c->last_late_rows += st.huge_rows;
    if (st.huge_rows != 0) rc = ctx_run_huge(c, s, st);
    return rc;
}

// Retire every pending slot (the stream must already be synchronised); the first error is what the caller gets.
static int ctx_drain(strsim_ctx *c)
{
    int rc = STRSIM_OK;
    c->last_long_rows = 0;
    c->last_late_rows = 0;
    for (int k = 0; k < strsim_ctx::RING; ++k) {
        const int s = (c->head + k) % strsim_ctx::RING; // oldest first
        if (!c->slot_pending[s]) continue;
        if (rc == STRSIM_OK) {
            rc = ctx_retire_slot(c, s);
        } else {
            c->slot_pending[s] = c->slot_timed[s] = c->slot_deferred[s] = false;
            for (int b = 0; b < strsim_ctx::MASKBUFS; ++b) if (c->maskbuf_owner[b] == s) c->maskbuf_owner[b] = -1;
        }
    }
    return rc;
}

static int ctx_reserve(void **p, size_t *cap, size_t bytes)
{
    using namespace strsim;
    if (bytes <= *cap) return STRSIM_OK;
    if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; *cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return STRSIM_OK;
}

extern "C" {

uint32_t strsim_abi_version(void) { return STRSIM_ABI_VERSION; }

const char *strsim_last_error_message(void) { return g_last_error.c_str(); }

int strsim_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n < 0 ? 0 : n;
}

int strsim_ctx_create(int device, void *hip_stream, strsim_ctx_t **out_ctx)
{
    if (!out_ctx) { set_error("strsim_ctx_create: out_ctx is NULL"); return STRSIM_ERR_ARG; }
    *out_ctx = nullptr;
    const int ndev = strsim_device_count();
    if (ndev == 0) {
        set_error("no HIP device is available; polars-strsim_amd has no CPU fallback");
        return STRSIM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("strsim_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
        return STRSIM_ERR_ARG;
    }
    strsim_ctx *c = new (std::nothrow) strsim_ctx();
    if (!c) { set_error("out of host memory"); return STRSIM_ERR_OOM; }
    c->device = device;
    for (int b = 0; b < strsim_ctx::MASKBUFS; ++b) c->maskbuf_owner[b] = -1;
    int rc = ctx_set_device(c);
    if (rc) { delete c; return rc; }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipGetDeviceProperties"); }
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *env = getenv("STRSIM_STAGE_WG_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 64) c->stage_wg_per_cu = v;
    }
    {
        const int resident = strsim::wave_lev_resident_per_cu(); // never more than fit: the grid is persistent
        if (resident >= 1 && resident < c->lev_waves_per_cu) c->lev_waves_per_cu = resident;
    }
    if (const char *env = getenv("STRSIM_LEV_WAVES_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 256) c->lev_waves_per_cu = v;
    }
    c->no_literal_path = getenv("STRSIM_NO_LITERAL_PATH") != nullptr;
    if (const char *env = getenv("STRSIM_WIDE_WG_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 4096) c->wide_cap_per_cu = v;
    }
    if (const char *env = getenv("STRSIM_HUGE_WAVES_PER_CU")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 32) c->huge_waves_per_cu = v;
    }
    if (const char *env = getenv("STRSIM_HOST_DIRECT_ROWS")) c->host_direct_rows = (uint64_t)strtoull(env, nullptr, 10);
    if (const char *env = getenv("STRSIM_FAULT_RETIRE_AT")) c->fault_retire_at = (uint64_t)strtoull(env, nullptr, 10);
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreateWithFlags"); }
        c->own_stream = true;
    }
    e = hipMalloc((void **)&c->status, sizeof(DevStatus) * strsim_ctx::RING);
    if (e == hipSuccess) e = hipMalloc((void **)&c->sched, sizeof(uint32_t) * 4 * strsim_ctx::RING);
    // (cleared on the context's own stream: a memset on the null stream would bring the legacy default stream into a process
    //  that may initialise torch's runtime later, tests/test_hip_runtime_sharing.py)
    if (e == hipSuccess) e = hipMemsetAsync(c->sched, 0, sizeof(uint32_t) * 4 * strsim_ctx::RING, c->stream);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&c->status_host, sizeof(DevStatus) * strsim_ctx::RING, hipHostMallocDefault);
    if (e != hipSuccess) { strsim_ctx_destroy(c); return hip_fail(e, "status allocation"); }
    memset(c->status_host, 0, sizeof(DevStatus) * strsim_ctx::RING);
    e = hipHostGetDevicePointer((void **)&c->status_host_dev, c->status_host, 0);
    if (e != hipSuccess) { strsim_ctx_destroy(c); return hip_fail(e, "hipHostGetDevicePointer"); }
    {   // integer quotients for the epilogues of the one-pair-per-lane kernels: the host's IEEE division is the device's
        static double q[QTAB_N * QTAB_N];
        for (int a = 0; a < QTAB_N; ++a)
            for (int b = 0; b < QTAB_N; ++b) q[a * QTAB_N + b] = b ? (double)a / (double)b : 0.0;
        e = hipMalloc((void **)&c->qtab, sizeof(q));
        if (e == hipSuccess) e = hipMemcpy(c->qtab, q, sizeof(q), hipMemcpyHostToDevice);
        if (e != hipSuccess) { strsim_ctx_destroy(c); return hip_fail(e, "quotient table"); }
    }
    *out_ctx = c;
    return STRSIM_OK;
}

void strsim_ctx_destroy(strsim_ctx_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int s = 0; s < strsim_ctx::RING; ++s)
        for (int i = 0; i < 3; ++i)
            if (c->ev[s][i]) (void)hipEventDestroy(c->ev[s][i]);
    for (int i = 0; i < 5; ++i) if (c->stage[i]) (void)hipFree(c->stage[i]);
    for (int i = 0; i < strsim_ctx::MASKBUFS; ++i) if (c->slowmask[i]) (void)hipFree(c->slowmask[i]);
    for (int i = 0; i < 2; ++i) if (c->ev_late[i]) (void)hipEventDestroy(c->ev_late[i]);
    if (c->qtab) (void)hipFree(c->qtab);
    if (c->sched) (void)hipFree(c->sched);
    if (c->huge_ws) (void)hipFree(c->huge_ws);
    if (c->scan_ws) (void)hipFree(c->scan_ws);
    if (c->lev_ws) (void)hipFree(c->lev_ws);
    if (c->status) (void)hipFree(c->status);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void *strsim_ctx_stream(strsim_ctx_t *c) { return c ? (void *)c->stream : nullptr; }

void strsim_split_offsets(uint64_t len, uint64_t n, uint64_t *out)
{
    // reference split_offsets, strsim.rs:21-39: len/n rows each, the remainder goes to the last part
    if (n == 0 || !out) return;
    if (n == 1) { out[0] = 0; out[1] = len; return; }
    const uint64_t chunk = len / n;
    for (uint64_t p = 0; p < n; ++p) {
        out[2 * p] = p * chunk;
        out[2 * p + 1] = (p == n - 1) ? len - p * chunk : chunk;
    }
}

static int pairs_device_impl(strsim_ctx_t *c, int measure, const uint32_t *a_off, const uint8_t *a_val, uint64_t a_rows,
                             const uint32_t *b_off, const uint8_t *b_val, uint64_t b_rows, double *const *outs,
                             uint64_t out_rows, bool eager = false)
{
    const bool all = measure == STRSIM_NUM_MEASURES;
    if (!c) { set_error("strsim_pairs_device: ctx is NULL"); return STRSIM_ERR_ARG; }
    if (measure < 0 || measure > STRSIM_NUM_MEASURES) {
        set_error("strsim_pairs_device: unknown measure %d", measure);
        return STRSIM_ERR_ARG;
    }
    // shape rule of parallel_apply, strsim.rs:48-52
    if (a_rows != b_rows && a_rows != 1 && b_rows != 1) {
        set_error("Inputs must have the same length, or one of them must be a Utf8 literal.");
        return STRSIM_ERR_SHAPE;
    }
    const uint64_t n = (a_rows == 1) ? b_rows : a_rows;
    if (out_rows != n) {
        set_error("strsim_pairs_device: out_rows=%llu but the inputs produce %llu rows", (unsigned long long)out_rows,
                  (unsigned long long)n);
        return STRSIM_ERR_ARG;
    }
    if (n == 0) return STRSIM_OK;
    if (n > 0xFFFFFFFFull) { // row and chunk indices are 32-bit inside the kernels (as are the offsets themselves)
        set_error("strsim_pairs_device: %llu rows in one call; split the column (at most 2^32 - 1 rows per call)", (unsigned long long)n);
        return STRSIM_ERR_ARG;
    }
    if (!a_off || !b_off || !outs) { set_error("strsim_pairs_device: NULL buffer"); return STRSIM_ERR_ARG; }
    for (int q = 0; q < (all ? STRSIM_NUM_MEASURES : 1); ++q)
        if (!outs[q]) { set_error("strsim_pairs_device: NULL output buffer"); return STRSIM_ERR_ARG; }
    int rc = ctx_set_device(c);
    if (rc) return rc;
    // the ring slot about to be reused must have been retired
    const int slot = c->head;
    if (c->slot_pending[slot]) {
        // the ring of status slots has wrapped (RING calls in flight without a retire): everything pending is retired here, and
        // what those calls finished late is kept for the caller's next strsim_ctx_synchronize / strsim_ctx_retire_oldest
        HIP_TRY(hipStreamSynchronize(c->stream));
        rc = ctx_drain(c);
        c->carry_late_rows += c->last_late_rows;
        c->carry_long_rows += c->last_long_rows;
        if (rc) { //  the failure belongs to an earlier call: say so, with a code of its own -- this call was not enqueued
            const std::string why = strsim_last_error_message();
            set_error("strsim_pairs_device: retiring earlier pending calls at the wrap of the ring failed (this call was not enqueued): %s", why.c_str());
            return STRSIM_ERR_EARLIER_CALL;
        }
    }
    const uint64_t nchunks = (n + 63) >> 6;
    // One launch (the lane kernel alone, the rest at retirement if it turns out to be needed) when the caller has opted in and the
    // context's last retired call left nothing behind its lane kernel.  Such a call owns a mask buffer until it is retired (there
    // is always a free one: as many as ring slots).
    bool defer = !eager && !c->expect_slow && !c->stream_ordered;
    int mb = strsim_ctx::SHARED_MASKBUF;
    if (defer) {
        defer = false;
        for (int b = 0; b < strsim_ctx::SHARED_MASKBUF; ++b)
            if (c->maskbuf_owner[b] < 0) { mb = b; defer = true; break; }
    }
    // mask + backup (five-measure call) + the work list of k_lane_utf8 (one u32 per chunk)
    {
        void *p = c->slowmask[mb];
        rc = ctx_reserve(&p, &c->slowmask_cap[mb], 2 * nchunks * sizeof(unsigned long long) + nchunks * sizeof(uint32_t));
        c->slowmask[mb] = static_cast<unsigned long long *>(p);
        if (rc) return rc;
    }
    // (the status block of the slot is cleared by the first kernel of the call)
    if (++c->ticket_seq == 0u) c->ticket_seq = 1u;
    c->slot_ticket[slot] = c->ticket_seq;
    memset(&c->status_host[slot], 0, sizeof(DevStatus)); // (the slot is not pending: nothing on the device writes this block now)
    c->status_host[slot].lane_left = LANE_LEFT_UNKNOWN;

    LaunchArgs la;
    la.offA = a_off; la.valA = a_val; la.rowsA = a_rows;
    la.offB = b_off; la.valB = b_val; la.rowsB = b_rows;
    la.out = outs[0]; la.n = n;
    la.slowmask = c->slowmask[mb]; la.status = c->status + slot; la.stream = c->stream;
    la.sched = c->sched + 4 * slot;
    la.publish_host = nullptr;
    la.publish_ticket = 0u;
    la.worklist = reinterpret_cast<uint32_t *>(c->slowmask[mb] + 2 * nchunks);
    la.qtab = c->qtab;
    la.stage_grid = c->num_cu * c->stage_wg_per_cu;
    la.no_literal_path = c->no_literal_path;
    la.long_rows = c->long_rows;
    la.wide_grid = c->num_cu * 3; // resident (LDS)
    la.wide_grid_cap = c->num_cu * c->wide_cap_per_cu;
    la.wave_grid = c->num_cu * 16;                       // k_wave_pairs, other measures: 4 waves per SIMD
    la.wave_grid_lev = c->num_cu * c->lev_waves_per_cu;
    {   // per-wave global scratch of k_wave_pairs (scalar-value arrays; Levenshtein: text arenas as well)
        const size_t waves = (size_t)std::max(la.wave_grid, la.wave_grid_lev);
        rc = ctx_reserve((void **)&c->lev_ws, &c->lev_ws_cap, waves * LEV_WS_WORDS * sizeof(uint32_t));
        if (rc) return rc;
        la.lev_ws = c->lev_ws;
    }
    la.ev_lane0 = la.ev_lane1 = la.ev_wave1 = nullptr;
    if (c->timing) {
        for (int i = 0; i < 3; ++i)
            // timing only: no system-scope fence when the event completes (a default event costs ~5 us of GPU time
            // between two kernels, three of them were 1/3 of a 1 M-row call)
            if (!c->ev[slot][i]) HIP_TRY(hipEventCreateWithFlags(&c->ev[slot][i], hipEventDisableSystemFence));
        la.ev_lane0 = c->ev[slot][0]; la.ev_lane1 = c->ev[slot][1]; la.ev_wave1 = c->ev[slot][2];
    }
    if (eager && !all && !c->timing) {
        // Small call the caller is going to wait for anyway: launch the one-pair-per-lane kernel alone, let its last
        // workgroup report how many rows it left (host-mapped status word) and wait for it.  Usually that is none --
        // short ASCII strings -- and the call is done after ONE kernel launch instead of five (the three slow-row kernels
        // and the status copy cost ~4 us each even when they find nothing to do).
        la.publish_host = c->status_host_dev + slot;
        hipError_t e0 = launch_lane_only(measure, la);
        if (e0 != hipSuccess) return hip_fail(e0, "kernel launch");
        c->enqueued_ops += (uint64_t)lane_kernel_launches(measure, la);
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint32_t left = *reinterpret_cast<volatile uint32_t *>(&c->status_host[slot].lane_left);
        if (left == 0u) {
            // nothing pending FOR THIS CALL: strsim_ctx_synchronize() has nothing to retire for it.  The last_* counters describe
            // the last retirement; they are only reset when no other call of this context is still pending (a caller
            // with one-launch calls in flight that copied results out early still has to see their late rows)
            bool others = false;
            for (int s2 = 0; s2 < strsim_ctx::RING; ++s2) others = others || c->slot_pending[s2];
            if (!others) {
                c->last_wave_rows = 0;
                c->last_long_rows = 0;
                c->last_late_rows = 0;
            }
            return STRSIM_OK;
        }
        la.publish_host = nullptr;
        e0 = launch_slow_only(measure, la);
        if (e0 != hipSuccess) return hip_fail(e0, "kernel launch");
        c->enqueued_ops += 4u;
        c->slot_timed[slot] = false;
        c->slot_deferred[slot] = false;
        c->slot_args[slot] = la;
        c->slot_measure[slot] = measure;
        for (int q = 0; q < STRSIM_NUM_MEASURES; ++q) c->slot_outs[slot][q] = nullptr;
        HIP_TRY(launch_publish_status(c->status + slot, c->status_host_dev + slot, c->slot_ticket[slot], c->stream));
        c->slot_pending[slot] = true;
        c->head = (slot + 1) % strsim_ctx::RING;
        return STRSIM_OK;
    }
    la.publish_host = c->status_host_dev + slot; // lane_left reaches the host either way: it sets expect_slow
    hipError_t e;
    if (defer) {
        la.publish_ticket = c->slot_ticket[slot];
        e = all ? launch_lane_all_only(la, outs) : launch_lane_only(measure, la);
    } else {
        e = all ? launch_pairs_all(la, outs, c->slowmask[mb] + nchunks) : launch_pairs(measure, la);
    }
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    // lane kernel (+ k_publish_lit behind k_lane_lit) | + three slow-row kernels (x 5, + 5 mask copies) + status copy
    c->enqueued_ops += (uint64_t)lane_kernel_launches(measure, la) + (defer ? 0u : (all ? 21u : 4u));
    c->slot_timed[slot] = c->timing;
    c->slot_deferred[slot] = defer;
    if (defer) c->maskbuf_owner[mb] = slot;
    c->slot_args[slot] = la;
    c->slot_measure[slot] = measure;
    for (int q = 0; q < STRSIM_NUM_MEASURES; ++q) c->slot_outs[slot][q] = all ? outs[q] : nullptr;
    if (!defer) HIP_TRY(launch_publish_status(c->status + slot, c->status_host_dev + slot, c->slot_ticket[slot], c->stream));
    c->slot_pending[slot] = true;
    c->head = (slot + 1) % strsim_ctx::RING;
    return STRSIM_OK;
}

int strsim_pairs_device(strsim_ctx_t *c, int measure, const uint32_t *a_off, const uint8_t *a_val, uint64_t a_rows,
                        const uint32_t *b_off, const uint8_t *b_val, uint64_t b_rows, double *out, uint64_t out_rows)
{
    if (measure == STRSIM_NUM_MEASURES) { set_error("strsim_pairs_device: unknown measure %d", measure); return STRSIM_ERR_ARG; }
    double *outs[1] = {out};
    return pairs_device_impl(c, measure, a_off, a_val, a_rows, b_off, b_val, b_rows, out ? outs : nullptr, out_rows);
}

int strsim_pairs_device_small(strsim_ctx_t *c, int measure, const uint32_t *a_off, const uint8_t *a_val, uint64_t a_rows,
                              const uint32_t *b_off, const uint8_t *b_val, uint64_t b_rows, double *out, uint64_t out_rows)
{
    if (measure == STRSIM_NUM_MEASURES) { set_error("strsim_pairs_device_small: unknown measure %d", measure); return STRSIM_ERR_ARG; }
    double *outs[1] = {out};
    return pairs_device_impl(c, measure, a_off, a_val, a_rows, b_off, b_val, b_rows, out ? outs : nullptr, out_rows, true);
}

int strsim_pairs_device_all(strsim_ctx_t *c, const uint32_t *a_off, const uint8_t *a_val, uint64_t a_rows,
                            const uint32_t *b_off, const uint8_t *b_val, uint64_t b_rows, double *const outs[5],
                            uint64_t out_rows)
{
    return pairs_device_impl(c, STRSIM_NUM_MEASURES, a_off, a_val, a_rows, b_off, b_val, b_rows, outs, out_rows);
}

int strsim_ctx_synchronize(strsim_ctx_t *c)
{
    if (!c) { set_error("strsim_ctx_synchronize: ctx is NULL"); return STRSIM_ERR_ARG; }
    int rc = ctx_set_device(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = ctx_drain(c);
    c->last_late_rows += c->carry_late_rows;
    c->last_long_rows += c->carry_long_rows;
    c->carry_late_rows = c->carry_long_rows = 0;
    return rc;
}

int strsim_internal_ctx_device(strsim_ctx *c) { return c ? c->device : 0; }

int strsim_internal_scan_workspace(strsim_ctx *c, uint32_t **p)
{
    int rc = ctx_set_device(c);
    if (rc) return rc;
    if (!c->scan_ws) HIP_TRY(hipMalloc((void **)&c->scan_ws, strsim::SCAN_WS_WORDS * sizeof(uint32_t)));
    *p = c->scan_ws;
    return STRSIM_OK;
}

int strsim_ctx_retire_oldest(strsim_ctx_t *c)
{
    if (!c) { set_error("strsim_ctx_retire_oldest: ctx is NULL"); return STRSIM_ERR_ARG; }
    int rc = ctx_set_device(c);
    if (rc) return rc;
    auto report_carry = [c] { // what calls retired by the library itself finished late goes out with this retirement
        c->last_long_rows = c->carry_long_rows;
        c->last_late_rows = c->carry_late_rows;
        c->carry_late_rows = c->carry_long_rows = 0;
    };
    for (int k = 0; k < strsim_ctx::RING; ++k) {
        const int s = (c->head + k) % strsim_ctx::RING; // oldest first
        if (!c->slot_pending[s]) continue;
        if (!ctx_slot_published(c, s)) {
            // the caller's promise does not hold (e.g. its event was recorded on another stream than strsim_ctx_stream()):
            // say so instead of reading a status block the device has not written; the call stays pending
            set_error("strsim_ctx_retire_oldest: the oldest call has not completed (wait on an event recorded on "
                      "strsim_ctx_stream() behind it, or call strsim_ctx_synchronize())");
            return STRSIM_ERR_ARG;
        }
        report_carry();
        return ctx_retire_slot(c, s);
    }
    report_carry();
    return STRSIM_OK; // nothing pending (e.g. a small call that completed inside strsim_pairs_device_small)
}

int strsim_pairs_host(strsim_ctx_t *c, int measure, const uint32_t *a_off, const uint8_t *a_val, uint64_t a_rows,
                      const uint32_t *b_off, const uint8_t *b_val, uint64_t b_rows, double *out, uint64_t out_rows)
{
    if (!c) { set_error("strsim_pairs_host: ctx is NULL"); return STRSIM_ERR_ARG; }
    if (a_rows != b_rows && a_rows != 1 && b_rows != 1) {
        set_error("Inputs must have the same length, or one of them must be a Utf8 literal.");
        return STRSIM_ERR_SHAPE;
    }
    const uint64_t n = (a_rows == 1) ? b_rows : a_rows;
    if (out_rows != n) { set_error("strsim_pairs_host: out_rows mismatch"); return STRSIM_ERR_ARG; }
    if (n == 0) return STRSIM_OK;
    if (!a_off || !b_off || !out) { set_error("strsim_pairs_host: NULL buffer"); return STRSIM_ERR_ARG; }
    int rc = ctx_set_device(c);
    if (rc) return rc;
    const size_t abytes = (size_t)a_off[a_rows] - a_off[0], bbytes = (size_t)b_off[b_rows] - b_off[0];
    if (n <= c->host_direct_rows && abytes <= HOST_DIRECT_BYTES && bbytes <= HOST_DIRECT_BYTES) {
        // Small call: gather the five buffers in one pinned block and let the kernels work on it in place through the
        // device's mapping of host memory -- no copy engine, whose hand-overs (four H2D, one D2H) are most of a small call.
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_aoff = 0, o_aval = o_aoff + up((a_rows + 1) * 4), o_boff = o_aval + up(abytes + 64),
                     o_bval = o_boff + up((b_rows + 1) * 4), o_out = o_bval + up(bbytes + 64), total = o_out + up(n * 8);
        if (total > c->pin_cap) {
            if (c->pin) { HIP_TRY(hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
            const size_t want = total + total / 4 + 4096;
            HIP_TRY(hipHostMalloc(&c->pin, want, hipHostMallocDefault));
            HIP_TRY(hipHostGetDevicePointer(&c->pin_dev, c->pin, 0));
            c->pin_cap = want;
        }
        uint8_t *h = static_cast<uint8_t *>(c->pin);
        const uint8_t *d = static_cast<const uint8_t *>(c->pin_dev);
        // the offsets are rebased to 0 on the way in, so that the kernels get the value blocks' real addresses (with the
        // caller's base kept and a pointer moved back by it, any kernel access at "offset 0" -- skipped rows of a block,
        // alignment of a staging copy -- would land a_off[0] bytes in front of the pinned block: a slice of a big column)
        {
            uint32_t *ha = reinterpret_cast<uint32_t *>(h + o_aoff), *hb = reinterpret_cast<uint32_t *>(h + o_boff);
            const uint32_t a0 = a_off[0], b0 = b_off[0];
            for (uint64_t i = 0; i <= a_rows; ++i) ha[i] = a_off[i] - a0;
            for (uint64_t i = 0; i <= b_rows; ++i) hb[i] = b_off[i] - b0;
        }
        if (abytes) memcpy(h + o_aval, a_val + a_off[0], abytes);
        if (bbytes) memcpy(h + o_bval, b_val + b_off[0], bbytes);
        rc = strsim_pairs_device_small(c, measure, (const uint32_t *)(d + o_aoff), d + o_aval, a_rows,
                                       (const uint32_t *)(d + o_boff), d + o_bval, b_rows,
                                       (double *)(const_cast<uint8_t *>(d) + o_out), n);
        if (rc) return rc;
        rc = strsim_ctx_synchronize(c); // also runs the long-string pass, which writes into the same block
        if (rc) return rc;
        memcpy(out, h + o_out, n * 8);
        return STRSIM_OK;
    }
    // (the copy path keeps the caller's offset base and uploads the values from byte 0 of the caller's buffer, so every
    //  offset the kernels can form, 0 included, is inside the device copy)
    const size_t need[5] = {(a_rows + 1) * 4, (size_t)a_off[a_rows] + 1, (b_rows + 1) * 4, (size_t)b_off[b_rows] + 1, n * 8};
    for (int i = 0; i < 5; ++i) {
        rc = ctx_reserve(&c->stage[i], &c->stage_cap[i], need[i]);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(c->stage[0], a_off, (a_rows + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (a_off[a_rows]) HIP_TRY(hipMemcpyAsync(c->stage[1], a_val, a_off[a_rows], hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->stage[2], b_off, (b_rows + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (b_off[b_rows]) HIP_TRY(hipMemcpyAsync(c->stage[3], b_val, b_off[b_rows], hipMemcpyHostToDevice, c->stream));
    rc = strsim_pairs_device(c, measure, (const uint32_t *)c->stage[0], (const uint8_t *)c->stage[1], a_rows,
                             (const uint32_t *)c->stage[2], (const uint8_t *)c->stage[3], b_rows, (double *)c->stage[4], n);
    if (rc) return rc;
    rc = strsim_ctx_synchronize(c); // also runs the long-string pass, which writes into the staged output
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->stage[4], n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return STRSIM_OK;
}

int strsim_ctx_timing_enable(strsim_ctx_t *c, int enable)
{
    if (!c) { set_error("strsim_ctx_timing_enable: ctx is NULL"); return STRSIM_ERR_ARG; }
    int rc = strsim_ctx_synchronize(c);
    if (rc) return rc;
    c->timing = enable != 0;
    return STRSIM_OK;
}

int strsim_ctx_timing_read(strsim_ctx_t *c, double *lane_ms, uint64_t *lane_launches, double *wave_ms,
                           uint64_t *wave_launches)
{
    if (!c) { set_error("strsim_ctx_timing_read: ctx is NULL"); return STRSIM_ERR_ARG; }
    int rc = strsim_ctx_synchronize(c);
    if (rc) return rc;
    if (lane_ms) *lane_ms = c->lane_ms;
    if (lane_launches) *lane_launches = c->lane_launches;
    if (wave_ms) *wave_ms = c->wave_ms;
    if (wave_launches) *wave_launches = c->wave_launches;
    c->lane_ms = c->wave_ms = 0;
    c->lane_launches = c->wave_launches = 0;
    return STRSIM_OK;
}

uint64_t strsim_ctx_last_wave_rows(strsim_ctx_t *c) { return c ? c->last_wave_rows : 0; }
uint64_t strsim_ctx_last_long_rows(strsim_ctx_t *c) { return c ? c->last_long_rows : 0; }
uint64_t strsim_ctx_last_late_rows(strsim_ctx_t *c) { return c ? c->last_late_rows : 0; }
uint64_t strsim_ctx_enqueued_ops(strsim_ctx_t *c) { return c ? c->enqueued_ops : 0; }

int strsim_ctx_set_stream_ordered(strsim_ctx_t *c, int enable)
{
    if (!c) { set_error("strsim_ctx_set_stream_ordered: ctx is NULL"); return STRSIM_ERR_ARG; }
    c->stream_ordered = enable != 0;
    return STRSIM_OK;
}

int strsim_ctx_get_stream_ordered(strsim_ctx_t *c) { return c ? (c->stream_ordered ? 1 : 0) : 1; }

} // extern "C"
