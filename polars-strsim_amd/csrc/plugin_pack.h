// plugin_pack.h -- the host's packing machinery: the per-thread fork-join pool, pinned staging slots and device pipelines, the device list,
// the CPU quota and the helper-thread budget of engine-parallel calls, and the slice packers that fill a slot.
// Included by polars_plugin.cpp inside its anonymous namespace, in this order ([r5] split out of polars_plugin.cpp along its seams).
// [r6] the pipelines are leased per call from one process-wide pool under a staging budget (StagingPool) instead of living with the thread.
// Reference: parallel_apply, /root/reference/src/expressions/strsim.rs:41-107.
#pragma once

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

// staging memory alive in this process right now, over every pipeline set (polars_strsim_staging_stats reports them)
std::atomic<uint64_t> g_live_pinned{0}, g_live_device{0};

// grow-only buffer: pinned host memory or device memory (given back as a whole when its pipeline set is released: StagingPool)
struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    bool device = false;
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
#ifdef STRSIM_TEST_HOOKS // (the test-hooks build runs the pool's and the combiner's logic without a device: plain host memory)
        p = malloc(want);
        if (!p) fail("out of host memory");
#else
        if (device) HIP_OR_FAIL(hipMalloc(&p, want)); else HIP_OR_FAIL(hipHostMalloc(&p, want, hipHostMallocDefault));
#endif
        cap = want;
        (device ? g_live_device : g_live_pinned).fetch_add(want, std::memory_order_relaxed);
    }
    void release()
    {
        if (p) {
#ifdef STRSIM_TEST_HOOKS
            free(p);
#else
            if (device) (void)hipFree(p); else (void)hipHostFree(p);
#endif
            (device ? g_live_device : g_live_pinned).fetch_sub(cap, std::memory_order_relaxed);
        }
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
    Buf h_bad; // pinned, one counter: view slots the device refused to copy (strsim_column_from_views_bounded); checked when the slice is finished
    uint64_t long_span[2] = {0, 0}; // bytes of h_long[s] to ship
    uint64_t r0 = 0, rows = 0;
    uint64_t bytes[2] = {0, 0};
    uint64_t live() const
    {
        uint64_t b = h_out.cap + d_out.cap + h_bad.cap;
        for (int i = 0; i < 2; ++i)
            b += h_off[i].cap + h_val[i].cap + d_off[i].cap + d_val[i].cap + h_len[i].cap + d_len[i].cap + d_land[i].cap + h_views[i].cap +
                 d_views[i].cap + h_long[i].cap + d_long[i].cap;
        return b;
    }
    bool direct = false; // this slice was computed in place on the pinned staging (see run(): launch)
    hipEvent_t ev_kernels = nullptr, ev_results = nullptr; // behind the slice's kernels (compute stream) / its D2H (copy stream)
    Slot() { for (int i = 0; i < 2; ++i) { d_off[i].device = true; d_val[i].device = true; d_len[i].device = true; d_land[i].device = true; d_views[i].device = true; d_long[i].device = true; } d_out.device = true; }
    void release()
    {
        for (int i = 0; i < 2; ++i) { h_off[i].release(); h_val[i].release(); d_off[i].release(); d_val[i].release(); h_len[i].release(); d_len[i].release(); d_land[i].release();
                                      h_views[i].release(); d_views[i].release(); h_long[i].release(); d_long[i].release(); }
        h_out.release(); d_out.release(); h_bad.release();
        if (ev_kernels) (void)hipEventDestroy(ev_kernels);
        if (ev_results) (void)hipEventDestroy(ev_results);
        ev_kernels = ev_results = nullptr;
    }
};

// ---- device pipelines, leased per call from one process-wide pool (Polars may call from many of its threads at once) -----
// One pipeline = one device context + its stream, a copy stream for the results, three slots and the literal's buffers.  A
// pipeline SET holds one pipeline per entry of the device list (an ordinal may repeat: two pipelines on one GPU).
struct Pipe {
    strsim_ctx_t *ctx = nullptr;
    int device = 0;
    Slot slot[3];             // two slices in flight on the GPU + the one being packed
    hipStream_t d2h = nullptr; // results travel back on their own stream: D2H of slice k runs beside H2D of slice k+1
    Buf lit_off, lit_val;     // device copy of a literal side
    Buf lit_h_off, lit_h_val; // its pinned host staging
    uint64_t live() const { return slot[0].live() + slot[1].live() + slot[2].live() + lit_off.cap + lit_val.cap + lit_h_off.cap + lit_h_val.cap; }
    void close()
    {
        if (ctx) {
            (void)hipSetDevice(device);
            if (d2h) (void)hipStreamDestroy(d2h);
            d2h = nullptr;
        }
        for (auto &s : slot) s.release();
        lit_off.release(); lit_val.release();
        lit_h_off.release(); lit_h_val.release();
        if (ctx) strsim_ctx_destroy(ctx);
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
struct PipeSet {
    std::vector<Pipe *> p;
    bool in_use = false;
    uint64_t reserved = 0;  // in use: max(what the call that holds the set asked for, what the set held when it was leased) -- written under
                            // the pool's lock; the set itself belongs to its holder until it comes back and is not looked at by anybody else
    uint64_t last_used = 0; // the pool's clock when the set came back
    Pipe &at(size_t i) { while (p.size() <= i) p.push_back(new Pipe); return *p[i]; }
    uint64_t live() const { uint64_t b = 0; for (const Pipe *q : p) b += q->live(); return b; }
    bool has_context() const { for (const Pipe *q : p) if (q->ctx) return true; return false; }
    void close()
    {   // (releasing a pipeline selects its device: put the calling thread's own choice back)
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        for (Pipe *q : p) q->close();
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    ~PipeSet() { close(); for (Pipe *q : p) delete q; }
};

// The staging of ALL calling threads against ONE budget (round 5 kept a set per calling thread, grow-only, for
// as long as the thread lived -- an engine pool of 64-256 threads that each saw one large morsel pinned tens of GB, where the
// reference's per-call scratch is three small vectors, strsim.rs:78-84, :109-123).  A call LEASES a set for its duration:
//   * the most recently used idle set if there is one (its buffers are sized and warm), else a new one;
//   * if the staging in use + this call's estimate would pass the budget, idle sets are released first (least recently used
//     first), and if that is not enough the call WAITS for a running call to return -- unless no call is running: one call always
//     proceeds, whatever it needs (the budget bounds what many calls pin together, never what one column requires);
//   * when a set comes back and the process is over budget, idle sets are released (least recently used first) until it is not.
// POLARS_STRSIM_STAGING_BUDGET_MB: pinned + device staging bytes, default 4096 (a 10 M-row call of cfg2's strings holds ~0.8 GB:
// five such calls at full speed; 0 = no budget).  Output columns are pinned memory of their own, bounded in PinnedPool.
class StagingPool {
  public:
    PipeSet *lease(uint64_t need)
    {
        std::unique_lock<std::mutex> lk(m_);
        bool waited = false;
        for (;;) {
            uint64_t in_use_bytes = 0, idle_bytes = 0;
            size_t running = 0;
            PipeSet *mru = nullptr;
            for (PipeSet *s : sets_) {
                // (a set in use is its holder's: its vectors and buffers change under the holder's hands -- only the figure written
                //  under this lock is read here.  [r6] TSan found lease() walking a running call's set: tests/run_sanitizers.sh)
                if (s->in_use) { in_use_bytes += s->reserved; ++running; continue; }
                idle_bytes += s->live();
                if (!mru || s->last_used > mru->last_used) mru = s;
            }
            {   // what the running calls have really allocated, from the global counters (atomic): whatever is alive and not idle
                const uint64_t alive = g_live_pinned.load(std::memory_order_relaxed) + g_live_device.load(std::memory_order_relaxed);
                if (alive > idle_bytes && alive - idle_bytes > in_use_bytes) in_use_bytes = alive - idle_bytes;
            }
            const uint64_t mine = mru ? std::max(need, mru->live()) : need;
            uint64_t others_idle = mru ? idle_bytes - mru->live() : 0;
            while (budget_ && in_use_bytes + mine + others_idle > budget_ && others_idle) { // make room: the longest-idle sets go first
                PipeSet *lru = nullptr;
                for (PipeSet *s : sets_)
                    if (!s->in_use && s != mru && s->live() && (!lru || s->last_used < lru->last_used)) lru = s;
                if (!lru) break;
                others_idle -= lru->live();
                lru->close();
                ++released_;
            }
            if (!budget_ || running == 0 || in_use_bytes + mine + others_idle <= budget_) {
                PipeSet *s = mru;
                if (!s) { s = new PipeSet; sets_.push_back(s); }
                s->in_use = true;
                s->reserved = mine;
                return s;
            }
            if (!waited) { waited = true; ++waits_; } // (calls that waited, not wake-ups)
            cv_.wait(lk); // a running call returns its set (give_back)
        }
    }
    void give_back(PipeSet *s)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            s->in_use = false;
            s->reserved = 0;
            s->last_used = ++clock_;
            peak_ = std::max(peak_, g_live_pinned.load(std::memory_order_relaxed) + g_live_device.load(std::memory_order_relaxed));
            while (budget_ && g_live_pinned.load(std::memory_order_relaxed) + g_live_device.load(std::memory_order_relaxed) > budget_) {
                PipeSet *lru = nullptr;
                for (PipeSet *q : sets_)
                    if (!q->in_use && q->live() && (!lru || q->last_used < lru->last_used)) lru = q;
                if (!lru) break; // (what is left belongs to running calls)
                lru->close();
                ++released_;
            }
            // (sets that hold nothing any more -- no staging and no device context: released above, or never used -- are dropped, so the
            //  list does not grow with every thread the engine ever had.  A set with a context but no buffers stays: a combined launch of
            //  small calls leases exactly that, and a context costs ~6 ms to make)
            for (size_t i = 0; i < sets_.size();)
                if (!sets_[i]->in_use && sets_[i]->live() == 0 && !sets_[i]->has_context() && sets_.size() > 1 && sets_[i] != s) {
                    delete sets_[i];
                    sets_.erase(sets_.begin() + (long)i);
                } else {
                    ++i;
                }
        }
        cv_.notify_all();
    }
    // out[0..7] = live pinned bytes, live device bytes, budget bytes, sets, sets in use, sets released so far, calls that waited, peak live bytes seen at a return
    void stats(uint64_t *out)
    {
        std::lock_guard<std::mutex> lk(m_);
        out[0] = g_live_pinned.load(std::memory_order_relaxed);
        out[1] = g_live_device.load(std::memory_order_relaxed);
        out[2] = budget_;
        out[3] = sets_.size();
        out[4] = 0;
        for (PipeSet *s : sets_) out[4] += s->in_use;
        out[5] = released_;
        out[6] = waits_;
        out[7] = peak_;
    }
    void set_budget(uint64_t bytes) { std::lock_guard<std::mutex> lk(m_); budget_ = bytes; }

  private:
    static uint64_t budget_from_env()
    {
        const char *e = getenv("POLARS_STRSIM_STAGING_BUDGET_MB");
        return (e && *e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)4096) << 20;
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<PipeSet *> sets_;
    uint64_t budget_ = budget_from_env(), clock_ = 0, released_ = 0, waits_ = 0, peak_ = 0;
};
// (never destroyed: releasing device memory from a static destructor races the HIP runtime's own teardown; the OS takes it back)
StagingPool &staging_pool() { static StagingPool *p = new StagingPool; return *p; }

struct PipeLease {
    PipeSet *set;
    explicit PipeLease(uint64_t need) : set(staging_pool().lease(need)) {}
    ~PipeLease() { staging_pool().give_back(set); }
    PipeLease(const PipeLease &) = delete;
    PipeLease &operator=(const PipeLease &) = delete;
};

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
