// plugin_sanitize_driver.cpp -- drives the HOST side of the Polars plugin layer (csrc/polars_plugin.cpp: Arrow import, the packers,
// the validity builder, the thread pool, input ownership) under AddressSanitizer + UndefinedBehaviorSanitizer and under
// ThreadSanitizer, without a GPU.  TEST INFRASTRUCTURE (tests/run_sanitizers.sh builds and runs it; logs under profiles/).
//
// What the reference gets from Rust's ownership rules (per-task private scratch, strsim.rs:78-84; borrowed chunk iterators,
// strsim.rs:46-47) this layer has to get right by hand: several caller threads at once (an engine's worker threads call a plugin
// concurrently), each call fanning out over the packing pool; chunks at odd Arrow offsets; validity bitmaps at odd bit offsets;
// inline and buffer-backed views; release callbacks called exactly once.
//
// Links the test-hooks build of the plugin layer (-DSTRSIM_TEST_HOOKS) compiled with the same sanitizer.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "polars_plugin_abi.h"

extern "C" {
int _strsim_test_pack_series(SeriesExport *series, uint64_t r0, uint64_t r1, uint32_t *off_out, uint8_t *val_out, uint64_t val_cap,
                             uint64_t *rows_out, uint64_t *bytes_out, uint8_t *valid_out, unsigned threads);
int _strsim_test_pack_onepass(SeriesExport *series, uint64_t r0, uint64_t r1, uint64_t bpr256, unsigned threads, uint8_t *val_out,
                              uint64_t val_cap, uint8_t *len_out, uint64_t *bytes_out, int *nseg_out);
int _strsim_test_pack_grants(int engine_parallel, int n_calls, uint64_t rows, unsigned *threads_out, int *helpers_out);
int _strsim_test_validity(SeriesExport *two_series, uint64_t *words, int64_t *null_count, double *vals, uint64_t *rows_out,
                          unsigned threads);
int _strsim_test_pack_views(SeriesExport *series, uint64_t r0, uint64_t r1, uint64_t lbpr256, unsigned threads, uint8_t *views_out,
                            uint8_t *long_out, uint64_t long_cap, uint64_t *span_out, uint64_t *bytes_out);
int _strsim_test_staging_lease(uint64_t budget_bytes, uint64_t need, uint64_t grow, unsigned hold_us, uint64_t *stats8);
int _strsim_test_combine(SeriesExport *two_series, int measure, double *out, uint64_t *rows_out);
}

namespace {

std::atomic<long> g_array_releases{0}, g_series_releases{0}, g_failures{0};

#define CHECK(cond, ...)                                                         \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "CHECK FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                                        \
            fprintf(stderr, "\n");                                               \
            g_failures++;                                                        \
            return;                                                              \
        }                                                                        \
    } while (0)

struct View { // Arrow Utf8View slot
    uint32_t len;
    uint8_t rest[12]; // len <= 12: the bytes; else prefix[4], buffer index u32, offset u32
};
static_assert(sizeof(View) == 16, "view slot");

enum Layout { L_U32, L_U64, L_VIEW };

// One exported Series: owns every buffer; the release callbacks only count (the memory goes with the object).
struct Exported {
    Layout layout;
    std::vector<std::string> rows;
    std::vector<uint8_t> valid; // one byte per row
    struct ChunkMem {
        std::vector<uint8_t> bitmap;
        std::vector<int32_t> off32;
        std::vector<int64_t> off64;
        std::vector<View> views;
        std::vector<std::vector<uint8_t>> data; // variadic buffers (views) / the values buffer
        std::vector<int64_t> sizes;             // views: the buffer-sizes buffer that closes the list
        std::vector<const void *> bufs;
        ArrowArray arr;
        int64_t pad; // rows in front of the chunk's first row (Arrow `offset`)
    };
    std::vector<std::unique_ptr<ChunkMem>> chunks;
    std::vector<ArrowArray *> arr_ptrs;
    ArrowSchema schema;
    SeriesExport se;

    static void rel_array(ArrowArray *a) { g_array_releases++; a->release = nullptr; }
    static void rel_schema(ArrowSchema *s) { s->release = nullptr; }
    static void rel_series(SeriesExport *s)
    {
        g_series_releases++;
        s->release = nullptr;
    }

    Exported(Layout l, std::vector<std::string> r, std::vector<uint8_t> v, std::mt19937 &rng) : layout(l), rows(std::move(r)), valid(std::move(v))
    {
        size_t at = 0;
        while (at < rows.size() || chunks.empty()) {
            const size_t n = rows.empty() ? 0 : std::min(rows.size() - at, (size_t)(1 + rng() % 700));
            auto c = std::make_unique<ChunkMem>();
            c->pad = (int64_t)(rng() % 70); // odd Arrow offsets, odd bit offsets in the bitmap
            const int64_t tot = c->pad + (int64_t)n;
            bool any_null = false;
            c->bitmap.assign((size_t)(tot + 7) / 8 + 8, 0xA5);
            for (int64_t i = 0; i < tot; ++i) {
                const bool ok = i < c->pad ? (rng() & 1) : valid[at + (size_t)(i - c->pad)] != 0;
                if (i >= c->pad && !ok) any_null = true;
                if (ok) c->bitmap[(size_t)i / 8] |= (uint8_t)(1u << (i & 7));
                else c->bitmap[(size_t)i / 8] &= (uint8_t)~(1u << (i & 7));
            }
            if (layout == L_VIEW) {
                c->views.resize((size_t)tot);
                c->data.resize(1 + rng() % 3);
                for (int64_t i = 0; i < tot; ++i) {
                    static const std::string junk = "padding-row-in-front-of-the-chunk";
                    const std::string &s = i < c->pad ? junk : rows[at + (size_t)(i - c->pad)];
                    View &w = c->views[(size_t)i];
                    memset(&w, 0, sizeof w);
                    w.len = (uint32_t)s.size();
                    if (s.size() <= 12) {
                        memcpy(w.rest, s.data(), s.size());
                    } else {
                        const uint32_t b = (uint32_t)(rng() % c->data.size());
                        const uint32_t o = (uint32_t)c->data[b].size();
                        c->data[b].insert(c->data[b].end(), s.begin(), s.end());
                        memcpy(w.rest, s.data(), 4);
                        memcpy(w.rest + 4, &b, 4);
                        memcpy(w.rest + 8, &o, 4);
                    }
                }
                c->bufs = {any_null ? (const void *)c->bitmap.data() : nullptr, c->views.data()};
                for (auto &d : c->data) { c->bufs.push_back(d.data()); c->sizes.push_back((int64_t)d.size()); }
                c->bufs.push_back(c->sizes.data());
            } else {
                c->data.resize(1);
                auto &d = c->data[0];
                std::vector<int64_t> o((size_t)tot + 1, 0);
                for (int64_t i = 0; i < tot; ++i) {
                    static const std::string junk = "xx";
                    const bool isnull = i >= c->pad && !valid[at + (size_t)(i - c->pad)];
                    const std::string &s = i < c->pad ? junk : rows[at + (size_t)(i - c->pad)];
                    if (!isnull) d.insert(d.end(), s.begin(), s.end()); // (a null slot of an offsets layout holds no bytes here)
                    o[(size_t)i + 1] = (int64_t)d.size();
                }
                d.resize(d.size() + 16, 0x5A);
                if (layout == L_U32) { c->off32.assign(o.begin(), o.end()); c->bufs = {any_null ? (const void *)c->bitmap.data() : nullptr, c->off32.data(), d.data()}; }
                else { c->off64 = o; c->bufs = {any_null ? (const void *)c->bitmap.data() : nullptr, c->off64.data(), d.data()}; }
            }
            memset(&c->arr, 0, sizeof c->arr);
            c->arr.length = (int64_t)n;
            c->arr.offset = c->pad;
            c->arr.null_count = any_null ? 1 : 0;
            c->arr.n_buffers = (int64_t)c->bufs.size();
            c->arr.buffers = c->bufs.data();
            c->arr.release = rel_array;
            chunks.push_back(std::move(c));
            at += n;
            if (rows.empty()) break;
        }
        for (auto &c : chunks) arr_ptrs.push_back(&c->arr);
        memset(&schema, 0, sizeof schema);
        schema.format = layout == L_VIEW ? "vu" : (layout == L_U32 ? "u" : "U");
        schema.name = "col";
        schema.flags = ARROW_FLAG_NULLABLE;
        schema.release = rel_schema;
        fill(se);
    }
    void fill(SeriesExport &out)
    {
        for (auto &c : chunks) c->arr.release = rel_array;
        schema.release = rel_schema;
        out.field = &schema;
        out.arrays = arr_ptrs.data();
        out.len = arr_ptrs.size();
        out.release = rel_series;
        out.private_data = nullptr;
    }
};

std::string rand_string(std::mt19937 &rng, int maxlen)
{
    static const char alpha[] = "abcdefghijklmnopqrstuvwxyz \xC3\xA9\xE6\x97\xA5";
    const int n = (int)(rng() % (unsigned)(maxlen + 1));
    std::string s;
    for (int i = 0; i < n; ++i) s.push_back(alpha[rng() % (sizeof alpha - 1)]);
    return s;
}

void one_round(unsigned seed)
{
    std::mt19937 rng(seed);
    const Layout layout = (Layout)(seed % 3);
    const size_t n = 1 + rng() % 6000;
    const int maxlen = (seed & 4) ? 12 : ((seed & 8) ? 300 : 40);
    std::vector<std::string> rows(n);
    std::vector<uint8_t> valid(n, 1);
    for (size_t i = 0; i < n; ++i) {
        rows[i] = rand_string(rng, maxlen);
        if ((seed & 16) && rng() % 10 == 0) valid[i] = 0;
    }
    const unsigned threads = 1 + rng() % 8;
    // ---- (1) the two-pass packer on a range of rows
    {
        Exported ex(layout, rows, valid, rng);
        const uint64_t r0 = rng() % n, r1 = r0 + rng() % (n - r0 + 1);
        std::string want;
        std::vector<uint32_t> want_off{0};
        for (uint64_t r = r0; r < r1; ++r) {
            if (valid[r]) want += rows[r];
            want_off.push_back((uint32_t)want.size());
        }
        std::vector<uint32_t> off(r1 - r0 + 1, 0xDEADBEEF);
        std::vector<uint8_t> val(want.size() + 64, 0xEE), ok(r1 - r0 + 1, 7);
        uint64_t nrows = 0, bytes = 0;
        const long a0 = g_array_releases, s0 = g_series_releases;
        const int rc = _strsim_test_pack_series(&ex.se, r0, r1, off.data(), val.data(), val.size(), &nrows, &bytes, ok.data(), threads);
        CHECK(rc == 0, "pack_series rc %d (%s)", rc, _polars_plugin_get_last_error_message());
        CHECK(nrows == n && bytes == want.size(), "rows %llu/%zu bytes %llu/%zu", (unsigned long long)nrows, n, (unsigned long long)bytes, want.size());
        CHECK(want.empty() || memcmp(val.data(), want.data(), want.size()) == 0, "packed bytes differ (layout %d, threads %u)", (int)layout, threads);
        CHECK(memcmp(off.data(), want_off.data(), want_off.size() * 4) == 0, "offsets differ");
        for (uint64_t r = r0; r < r1; ++r) CHECK(ok[r - r0] == valid[r], "validity of row %llu", (unsigned long long)r);
        CHECK(g_series_releases >= s0 + 1 && g_array_releases >= a0 + (long)ex.chunks.size(), "the input was not released");
        for (auto &c : ex.chunks) CHECK(c->arr.release == nullptr, "a chunk was not released");
        CHECK(ex.se.release == nullptr, "the series was not released");
    }
    // ---- (2) the one-pass packer (view columns, strings of at most 255 bytes)
    if (layout == L_VIEW) {
        Exported ex(layout, rows, valid, rng);
        const uint64_t r0 = rng() % n, r1 = r0 + rng() % (n - r0 + 1);
        std::string want;
        std::vector<uint8_t> want_len;
        bool fits = true;
        for (uint64_t r = r0; r < r1; ++r) {
            const std::string s = valid[r] ? rows[r] : std::string();
            if (s.size() > 255) fits = false;
            want += s;
            want_len.push_back((uint8_t)s.size());
        }
        std::vector<uint8_t> val(want.size() + 4096, 0xEE), len(r1 - r0 + 64, 0xEE);
        uint64_t bytes = 0;
        int nseg = 0;
        const uint64_t bpr256 = (uint64_t)((want.size() + 1) * 256 / (r1 - r0 + 1)) + 256 * (rng() % 3);
        const int rc = _strsim_test_pack_onepass(&ex.se, r0, r1, bpr256, threads, val.data(), val.size(), len.data(), &bytes, &nseg);
        CHECK(rc >= 0, "pack_onepass rc %d (%s)", rc, _polars_plugin_get_last_error_message());
        if (rc == 1) {
            CHECK(fits, "the one-pass packer took a slice with a string beyond 255 bytes");
            CHECK(bytes == want.size() && (want.empty() || memcmp(val.data(), want.data(), want.size()) == 0), "one-pass bytes differ");
            CHECK(want_len.empty() || memcmp(len.data(), want_len.data(), want_len.size()) == 0, "one-pass lengths differ");
        }
        CHECK(ex.se.release == nullptr, "the series was not released");
    }
    // ---- (2b) the view-native packer: the views as they lie + the strings beyond 12 bytes in per-thread segments
    if (layout == L_VIEW) {
        Exported ex(layout, rows, valid, rng);
        const uint64_t r0 = rng() % n, r1 = r0 + rng() % (n - r0 + 1);
        uint64_t want_bytes = 0;
        for (uint64_t r = r0; r < r1; ++r) want_bytes += valid[r] ? rows[r].size() : 0;
        std::vector<uint8_t> views((r1 - r0 + 8) * 16, 0xEE), lng(want_bytes + (r1 - r0) * 2 + (1u << 20), 0xEE);
        uint64_t span = 0, bytes = 0;
        const uint64_t est = (seed & 64) ? ~0ull : (uint64_t)(rng() % 3) * 256 * 8; // exact / an estimate that may not hold
        const int rc = _strsim_test_pack_views(&ex.se, r0, r1, est, threads, views.data(), lng.data(), lng.size(), &span, &bytes);
        CHECK(rc == 0, "pack_views rc %d (%s)", rc, _polars_plugin_get_last_error_message());
        CHECK(bytes == want_bytes, "view-native packed size %llu / %llu", (unsigned long long)bytes, (unsigned long long)want_bytes);
        for (uint64_t r = r0; r < r1; ++r) {
            const std::string want = valid[r] ? rows[r] : std::string();
            View w;
            memcpy(&w, views.data() + (r - r0) * 16, 16);
            CHECK(w.len == want.size(), "view length of row %llu", (unsigned long long)r);
            uint32_t at;
            memcpy(&at, w.rest + 8, 4);
            const uint8_t *src = w.len <= 12 ? w.rest : lng.data() + at;
            CHECK(w.len <= 12 || (uint64_t)at + w.len <= span, "long string of row %llu outside the shipped span", (unsigned long long)r);
            CHECK(want.empty() || memcmp(src, want.data(), want.size()) == 0, "view-native bytes of row %llu", (unsigned long long)r);
        }
        CHECK(ex.se.release == nullptr, "the series was not released");
    }
    // ---- (3) the validity of a call over two columns (one of them may be a literal, the literal may be null)
    {
        const int lit = (int)(rng() % 4); // 0, 1: that side is the literal; else none
        std::vector<std::string> ra = lit == 0 ? std::vector<std::string>{"lit"} : rows, rb = lit == 1 ? std::vector<std::string>{"lit"} : rows;
        std::vector<uint8_t> va = lit == 0 ? std::vector<uint8_t>{(uint8_t)(rng() % 4 != 0)} : valid, vb(rb.size(), 1);
        if (lit != 1) for (auto &x : vb) x = (seed & 32) ? (uint8_t)(rng() % 7 != 0) : 1;
        else vb[0] = (uint8_t)(rng() % 4 != 0);
        Exported ea(layout, ra, va, rng), eb((Layout)((seed / 3) % 3), rb, vb, rng);
        SeriesExport two[2];
        ea.fill(two[0]);
        eb.fill(two[1]);
        const uint64_t rows_out_want = lit == 0 ? rb.size() : ra.size();
        std::vector<uint64_t> words((rows_out_want + 63) / 64 + 1, 0x1234567890ABCDEFull);
        std::vector<double> vals(rows_out_want, 1.5);
        int64_t nulls = -1;
        uint64_t rows_out = 0;
        const int rc = _strsim_test_validity(two, words.data(), &nulls, vals.data(), &rows_out, threads);
        CHECK(rc == 0, "validity rc %d (%s)", rc, _polars_plugin_get_last_error_message());
        CHECK(rows_out == rows_out_want, "rows");
        int64_t want_nulls = 0;
        for (uint64_t r = 0; r < rows_out_want; ++r) {
            const bool ok = va[lit == 0 ? 0 : r] && vb[lit == 1 ? 0 : r];
            want_nulls += !ok;
            CHECK((((words[r / 64] >> (r & 63)) & 1) != 0) == ok, "validity bit of row %llu", (unsigned long long)r);
            // (a null literal makes every row null: the bitmap is all the plugin writes then, the slots keep what they held)
            const bool all_null = (lit == 0 && !va[0]) || (lit == 1 && !vb[0]);
            CHECK(ok ? vals[r] == 1.5 : (all_null || vals[r] == 0.0), "value under a null slot, row %llu", (unsigned long long)r);
        }
        CHECK(nulls == want_nulls, "null count %lld / %lld", (long long)nulls, (long long)want_nulls);
        CHECK(two[0].release == nullptr && two[1].release == nullptr, "the inputs were not released");
    }
    // ---- (3b) [r6] the staging pool: leases of pipeline sets from several caller threads at once under a budget that holds about two
    //      of them (the calls WAIT for one another, idle sets are released); host memory stands in for pinned / device memory
    {
        const uint64_t budget = 24u << 20;
        uint64_t st[8];
        const uint64_t need = (uint64_t)(4 + rng() % 12) << 20, grow = (uint64_t)(1 + rng() % 10) << 20;
        const int rc = _strsim_test_staging_lease(budget, need, grow, (unsigned)(rng() % 300), st);
        CHECK(rc == 0, "staging lease rc %d (%s)", rc, _polars_plugin_get_last_error_message());
        CHECK(st[2] == budget, "the pool's budget");
        CHECK(st[3] >= 1 && st[4] <= st[3], "sets %llu, in use %llu", (unsigned long long)st[3], (unsigned long long)st[4]);
    }
    // ---- (3c) [r6] the combiner of concurrent small calls: this thread's call joins whatever batch the other threads have open; a CPU
    //      stand-in for the combined launch writes a value that depends on the row's two lengths and first byte
    {
        const size_t m = 1 + rng() % 300;
        std::vector<std::string> ra(rows.begin(), rows.begin() + (long)std::min(m, rows.size())), rb = ra;
        for (auto &x : ra) if (x.size() > 200) x.resize(200);
        for (auto &x : rb) { if (x.size() > 150) x.resize(150); if (!x.empty() && (rng() & 1)) x.pop_back(); }
        std::vector<uint8_t> va(ra.size(), 1), vb(rb.size(), 1);
        for (auto &x : va) x = (uint8_t)(rng() % 9 != 0);
        Exported ea(layout, ra, va, rng), eb((Layout)((seed / 5) % 3), rb, vb, rng);
        SeriesExport two[2];
        ea.fill(two[0]);
        eb.fill(two[1]);
        std::vector<double> out(ra.size(), -1.0);
        uint64_t rows_out = 0;
        const int rc = _strsim_test_combine(two, (int)(seed % 5), out.data(), &rows_out);
        CHECK(rc == 1, "combine rc %d (%s)", rc, _polars_plugin_get_last_error_message());
        CHECK(rows_out == ra.size(), "rows");
        for (size_t r = 0; r < ra.size(); ++r) {
            const std::string a = va[r] ? ra[r] : std::string(), b = rb[r]; // (a null slot packs as an empty string)
            const double want = 4096.0 * (double)a.size() + (double)b.size() + (a.empty() ? 0.0 : (unsigned char)a[0] / 256.0);
            CHECK(out[r] == want, "combined call, row %llu: %f / %f", (unsigned long long)r, out[r], want);
        }
        CHECK(two[0].release == nullptr && two[1].release == nullptr, "the inputs were not released");
    }
    // ---- (4) the helper-thread budget of engine-parallel calls, borrowed and returned by several caller threads at once: what is
    //      lent out at any moment never exceeds half the CPUs (the grant of every call in flight, other threads' included)
    {
        unsigned t3[3] = {0, 0, 0};
        int lent = -1;
        (void)_strsim_test_pack_grants(1, 3, 10000000ull, t3, &lent);
        const int budget = (int)(std::thread::hardware_concurrency() < 32u ? std::thread::hardware_concurrency() : 32u) / 2;
        CHECK(lent >= 0 && lent <= budget, "helpers lent out: %d of a budget of at most %d", lent, budget);
        CHECK(t3[0] >= 1 && t3[1] >= 1 && t3[2] >= 1, "a call packs on at least its own thread");
    }
}

} // namespace

int main(int argc, char **argv)
{
    const int callers = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 60;
    std::vector<std::thread> th;
    for (int t = 0; t < callers; ++t)
        th.emplace_back([=] { for (int r = 0; r < rounds; ++r) one_round((unsigned)(t * 1000 + r)); });
    for (auto &x : th) x.join();
    {   // every permit came back
        unsigned t1[1];
        int lent = 0;
        if (_strsim_test_pack_grants(1, 0, 0, t1, &lent) != 0) { fprintf(stderr, "CHECK FAILED: helper permits were not returned\n"); ++g_failures; }
    }
    printf("plugin_sanitize_driver: %d caller threads x %d rounds, %ld chunk releases, %ld series releases, %ld check failures\n", callers,
           rounds, (long)g_array_releases, (long)g_series_releases, (long)g_failures);
    return g_failures ? 1 : 0;
}
