/*
 * strsim_oracle.c -- CPU restatement of the reference's pairwise string-similarity path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg may build, load or call it.  The product (polars-strsim_amd/)
 * never links or falls back to it.
 *
 * What it restates (reference = foxcroftjn/polars-strsim @ v0.2.3, /root/reference):
 *   src/expressions/strsim.rs:21-39    split_offsets
 *   src/expressions/strsim.rs:41-107   parallel_apply  (shape rule, literal broadcast, row split)
 *   src/expressions/strsim.rs:125-162  Levenshtein::compute
 *   src/expressions/strsim.rs:180-245  Jaro::compute
 *   src/expressions/strsim.rs:257-272  JaroWinkler::compute
 *   src/expressions/strsim.rs:286-308  Jaccard::compute
 *   src/expressions/strsim.rs:322-345  SorensenDice::compute
 *
 * Pinning: the reference is Rust and cannot be built in this image (no cargo/rustc), so this oracle is
 * pinned by the reference's own 1 115 known-answer vectors (strsim.rs:371-1534, abs tol 1e-8 per
 * strsim.rs:350) and the README demo table (README.md:59-72), both committed as data under
 * tests/golden/.  Those vectors are lowercase ASCII, length <= 27, and fix results to 1e-8 -- not to the
 * bit.  Bit-level results follow from construction: identical integer intermediates and the same IEEE-754
 * double operations in the same order as the Rust source (compile with -ffp-contract=off; Rust never
 * contracts a*b+c).  Non-ASCII behaviour (Rust `str::chars()` = Unicode scalar values), strings longer
 * than 27, nulls and literal broadcast are NOT pinned by any reference test: "parity unpinned" for those,
 * defined here by reading the source.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fPIC -shared -pthread).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORACLE_API __attribute__((visibility("default")))

enum { M_LEVENSHTEIN = 0, M_JARO = 1, M_JARO_WINKLER = 2, M_JACCARD = 3, M_SORENSEN_DICE = 4 };

/* ------------------------------------------------------------------------------------------------
 * `str::chars()` -- decode UTF-8 into Unicode scalar values (strsim.rs:133,138,189,194,262,297-300).
 * Input is valid UTF-8 by the Rust `&str` contract; a malformed tail is clamped, never over-read.
 * ---------------------------------------------------------------------------------------------- */
static size_t decode_chars(const uint8_t *s, size_t n, uint32_t *out)
{
    size_t i = 0, k = 0;
    while (i < n) {
        uint8_t c = s[i];
        uint32_t cp;
        size_t need;
        if (c < 0x80)      { cp = c;        need = 0; }
        else if (c < 0xE0) { cp = c & 0x1F; need = 1; }
        else if (c < 0xF0) { cp = c & 0x0F; need = 2; }
        else               { cp = c & 0x07; need = 3; }
        i++;
        while (need && i < n) { cp = (cp << 6) | (s[i] & 0x3F); i++; need--; }
        out[k++] = cp;
    }
    return k;
}

typedef struct {
    uint32_t *a, *b;       /* decoded strings                          */
    uint64_t *row0, *row1; /* two DP rows (strsim.rs:112 `matrix`)      */
    uint8_t *fa, *fb;      /* Jaro flags (strsim.rs:167 `flagged`)      */
    size_t cap;
} scratch_t;

static void scratch_reserve(scratch_t *s, size_t n)
{
    if (n + 1 <= s->cap) return;
    size_t cap = s->cap ? s->cap : 64;
    while (cap < n + 1) cap *= 2;
    s->a = (uint32_t *)realloc(s->a, cap * sizeof(uint32_t));
    s->b = (uint32_t *)realloc(s->b, cap * sizeof(uint32_t));
    s->row0 = (uint64_t *)realloc(s->row0, cap * sizeof(uint64_t));
    s->row1 = (uint64_t *)realloc(s->row1, cap * sizeof(uint64_t));
    s->fa = (uint8_t *)realloc(s->fa, cap);
    s->fb = (uint8_t *)realloc(s->fb, cap);
    s->cap = cap;
}

static void scratch_free(scratch_t *s)
{
    free(s->a); free(s->b); free(s->row0); free(s->row1); free(s->fa); free(s->fb);
    memset(s, 0, sizeof *s);
}

static int bytes_equal(const uint8_t *a, size_t la, const uint8_t *b, size_t lb)
{
    return la == lb && (la == 0 || memcmp(a, b, la) == 0);
}

static uint64_t min_u64(uint64_t x, uint64_t y) { return x < y ? x : y; }

/* strsim.rs:125-162.  Two-row DP; result 1 - dist/max(la,lb) in chars. */
static double lev_compute(scratch_t *s, const uint8_t *a8, size_t na, const uint8_t *b8, size_t nb,
                          uint64_t *dist_out, uint64_t *den_out)
{
    if ((na == 0 && nb == 0) || bytes_equal(a8, na, b8, nb)) {        /* :128-130 */
        if (dist_out) { *dist_out = 0; *den_out = 0; }
        return 1.0;
    }
    scratch_reserve(s, na > nb ? na : nb);
    size_t la = decode_chars(a8, na, s->a);                            /* :131-135 */
    size_t lb = decode_chars(b8, nb, s->b);                            /* :136-140 */
    uint64_t *prev = s->row0, *cur = s->row1;
    for (size_t j = 0; j <= lb; j++) prev[j] = j;                      /* :141-145 */
    for (size_t i = 0; i < la; i++) {                                  /* :146-159 */
        cur[0] = i + 1;
        uint32_t ai = s->a[i];
        for (size_t j = 0; j < lb; j++) {
            uint64_t sub = prev[j] + (ai == s->b[j] ? 0 : 1);
            cur[j + 1] = min_u64(min_u64(sub, prev[j + 1] + 1), cur[j] + 1);
        }
        uint64_t *t = prev; prev = cur; cur = t;
    }
    uint64_t dist = prev[lb];
    uint64_t den = la > lb ? la : lb;
    if (dist_out) { *dist_out = dist; *den_out = den; }
    return 1.0 - ((double)dist / (double)den);                         /* :160 */
}

/* strsim.rs:180-245. */
static double jaro_compute(scratch_t *s, const uint8_t *a8, size_t na, const uint8_t *b8, size_t nb)
{
    if ((na == 0 && nb == 0) || bytes_equal(a8, na, b8, nb)) return 1.0;   /* :182-183 */
    if (na == 0 || nb == 0) return 0.0;                                    /* :184-186 */
    scratch_reserve(s, na > nb ? na : nb);
    size_t la = decode_chars(a8, na, s->a);
    size_t lb = decode_chars(b8, nb, s->b);
    const uint32_t *a = s->a, *b = s->b;
    if (la == 1 && lb == 1) return a[0] == b[0] ? 1.0 : 0.0;               /* :197-199 */
    size_t mx = la > lb ? la : lb;
    size_t bound = mx / 2 - 1;                                             /* :200 */
    size_t m = 0;
    memset(s->fa, 0, mx);                                                  /* :202-207 */
    memset(s->fb, 0, mx);
    size_t take = lb + bound;                                              /* :208 `.take(b.len()+bound)` */
    size_t ni = la < take ? la : take;
    for (size_t i = 0; i < ni; i++) {
        size_t lo = bound > i ? 0 : i - bound;                             /* :209 */
        size_t hi = i + bound < lb - 1 ? i + bound : lb - 1;               /* :210 */
        for (size_t j = lo; j <= hi; j++) {                                /* :211-218 */
            if (a[i] == b[j] && !s->fb[j]) {
                m++;
                s->fa[i] = 1;
                s->fb[j] = 1;
                break;
            }
        }
    }
    /* :220-237 -- zip the flagged positions of a and of b in ascending order, count unequal chars */
    size_t t = 0, j = 0;
    for (size_t i = 0; i < mx; i++) {
        if (!s->fa[i]) continue;
        while (j < mx && !s->fb[j]) j++;
        if (j >= mx) break;
        if (a[i] != b[j]) t++;
        j++;
    }
    if (m == 0) return 0.0;                                                /* :238-239 */
    double dm = (double)m;
    return (dm / (double)la + dm / (double)lb + (double)(m - t / 2) / dm) / 3.0;   /* :241-242 */
}

/* strsim.rs:257-272. */
static double jw_compute(scratch_t *s, const uint8_t *a8, size_t na, const uint8_t *b8, size_t nb)
{
    double j = jaro_compute(s, a8, na, b8, nb);
    if (j > 0.7) {                                                         /* :260 */
        /* a.chars().zip(b.chars()).take(4).take_while(eq).count()  :261-266 */
        uint32_t ca[4], cb[4];
        uint32_t tmp[16];
        size_t la = 0, lb = 0;
        /* at most 4 chars = at most 16 bytes */
        size_t ta = na < 16 ? na : 16, tb = nb < 16 ? nb : 16;
        /* do not split a trailing multi-byte sequence: decode whole prefix, keep first 4 */
        size_t n = decode_chars(a8, ta, tmp);
        /* a sequence cut at byte 16 can only be char index >= 4 (4 chars need <= 16 bytes) */
        for (size_t i = 0; i < n && la < 4; i++) ca[la++] = tmp[i];
        n = decode_chars(b8, tb, tmp);
        for (size_t i = 0; i < n && lb < 4; i++) cb[lb++] = tmp[i];
        size_t lim = la < lb ? la : lb, p = 0;
        while (p < lim && ca[p] == cb[p]) p++;
        double pl = (double)p;
        return j + (pl * 0.1 * (1.0 - j));                                 /* :267 */
    }
    return j;
}

static int cmp_u32(const void *x, const void *y)
{
    uint32_t a = *(const uint32_t *)x, b = *(const uint32_t *)y;
    return a < b ? -1 : a > b;
}

/* Character-multiset counts shared by Jaccard and Sorensen-Dice (strsim.rs:297-300, :333-336):
 * the HashMap<char,[usize;2]> becomes two sorted code-point arrays merged once; the sums the
 * reference folds over map values (:301-305, :337-342) are order-independent integers. */
static void multiset_sums(scratch_t *s, const uint8_t *a8, size_t na, const uint8_t *b8, size_t nb,
                          uint64_t *sum_min, uint64_t *sum_max, uint64_t *sum_all)
{
    scratch_reserve(s, na > nb ? na : nb);
    size_t la = decode_chars(a8, na, s->a);
    size_t lb = decode_chars(b8, nb, s->b);
    qsort(s->a, la, sizeof(uint32_t), cmp_u32);
    qsort(s->b, lb, sizeof(uint32_t), cmp_u32);
    uint64_t mn = 0, mxs = 0;
    size_t i = 0, j = 0;
    while (i < la || j < lb) {
        uint32_t c;
        if (j >= lb || (i < la && s->a[i] <= s->b[j])) c = s->a[i]; else c = s->b[j];
        uint64_t ca = 0, cb = 0;
        while (i < la && s->a[i] == c) { ca++; i++; }
        while (j < lb && s->b[j] == c) { cb++; j++; }
        mn += ca < cb ? ca : cb;
        mxs += ca > cb ? ca : cb;
    }
    *sum_min = mn; *sum_max = mxs; *sum_all = (uint64_t)la + (uint64_t)lb;
}

/* strsim.rs:286-308. */
static double jaccard_compute(scratch_t *s, const uint8_t *a8, size_t na, const uint8_t *b8, size_t nb)
{
    if ((na == 0 && nb == 0) || bytes_equal(a8, na, b8, nb)) return 1.0;   /* :288-289 */
    if (na == 0 || nb == 0) return 0.0;                                    /* :290-292 */
    uint64_t mn, mx, all;
    multiset_sums(s, a8, na, b8, nb, &mn, &mx, &all);
    return (double)mn / (double)mx;                                        /* :306 */
}

/* strsim.rs:322-345.  frac[1] accumulates v[0] and v[1]; frac[2] stays 0 (:337-342). */
static double dice_compute(scratch_t *s, const uint8_t *a8, size_t na, const uint8_t *b8, size_t nb)
{
    if ((na == 0 && nb == 0) || bytes_equal(a8, na, b8, nb)) return 1.0;   /* :324-325 */
    if (na == 0 || nb == 0) return 0.0;                                    /* :326-328 */
    uint64_t mn, mx, all;
    multiset_sums(s, a8, na, b8, nb, &mn, &mx, &all);
    return 2.0 * (double)mn / (double)all;                                 /* :343 */
}

static double compute_one(scratch_t *s, int measure, const uint8_t *a, size_t na, const uint8_t *b, size_t nb)
{
    switch (measure) {
    case M_LEVENSHTEIN:   return lev_compute(s, a, na, b, nb, NULL, NULL);
    case M_JARO:          return jaro_compute(s, a, na, b, nb);
    case M_JARO_WINKLER:  return jw_compute(s, a, na, b, nb);
    case M_JACCARD:       return jaccard_compute(s, a, na, b, nb);
    default:              return dice_compute(s, a, na, b, nb);
    }
}

/* ---- exported single-pair entry points (what the reference's `Test::test` exercises, :352-363) ---- */
ORACLE_API double oracle_pair(int measure, const uint8_t *a, size_t na, const uint8_t *b, size_t nb)
{
    scratch_t s; memset(&s, 0, sizeof s);
    double r = compute_one(&s, measure, a, na, b, nb);
    scratch_free(&s);
    return r;
}

/* integer edit distance + denominator, for rational (bit-exactness) checks */
ORACLE_API int oracle_lev_rational(const uint8_t *a, size_t na, const uint8_t *b, size_t nb,
                                   uint64_t *dist, uint64_t *den)
{
    scratch_t s; memset(&s, 0, sizeof s);
    lev_compute(&s, a, na, b, nb, dist, den);
    scratch_free(&s);
    return 0;
}

/* strsim.rs:21-39.  Writes n (offset,len) pairs. */
ORACLE_API void oracle_split_offsets(uint64_t len, uint64_t n, uint64_t *out_offset_len)
{
    if (n == 1) { out_offset_len[0] = 0; out_offset_len[1] = len; return; }
    uint64_t chunk = len / n;
    for (uint64_t p = 0; p < n; p++) {
        uint64_t off = p * chunk;
        out_offset_len[2 * p] = off;
        out_offset_len[2 * p + 1] = (p == n - 1) ? len - off : chunk;
    }
}

/* ---- batch entry point: the row loop of parallel_apply (strsim.rs:41-107) ------------------------
 * Columns are Arrow-style: offsets[rows+1] (uint64 here so any size works) + packed values.
 * rows_a == 1 or rows_b == 1 broadcasts that side (the "Utf8 literal" rule, :48-52, :61-66, :85-92;
 * the literal-vs-column case follows the parallel-context branch :64-66 -- the rayon branch's
 * split by a.len() at :73 is a reference defect).
 * Returns 0, or -1 on the ShapeMismatch condition of :48-52.
 * Rows are split over `nthreads` by split_offsets exactly like :73-77; results are identical for any
 * thread count (pure per-row function). */
typedef struct {
    int measure;
    const uint64_t *oa; const uint8_t *va; uint64_t ra;
    const uint64_t *ob; const uint8_t *vb; uint64_t rb;
    double *out; uint64_t off, len;
} job_t;

static void *job_run(void *p)
{
    job_t *j = (job_t *)p;
    scratch_t s; memset(&s, 0, sizeof s);
    for (uint64_t r = j->off; r < j->off + j->len; r++) {
        uint64_t ia = j->ra == 1 ? 0 : r, ib = j->rb == 1 ? 0 : r;
        j->out[r] = compute_one(&s, j->measure,
                                j->va + j->oa[ia], (size_t)(j->oa[ia + 1] - j->oa[ia]),
                                j->vb + j->ob[ib], (size_t)(j->ob[ib + 1] - j->ob[ib]));
    }
    scratch_free(&s);
    return NULL;
}

ORACLE_API int oracle_batch(int measure,
                            const uint64_t *offs_a, const uint8_t *vals_a, uint64_t rows_a,
                            const uint64_t *offs_b, const uint8_t *vals_b, uint64_t rows_b,
                            double *out, int nthreads)
{
    if (rows_a != rows_b && rows_a != 1 && rows_b != 1) return -1;        /* :48-52 */
    uint64_t n = rows_a == 1 ? rows_b : rows_a;
    if (rows_a == 1 && rows_b == 1) n = 1;
    if (nthreads < 1) nthreads = 1;
    uint64_t *splits = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)nthreads);
    oracle_split_offsets(n, (uint64_t)nthreads, splits);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        job_t jb = { measure, offs_a, vals_a, rows_a, offs_b, vals_b, rows_b, out, splits[2 * t], splits[2 * t + 1] };
        jobs[t] = jb;
        if (nthreads == 1) job_run(&jobs[t]);
        else pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs); free(splits);
    return 0;
}

/* u32-offset variant (the device layout of the product: u32 offsets[rows+1]); same semantics, same row partition.
 * Reads the 32-bit offsets in place -- an earlier version widened both offset arrays on one thread first, which took
 * longer than the 256-thread compute it preceded and made the CPU baseline look 10-20x slower than it is. */
typedef struct {
    int measure;
    const uint32_t *oa; const uint8_t *va; uint64_t ra;
    const uint32_t *ob; const uint8_t *vb; uint64_t rb;
    double *out; uint64_t off, len;
} job32_t;

static void *job32_run(void *p)
{
    job32_t *j = (job32_t *)p;
    scratch_t s; memset(&s, 0, sizeof s);
    for (uint64_t r = j->off; r < j->off + j->len; r++) {
        uint64_t ia = j->ra == 1 ? 0 : r, ib = j->rb == 1 ? 0 : r;
        j->out[r] = compute_one(&s, j->measure,
                                j->va + j->oa[ia], (size_t)(j->oa[ia + 1] - j->oa[ia]),
                                j->vb + j->ob[ib], (size_t)(j->ob[ib + 1] - j->ob[ib]));
    }
    scratch_free(&s);
    return NULL;
}

ORACLE_API int oracle_batch_u32(int measure,
                                const uint32_t *offs_a, const uint8_t *vals_a, uint64_t rows_a,
                                const uint32_t *offs_b, const uint8_t *vals_b, uint64_t rows_b,
                                double *out, int nthreads)
{
    if (rows_a != rows_b && rows_a != 1 && rows_b != 1) return -1;        /* :48-52 */
    uint64_t n = rows_a == 1 ? rows_b : rows_a;
    if (rows_a == 1 && rows_b == 1) n = 1;
    if (nthreads < 1) nthreads = 1;
    uint64_t *splits = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)nthreads);
    oracle_split_offsets(n, (uint64_t)nthreads, splits);                  /* :73-77: one contiguous range per thread */
    job32_t *jobs = (job32_t *)malloc(sizeof(job32_t) * (size_t)nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        job32_t jb = { measure, offs_a, vals_a, rows_a, offs_b, vals_b, rows_b, out, splits[2 * t], splits[2 * t + 1] };
        jobs[t] = jb;
        if (nthreads == 1) job32_run(&jobs[t]);
        else pthread_create(&th[t], NULL, job32_run, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs); free(splits);
    return 0;
}
