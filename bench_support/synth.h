// synth.h -- deterministic synthetic string-pair columns for benchmarks and full-size parity checks.
//
// Counter-based: every row is a pure function of (seed, row), so any shard can be produced on any
// rank, on the GPU (kernels in synth.hip) or on the CPU (host loops in synth.hip) with identical bytes.
// Input law (BASELINE.md section 3 / SURVEY.md 8d): alphabet a-z; column b is an edited copy of a
// (1-3 random ins/del/sub) with p = 0.5, identical with p = 0.05, else independent.
// Length laws: UNIFORM lo..hi, or ZIPF: lo + r - 1 with r ~ Zipf(s=1) truncated to 1..(hi-lo+1).
// Bench/test tooling, not part of the product library.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SYNTH_HD __host__ __device__ inline
#else
#define SYNTH_HD inline
#endif

namespace synth {

enum LengthLaw : int { UNIFORM = 0, ZIPF = 1 };

struct Config {
    uint64_t seed;
    int law;         // LengthLaw
    uint32_t lo, hi; // inclusive length range (bytes)
};

constexpr int MAX_LEN = 1024;

struct Rng {
    uint64_t s;
    SYNTH_HD uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    SYNTH_HD uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); } // [0, n)
    SYNTH_HD double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

SYNTH_HD Rng row_rng(uint64_t seed, uint64_t row, uint64_t stream)
{
    Rng r{seed * 0xD1342543DE82EF95ull + row * 0x9E3779B97F4A7C15ull + stream * 0xC2B2AE3D27D4EB4Full};
    r.next();
    return r;
}

SYNTH_HD uint32_t draw_len(const Config &c, Rng &r)
{
    const uint32_t span = c.hi - c.lo + 1u;
    if (c.law == UNIFORM) return c.lo + r.below(span);
    // Zipf(s=1) on 1..span by inverse CDF (linear scan; the mass sits at small ranks)
    double h = 0.0;
    for (uint32_t k = 1; k <= span; ++k) h += 1.0 / (double)k;
    const double u = r.unit() * h;
    double acc = 0.0;
    for (uint32_t k = 1; k <= span; ++k) {
        acc += 1.0 / (double)k;
        if (u < acc) return c.lo + k - 1u;
    }
    return c.hi;
}

// kind of row: 0 = b is an edited copy, 1 = identical, 2 = independent
SYNTH_HD int draw_kind(Rng &r)
{
    const uint32_t x = r.below(100u);
    return x < 50u ? 0 : (x < 55u ? 1 : 2);
}

SYNTH_HD uint8_t draw_char(Rng &r) { return (uint8_t)('a' + r.below(26u)); }

// Lengths of both strings of a row.
SYNTH_HD void row_lengths(const Config &c, uint64_t row, uint32_t &la, uint32_t &lb)
{
    Rng h = row_rng(c.seed, row, 0);
    la = draw_len(c, h);
    const int kind = draw_kind(h);
    if (kind == 1) { lb = la; return; }
    if (kind == 2) { lb = draw_len(c, h); return; }
    uint32_t l = la;
    const uint32_t nedit = 1u + h.below(3u);
    for (uint32_t e = 0; e < nedit; ++e) {
        const uint32_t op = h.below(3u);
        (void)h.next(); // position draw (consumed identically in row_fill)
        (void)h.next(); // char draw
        if (op == 0u) { if (l < c.hi) ++l; }
        else if (op == 1u) { if (l > 0u) --l; }
    }
    lb = l;
}

// Bytes of both strings (a: la bytes, b: lb bytes).  `tmp` needs MAX_LEN + 4 bytes when editing.
SYNTH_HD void row_fill(const Config &c, uint64_t row, uint8_t *a, uint8_t *b)
{
    Rng h = row_rng(c.seed, row, 0);
    const uint32_t la = draw_len(c, h);
    const int kind = draw_kind(h);
    Rng ca = row_rng(c.seed, row, 1);
    for (uint32_t i = 0; i < la; ++i) a[i] = draw_char(ca);
    if (kind == 1) { for (uint32_t i = 0; i < la; ++i) b[i] = a[i]; return; }
    if (kind == 2) {
        const uint32_t lb = draw_len(c, h);
        Rng cb = row_rng(c.seed, row, 2);
        for (uint32_t i = 0; i < lb; ++i) b[i] = draw_char(cb);
        return;
    }
    uint32_t l = la;
    for (uint32_t i = 0; i < la; ++i) b[i] = a[i];
    const uint32_t nedit = 1u + h.below(3u);
    for (uint32_t e = 0; e < nedit; ++e) {
        const uint32_t op = h.below(3u);
        const uint64_t pr = h.next();
        const uint8_t ch = (uint8_t)('a' + (uint32_t)((h.next() >> 32) * 26ull >> 32));
        if (op == 0u) {
            if (l < c.hi) {
                const uint32_t pos = (uint32_t)((pr >> 32) * (uint64_t)(l + 1u) >> 32);
                for (uint32_t i = l; i > pos; --i) b[i] = b[i - 1];
                b[pos] = ch;
                ++l;
            }
        } else if (op == 1u) {
            if (l > 0u) {
                const uint32_t pos = (uint32_t)((pr >> 32) * (uint64_t)l >> 32);
                for (uint32_t i = pos; i + 1u < l; ++i) b[i] = b[i + 1];
                --l;
            }
        } else if (l > 0u) {
            const uint32_t pos = (uint32_t)((pr >> 32) * (uint64_t)l >> 32);
            b[pos] = ch;
        }
    }
}

} // namespace synth
