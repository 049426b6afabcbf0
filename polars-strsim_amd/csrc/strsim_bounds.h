// strsim_bounds.h -- LAB ONLY: address checks for the kernels (make EXTRA="-DSTRSIM_LAB -DSTRSIM_BOUNDS"; the product Makefile
// never sets either, and without them every macro below expands to nothing: the product library is bit-identical with and
// without this header).
//
// Why: round 4 shipped a GPU memory fault (a cooperative fetch that ran with lanes switched off and handed out null addresses)
// that only fresh fuzz seeds found, and an out-of-range read that lands in mapped memory is silent.  GPU AddressSanitizer is
// not available on the pool, so the lab build checks the addresses itself: every global and LDS address the kernels FORM -- also
// in lanes that a partial EXEC mask is about to switch off -- is compared with the extents the launch was given (the columns'
// byte ranges by the over-read contract of include/strsim_amd.h, the offsets arrays, the output column, the mask, the LDS arrays),
// and a violation is counted in a device record with the first offender's (kernel, site, row, value, bounds).  Read back with
// strsim_debug_bounds() (tests/fuzz_gpu.py does after every round); bench_support/jobs/r5_bounds_fuzz.sh is the soak.
#pragma once
#include <stdint.h>

#if defined(STRSIM_LAB) && defined(STRSIM_BOUNDS)
#include <hip/hip_runtime.h>
namespace strsim {
namespace bounds {
struct Record {
    unsigned long long hits;   // violations since the last read
    unsigned long long site;   // first one: kernel << 32 | site
    unsigned long long row;    //   the row (or block / chunk) it belongs to, ~0 if none
    unsigned long long value;  //   the offending address / index
    unsigned long long lo, hi; //   the inclusive range it had to lie in
};
static __device__ Record g_rec; // (one per translation unit: no relocatable device code in this build)
enum Kernel : uint32_t { K_STAGE = 1, K_LIT = 2, K_WIDE = 3, K_UTF8 = 4, K_WAVE = 5, K_VIEWS = 6 };
__device__ __forceinline__ void hit(uint32_t kernel, uint32_t site, unsigned long long row, unsigned long long v, unsigned long long lo,
                                    unsigned long long hi)
{
    if (atomicAdd(&g_rec.hits, 1ull) == 0ull) {
        g_rec.site = ((unsigned long long)kernel << 32) | site;
        g_rec.row = row;
        g_rec.value = v;
        g_rec.lo = lo;
        g_rec.hi = hi;
    }
}
// [p, p + bytes) inside [base, base + extent)?
__device__ __forceinline__ void span(uint32_t kernel, uint32_t site, unsigned long long row, unsigned long long p, unsigned long long bytes,
                                     unsigned long long base, unsigned long long extent)
{
    if (p < base || p + bytes > base + extent) hit(kernel, site, row, p, base, base + extent - (bytes < extent ? bytes : extent));
}
// the bytes a kernel may read of a column whose strings lie in [val + first, val + total): the 16-byte chunks they touch
__device__ __forceinline__ void column(uint32_t kernel, uint32_t site, unsigned long long row, const void *p, unsigned long long bytes,
                                       const void *val, unsigned long long first, unsigned long long total)
{
    const unsigned long long lo = ((unsigned long long)(uintptr_t)val + first) & ~15ull;
    const unsigned long long hi = ((unsigned long long)(uintptr_t)val + total + 15ull) & ~15ull;
    span(kernel, site, row, (unsigned long long)(uintptr_t)p, bytes, lo, hi > lo ? hi - lo : 0ull);
}
} // namespace bounds
} // namespace strsim
#define STRSIM_BOUNDS_ON 1
// index / address v inside [lo, hi] (inclusive)?
#define STRSIM_CHECK_RANGE(kernel, site, row, v, lo, hi)                                                              \
    do {                                                                                                              \
        const unsigned long long v__ = (unsigned long long)(v), lo__ = (unsigned long long)(lo), hi__ = (unsigned long long)(hi); \
        if (v__ < lo__ || v__ > hi__) ::strsim::bounds::hit(::strsim::bounds::kernel, site, (unsigned long long)(row), v__, lo__, hi__); \
    } while (0)
// element index i of an array of n elements
#define STRSIM_CHECK_INDEX(kernel, site, row, i, n) STRSIM_CHECK_RANGE(kernel, site, row, i, 0, (unsigned long long)(n) - 1ull)
// [p, p + bytes) inside the buffer [base, base + extent)
#define STRSIM_CHECK_SPAN(kernel, site, row, p, bytes, base, extent)                                                  \
    ::strsim::bounds::span(::strsim::bounds::kernel, site, (unsigned long long)(row), (unsigned long long)(uintptr_t)(p), (unsigned long long)(bytes), \
                           (unsigned long long)(uintptr_t)(base), (unsigned long long)(extent))
// [p, p + bytes) inside the 16-byte chunks a column's strings touch (include/strsim_amd.h: reads beyond the strings)
#define STRSIM_CHECK_COLUMN(kernel, site, row, p, bytes, val, first, total)                                           \
    ::strsim::bounds::column(::strsim::bounds::kernel, site, (unsigned long long)(row), p, (unsigned long long)(bytes), val, (unsigned long long)(first), (unsigned long long)(total))
#else
#define STRSIM_BOUNDS_ON 0
#define STRSIM_CHECK_RANGE(kernel, site, row, v, lo, hi) do { } while (0)
#define STRSIM_CHECK_INDEX(kernel, site, row, i, n) do { } while (0)
#define STRSIM_CHECK_SPAN(kernel, site, row, p, bytes, base, extent) do { } while (0)
#define STRSIM_CHECK_COLUMN(kernel, site, row, p, bytes, val, first, total) do { } while (0)
#endif
