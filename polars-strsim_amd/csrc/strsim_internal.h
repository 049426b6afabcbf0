// strsim_internal.h -- helpers shared by the C-ABI translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace strsim {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char *what);

// 64 KB of device scratch owned by the context (strsim_capi.cpp), for strsim_offsets_from_lengths' block sums
constexpr size_t SCAN_WS_WORDS = 16384;

} // namespace strsim

struct strsim_ctx;
extern "C" int strsim_internal_scan_workspace(strsim_ctx *ctx, uint32_t **p); // (hidden visibility: not exported)
extern "C" int strsim_internal_ctx_device(strsim_ctx *ctx);                   // the HIP device ordinal of a context
