// strsim_internal.h -- helpers shared by the C-ABI translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace strsim {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char *what);

} // namespace strsim
