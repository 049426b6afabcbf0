// stage_probe.hip -- compiles k_lane_stage alone (register budget / ISA inspection; not part of the product library)
#include <hip/hip_runtime.h>
#include "strsim_kernels.h"
#include "strsim_lane_common.h"
namespace strsim {
constexpr int ALL_MEASURES = 5;
struct OutPtrs { double *p[5]; };
#include "strsim_lane_stage.h"
#ifndef PROBE_M
#define PROBE_M 0
#endif
#if PROBE_M != 5
template __global__ void k_lane_stage<PROBE_M>(const uint32_t *, const uint8_t *, uint64_t, const uint32_t *, const uint8_t *, uint64_t,
                                        OutPtrs, uint64_t, unsigned long long *, DevStatus *, const double *, uint32_t *, DevStatus *);
#endif
}
