// pipe_probe.hip -- compiles k_lane_pipe alone (register budget / ISA inspection; not part of the product library)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ipolars-strsim_amd/csrc -Iinclude \
//         -Rpass-analysis=kernel-resource-usage -save-temps -c bench_support/micro/pipe_probe.hip -o /tmp/pipe_probe.o
#include <hip/hip_runtime.h>
#include "strsim_kernels.h"
#include "strsim_lane_common.h"
namespace strsim {
#include "strsim_lane_pipe.h"
#ifndef PROBE_M
#define PROBE_M 0
#endif
#ifndef PROBE_LIT
#define PROBE_LIT false
#endif
template __global__ void k_lane_pipe<PROBE_M, PROBE_LIT>(const uint32_t *, const uint8_t *, uint64_t, const uint32_t *, const uint8_t *, uint64_t,
                                        double *, uint64_t, unsigned long long *, DevStatus *, const double *);
}
