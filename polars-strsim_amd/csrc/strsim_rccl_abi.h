// strsim_rccl_abi.h -- the slice of RCCL's C ABI that strsim_gather.cpp resolves with dlsym, declared by hand: the library must build
// (and load) without RCCL's header or shared object (csrc/strsim_gather.cpp says why).  RCCL keeps NCCL's ABI: enums travel as int,
// a communicator is a pointer, the unique id is a 128-byte struct passed BY VALUE.  tests/cpu_harness/rccl_abi_check.cpp holds these
// declarations against /opt/rocm/include/rccl/rccl.h (sizes, alignment, arity, pointer-ness of every parameter, the two constants).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace strsim_rccl {

constexpr int ID_BYTES = 128; // NCCL_UNIQUE_ID_BYTES (== STRSIM_GATHER_ID_BYTES)
typedef struct { char internal[ID_BYTES]; } UniqueId;
typedef void *Comm;
constexpr int SUCCESS = 0;  // ncclSuccess
constexpr int FLOAT64 = 8;  // ncclFloat64

using GetUniqueIdFn = int (*)(UniqueId *);
using CommInitRankFn = int (*)(Comm *, int, UniqueId, int);
using CommDestroyFn = int (*)(Comm);
using SendFn = int (*)(const void *, size_t, int, int, Comm, hipStream_t);
using RecvFn = int (*)(void *, size_t, int, int, Comm, hipStream_t);
using GroupFn = int (*)();
using GetErrorStringFn = const char *(*)(int);
using CommCountFn = int (*)(const Comm, int *);

} // namespace strsim_rccl
