// rccl_abi_check.cpp -- TEST INFRASTRUCTURE (compiled with -fsyntax-only by tests/test_gather_abi.py, never linked or shipped):
// the hand-written declarations of csrc/strsim_rccl_abi.h held against the real header of the RCCL in this image.
#include <rccl/rccl.h>

#include <type_traits>

#include "strsim_rccl_abi.h"

namespace {

template <class A, class B> constexpr bool same_slot()
{ // what the calling convention sees of one parameter: its size, its alignment, pointer or not, aggregate (memory class) or not
    return sizeof(A) == sizeof(B) && alignof(A) == alignof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value &&
           std::is_class<A>::value == std::is_class<B>::value &&
           (std::is_integral<A>::value || std::is_enum<A>::value) == (std::is_integral<B>::value || std::is_enum<B>::value);
}
template <class F, class G> struct same_shape : std::false_type {};
template <class R1, class... A, class R2, class... B> struct same_shape<R1 (*)(A...), R2 (*)(B...)> {
    template <bool Same, class Dummy = void> struct args { static constexpr bool value = false; };
    template <class Dummy> struct args<true, Dummy> { static constexpr bool value = (same_slot<A, B>() && ...); };
    static constexpr bool value = same_slot<R1, R2>() && args<sizeof...(A) == sizeof...(B)>::value;
};
#define SAME(ours, theirs) static_assert(same_shape<strsim_rccl::ours, decltype(&theirs)>::value, #ours " does not match " #theirs)

SAME(GetUniqueIdFn, ncclGetUniqueId);
SAME(CommInitRankFn, ncclCommInitRank);
SAME(CommDestroyFn, ncclCommDestroy);
SAME(SendFn, ncclSend);
SAME(RecvFn, ncclRecv);
SAME(GroupFn, ncclGroupStart);
SAME(GroupFn, ncclGroupEnd);
SAME(GetErrorStringFn, ncclGetErrorString);
SAME(CommCountFn, ncclCommCount);
static_assert(strsim_rccl::ID_BYTES == NCCL_UNIQUE_ID_BYTES && sizeof(strsim_rccl::UniqueId) == sizeof(ncclUniqueId), "unique id");
static_assert(strsim_rccl::SUCCESS == (int)ncclSuccess && strsim_rccl::FLOAT64 == (int)ncclFloat64, "constants");
// (the self-test of the checker: a declaration with a parameter missing, or an id of another size, must NOT pass)
static_assert(!same_shape<int (*)(const void *, size_t, int, int, strsim_rccl::Comm), decltype(&ncclSend)>::value, "checker is blind");
struct ShortId { char internal[64]; };
static_assert(!same_shape<int (*)(strsim_rccl::Comm *, int, ShortId, int), decltype(&ncclCommInitRank)>::value, "checker is blind");

} // namespace
