// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product's default path).
//
// A stand-in for librccl.so that lets SEVERAL RANKS SHARE ONE GPU, selected with STRSIM_RCCL_LIB=<this library>: real RCCL refuses
// two ranks on one device ("Duplicate GPU detected"), and the test boxes have one.  It implements the nine entry points
// csrc/strsim_gather.cpp resolves, with the prototypes of the real <rccl/rccl.h> (so a drift of that header breaks this build), over
// a POSIX shared-memory segment: a send is a device-to-host copy into the (src, dst) mailbox, a receive the host-to-device copy out
// of it.  What it proves about the product: strsim_gather_f64 / _f64_ranges post the right sends and receives -- peers, counts,
// `column + offset` addresses, ragged last shard, root != 0, the datatype constant, the by-value unique id -- for N > 1 ranks.  What it
// cannot prove: anything about real RCCL or xGMI (tests/test_gather_abi.py has that test too; it needs two GPUs).
//
// Semantics kept from NCCL: calls between ncclGroupStart / ncclGroupEnd are issued together (no ordering deadlock), operations are
// ordered behind the work already on their stream.  Stronger than NCCL: ncclGroupEnd returns when the data has arrived.
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t BOX_BYTES = 1 << 20;
constexpr double TIMEOUT_S = 120.0;

struct Mailbox {
    std::atomic<uint64_t> written, read; // messages put in / taken out (one message in flight per direction)
    uint64_t bytes;
    alignas(64) unsigned char data[BOX_BYTES];
};
struct Segment {
    std::atomic<int> arrived, departed;
    Mailbox box[MAX_RANKS][MAX_RANKS]; // [src][dst]
};

struct FakeComm {
    Segment *seg = nullptr;
    int world = 0, rank = 0;
    char name[NCCL_UNIQUE_ID_BYTES] = {};
};

struct Op {
    bool send;
    unsigned char *dev;
    size_t left;
    int peer;
    FakeComm *comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

ncclResult_t run_ops()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    for (const Op &o : ops) // ordered behind what the stream already holds (the kernels that produced the shard)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    const double t0 = now();
    size_t open = 0;
    for (const Op &o : ops) open += o.left != 0;
    while (open) {
        bool moved = false;
        for (Op &o : ops) {
            if (!o.left) continue;
            Mailbox &b = o.send ? o.comm->seg->box[o.comm->rank][o.peer] : o.comm->seg->box[o.peer][o.comm->rank];
            const bool full = b.written.load(std::memory_order_acquire) != b.read.load(std::memory_order_acquire);
            if (o.send && !full) {
                const size_t n = o.left < BOX_BYTES ? o.left : BOX_BYTES;
                if (hipMemcpyAsync(b.data, o.dev, n, hipMemcpyDeviceToHost, o.stream) != hipSuccess ||
                    hipStreamSynchronize(o.stream) != hipSuccess)
                    return ncclUnhandledCudaError;
                b.bytes = n;
                b.written.fetch_add(1, std::memory_order_release);
                o.dev += n; o.left -= n; moved = true;
            } else if (!o.send && full) {
                const size_t n = b.bytes;
                if (n > o.left) { fprintf(stderr, "fake_rccl: rank %d received %zu bytes from %d, expected at most %zu\n", o.comm->rank, n, o.peer, o.left); return ncclInvalidUsage; }
                if (hipMemcpyAsync(o.dev, b.data, n, hipMemcpyHostToDevice, o.stream) != hipSuccess ||
                    hipStreamSynchronize(o.stream) != hipSuccess)
                    return ncclUnhandledCudaError;
                b.read.fetch_add(1, std::memory_order_release);
                o.dev += n; o.left -= n; moved = true;
            }
            if (!o.left) --open;
        }
        if (!moved) {
            if (now() - t0 > TIMEOUT_S) { fprintf(stderr, "fake_rccl: timed out with %zu operations open\n", open); return ncclSystemError; }
            std::this_thread::yield();
        }
    }
    return ncclSuccess;
}

ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c || !buf) return ncclInvalidArgument;
    if (dt != ncclFloat64) { fprintf(stderr, "fake_rccl: datatype %d is not ncclFloat64 (%d)\n", (int)dt, (int)ncclFloat64); return ncclInvalidArgument; }
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{send, static_cast<unsigned char *>(buf), count * sizeof(double), peer, c, stream});
    return g_depth ? ncclSuccess : run_ops();
}

} // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<unsigned> seq{0};
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "/strsim_fake_rccl_%ld_%u_%lld", (long)getpid(), seq.fetch_add(1),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (memchr(id.internal, 0, sizeof id.internal) == nullptr || strncmp(id.internal, "/strsim_fake_rccl_", 18) != 0) return ncclInvalidArgument;
    FakeComm *c = new (std::nothrow) FakeComm;
    if (!c) return ncclSystemError;
    c->world = nranks; c->rank = rank;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); // (a fresh segment reads as zeros)
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->seg = static_cast<Segment *>(p);
    c->seg->arrived.fetch_add(1);
    const double t0 = now();
    while (c->seg->arrived.load() < nranks) { // collective, like the real one
        if (now() - t0 > TIMEOUT_S) { munmap(p, sizeof(Segment)); shm_unlink(c->name); delete c; return ncclSystemError; }
        std::this_thread::yield();
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c) return ncclInvalidArgument;
    const bool last = c->seg->departed.fetch_add(1) + 1 == c->world;
    munmap(c->seg, sizeof(Segment));
    if (last) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    const FakeComm *c = reinterpret_cast<const FakeComm *>(comm);
    if (!c || !count) return ncclInvalidArgument;
    *count = c->world;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(true, const_cast<void *>(buf), count, dt, peer, comm, stream);
}

__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(false, buf, count, dt, peer, comm, stream);
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth ? ncclSuccess : run_ops();
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclUnhandledCudaError: return "a HIP call failed (fake_rccl)";
    case ncclSystemError: return "shared memory or time-out (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "error (fake_rccl)";
    }
}

} // extern "C"
