// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that moves an all-gather through POSIX shared memory, so that the
// graph-sharded search (csrc/comm.inc: dr_comm_init, the unique-id hand-off, grouped exchanges, the failure protocol, the bounded wait)
// can run with TWO REAL PROCESSES on a box that has ONE GPU (RCCL itself refuses two ranks on one device). Loaded through the
// library's own hook (DR_RCCL_LIB=<this .so>); exports the six symbols comm.inc looks up. Nothing in the product links or ships it.
//
// Stream semantics kept: ncclAllGather returns at once; on `stream` it queues  D2H(send -> pinned bounce)  ->  host function (bounce ->
// my slot of the shared segment, barrier over the ranks, all slots -> pinned bounce)  ->  H2D(bounce -> recv). Slots are double-buffered
// by the collective's sequence number: one barrier per collective is enough (a rank reaches collective n + 1's write only after every
// rank has reached collective n's barrier, which is stream-ordered behind its reads of collective n - 1).
// Like the real library, a rank whose peer never arrives stays blocked until ITS OWN ncclCommAbort (or until the peer arrives).
// FAKE_RCCL_FAIL_RANK=<r> FAKE_RCCL_FAIL_AT=<n>: rank r's n-th ncclAllGather (0-based) returns ncclInternalError without queuing anything.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {
constexpr size_t SLOT_BYTES = 8u << 20;      // per rank and parity: 32768 queries x k = 32 x 8 bytes fits
constexpr int MAX_RANKS = 16;

struct Shared {
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> arrived[2];        // ranks that reached the barrier of a collective of this parity
    std::atomic<uint32_t> generation[2];     // bumped by the last arriver
    unsigned char data[1];                   // [2][nranks][SLOT_BYTES]
};

struct Op { struct FakeComm *c; size_t bytes; uint64_t seq; };

struct FakeComm {
    int nranks = 1, rank = 0;
    Shared *sh = nullptr;
    size_t map_bytes = 0;
    char name[64] = {};
    unsigned char *bounce = nullptr;         // pinned: [nranks][SLOT_BYTES]
    uint64_t seq = 0;
    int calls = 0;
    std::atomic<int> aborted{0};
    Op ops[64];
};

unsigned char *slot(FakeComm *c, int parity, int r) { return c->sh->data + ((size_t)parity * c->nranks + r) * SLOT_BYTES; }

void host_step(void *arg)
{
    Op *op = static_cast<Op *>(arg);
    FakeComm *c = op->c;
    const int par = (int)(op->seq & 1);
    memcpy(slot(c, par, c->rank), c->bounce + (size_t)c->rank * SLOT_BYTES, op->bytes);
    const uint32_t gen = c->sh->generation[par].load();
    if (c->sh->arrived[par].fetch_add(1) + 1 == (uint32_t)c->nranks) {
        c->sh->arrived[par].store(0);
        c->sh->generation[par].fetch_add(1);
    } else {
        while (c->sh->generation[par].load() == gen && !c->aborted.load()) { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
    }
    for (int r = 0; r < c->nranks; r++) memcpy(c->bounce + (size_t)r * SLOT_BYTES, slot(c, par, r), op->bytes);
}
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/dr_fake_rccl_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    FakeComm *c = new FakeComm();
    c->nranks = nranks; c->rank = rank;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->map_bytes = sizeof(Shared) + 2 * (size_t)nranks * SLOT_BYTES;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared *>(p);            // (a fresh segment is zero-filled: the atomics start at 0)
    if (hipHostMalloc(reinterpret_cast<void **>(&c->bounce), (size_t)nranks * SLOT_BYTES, hipHostMallocDefault) != hipSuccess) { munmap(p, c->map_bytes); delete c; return ncclUnhandledCudaError; }
    // collective like the real call: returns when every rank has attached (60 s at most)
    c->sh->attached.fetch_add(1);
    for (int i = 0; i < 3000000 && c->sh->attached.load() < (uint32_t)nranks; i++) { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
    if (c->sh->attached.load() < (uint32_t)nranks) { (void)hipHostFree(c->bounce); munmap(p, c->map_bytes); delete c; return ncclSystemError; }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t es = (dt == ncclUint64 || dt == ncclInt64 || dt == ncclFloat64) ? 8 : (dt == ncclUint8 || dt == ncclInt8) ? 1 : 4;
    const size_t bytes = count * es;
    if (bytes > SLOT_BYTES) return ncclInvalidArgument;
    const int call = c->calls++;
    const char *fr = getenv("FAKE_RCCL_FAIL_RANK"), *fa = getenv("FAKE_RCCL_FAIL_AT");
    if (fr && fa && atoi(fr) == c->rank && atoi(fa) == call) return ncclInternalError;
    Op *op = &c->ops[c->seq % 64];
    op->c = c; op->bytes = bytes; op->seq = c->seq++;
    if (hipMemcpyAsync(c->bounce + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, host_step, op) != hipSuccess) return ncclUnhandledCudaError;
    for (int r = 0; r < c->nranks; r++)
        if (hipMemcpyAsync(static_cast<unsigned char *>(recv) + (size_t)r * bytes, c->bounce + (size_t)r * SLOT_BYTES, bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    c->aborted.store(1);                         // releases THIS rank's host function, as the real abort ends this rank's kernels
    shm_unlink(c->name);                         // (the mapping stays valid; the segment goes away with the last process)
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    c->aborted.store(1);
    (void)hipDeviceSynchronize();
    shm_unlink(c->name);
    if (c->bounce) (void)hipHostFree(c->bounce);
    if (c->sh) munmap(c->sh, c->map_bytes);
    delete c;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "success";
    case ncclInternalError: return "internal error (fake_rccl: injected)";
    case ncclInvalidArgument: return "invalid argument";
    case ncclSystemError: return "system error";
    default: return "error";
    }
}
}
