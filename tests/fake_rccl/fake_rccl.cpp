// fake_rccl.cpp — TEST DOUBLE for the ten nccl* entry points libns3d.so resolves with dlopen (ns3d_mgpu.cpp, load_rccl()):
// ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommCount, ncclSend, ncclRecv, ncclAllReduce, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString.  Test infrastructure only (tests/test_gpu_fake_rccl.py points NS3D_RCCL_LIB at it);
// nothing under navierstokes3d_amd/ knows it exists.
//
// Why: RCCL refuses two ranks on one device, and the builder's box has one GPU — so the one-process-per-GPU arm of the
// multi-GPU layer (exchange_begin's send/recv group, gather_impl's send/recv, slab_plan's and the residual's all-reduce)
// would first run on the driver's 8-GPU node.  This library lets SEVERAL PROCESSES ON ONE GPU run that arm for real:
// pairing, grouping, byte counts, in-order matching per peer, stream ordering and the all-reduce — everything but xGMI.
//
// How: a POSIX shared-memory segment per communicator, registered with hipHostRegister in every rank.
//   * send  = hipMemcpyAsync(device → the (src,dst) channel's arena in the segment) on the caller's stream, then a stream
//             host function that publishes the message (sequence number, byte count).
//   * recv  = a stream host function that waits for that message (and checks its byte count against the receive's),
//             then hipMemcpyAsync(arena → device) on the caller's stream, then a host function that frees the arena space.
//   * all-reduce = device → pinned slot, host function {contribute; wait for every rank; reduce}, pinned slot → device.
// Everything is ordered ON THE CALLER'S STREAM exactly like the real library's kernels: work enqueued behind a receive sees
// the data, work enqueued before a send has completed when the bytes leave — and nothing else is synchronised, so a
// missing event wait in the schedule shows up as wrong planes.  Inside ncclGroupStart/End the sends of the group are issued
// before its receives (as RCCL progresses them concurrently), so two ranks that both post {send; recv} to each other cannot
// deadlock; a channel's arena must hold one group's sends (FAKE_RCCL_ARENA_MB, default 48).  Waits give up after
// FAKE_RCCL_TIMEOUT_S (default 60) and abort the process with a message — a test fails, nothing hangs.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

constexpr uint64_t MAGIC = 0x46414b4552434c31ull;      // "FAKERCL1"
constexpr int AR_SLOT_BYTES = 4096;                    // largest all-reduce payload
constexpr int AR_RING = 64;                            // pinned staging slots per communicator (all-reduces in flight)

double timeout_s()
{
    static double t = [] { const char *e = std::getenv("FAKE_RCCL_TIMEOUT_S"); return e ? std::atof(e) : 60.0; }();
    return t;
}
size_t arena_bytes()
{
    static size_t b = [] { const char *e = std::getenv("FAKE_RCCL_ARENA_MB"); return (size_t)(e ? std::atol(e) : 48) << 20; }();
    return b;
}

struct Channel {                                       // one directed pair src → dst
    std::atomic<uint64_t> published;                   // messages the sender has made visible
    std::atomic<uint64_t> freed;                       // arena bytes (with wrap padding) the receiver has released
    std::atomic<uint64_t> size[256];                   // byte count of message i (mod 256)
    char pad[64];
};
struct Header {
    std::atomic<uint64_t> magic;
    std::atomic<int> attached, detached;
    int nranks;
    std::atomic<uint64_t> ar_flag[2][64];              // all-reduce: sequence number per parity and rank
    std::atomic<int> failed;                           // a rank saw a protocol violation
};

struct Comm {
    int nranks = 0, rank = 0;
    char name[64] = {0};
    void *base = nullptr;
    size_t total = 0;
    Header *hdr = nullptr;
    Channel *chan = nullptr;                           // [src * nranks + dst]
    char *ar_data = nullptr;                           // [2][nranks][AR_SLOT_BYTES]
    char *arena = nullptr;                             // [src * nranks + dst][arena_bytes()]
    // per directed pair, this process's view of the byte ring (identical on both ends: same message sizes in the same order)
    std::vector<uint64_t> send_seq, send_alloc, recv_seq, recv_alloc;
    uint64_t ar_seq = 0;
    char *ar_pinned = nullptr;                         // AR_RING slots of AR_SLOT_BYTES, hipHostMalloc
    bool registered = false;
};

struct Op { bool send; void *buf; size_t bytes; int peer; Comm *c; hipStream_t s; };
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;

[[noreturn]] void die(const Comm *c, const char *what)
{
    std::fprintf(stderr, "fake_rccl[rank %d of %d]: %s\n", c ? c->rank : -1, c ? c->nranks : -1, what);
    std::fflush(stderr);
    if (c && c->hdr) c->hdr->failed.store(1);
    std::abort();
}

template <class F>
void spin_until(const Comm *c, const char *what, F cond)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0; !cond(); ++it) {
        if (c->hdr->failed.load()) die(c, "another rank reported a protocol violation");
        if ((it & 1023) == 1023) {
            std::this_thread::yield();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) die(c, what);
        }
    }
}

size_t round64(size_t n) { return (n + 63) & ~(size_t)63; }

// where message of `bytes` lands in a channel's byte ring whose allocation counter stands at `alloc`; advances alloc
size_t place(uint64_t &alloc, size_t bytes)
{
    const size_t A = arena_bytes(), need = round64(bytes);
    size_t off = (size_t)(alloc % A);
    if (off + need > A) { alloc += A - off; off = 0; }     // wrap: the tail is skipped (and counted as used)
    alloc += need;
    return off;
}

struct SendDone { Comm *c; int ch; uint64_t seq; size_t bytes; };
struct RecvWait { Comm *c; int ch; uint64_t seq; size_t bytes; };
struct RecvDone { Comm *c; int ch; uint64_t alloc_end; };
struct SendSpace { Comm *c; int ch; uint64_t alloc_end; };

void cb_send_space(void *p)
{
    SendSpace *a = (SendSpace *)p;
    spin_until(a->c, "send: the receiver never freed arena space (a group sends more than FAKE_RCCL_ARENA_MB to one peer, or the peer never posted its receives)",
               [&] { return a->alloc_end - a->c->chan[a->ch].freed.load(std::memory_order_acquire) <= arena_bytes(); });
    delete a;
}
void cb_send_done(void *p)
{
    SendDone *a = (SendDone *)p;
    Channel &ch = a->c->chan[a->ch];
    ch.size[a->seq & 255].store(a->bytes, std::memory_order_relaxed);
    ch.published.store(a->seq + 1, std::memory_order_release);
    delete a;
}
void cb_recv_wait(void *p)
{
    RecvWait *a = (RecvWait *)p;
    Channel &ch = a->c->chan[a->ch];
    spin_until(a->c, "recv: no matching send arrived (pairing or ordering of ncclSend/ncclRecv differs between the ranks)",
               [&] { return ch.published.load(std::memory_order_acquire) > a->seq; });
    const uint64_t got = ch.size[a->seq & 255].load(std::memory_order_relaxed);
    if (got != a->bytes) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "recv of %zu bytes matched a send of %llu bytes (message %llu of the pair)", a->bytes,
                      (unsigned long long)got, (unsigned long long)a->seq);
        die(a->c, msg);
    }
    delete a;
}
void cb_recv_done(void *p)
{
    RecvDone *a = (RecvDone *)p;
    a->c->chan[a->ch].freed.store(a->alloc_end, std::memory_order_release);
    delete a;
}

ncclResult_t hipfail(const Comm *c, hipError_t e, const char *what)
{
    std::fprintf(stderr, "fake_rccl[rank %d]: %s: %s\n", c ? c->rank : -1, what, hipGetErrorString(e));
    return ncclUnhandledCudaError;
}
#define HIPOK(c, expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return hipfail((c), e_, #expr);                                                \
    } while (0)

ncclResult_t issue(const Op &o)
{
    Comm *c = o.c;
    if (o.peer < 0 || o.peer >= c->nranks || o.peer == c->rank) return ncclInvalidArgument;
    if (round64(o.bytes) > arena_bytes()) {
        std::fprintf(stderr, "fake_rccl: a message of %zu bytes exceeds the channel arena (FAKE_RCCL_ARENA_MB)\n", o.bytes);
        return ncclInvalidArgument;
    }
    if (o.send) {
        const int ch = c->rank * c->nranks + o.peer;
        const size_t off = place(c->send_alloc[o.peer], o.bytes);
        HIPOK(c, hipLaunchHostFunc(o.s, cb_send_space, new SendSpace{c, ch, c->send_alloc[o.peer]}));
        if (o.bytes) HIPOK(c, hipMemcpyAsync(c->arena + (size_t)ch * arena_bytes() + off, o.buf, o.bytes, hipMemcpyDeviceToHost, o.s));
        HIPOK(c, hipLaunchHostFunc(o.s, cb_send_done, new SendDone{c, ch, c->send_seq[o.peer]++, o.bytes}));
    } else {
        const int ch = o.peer * c->nranks + c->rank;
        const size_t off = place(c->recv_alloc[o.peer], o.bytes);
        HIPOK(c, hipLaunchHostFunc(o.s, cb_recv_wait, new RecvWait{c, ch, c->recv_seq[o.peer]++, o.bytes}));
        if (o.bytes) HIPOK(c, hipMemcpyAsync(o.buf, c->arena + (size_t)ch * arena_bytes() + off, o.bytes, hipMemcpyHostToDevice, o.s));
        HIPOK(c, hipLaunchHostFunc(o.s, cb_recv_done, new RecvDone{c, ch, c->recv_alloc[o.peer]}));
    }
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

struct ArArgs { Comm *c; uint64_t seq; char *slot; size_t count; ncclDataType_t type; ncclRedOp_t op; };

template <class T>
void reduce_typed(ArArgs *a)
{
    Comm *c = a->c;
    const int par = (int)(a->seq & 1);
    T *out = (T *)a->slot;
    for (int r = 0; r < c->nranks; ++r) {                       // same order on every rank: identical results everywhere
        const T *in = (const T *)(c->ar_data + ((size_t)par * c->nranks + r) * AR_SLOT_BYTES);
        for (size_t i = 0; i < a->count; ++i) {
            if (r == 0) { out[i] = in[i]; continue; }
            switch (a->op) {
            case ncclSum: out[i] = (T)(out[i] + in[i]); break;
            case ncclProd: out[i] = (T)(out[i] * in[i]); break;
            case ncclMax: out[i] = in[i] > out[i] ? in[i] : out[i]; break;
            case ncclMin: out[i] = in[i] < out[i] ? in[i] : out[i]; break;
            default: die(c, "all-reduce: unsupported reduction operator");
            }
        }
    }
}
void cb_allreduce(void *p)
{
    ArArgs *a = (ArArgs *)p;
    Comm *c = a->c;
    const int par = (int)(a->seq & 1);
    const size_t bytes = a->count * type_bytes(a->type);
    std::memcpy(c->ar_data + ((size_t)par * c->nranks + c->rank) * AR_SLOT_BYTES, a->slot, bytes);
    c->hdr->ar_flag[par][c->rank].store(a->seq + 1, std::memory_order_release);
    spin_until(c, "all-reduce: a rank never arrived (the ranks do not call their collectives in the same order)", [&] {
        for (int r = 0; r < c->nranks; ++r)
            if (c->hdr->ar_flag[par][r].load(std::memory_order_acquire) < a->seq + 1) return false;
        return true;
    });
    switch (a->type) {
    case ncclUint64: reduce_typed<uint64_t>(a); break;
    case ncclInt64: reduce_typed<int64_t>(a); break;
    case ncclUint32: reduce_typed<uint32_t>(a); break;
    case ncclInt32: reduce_typed<int32_t>(a); break;
    case ncclFloat64: reduce_typed<double>(a); break;
    case ncclFloat32: reduce_typed<float>(a); break;
    default: die(c, "all-reduce: unsupported element type");
    }
    delete a;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    unsigned long long r[2] = {0, 0};
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { if (read(fd, r, sizeof r) != (ssize_t)sizeof r) r[0] = (unsigned long long)getpid(); close(fd); }
    std::snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%016llx%016llx", (int)getpid(), r[0], r[1]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (std::strncmp(id.internal, "/fake_rccl_", 11) != 0) return ncclInvalidArgument;     // an id of the real library
    Comm *c = new Comm();
    c->nranks = nranks; c->rank = rank;
    std::snprintf(c->name, sizeof c->name, "%.63s", id.internal);
    const size_t nch = (size_t)nranks * nranks;
    const size_t off_chan = round64(sizeof(Header)), off_ar = off_chan + nch * sizeof(Channel);
    const size_t off_arena = (off_ar + 2ull * nranks * AR_SLOT_BYTES + 4095) & ~(size_t)4095;
    c->total = off_arena + nch * arena_bytes();
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { perror("fake_rccl: shm_open"); delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->total) != 0) { perror("fake_rccl: ftruncate"); close(fd); delete c; return ncclSystemError; }
    c->base = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);      // fresh segments read as zeros
    close(fd);
    if (c->base == MAP_FAILED) { perror("fake_rccl: mmap"); delete c; return ncclSystemError; }
    c->hdr = (Header *)c->base;
    c->chan = (Channel *)((char *)c->base + off_chan);
    c->ar_data = (char *)c->base + off_ar;
    c->arena = (char *)c->base + off_arena;
    c->send_seq.assign(nranks, 0); c->send_alloc.assign(nranks, 0); c->recv_seq.assign(nranks, 0); c->recv_alloc.assign(nranks, 0);
    if (rank == 0) { c->hdr->nranks = nranks; c->hdr->magic.store(MAGIC, std::memory_order_release); }
    // only the arenas this rank touches are registered (a channel is used by its two ends): sends of `rank`, receives of `rank`
    hipError_t e = hipHostRegister(c->base, c->total, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); std::fprintf(stderr, "fake_rccl: hipHostRegister(%zu MB): %s — copies will be staged by the runtime\n", c->total >> 20, hipGetErrorString(e)); }
    else c->registered = true;
    if (hipHostMalloc((void **)&c->ar_pinned, (size_t)AR_RING * AR_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return ncclUnhandledCudaError;
    }
    c->hdr->attached.fetch_add(1);
    spin_until(c, "ncclCommInitRank: not every rank attached", [&] {
        return c->hdr->magic.load(std::memory_order_acquire) == MAGIC && c->hdr->attached.load() >= nranks;
    });
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return ncclInvalidArgument;
    (void)hipDeviceSynchronize();                       // nothing of this communicator is still queued on a stream
    const int gone = c->hdr->detached.fetch_add(1) + 1;
    if (gone == c->nranks) shm_unlink(c->name);
    if (c->registered) (void)hipHostUnregister(c->base);
    if (c->ar_pinned) (void)hipHostFree(c->ar_pinned);
    munmap(c->base, c->total);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((const Comm *)comm)->nranks;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++g_group;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_group <= 0) return ncclInvalidUsage;
    if (--g_group > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    ncclResult_t rc = ncclSuccess;
    for (int pass = 0; pass < 2 && rc == ncclSuccess; ++pass)       // every send of the group before its receives
        for (const Op &o : ops)
            if (o.send == (pass == 0) && (rc = issue(o)) != ncclSuccess) break;
    return rc;
}

static ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    const size_t es = type_bytes(t);
    if (!c || !es || (!buf && count)) return ncclInvalidArgument;
    if (c->hdr->failed.load()) return ncclRemoteError;
    const Op o{send, buf, count * es, peer, c, s};
    if (g_group > 0) { g_ops.push_back(o); return ncclSuccess; }
    return issue(o);
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(true, (void *)sendbuff, count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return p2p(false, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_bytes(datatype);
    if (!c || !sendbuff || !recvbuff || !bytes || bytes > (size_t)AR_SLOT_BYTES) return ncclInvalidArgument;
    if (g_group > 0) return ncclInvalidUsage;          // not needed by libns3d; keeps the double simple
    if (c->hdr->failed.load()) return ncclRemoteError;
    const uint64_t seq = c->ar_seq++;
    char *slot = c->ar_pinned + (size_t)(seq % AR_RING) * AR_SLOT_BYTES;
    HIPOK(c, hipMemcpyAsync(slot, sendbuff, bytes, hipMemcpyDeviceToHost, stream));
    HIPOK(c, hipLaunchHostFunc(stream, cb_allreduce, new ArArgs{c, seq, slot, count, datatype, op}));
    HIPOK(c, hipMemcpyAsync(recvbuff, slot, bytes, hipMemcpyHostToDevice, stream));
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclSystemError: return "unhandled system error (fake_rccl)";
    case ncclInternalError: return "internal error (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    case ncclRemoteError: return "remote error (fake_rccl)";
    default: return "unknown result code (fake_rccl)";
    }
}

// so that a test can tell the double from the real library
int fake_rccl_marker(void) { return 1; }

} // extern "C"
