// rccl_stub.cpp -- TEST-ONLY stand-in for librccl.so.1 so that the multi-rank code of xfeatslam_amd/csrc/comm.cpp
// (ncclAllGather, the ncclSend / ncclRecv group with r * nb offsets, the compact size exchange) can run with MORE THAN ONE
// RANK on a box that has ONE GPU: real RCCL refuses two ranks on one device.  Built into tests/stubs/librccl.so.1 and
// injected through the dlopen search comm.cpp already performs (LD_LIBRARY_PATH of the worker processes).  Never shipped,
// never linked into libxfeat_hip.so; the product talks to the real librccl on a multi-GPU node.
//
// Transport: one mmap'ed file in /tmp per communicator (named by the unique id), one mailbox per ordered rank pair, messages
// cut into chunks; a collective first synchronises the caller's stream (so the producers have finished), then moves the bytes
// with blocking hipMemcpy D2H / H2D through the mailboxes while polling all of its pending sends and receives round-robin
// (so groups and all-gathers of any size cannot dead-lock), and returns when its own part is complete.  That is a legal
// (if slow) implementation of the stream-ordered contract: everything queued on the stream afterwards sees the result.
// Every wait has a deadline and returns ncclSystemError instead of hanging.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {
enum { kSuccess = 0, kUnhandledCuda = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
const int MAXW = 8;
const size_t CHUNK = 1u << 20;
const double DEADLINE_S = 60.0;

struct Box {
    std::atomic<unsigned long long> written;      // chunks published by the sender
    std::atomic<unsigned long long> consumed;     // chunks taken by the receiver
    unsigned long long msg_bytes;                 // total size of the message the current chunk belongs to (checked by the receiver)
    unsigned long long chunk_bytes;
    char pad[64 - 32];
};
struct Shm {
    std::atomic<int> attached;
    std::atomic<int> detached;
    char pad[56];
    Box box[MAXW][MAXW];                          // [src][dst]
    // followed by MAXW * MAXW chunks of CHUNK bytes (sparse: only the pairs in use are ever touched)
};
size_t shm_bytes() { return sizeof(Shm) + (size_t)MAXW * MAXW * CHUNK; }

struct Comm {
    Shm* shm = nullptr;
    int rank = 0, world = 1, device = 0;
    char path[160];
    char* data(int src, int dst) { return (char*)shm + sizeof(Shm) + ((size_t)src * MAXW + dst) * CHUNK; }
};

struct Op { bool send; int peer; char* dptr; size_t total, done; Comm* comm; hipStream_t stream; bool finished; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

size_t dtype_size(int dt) {
    switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}

// one attempt to move the next chunk of an operation; true if progress was made
bool progress(Op& o, int* err) {
    Comm* c = o.comm;
    if (o.send) {
        Box& b = c->shm->box[c->rank][o.peer];
        if (b.written.load(std::memory_order_acquire) != b.consumed.load(std::memory_order_acquire)) return false;     // mailbox still full
        const size_t n = o.total - o.done < CHUNK ? o.total - o.done : CHUNK;
        if (n && hipMemcpy(c->data(c->rank, o.peer), o.dptr + o.done, n, hipMemcpyDeviceToHost) != hipSuccess) { *err = kUnhandledCuda; return false; }
        b.msg_bytes = o.total; b.chunk_bytes = n;
        b.written.fetch_add(1, std::memory_order_release);
        o.done += n;
        o.finished = o.done == o.total;                       // a zero-byte message still takes one (empty) chunk
        return true;
    }
    Box& b = c->shm->box[o.peer][c->rank];
    if (b.written.load(std::memory_order_acquire) == b.consumed.load(std::memory_order_acquire)) return false;         // nothing there yet
    const size_t want = o.total;
    if (b.msg_bytes != want) { fprintf(stderr, "rccl_stub: rank %d expects %zu bytes from rank %d, which sends %llu\n", c->rank, want, o.peer, b.msg_bytes); *err = kInvalidUsage; return false; }
    const size_t n = (size_t)b.chunk_bytes;
    if (n && hipMemcpy(o.dptr + o.done, c->data(o.peer, c->rank), n, hipMemcpyHostToDevice) != hipSuccess) { *err = kUnhandledCuda; return false; }
    b.consumed.fetch_add(1, std::memory_order_release);
    o.done += n;
    o.finished = o.done == o.total;
    return true;
}

int run_ops(std::vector<Op>& ops) {
    // the producers of every send buffer (and earlier users of every receive buffer) have finished
    for (size_t i = 0; i < ops.size(); ++i) {
        bool seen = false;
        for (size_t j = 0; j < i; ++j) seen = seen || ops[j].stream == ops[i].stream;
        if (!seen && hipStreamSynchronize(ops[i].stream) != hipSuccess) return kUnhandledCuda;
    }
    const double t_end = now() + DEADLINE_S;
    for (;;) {
        bool all = true, moved = false;
        for (Op& o : ops) {
            if (o.finished) continue;
            all = false;
            int err = 0;
            if (progress(o, &err)) moved = true;
            if (err) return err;
        }
        if (all) return kSuccess;
        if (!moved) {
            if (now() > t_end) { fprintf(stderr, "rccl_stub: rank %d timed out inside a collective\n", ops[0].comm->rank); return kSystemError; }
            usleep(50);
        }
    }
}

int submit(const Op& o) {
    g_ops.push_back(o);
    if (g_depth > 0) return kSuccess;
    std::vector<Op> ops; ops.swap(g_ops);
    return run_ops(ops);
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };
typedef Comm* ncclComm_t;

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof *id);
    unsigned long long r[2] = {0, 0};
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(r, sizeof r, 1, f) != 1) r[0] = (unsigned long long)getpid(); fclose(f); }
    snprintf(id->internal, sizeof id->internal, "/tmp/xfh_rccl_stub_%d_%016llx%016llx", (int)getpid(), r[0], r[1]);
    return kSuccess;
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAXW || rank < 0 || rank >= nranks) return kInvalidArgument;
    if (strncmp(id.internal, "/tmp/xfh_rccl_stub_", 19) != 0) return kInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    snprintf(c->path, sizeof c->path, "%s", id.internal);
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return kUnhandledCuda; }
    const int fd = open(c->path, O_RDWR | O_CREAT, 0600);
    if (fd < 0) { delete c; return kSystemError; }
    if (ftruncate(fd, (off_t)shm_bytes()) != 0) { close(fd); delete c; return kSystemError; }       // zero-filled: every counter starts at 0
    void* p = mmap(nullptr, shm_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return kSystemError; }
    c->shm = (Shm*)p;
    c->shm->attached.fetch_add(1);
    const double t_end = now() + DEADLINE_S;
    while (c->shm->attached.load() < nranks) {
        if (now() > t_end) { fprintf(stderr, "rccl_stub: rank %d: only %d of %d ranks attached\n", rank, c->shm->attached.load(), nranks); munmap(p, shm_bytes()); delete c; return kSystemError; }
        usleep(200);
    }
    if (rank == 0) fprintf(stderr, "rccl_stub: TEST-ONLY librccl stand-in, %d ranks over %s\n", nranks, c->path);
    *out = c;
    return kSuccess;
}

int ncclCommDestroy(ncclComm_t c) {
    if (!c) return kSuccess;
    const bool last = c->shm->detached.fetch_add(1) + 1 == c->world;
    munmap(c->shm, shm_bytes());
    if (last) unlink(c->path);
    delete c;
    return kSuccess;
}

int ncclGroupStart() { ++g_depth; return kSuccess; }
int ncclGroupEnd() {
    if (g_depth <= 0) return kInvalidUsage;
    if (--g_depth > 0) return kSuccess;
    if (g_ops.empty()) return kSuccess;
    std::vector<Op> ops; ops.swap(g_ops);
    return run_ops(ops);
}

int ncclSend(const void* buf, size_t count, int dt, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || !dtype_size(dt) || (!buf && count)) return kInvalidArgument;
    return submit(Op{true, peer, (char*)buf, count * dtype_size(dt), 0, c, s, false});
}
int ncclRecv(void* buf, size_t count, int dt, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || !dtype_size(dt) || (!buf && count)) return kInvalidArgument;
    return submit(Op{false, peer, (char*)buf, count * dtype_size(dt), 0, c, s, false});
}

int ncclAllGather(const void* send, void* recv, size_t count, int dt, ncclComm_t c, hipStream_t s) {
    if (!c || !send || !recv || !dtype_size(dt)) return kInvalidArgument;
    const size_t nb = count * dtype_size(dt);
    if (hipStreamSynchronize(s) != hipSuccess) return kUnhandledCuda;
    char* own = (char*)recv + (size_t)c->rank * nb;
    if (own != (const char*)send && nb && hipMemcpy(own, send, nb, hipMemcpyDeviceToDevice) != hipSuccess) return kUnhandledCuda;
    if (c->world == 1) return kSuccess;
    ++g_depth;
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        g_ops.push_back(Op{true, p, (char*)send, nb, 0, c, s, false});
        g_ops.push_back(Op{false, p, (char*)recv + (size_t)p * nb, nb, 0, c, s, false});
    }
    return ncclGroupEnd();
}

const char* ncclGetErrorString(int r) {
    switch (r) {
        case kSuccess: return "no error";
        case kUnhandledCuda: return "unhandled hip error (rccl_stub)";
        case kSystemError: return "system error / timeout (rccl_stub)";
        case kInvalidArgument: return "invalid argument (rccl_stub)";
        case kInvalidUsage: return "invalid usage: message sizes of a send / recv pair differ (rccl_stub)";
        default: return "internal error (rccl_stub)";
    }
}
}  // extern "C"
