// comm.cpp -- the one exchange step of the path behind the C ABI: frames are sharded over GPUs
// (frame i -> rank i mod R, one ctx and one process per GPU) and the fixed-size records travel to rank 0, where the
// sequential SLAM state machine lives (reference: one process, src/System.cc:197-233), with RCCL over xGMI:
//   xfh_allgather_records     ncclAllGather of B records per rank (what BASELINE.json's configs[3] names)
//   xfh_gather_records_root   ncclSend / ncclRecv to one root only (the other GPUs receive nothing)
//   xfh_gather_compact_root   the same with only header + valid rows on the wire (sizes exchanged first)
// librccl is opened with dlopen at xfh_comm_create, so the library has no link-time dependency on it.  WHICH librccl is deterministic and
// recorded (round 6): $XFH_RCCL_LIB if set (the tests' stand-in), else /opt/rocm/lib/librccl.so.1 -- the RCCL of the ROCm this library was built
// with -- else whatever "librccl.so.1" / "librccl.so" resolve to; xfh_comm_library() names the file (dladdr of ncclAllGather), its ncclGetVersion and the
// HIP runtime it calls, and xfh_comm_create refuses an RCCL that would run on ANOTHER HIP runtime than this library's streams live in (a process that
// imported PyTorch first carries the wheel's own libamdhip64 + librccl: a stream handle of one runtime means nothing to the other).  Collectives run on a communication stream of the
// ctx, ordered after the extraction by an event, so the next extraction overlaps them; xfh_comm_fence orders a later
// extraction after the collective that read a record buffer.
#include "ctx.h"
#include <dlfcn.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {
typedef int ncclResult_t;                          // rccl.h: ncclSuccess == 0
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };       // NCCL_UNIQUE_ID_BYTES
enum { ncclUint8 = 1 };                            // rccl.h ncclDataType_t

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string path, hip_of_rccl, hip_of_lib, describe;
    int version = 0, hip_ver_rccl = 0, hip_ver_lib = 0;
};
// path of the shared object a function lives in ("?" if the loader does not know)
static std::string so_of(void* fn) {
    Dl_info di;
    if (fn && dladdr(fn, &di) && di.dli_fname) { char buf[4096]; const char* rp = realpath(di.dli_fname, buf); return rp ? rp : di.dli_fname; }
    return "?";
}
Rccl* rccl() {
    static Rccl* R = []() -> Rccl* {               // thread-safe one-time initialisation (C++11 static)
        static Rccl r;
        const char* forced = getenv("XFH_RCCL_LIB");
        if (forced && *forced) r.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);          // explicit: no fallback behind it
        else for (const char* n : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) { r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.h) break; }
        if (!r.h) return nullptr;
#define SYM(f, s) do { *(void**)(&r.f) = dlsym(r.h, s); if (!r.f) return nullptr; } while (0)
        SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
        SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
        SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        *(void**)(&r.GetVersion) = dlsym(r.h, "ncclGetVersion");                     // optional (the tests' stand-in has none)
        r.path = so_of((void*)r.AllGather);
        int v = 0;
        if (r.GetVersion && r.GetVersion(&v) == 0) r.version = v;
        // the HIP runtime this RCCL calls = the one its own symbol lookup finds, against the one this library's streams belong to
        void* rccl_hip = dlsym(r.h, "hipRuntimeGetVersion");
        r.hip_of_rccl = rccl_hip ? so_of(rccl_hip) : "?";
        r.hip_of_lib = so_of((void*)&hipRuntimeGetVersion);
        int (*rv)(int*) = nullptr; *(void**)(&rv) = rccl_hip;
        int a = 0, b = 0;
        if (rv && rv(&a) == 0) r.hip_ver_rccl = a;
        if (hipRuntimeGetVersion(&b) == hipSuccess) r.hip_ver_lib = b;
        char buf[8192];
        snprintf(buf, sizeof buf, "%s (RCCL %d.%d.%d; its HIP runtime %d.%d at %s; libxfeat_hip's HIP runtime %d.%d at %s)", r.path.c_str(),
                 r.version / 10000, (r.version / 100) % 100, r.version % 100, r.hip_ver_rccl / 10000000, (r.hip_ver_rccl / 100000) % 100, r.hip_of_rccl.c_str(),
                 r.hip_ver_lib / 10000000, (r.hip_ver_lib / 100000) % 100, r.hip_of_lib.c_str());
        r.describe = buf;
        return &r;
    }();
    return R;
}
// an RCCL bound to another HIP runtime than ours (another file, or another major.minor) cannot take our streams
static bool rccl_hip_mismatch(const Rccl* r) {
    if (r->hip_of_rccl == "?" || r->hip_ver_rccl == 0) return false;            // a stand-in without HIP inside (tests): nothing to compare
    return r->hip_of_rccl != r->hip_of_lib || r->hip_ver_rccl / 100000 != r->hip_ver_lib / 100000;
}
}  // namespace

struct XfhComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;                  // communication stream
    hipEvent_t ev_ready = nullptr;                 // extraction done -> collective may start
    hipEvent_t ev_done[2] = {nullptr, nullptr};    // collective of buffer generation g finished
    bool used[2] = {false, false};
    unsigned long long* h_sizes = nullptr;         // pinned: compact sizes of all ranks
    unsigned long long* d_sizes = nullptr;         // [world] + [1] own
    uint8_t* d_pack = nullptr; size_t cap_pack = 0;
    int* d_prefix = nullptr; int cap_prefix = 0;   // [frames + 1] rows before each frame of a compact shard; grown on demand (several ctx may feed one gather)
};

#define HIPCK(c, x) do { hipError_t _e = (x); if (_e != hipSuccess) { (c)->hip_err = std::string(#x) + ": " + hipGetErrorString(_e); return XFH_ERR_HIP; } } while (0)
#define NCK(c, x) do { ncclResult_t _r = (x); if (_r != 0) { (c)->hip_err = std::string(#x) + ": " + rccl()->GetErrorString(_r); return XFH_ERR_COMM; } } while (0)

// header of a packed shard (xfh_gather_compact_root): B frames, then per frame its RecordHeader, then all valid
// keypoints (28 B each, frame after frame, front segment then back segment), then all valid descriptors (256 B each)
__global__ void k_pack_offsets(const uint8_t* __restrict__ rec, size_t rec_bytes, int B, int nf, unsigned long long* __restrict__ total,
                               int* __restrict__ prefix /* [B+1] rows before frame b */) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) { prefix[b] = run; int nv = ((const RecordHeader*)(rec + (size_t)b * rec_bytes))->n_valid; run += nv < 0 ? 0 : (nv > nf ? nf : nv); }
        prefix[B] = run;
        *total = 256 + (((size_t)B * 16 + 255) & ~(size_t)255) + (((size_t)run * 28 + 255) & ~(size_t)255) + (size_t)run * 256;
    }
}
__global__ __launch_bounds__(256)
void k_pack_rows(const uint8_t* __restrict__ rec, size_t rec_bytes, int B, int nf, size_t kps_off, size_t desc_off,
                 const int* __restrict__ prefix, uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const uint8_t* r = rec + (size_t)b * rec_bytes;
    const RecordHeader h = *(const RecordHeader*)r;
    const int nv = h.n_valid < 0 ? 0 : (h.n_valid > nf ? nf : h.n_valid), front = h.mono_index < 0 ? 0 : (h.mono_index > nv ? nv : h.mono_index);
    const int total = prefix[B];
    uint8_t* o_hdr = out + 256;
    uint8_t* o_kps = o_hdr + (((size_t)B * 16 + 255) & ~(size_t)255);
    uint8_t* o_desc = o_kps + (((size_t)total * 28 + 255) & ~(size_t)255);
    if (blockIdx.x == 0 && threadIdx.x < 4) ((int*)(o_hdr + (size_t)b * 16))[threadIdx.x] = ((const int*)r)[threadIdx.x];
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0) { ((int*)out)[0] = B; ((int*)out)[1] = nf; ((int*)out)[2] = total; }
    // one wave per valid row: row j of the compact order = slot j (front) or slot nf - nv + j (back)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = blockIdx.x * 4 + wave; j < nv; j += gridDim.x * 4) {
        const int slot = j < front ? j : nf - nv + j;
        const size_t dst = (size_t)(prefix[b] + j);
        ((float*)(o_desc + dst * 256))[lane] = ((const float*)(r + desc_off + (size_t)slot * 256))[lane];
        if (lane < 7) ((int*)(o_kps + dst * 28))[lane] = ((const int*)(r + kps_off + (size_t)slot * 28))[lane];
    }
}

extern "C" {

// which librccl the exchange runs on: "<file> (RCCL a.b.c; its HIP runtime x.y at <file>; libxfeat_hip's HIP runtime x.y at <file>)"; "" if none can be loaded
const char* xfh_comm_library(void) {
    Rccl* R = rccl();
    return R ? R->describe.c_str() : "";
}

int xfh_comm_unique_id(void* id_out) {
    if (!id_out) return XFH_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R) return XFH_ERR_COMM;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != 0) return XFH_ERR_COMM;
    memcpy(id_out, &id, sizeof id);
    return XFH_OK;
}

int xfh_comm_destroy(xfh_ctx* c) {
    if (!c) return XFH_ERR_INVALID_ARG;
    XfhComm* m = c->comm;
    if (!m) return XFH_OK;
    hipSetDevice(c->cfg.device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->comm && rccl()) rccl()->CommDestroy(m->comm);
    if (m->ev_ready) hipEventDestroy(m->ev_ready);
    for (int g = 0; g < 2; ++g) if (m->ev_done[g]) hipEventDestroy(m->ev_done[g]);
    if (m->h_sizes) hipHostFree(m->h_sizes);
    if (m->d_sizes) hipFree(m->d_sizes);
    if (m->d_pack) hipFree(m->d_pack);
    if (m->d_prefix) hipFree(m->d_prefix);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    c->comm = nullptr;
    return XFH_OK;
}

int xfh_comm_create(xfh_ctx* c, const void* unique_id, int rank, int world) {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world || c->comm) return XFH_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R) { c->hip_err = "librccl not found ($XFH_RCCL_LIB, /opt/rocm/lib/librccl.so.1, librccl.so.1, librccl.so)"; return XFH_ERR_COMM; }
    if (rccl_hip_mismatch(R) && !(getenv("XFH_RCCL_ALLOW_HIP_MISMATCH") && *getenv("XFH_RCCL_ALLOW_HIP_MISMATCH") == '1')) {
        c->hip_err = "RCCL runs on another HIP runtime than libxfeat_hip: " + R->describe + " -- set XFH_RCCL_LIB to an RCCL of this runtime";
        fprintf(stderr, "[xfh] %s\n", c->hip_err.c_str());
        return XFH_ERR_COMM;
    }
    HIPCK(c, hipSetDevice(c->cfg.device));
    XfhComm* m = new XfhComm();
    m->rank = rank; m->world = world;
    c->comm = m;
    auto bail = [&](int code) { xfh_comm_destroy(c); return code; };
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) return bail(XFH_ERR_HIP);
    if (hipEventCreateWithFlags(&m->ev_ready, hipEventDisableTiming) != hipSuccess) return bail(XFH_ERR_HIP);
    for (int g = 0; g < 2; ++g) if (hipEventCreateWithFlags(&m->ev_done[g], hipEventDisableTiming) != hipSuccess) return bail(XFH_ERR_HIP);
    if (hipHostMalloc((void**)&m->h_sizes, sizeof(unsigned long long) * (world + 1), hipHostMallocDefault) != hipSuccess) return bail(XFH_ERR_OUT_OF_MEMORY);
    if (hipMalloc((void**)&m->d_sizes, sizeof(unsigned long long) * (world + 1)) != hipSuccess) return bail(XFH_ERR_OUT_OF_MEMORY);
    ncclUniqueId id; memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = R->CommInitRank(&m->comm, world, id, rank);
    if (r != 0) { c->hip_err = std::string("ncclCommInitRank: ") + R->GetErrorString(r); return bail(XFH_ERR_COMM); }
    if (xfh_verbose()) fprintf(stderr, "[xfh] ctx %p: communicator rank %d of %d on device %d over %s\n", (void*)c, rank, world, c->cfg.device, R->describe.c_str());
    return XFH_OK;
}

int xfh_comm_rank(xfh_ctx* c) { return c && c->comm ? c->comm->rank : -1; }
int xfh_comm_world(xfh_ctx* c) { return c && c->comm ? c->comm->world : 0; }

// the collective starts when the ctx stream reaches this point (the records are complete)
static int comm_begin(xfh_ctx* c, int gen) {
    XfhComm* m = c->comm;
    XfhRange range("xfh:gather_begin");
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipEventRecord(m->ev_ready, c->stream));
    HIPCK(c, hipStreamWaitEvent(m->stream, m->ev_ready, 0));
    (void)gen;
    return XFH_OK;
}
static int comm_end(xfh_ctx* c, int gen) {
    XfhComm* m = c->comm;
    HIPCK(c, hipEventRecord(m->ev_done[gen], m->stream));
    m->used[gen] = true;
    return XFH_OK;
}

int xfh_allgather_bytes(xfh_ctx* c, const void* d_send, size_t nbytes, void* d_recv, int gen) {
    if (!c || !c->comm || !d_send || !d_recv || gen < 0 || gen > 1) return XFH_ERR_INVALID_ARG;
    int rc = comm_begin(c, gen);
    if (rc != XFH_OK) return rc;
    NCK(c, rccl()->AllGather(d_send, d_recv, nbytes, ncclUint8, c->comm->comm, c->comm->stream));
    return comm_end(c, gen);
}

int xfh_allgather_records(xfh_ctx* c, const void* d_records, int B, void* d_all, int gen) {
    if (!c || B < 1) return XFH_ERR_INVALID_ARG;
    return xfh_allgather_bytes(c, d_records, (size_t)B * xfh_record_bytes(c->cfg.nfeatures), d_all, gen);
}

int xfh_gather_records_root(xfh_ctx* c, const void* d_records, int B, void* d_all, int root, int gen) {
    if (!c || !c->comm || !d_records || B < 1 || gen < 0 || gen > 1) return XFH_ERR_INVALID_ARG;
    XfhComm* m = c->comm;
    if (root < 0 || root >= m->world || (m->rank == root && !d_all)) return XFH_ERR_INVALID_ARG;
    const size_t nb = (size_t)B * xfh_record_bytes(c->cfg.nfeatures);
    int rc = comm_begin(c, gen);
    if (rc != XFH_OK) return rc;
    Rccl* R = rccl();
    if (m->rank == root) {
        NCK(c, R->GroupStart());
        for (int r = 0; r < m->world; ++r)
            if (r != root) NCK(c, R->Recv((uint8_t*)d_all + (size_t)r * nb, nb, ncclUint8, r, m->comm, m->stream));
        NCK(c, R->GroupEnd());
        HIPCK(c, hipMemcpyAsync((uint8_t*)d_all + (size_t)root * nb, d_records, nb, hipMemcpyDeviceToDevice, m->stream));
    } else {
        NCK(c, R->Send(d_records, nb, ncclUint8, root, m->comm, m->stream));
    }
    return comm_end(c, gen);
}

size_t xfh_compact_bytes_max(int nfeatures, int B) {
    return 256 + (((size_t)B * 16 + 255) & ~(size_t)255) + (((size_t)B * nfeatures * 28 + 255) & ~(size_t)255) + (size_t)B * nfeatures * 256;
}

int xfh_gather_compact_root(xfh_ctx* c, const void* d_records, int B, void* d_all, size_t* shard_bytes, int root, int gen) {
    if (!c || !c->comm || !d_records || B < 1 || gen < 0 || gen > 1) return XFH_ERR_INVALID_ARG;
    XfhComm* m = c->comm;
    if (root < 0 || root >= m->world || (m->rank == root && (!d_all || !shard_bytes))) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int nf = c->cfg.nfeatures;
    const size_t cap = xfh_compact_bytes_max(nf, B), rec = xfh_record_bytes(nf);
    if (m->cap_pack < cap) {
        if (m->d_pack) { HIPCK(c, hipStreamSynchronize(m->stream)); hipFree(m->d_pack); m->d_pack = nullptr; m->cap_pack = 0; }
        HIPCK(c, hipMalloc((void**)&m->d_pack, cap));
        m->cap_pack = cap;
    }
    if (m->cap_prefix < B + 1) {                      // B is the caller's frame count (bench.py: the sub-batches of several ctx), not this ctx' max_batch
        if (m->d_prefix) { HIPCK(c, hipStreamSynchronize(m->stream)); hipFree(m->d_prefix); m->d_prefix = nullptr; m->cap_prefix = 0; }
        HIPCK(c, hipMalloc((void**)&m->d_prefix, sizeof(int) * (size_t)(B + 1)));
        m->cap_prefix = B + 1;
    }
    int rc = comm_begin(c, gen);
    if (rc != XFH_OK) return rc;
    Rccl* R = rccl();
    unsigned long long* d_own = m->d_sizes + m->world;
    int* d_prefix = m->d_prefix;
    hipLaunchKernelGGL(k_pack_offsets, dim3(1), dim3(64), 0, m->stream, (const uint8_t*)d_records, rec, B, nf, d_own, d_prefix);
    hipLaunchKernelGGL(k_pack_rows, dim3(64, B), dim3(256), 0, m->stream, (const uint8_t*)d_records, rec, B, nf, xfh_record_kps_offset(), xfh_record_desc_offset(nf),
                       (const int*)d_prefix, m->d_pack);
    HIPCK(c, hipGetLastError());
    // sizes first (8 bytes per rank), read back on every rank: send / recv counts must be known to the host
    NCK(c, R->AllGather(d_own, m->d_sizes, sizeof(unsigned long long), ncclUint8, m->comm, m->stream));
    HIPCK(c, hipMemcpyAsync(m->h_sizes, m->d_sizes, sizeof(unsigned long long) * m->world, hipMemcpyDeviceToHost, m->stream));
    HIPCK(c, hipStreamSynchronize(m->stream));
    for (int r = 0; r < m->world; ++r) if (m->h_sizes[r] > cap) { c->hip_err = "compact shard larger than its bound"; return XFH_ERR_COMM; }
    if (m->rank == root) {
        NCK(c, R->GroupStart());
        for (int r = 0; r < m->world; ++r) {
            shard_bytes[r] = (size_t)m->h_sizes[r];
            if (r != root) NCK(c, R->Recv((uint8_t*)d_all + (size_t)r * cap, (size_t)m->h_sizes[r], ncclUint8, r, m->comm, m->stream));
        }
        NCK(c, R->GroupEnd());
        HIPCK(c, hipMemcpyAsync((uint8_t*)d_all + (size_t)root * cap, m->d_pack, (size_t)m->h_sizes[root], hipMemcpyDeviceToDevice, m->stream));
    } else {
        NCK(c, R->Send(m->d_pack, (size_t)m->h_sizes[m->rank], ncclUint8, root, m->comm, m->stream));
    }
    return comm_end(c, gen);
}

int xfh_unpack_compact(const void* shard, size_t nbytes, int frame, int nfeatures, xfh_keypoint* kps_out, float* desc_out, int* n_valid, int* mono_index) {
    // host-side reader of one frame of a packed shard (rank 0 after a D2H copy): restores the padded nfeatures-row form
    if (!shard || nbytes < 256 || !kps_out || !desc_out) return XFH_ERR_INVALID_ARG;
    const uint8_t* p = (const uint8_t*)shard;
    const int B = ((const int*)p)[0], nf = ((const int*)p)[1], total = ((const int*)p)[2];
    if (nf != nfeatures || frame < 0 || frame >= B || total < 0) return XFH_ERR_INVALID_ARG;
    const uint8_t* hdr = p + 256;
    const uint8_t* kps = hdr + (((size_t)B * 16 + 255) & ~(size_t)255);
    const uint8_t* desc = kps + (((size_t)total * 28 + 255) & ~(size_t)255);
    if ((size_t)(desc - p) + (size_t)total * 256 > nbytes) return XFH_ERR_INVALID_ARG;
    int before = 0;
    for (int b = 0; b < frame; ++b) { const int nv = ((const int*)(hdr + (size_t)b * 16))[0]; before += nv < 0 ? 0 : (nv > nf ? nf : nv); }
    const int* h = (const int*)(hdr + (size_t)frame * 16);
    const int nv = h[0] < 0 ? 0 : (h[0] > nf ? nf : h[0]), front = h[1] < 0 ? 0 : (h[1] > nv ? nv : h[1]), back = nv - front;
    if (before + nv > total) return XFH_ERR_INVALID_ARG;
    const xfh_keypoint dk = {0.f, 0.f, 0.f, -1.f, 0.f, 0, -1};
    memcpy(kps_out, kps + (size_t)before * 28, (size_t)front * 28);
    memcpy(desc_out, desc + (size_t)before * 256, (size_t)front * 256);
    for (int i = front; i < nf - back; ++i) kps_out[i] = dk;
    memset(desc_out + (size_t)front * 64, 0, (size_t)(nf - nv) * 256);
    memcpy(kps_out + (nf - back), kps + (size_t)(before + front) * 28, (size_t)back * 28);
    memcpy(desc_out + (size_t)(nf - back) * 64, desc + (size_t)(before + front) * 256, (size_t)back * 256);
    if (n_valid) *n_valid = h[0];
    if (mono_index) *mono_index = h[1];
    return XFH_OK;
}

int xfh_comm_fence(xfh_ctx* c, int gen) {
    if (!c || !c->comm || gen < 0 || gen > 1) return XFH_ERR_INVALID_ARG;
    if (c->comm->used[gen]) HIPCK(c, hipStreamWaitEvent(c->stream, c->comm->ev_done[gen], 0));
    return XFH_OK;
}

// several ctx feed one communicator (sub-batches of a step, each on its own ctx): the collective also waits for `other`'s
// stream, and `other` does not overwrite generation `gen` before the collective that read it has finished -- both without
// a host synchronisation and without tying the ctx streams to each other
int xfh_comm_wait_ctx(xfh_ctx* c, xfh_ctx* other) {
    if (!c || !c->comm || !other || other->cfg.device != c->cfg.device) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipEventRecord(other->ev_out, other->stream));
    HIPCK(c, hipStreamWaitEvent(c->comm->stream, other->ev_out, 0));
    return XFH_OK;
}
int xfh_comm_fence_ctx(xfh_ctx* c, xfh_ctx* other, int gen) {
    if (!c || !c->comm || !other || gen < 0 || gen > 1 || other->cfg.device != c->cfg.device) return XFH_ERR_INVALID_ARG;
    if (c->comm->used[gen]) HIPCK(c, hipStreamWaitEvent(other->stream, c->comm->ev_done[gen], 0));
    return XFH_OK;
}

int xfh_comm_synchronize(xfh_ctx* c) {
    if (!c || !c->comm) return XFH_ERR_INVALID_ARG;
    HIPCK(c, hipStreamSynchronize(c->comm->stream));
    return XFH_OK;
}

}  // extern "C"
