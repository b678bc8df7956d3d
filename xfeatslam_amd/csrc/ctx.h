// ctx.h -- the extractor/matcher context: all device memory is allocated once in xfh_create.
#pragma once
#include "common.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <string>

// packed device weights
struct DevWeights {
    float* direct[3] = {nullptr, nullptr, nullptr};            // layers 0..2: [ky][kx][ci][co]
    float* mfma[XFH_NUM_LAYERS] = {};                          // layers 3..22: [chunk][n][CB], k-permuted
    float* alt2[XFH_NUM_LAYERS] = {};                          // third packing: three taps per chunk (small-batch configurations)
    float* m16[XFH_NUM_LAYERS] = {};                           // k_conv_mfma16 packing (single-frame form of the 3x3 stride-1 layers with >= 64 channels): channels permuted in groups of 16
    float* alt[XFH_NUM_LAYERS] = {};                           // second packing of some layers (32-channel chunks / all taps in one chunk), see launch_basic_layer
    float* bn_bias[XFH_NUM_LAYERS] = {};                       // XFH_BN_RUNNING_FOLDED: -running_mean * rstd per output channel (rstd is folded into the weights)
    float* fus2 = nullptr;                                     // block_fusion.2 packed like an MFMA layer
    float* fus2_bias = nullptr;                                // [64]
    float* skip_w = nullptr; float* skip_b = nullptr;          // [24] each
    float* heat2_w = nullptr; float* heat2_b = nullptr;        // [64], [1]
    float* kp3_w = nullptr; float* kp3_b = nullptr;            // [64][68] (k-major, 65 used), [65]
    bool loaded = false;
};

struct XfhComm;
struct xfh_ctx;

// xfh_extract_batch / _submit / _wait (host-visible frames in, host-visible records out; pipeline.cpp).  A call of
// B frames is cut into sub-batches of cfg.max_batch frames.  One sub-batch: on the ctx itself, in order on its stream (H2D, kernels, D2H, an event).
// More: the sub-batches go into ONE queue that up to XFH_PIPE_MAX_LANES worker lanes drain.  A lane = a child ctx (own activations and streams,
// the parent's weights) + a copy stream + a host thread that drives, per sub-batch, copy in -> wait -> kernels -> wait -> copy out -> wait: no copy
// command ever sits in a stream in front of a kernel and no stream waits for a COPY on the GPU (the only event waits left are the fork / join of a lane's own
// keypoint branch onto its second kernel stream for sub-batches above 8 frames, kernel to kernel); the lanes overlap each other.  pipeline.cpp says why.
// Cost: the first submit of more than one sub-batch builds min(sub-batches, xfh_pipeline_lanes) lanes, each a FULL ctx with activations for
// cfg.max_batch frames (29 MB per VGA frame: 1.9 GB per lane at max_batch 64) and a host thread; XFH_ERR_OUT_OF_MEMORY surfaces at that submit.
#define XFH_PIPE_MAX_LANES 8
#define XFH_PIPE_MAX_BATCHES 8                   // xfh_extract_batch_submit calls outstanding (include/xfeat_hip.h: XFH_MAX_BATCHES_INFLIGHT)
struct PipeShared;                               // queue, mutex, condition variables, per-slot counters (pipeline.cpp)
struct PipeLane {
    xfh_ctx* ctx = nullptr;                      // child ctx of the lane
    hipStream_t copy = nullptr;                  // its copy stream (both directions)
    void* thread = nullptr;                      // std::thread*
};
struct Pipe {
    int nlanes = 0, max_lanes = 6;               // 512 VGA frames per step on one box: 26.6 k frames/s with 4 lanes, 29.3 k with 6, 29.2 k with 8 (profiles/r04_host_batch_probe.log)
    PipeLane lane[XFH_PIPE_MAX_LANES];
    PipeShared* sh = nullptr;
    hipEvent_t inline_done[XFH_PIPE_MAX_BATCHES] = {};   // one-sub-batch submits run on the ctx itself: their last download
    bool slot_inline[XFH_PIPE_MAX_BATCHES] = {};
    bool inline_busy = false;
    int b_head = 0, b_count = 0;                 // ring of outstanding submits, oldest first
};

struct xfh_ctx {
    xfh_config cfg;
    XfhComm* comm = nullptr;        // RCCL communicator + communication stream (comm.cpp), created by xfh_comm_create
    int Hmax = 0, Wmax = 0;         // resized maxima (multiples of 32)
    int num_cu = 256;               // compute units of the device (workgroups of the persistent match GEMM)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // own_stream or an external one
    hipStream_t aux_stream = nullptr;                  // keypoint branch of run_extract (forked / joined with the two events)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_out = nullptr;                       // "everything queued on the ctx stream so far", for another ctx' communicator (xfh_comm_wait_ctx)
    std::string hip_err;

    DevWeights w;

    // geometry of the last call
    int B = 0, H0 = 0, W0 = 0, H = 0, W = 0;
    int lh[XFH_NUM_LAYERS] = {}, lw[XFH_NUM_LAYERS] = {};     // output dims per layer
    int npart[XFH_NUM_LAYERS] = {};                            // stat partials per frame per layer

    // device buffers, all [max_batch][...]
    uint8_t* d_gray = nullptr;                  // staging of host frames
    float* X = nullptr;                         // [H][W]
    double* pre_part = nullptr; int pre_npart = 0;
    float* xstat = nullptr;                     // [B][2]
    float* raw[XFH_NUM_LAYERS] = {};
    double* part[XFH_NUM_LAYERS] = {};          // [B][npart][C][2]
    float* stat[XFH_NUM_LAYERS] = {};           // [B][2*C]
    size_t raw_stride[XFH_NUM_LAYERS] = {};     // floats per frame (at max size)
    size_t part_stride[XFH_NUM_LAYERS] = {};    // doubles per frame
    float* skip_pool = nullptr; float* feats = nullptr;
    float* act4 = nullptr; float* act5 = nullptr;              // relu(bn(x4)), relu(bn(x5)) for block_fusion.0 at batches > 32 (k_act_pyramid)
    float* H1 = nullptr; float* K1h = nullptr;
    bool no_nms_heat = false;                   // XFH_NO_NMS_HEAT=1 (tests)
    bool no_ride = false;                       // XFH_NO_RIDE=1: no riders (tests)
    bool select_legacy = false;                 // XFH_SELECT_LEGACY=1: k_select takes its radix-select + bitonic form (tests)
    float* feat_nrm = nullptr;                  // [h8*w8] L2 norm of every feature pixel (k_feat_norm)
    u64* cand = nullptr; size_t cand_cap = 0;   // keys per frame (power of two >= Hmax*Wmax)
    int* cand_count = nullptr;                  // [B]
    int* slot_src = nullptr;                    // [B][nfeatures]
    u64* sel_key = nullptr;                     // [B][nfeatures]
    int* sel_n = nullptr;                       // [B]
    uint8_t* d_records = nullptr;               // [B][record_bytes]: record buffer 0 of the batch pipeline (xfh_extract_batch, pipeline.cpp)
    uint8_t* h_records = nullptr;               // pinned, ONE record: slot 0 of the submit / collect ring
    uint8_t* h_gray = nullptr;                  // pinned, ONE frame: slot 0 of the submit / collect ring

    // xfh_extract_submit / _collect: a ring of SLOTS single-frame submissions, collected in order.  Slot buffers: pinned
    // image, device image, and a pinned record that the kernels write DIRECTLY (host memory is device visible: no D2H
    // command, the stores cross PCIe while k_desc runs).  Slot 0 runs on this ctx; slot 1 runs on `twin`, a second
    // single-frame ctx (own activations, own streams, shared weights; created at the first overlapping submit), so two
    // frames in flight really execute side by side -- one frame's 40-workgroup kernels leave most of the 256 CUs idle.
    static const int SLOTS = 2;
    uint8_t* s_hgray[SLOTS] = {}; uint8_t* s_dgray[SLOTS] = {}; uint8_t* s_hrec[SLOTS] = {};
    hipEvent_t s_done[SLOTS] = {};
    xfh_ctx* twin = nullptr;
    bool is_twin = false;                       // a twin does not own its weights
    int s_head = 0, s_count = 0;                // oldest outstanding slot, number outstanding

    Pipe pipe;                                  // xfh_extract_batch: lanes of the host-visible batch pipeline (pipeline.cpp)
    bool is_lane = false;                       // a lane ctx borrows the weights of its parent, like a twin

    MatchWs mws;
    KTimer timer;
};

// ---- tracing (the reference has none; these are this build's equivalents) ---------------------------------------
// XFH_ROCTX=1 in the environment: named roctx ranges around every public phase (librocprofiler-sdk-roctx / libroctx64 opened with
// dlopen, so there is no link-time dependency): `rocprofv3 --marker-trace` then shows xfh:extract / xfh:match / xfh:batch_submit /
// xfh:gather with the kernels under them.  XFH_VERBOSE=1: one stderr line per ctx / lane / communicator created and per weight load.
void xfh_trace_push(const char* name);
void xfh_trace_pop();
bool xfh_verbose();
struct XfhRange { explicit XfhRange(const char* n) { xfh_trace_push(n); } ~XfhRange() { xfh_trace_pop(); } };

// helpers shared by capi.cpp / pipeline.cpp
int ctx_share_weights(xfh_ctx* parent, xfh_ctx* child);      // child borrows the parent's packed weights (and its eval()-mode statistics)
void pipe_destroy(xfh_ctx* c);
int pipe_reshare_weights(xfh_ctx* c);
void pipe_wait_idle(xfh_ctx* c);

// helpers implemented in capi.cpp
// kernel timing: when the timer is armed for (kernel_id, layer) the launch goes through
// hipExtLaunchKernelGGL with a start/stop event pair attached to the dispatch itself, so the
// measured time is the kernel's own begin..end (what rocprofv3 --kernel-trace reports), not the
// gap between two stream markers.
bool ktimer_slot(xfh_ctx* c, int kernel_id, int layer, hipEvent_t* e0, hipEvent_t* e1);
template <typename K, typename... Args>
inline void launch_k(xfh_ctx* c, int kernel_id, int layer, K kern, dim3 grid, dim3 block, size_t shmem, Args... args) {
    hipEvent_t e0, e1;
    if (ktimer_slot(c, kernel_id, layer, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)shmem, c->stream, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kern, grid, block, shmem, c->stream, args...);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember per kernel which devices have it
// (one 64-bit mask per call site; device ids < 64)
#define XFH_SET_LDS_ATTR_ONCE(c, kern, bytes)                                                                   \
    do {                                                                                                       \
        static std::atomic<unsigned long long> done_mask_{0ull};   /* several ctx may be driven from several threads */ \
        const unsigned long long bit_ = 1ull << ((c)->cfg.device & 63);                                        \
        if (!(done_mask_.load(std::memory_order_acquire) & bit_)) {                                            \
            hipError_t e_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess) return e_;                                                                   \
            done_mask_.fetch_or(bit_, std::memory_order_release);                                              \
        }                                                                                                      \
    } while (0)

// launchers (kernels_*.hip)
bool ride_mode(const xfh_ctx* c, int B);      // batches <= 8, batch statistics: the keypoint branch rides on block1.3 .. block3.0 (kernels_conv.hip)
hipError_t launch_layer_with_rider(xfh_ctx* c, int li, const float* in, size_t in_stride, int src, int pro, int Hin, int Win, int B, const float* K1h, size_t k1h_stride);
hipError_t run_extract(xfh_ctx* c, const uint8_t* d_gray, int B, int H0, int W0, int lap0, int lap1, uint8_t* d_records, bool write_padding = true,
                       float* d_images = nullptr);
