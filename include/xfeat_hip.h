/*
 * xfeat_hip.h -- C ABI of libxfeat_hip.so: the MI355X (gfx950) XFeat feature-extraction
 * and descriptor-matching front end for xfeatSLAM.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it
 * replaces (paths relative to udaysankar01/xfeatSLAM).  Plain pointers and sizes only;
 * no C++ or torch types; nothing throws or aborts across this ABI -- every call returns
 * an xfh_status (0 = OK) except where noted.  A ctx is single-caller (one HIP stream per
 * ctx), exactly like the reference's XFextractor object (Tracking.h:265); create one ctx
 * per GPU for multi-GPU use.  xfh_descriptor_distance is stateless and thread-safe like
 * the static ORBmatcher::DescriptorDistance.
 *
 * The C++ wrappers that restore the reference's class surface on top of this ABI are
 * include/xfeat/XFextractor.h and include/xfeat/ORBmatcher_xfeat.h; INTEGRATION.md shows
 * the edits a maintainer makes in the reference tree.
 */
#ifndef XFEAT_HIP_H
#define XFEAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xfh_ctx xfh_ctx;

typedef enum {
    XFH_OK = 0,
    XFH_ERR_INVALID_ARG = 1,
    XFH_ERR_EMPTY_IMAGE = 2,      /* reference returns -1 (XFextractor.cc:253-254)            */
    XFH_ERR_BAD_SIZE = 3,         /* image smaller than 32x32 or larger than the ctx maximum  */
    XFH_ERR_NO_WEIGHTS = 4,       /* extract called before xfh_load_weights                   */
    XFH_ERR_BAD_WEIGHTS = 5,      /* blob magic / tensor table / shapes wrong                 */
    XFH_ERR_HIP = 6,              /* a HIP runtime call failed; see xfh_last_hip_error        */
    XFH_ERR_NO_DEVICE = 7,        /* device ordinal absent or not gfx950: the library never falls back */
    XFH_ERR_OUT_OF_MEMORY = 8,
    XFH_ERR_BATCH_TOO_LARGE = 9,
    XFH_ERR_IO = 10,
    XFH_ERR_COMM = 11             /* RCCL missing or a collective failed; see xfh_last_hip_error */
} xfh_status;

/* mirrors cv::KeyPoint field for field (28 bytes): pt.x, pt.y, size, angle, response,
 * octave, class_id.  Written keypoints are KeyPoint(x, y, 1, -1, score) and unwritten
 * slots are the default cv::KeyPoint() -- XFextractor.cc:312,329 */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} xfh_keypoint;

/* BatchNorm behaviour.  BATCH_STATS (default) reproduces the reference: the module is never put in
 * eval(), so every BasicLayer normalises with the statistics of the current frame;
 * statistics are always per frame, also in batched calls.  RUNNING_STATS is the upstream-XFeat
 * eval() behaviour: the running_mean / running_var buffers of the weight file are used instead (the
 * blob must carry them, otherwise xfh_load_weights returns XFH_ERR_BAD_WEIGHTS); the InstanceNorm of
 * the input image is per frame in all modes.
 * RUNNING_FOLDED is the same eval() network with every BatchNorm folded into the preceding convolution when the weights
 * are loaded (W' = W * rstd, b' = -mean * rstd, ReLU in the epilogue): no statistics are computed or applied at run time.
 * Folding re-rounds the weights, so this mode equals RUNNING_STATS to ~1e-6, not bit for bit. */
enum { XFH_BN_BATCH_STATS = 0, XFH_BN_RUNNING_STATS = 1, XFH_BN_RUNNING_FOLDED = 2 };

typedef struct {
    int32_t device;        /* HIP device ordinal                                              */
    int32_t max_height;    /* largest input image accepted (before the /32 resize)            */
    int32_t max_width;
    int32_t nfeatures;     /* rows of every output (XFextractor ctor arg, Tracking.cc:597)    */
    int32_t max_batch;     /* frames per xfh_extract_batch* call                              */
    int32_t bn_mode;       /* XFH_BN_*                                                        */
    float nms_threshold;   /* 0.05 in the reference (XFextractor.cc:277)                      */
    int32_t flags;         /* XFH_FLAG_*; 0 = the reference's behaviour                        */
    int32_t reserved[7];
} xfh_config;

/* flags.  XFH_FLAG_RESCALE_KEYPOINTS: report keypoints in INPUT-image coordinates, x * (W/W32), y * (H/H32) in fp32
 * as upstream XFeat does.  The reference multiplies by a Long-typed factor, i.e. by 1 (XFextractor.cc:304-305,
 * SURVEY.md Q2): for inputs whose sides are not multiples of 32 its keypoints stay in the resized frame. */
#define XFH_FLAG_RESCALE_KEYPOINTS 1
/* XFH_FLAG_SERIAL_BRANCH: run the keypoint-head branch on the ctx stream instead of the ctx's second stream (same results;
 * no two kernels overlap, so a profiler's per-launch durations are the kernels' own: profiles/r02_roofline_table.md). */
#define XFH_FLAG_SERIAL_BRANCH 2

/* fills the defaults: device 0, 480x640, nfeatures 4096, max_batch 1, threshold 0.05 */
void xfh_config_default(xfh_config* cfg);

/* XFextractor::XFextractor (XFextractor.cc:75-149) minus the host-only scale tables,
 * which live in the C++ wrapper.  Allocates all device memory up front. */
int xfh_create(const xfh_config* cfg, xfh_ctx** out);
int xfh_destroy(xfh_ctx* ctx);

/* Weight loading (replaces InputArchive::load_from + model->load, XFextractor.cc:133-137).
 * blob format: xfeatslam_amd/weights.py ("XFHWGT01" + tensor table + fp32 OIHW data). */
int xfh_load_weights(xfh_ctx* ctx, const void* blob, size_t nbytes);
int xfh_load_weights_file(xfh_ctx* ctx, const char* path);

/* ---- extraction: XFextractor::operator() (XFextractor.cc:250-356) -------------------
 * gray: H x W CV_8UC1 in host memory, stride_bytes between rows (the reference assumes a
 * dense Mat, :166).  lap_x0/lap_x1 = vLappingArea (Frame.cc:311 passes {0,0}, :495
 * {0,1000}).  kps_out: nfeatures records, desc_out: nfeatures x 64 floats row-major; both
 * fully written (padding = default KeyPoint / zero rows).  *mono_index is operator()'s
 * return value, *n_valid the number of keypoints with score > 0 that were written. */
int xfh_extract(xfh_ctx* ctx, const uint8_t* gray, int H, int W, int stride_bytes, int lap_x0, int lap_x1,
                xfh_keypoint* kps_out, float* desc_out, int* n_valid, int* mono_index);

/* Split form of xfh_extract: submit copies the image into a pinned staging buffer of the ctx and
 * enqueues H2D and the kernels, which write the record straight into pinned host memory, then returns; collect waits
 * for the oldest submission and unpacks it (only the valid rows are copied, the padding is filled on the host).  Up to
 * XFH_MAX_INFLIGHT submissions may be outstanding per ctx (a further submit returns XFH_ERR_INVALID_ARG until one is
 * collected), so frame t+1 can be uploaded and computed while the caller still copies out / tracks frame t; or submit the
 * right image of a stereo pair on a second ctx. */
#define XFH_MAX_INFLIGHT 2
int xfh_extract_submit(xfh_ctx* ctx, const uint8_t* gray, int H, int W, int stride_bytes, int lap_x0, int lap_x1);
int xfh_extract_collect(xfh_ctx* ctx, xfh_keypoint* kps_out, float* desc_out, int* n_valid, int* mono_index);

/* the upstream name of the same call (README.md:9, xfeat_cpp `detectAndCompute`) */
int xfh_detect_and_compute(xfh_ctx* ctx, const uint8_t* gray, int H, int W, int stride_bytes,
                           int lap_x0, int lap_x1, xfh_keypoint* kps_out, float* desc_out,
                           int* n_valid, int* mono_index);

/* One output record per frame, device or host resident, fixed size (the reference output
 * is already padded to nfeatures rows -- SURVEY.md Q3):
 *   int32 n_valid, mono_index, n_candidates, reserved;
 *   xfh_keypoint kps[nfeatures];  float desc[nfeatures*64];                              */
size_t xfh_record_bytes(int nfeatures);
size_t xfh_record_kps_offset(void);
size_t xfh_record_desc_offset(int nfeatures);

/* B dense frames [B][H][W] u8 in HOST memory -> B records in HOST memory: the batched form of operator()'s contract (host image
 * in, host keypoints / descriptors out, XFextractor.cc:250-356; SURVEY.md 8d "host-visible").  B is NOT limited by cfg.max_batch:
 * the call is cut into sub-batches of cfg.max_batch frames.  One sub-batch (B <= cfg.max_batch) runs on the ctx itself, in order on its stream.
 * More go into one queue that up to xfh_pipeline_lanes (default 6) internal lanes drain: a lane = a child ctx (own activations and HIP streams,
 * the ctx' weights, built at the first call that needs it) + a copy stream + a host THREAD of the library that drives one sub-batch at a time --
 * copy in, kernels, copy out, each waited for on the host -- so that no copy command ever sits in a stream in front of a kernel and no stream
 * waits for a copy on the GPU; the lanes overlap each other (NOTES.md 6: 0.97-0.98 of the device-resident rate, 0.74-0.85 with in-order streams).
 * Cost: every lane is a full child ctx -- activations for cfg.max_batch frames (about 29 MB per VGA frame, i.e. 1.9 GB per lane at max_batch 64) --
 * plus one host thread; they are built at the first submit that spans more than one sub-batch (min(sub-batches, lanes) of them), and that submit is
 * where XFH_ERR_OUT_OF_MEMORY surfaces if the device cannot hold them.  xfh_pipeline_lanes(ctx, n) bounds the number before that call.
 * gray / records_out should be pinned (xfh_host_alloc, or the caller's own buffers through xfh_host_register): pageable
 * memory works, but the runtime then stages every copy and the stages serialise.
 *   xfh_extract_batch_submit  returns when everything is queued; both buffers must stay untouched until the batch is complete.
 *                             Up to XFH_MAX_BATCHES_INFLIGHT submits may be outstanding (one more: XFH_ERR_INVALID_ARG); their
 *                             sub-batches simply queue up on the lanes, so a consumer that double-buffers records_out keeps
 *                             the GPU busy across calls: submit(t + 1); wait() -> batch t is complete; ...
 *   xfh_extract_batch_wait    the OLDEST outstanding submit is complete in its records_out (XFH_ERR_INVALID_ARG if none is)
 *   xfh_extract_batch_drain   every submit so far is complete
 *   xfh_extract_batch         = submit + drain.
 *                             A submit that FAILS has waited for whatever part of it was already queued: nothing of it is in flight when
 *                             the error comes back, and the buffers are the caller's again.
 * These calls and the single-frame ring (xfh_extract_submit) share the ctx' first frame buffer (a one-sub-batch submit runs on the ctx itself): a
 * batch call while a single-frame submission is outstanding returns XFH_ERR_INVALID_ARG, and so does xfh_extract (it would collect the OLDER
 * submission's result).  The other direction needs no guard: xfh_extract_submit while batches are outstanding queues on the ctx' own stream BEHIND a
 * one-sub-batch submit (same stream, in order), and the lanes of larger submits have buffers of their own.  The ctx stays single-caller: the worker
 * threads touch the lanes only, never the ctx' own buffers or streams. */
#define XFH_MAX_BATCHES_INFLIGHT 8
int xfh_extract_batch(xfh_ctx* ctx, const uint8_t* gray, int B, int H, int W, int lap_x0, int lap_x1,
                      void* records_out);
int xfh_extract_batch_submit(xfh_ctx* ctx, const uint8_t* gray, int B, int H, int W, int lap_x0, int lap_x1,
                             void* records_out);
int xfh_extract_batch_wait(xfh_ctx* ctx);
int xfh_extract_batch_drain(xfh_ctx* ctx);
int xfh_pipeline_lanes(xfh_ctx* ctx, int lanes);      /* 1 .. 8 (default 6); sub-batches in flight side by side, one worker thread each */
/* pinned host memory for the calls above (hipHostMalloc / hipHostRegister behind the ABI, for host languages without a HIP binding) */
int xfh_host_alloc(void** p, size_t nbytes);
int xfh_host_free(void* p);
int xfh_host_register(void* p, size_t nbytes);
int xfh_host_unregister(void* p);
/* same with DEVICE pointers (frames resident in HBM, records stay in HBM for the matcher or
 * an RCCL all-gather); asynchronous on the ctx stream -- call xfh_synchronize to wait. */
int xfh_extract_batch_device(xfh_ctx* ctx, const uint8_t* d_gray, int B, int H, int W, int lap_x0,
                             int lap_x1, void* d_records_out);
/* the same, and each frame's descriptor block is also written as the matcher's PREPARED IMAGE (see xfh_match_prepare_device):
 * d_images_out holds B images of xfh_match_image_bytes(nfeatures) bytes, image b = what xfh_match_prepare_device makes of the
 * nfeatures descriptor rows of record b, bit for bit (padding slots are rows of zeros).  The frame-to-frame match of the tracker
 * (frame t against t-1) is then xfh_match_mnn_prepared_device on two of these images with n1 = n2 = nfeatures: two launches, no
 * normalisation pass over descriptors that k_desc has only just written. */
int xfh_extract_batch_device_images(xfh_ctx* ctx, const uint8_t* d_gray, int B, int H, int W, int lap_x0,
                                    int lap_x1, void* d_records_out, void* d_images_out);

/* ---- matching ----------------------------------------------------------------------
 * ORBmatcher::match (declared ORBmatcher.h:77; its definition is commented out at
 * ORBmatcher.cc:340-405; these are the semantics of that code with float descriptors):
 * rows L2-normalised, cosine similarity, mutual nearest neighbours, first maximum wins
 * ties, matches in ascending idx1 order, dist = sqrt(2 (1 - cos)).  min_cossim <= 0
 * disables the gate as the reference does (:361).  idx1/idx2/dist need min(n1,n2) slots.
 * Descriptors must be finite.  Rows holding NaN or Inf are outside the contract: the calls still return (a bounded wait, never a
 * hang) and the pairs among finite rows that do not compete with a poisoned row are unaffected, but which partner a poisoned row
 * gets -- the reference's torch::max would propagate the NaN -- is unspecified (the match GEMM is built with -fno-honor-nans).
 * The extraction never produces such rows. */
int xfh_match_mnn(xfh_ctx* ctx, const float* d1, int n1, const float* d2, int n2, float min_cossim,
                  int* idx1, int* idx2, float* dist, int* n_matches);
/* device-resident variant: d1/d2 device pointers (e.g. the desc block of two records),
 * outputs device pointers; *d_n_matches is one int in device memory.  Asynchronous. */
int xfh_match_mnn_device(xfh_ctx* ctx, const float* d_d1, int n1, const float* d_d2, int n2,
                         float min_cossim, int* d_idx1, int* d_idx2, float* d_dist, int* d_n_matches);

/* Prepared descriptor sets (device-resident hand-off).  A tracker matches every frame against several
 * others (previous frame, key frames, loop candidates): xfh_match_prepare_device normalises the n x 64 rows once
 * (F::normalize, ORBmatcher.cc:358-359) and stores them as the "panel image" the GEMM kernel reads
 * (xfh_match_image_bytes(n) bytes of device memory owned by the caller); xfh_match_mnn_prepared_device then runs
 * ORBmatcher::match on two images with the same results as xfh_match_mnn_device on the rows they were made from,
 * in two kernel launches instead of three.  Device pointers, asynchronous on the ctx stream. */
size_t xfh_match_image_bytes(int n);
int xfh_match_prepare_device(xfh_ctx* ctx, const float* d_desc, int n, void* d_image);
int xfh_match_mnn_prepared_device(xfh_ctx* ctx, const void* d_image1, int n1, const void* d_image2, int n2,
                                  float min_cossim, int* d_idx1, int* d_idx2, float* d_dist, int* d_n_matches);

/* Many pairs in one call.  The reference's consumers meet one frame with several partners -- the previous frame, key frames, loop
 * candidates: one ORBmatcher::match per frame pair (ORBmatcher.cc:358-372 per call; SURVEY.md 8e "for many frame pairs, shard pairs") --
 * and one 4096 x 4096 pair is exactly one tile per CU, so a call per pair pays its launch ramp, staging wait and epilogue latency with
 * nothing to overlap them.  This entry runs the similarity GEMM of ALL pairs as one persistent launch (equal shares of the tiles of all
 * pairs per workgroup, the next tile's panel arriving while the current one is multiplied) and the mutual check / output of all
 * pairs as a second one.  Pair p: prepared images d_image1[p] (n1[p] rows) and d_image2[p] (n2[p] rows) -- the same image may appear
 * in any number of pairs, on either side -- match list to d_idx1[p] / d_idx2[p] / d_dist[p] (min(n1[p], n2[p]) slots each) and its
 * length to d_n_matches[p]; results are those of xfh_match_mnn_prepared_device pair by pair, bit for bit.  The pointer and size arrays
 * are HOST arrays (read before the call returns); everything they point to is device memory.  Asynchronous on the ctx stream. */
int xfh_match_mnn_prepared_batch_device(xfh_ctx* ctx, int n_pairs, const void* const* d_image1, const int* n1, const void* const* d_image2, const int* n2,
                                        float min_cossim, int* const* d_idx1, int* const* d_idx2, float* const* d_dist, int* d_n_matches);

/* n_valid-aware match of two extraction records (option).  ORBmatcher::match treats the zero rows that pad a record
 * like descriptors (they have similarity 0 with everything and can end up in mutual pairs); here every pair that touches a padding
 * slot is dropped.  d_record1/2: records of this ctx' nfeatures (only their headers are read: valid slots are [0, mono_index) and
 * [nfeatures - (n_valid - mono_index), nfeatures)); d_image1/2: their prepared images (xfh_extract_batch_device_images).  Indices
 * are slot numbers; everything else as xfh_match_mnn_prepared_device. */
int xfh_match_records_device(xfh_ctx* ctx, const void* d_record1, const void* d_image1, const void* d_record2, const void* d_image2,
                             float min_cossim, int* d_idx1, int* d_idx2, float* d_dist, int* d_n_matches);

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2242-2250), XFeat branch:
 * (int)(float(cv::norm(a, b, NORM_L2SQR)) * 512).  Scalar host version, stateless. */
int xfh_descriptor_distance(const float* a, const float* b);
/* dense n1 x n2 table of the same integer metric, computed on the GPU (host pointers) */
int xfh_distance_i32(xfh_ctx* ctx, const float* d1, int n1, const float* d2, int n2, int32_t* out);
int xfh_distance_i32_device(xfh_ctx* ctx, const float* d_d1, int n1, const float* d_d2, int n2, int32_t* d_out);

/* Guided ("windowed") matching, the inner loop of ORBmatcher::SearchByProjection / SearchByBoW /
 * SearchForTriangulation / Fuse (ORBmatcher.cc:82-119, 1928-1953, 450-500, ...): for query q the
 * candidates are indices[offsets[q] .. offsets[q+1]) into the target descriptors, visited in that
 * order with
 *     dist = DescriptorDistance(query_q, target_idx);
 *     if (dist < best)        { second = best; best = dist; best_idx = idx; }
 *     else if (dist < second) { second = dist; }
 * starting from best = second = init_dist (the reference keeps ORB's 256) and
 * best_idx = -1.  second_idx is the candidate that holds `second` at the end (the reference keeps
 * its pyramid level, always 0 for XFeat).  The map-state filters of the reference (already-matched
 * map points, stereo consistency) are applied by the caller when it builds the candidate lists.
 * All pointers host memory (the _device variant: device memory, asynchronous). */
int xfh_best2_csr(xfh_ctx* ctx, const float* queries, int nq, const float* targets, int nt,
                  const int* offsets, const int* indices, int init_dist,
                  int* best_idx, int* best_dist, int* second_idx, int* second_dist);
int xfh_best2_csr_device(xfh_ctx* ctx, const float* d_queries, int nq, const float* d_targets, int nt,
                         const int* d_offsets, const int* d_indices, int init_dist,
                         int* d_best_idx, int* d_best_dist, int* d_second_idx, int* d_second_dist);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403), batched over map points: group g observes the
 * descriptor rows indices[offsets[g] .. offsets[g+1]) of `table` (n_rows x 64).  Pairwise DescriptorDistance inside
 * the group (diagonal 0), per row the median sorted[(N-1)/2], and the FIRST row with the least median wins:
 * best_pos[g] = its position inside the group, best_median[g] = that median; an empty group gives -1 / INT_MAX
 * (the reference returns without touching mDescriptor).  Groups may hold at most XFH_MAX_GROUP rows.
 * All pointers host memory (the _device variant: device memory, asynchronous, max_group = largest group size). */
#define XFH_MAX_GROUP 256
int xfh_distinctive_csr(xfh_ctx* ctx, const float* table, int n_rows, const int* offsets, const int* indices, int n_groups,
                        int* best_pos, int* best_median);
int xfh_distinctive_csr_device(xfh_ctx* ctx, const float* d_table, int n_rows, const int* d_offsets, const int* d_indices,
                               int n_groups, int max_group, int* d_best_pos, int* d_best_median);

/* ---- multi-GPU exchange (BASELINE.json configs[3]) ---------------------------------------------------
 * Frames are independent: frame i of a batch goes to rank i mod R (one process and one ctx per GPU, every rank holds the
 * weights) and the fixed-size records travel to the rank that runs the sequential SLAM state machine (the reference is
 * one process, src/System.cc:197-233) with RCCL over xGMI.  These calls wrap librccl directly (dlopen at
 * xfh_comm_create); no Python or torch is involved.  Rank 0 calls xfh_comm_unique_id and ships the 128 bytes to the
 * other ranks by any means (TCP, MPI, a file); then every rank calls xfh_comm_create.
 *   xfh_allgather_records   : ncclAllGather -- d_all receives world x B records, rank r's at offset r * B * record_bytes
 *   xfh_gather_records_root : ncclSend / ncclRecv -- only `root` receives (same layout); cheaper when one rank consumes
 *   xfh_gather_compact_root : only header + valid rows travel (a frame with n_valid of nfeatures rows sends n_valid * 284 B);
 *                             shard r lands at d_all + r * xfh_compact_bytes_max(nfeatures, B) with shard_bytes[r] bytes, and
 *                             xfh_unpack_compact restores a frame's padded form on the host.  Needs one host round trip for
 *                             the sizes (the only blocking call of the three).
 * All of them start when the ctx stream reaches the call (the records are complete) and run on the ctx's communication
 * stream, so the next extraction overlaps them; `gen` (0 / 1) names the record buffer generation the call reads:
 * xfh_comm_fence(ctx, gen) makes the ctx stream wait for the last collective on that generation before the buffer is
 * overwritten, xfh_comm_synchronize waits on the host. */
#define XFH_UNIQUE_ID_BYTES 128
int xfh_comm_unique_id(void* id_out /* XFH_UNIQUE_ID_BYTES */);
/* Which librccl the exchange runs on and what it is: "<file> (RCCL a.b.c; its HIP runtime x.y at <file>; libxfeat_hip's HIP runtime x.y at <file>)",
 * "" when none can be loaded.  Search order: $XFH_RCCL_LIB (explicit, no fallback), /opt/rocm/lib/librccl.so.1, librccl.so.1, librccl.so.
 * xfh_comm_create returns XFH_ERR_COMM when that RCCL is bound to another HIP runtime (file or major.minor) than this library. */
const char* xfh_comm_library(void);
int xfh_comm_create(xfh_ctx* ctx, const void* unique_id, int rank, int world);
int xfh_comm_destroy(xfh_ctx* ctx);
int xfh_comm_rank(xfh_ctx* ctx);
int xfh_comm_world(xfh_ctx* ctx);
int xfh_allgather_records(xfh_ctx* ctx, const void* d_records, int B, void* d_all, int gen);
int xfh_gather_records_root(xfh_ctx* ctx, const void* d_records, int B, void* d_all, int root, int gen);
size_t xfh_compact_bytes_max(int nfeatures, int B);
int xfh_gather_compact_root(xfh_ctx* ctx, const void* d_records, int B, void* d_all, size_t* shard_bytes /* [world], root only */, int root, int gen);
int xfh_unpack_compact(const void* shard, size_t nbytes, int frame, int nfeatures, xfh_keypoint* kps_out, float* desc_out, int* n_valid, int* mono_index);
int xfh_allgather_bytes(xfh_ctx* ctx, const void* d_send, size_t nbytes, void* d_recv, int gen);   /* e.g. timings, barriers */
int xfh_comm_fence(xfh_ctx* ctx, int gen);
/* Several ctx of one GPU feeding one communicator (sub-batches of a step, each extracted by its own ctx into one record buffer):
 * xfh_comm_wait_ctx makes the NEXT collective of `ctx` also wait for the work queued on `other` so far; xfh_comm_fence_ctx is
 * xfh_comm_fence for `other`'s stream.  Neither synchronises the host nor orders the two ctx streams against each other. */
int xfh_comm_wait_ctx(xfh_ctx* ctx, xfh_ctx* other);
int xfh_comm_fence_ctx(xfh_ctx* ctx, xfh_ctx* other, int gen);
int xfh_comm_synchronize(xfh_ctx* ctx);

/* ---- plumbing ----------------------------------------------------------------------- */
int xfh_synchronize(xfh_ctx* ctx);
/* run the ctx on an externally owned hipStream_t (e.g. torch's current stream); NULL
 * restores the ctx's own stream */
int xfh_set_stream(xfh_ctx* ctx, void* hip_stream);
const char* xfh_strerror(int status);
const char* xfh_last_hip_error(xfh_ctx* ctx);
/* "xfeat_hip 0.1 (gfx950); built with clang <x.y.z>, HIP <x.y.z>; runtime HIP <n>, driver <n>": the compiler the library was built with and the HIP runtime it
 * met in this process (the library carries hand-counted MFMA hazard padding; tests/test_gpu_hazard.py re-checks it with the GPU box's own compiler) */
const char* xfh_version(void);
int xfh_device_count(void);

/* device-memory helpers so that a host language without a HIP binding can keep inputs and
 * records resident in HBM (used by bench.py and the tests through ctypes) */
int xfh_dev_alloc(void** dptr, size_t nbytes);
int xfh_dev_free(void* dptr);
int xfh_memcpy_h2d(void* dst, const void* src, size_t nbytes);
int xfh_memcpy_d2h(void* dst, const void* src, size_t nbytes);

/* Measurement and debugging entry points (kernel timers, back-to-back timing loops, intermediate tensors, counter
 * calibration kernels) are declared in xfeat_hip_bench.h: they are exported by the same library but are not part of the
 * drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* XFEAT_HIP_H */
