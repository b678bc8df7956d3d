/*
 * xfeat_oracle.h -- CPU oracle for the XFeat extraction + descriptor matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker.  The shipped path is the HIP library behind include/xfeat_hip.h.
 *
 * PARITY STATUS: "parity unpinned" by execution of the reference.  The reference
 * (udaysankar01/xfeatSLAM) has no tests or golden vectors for this path,
 * is C++ (not importable), and its two source files include <opencv2/...> headers that
 * this image does not have, so it cannot be compiled here without writing stand-in
 * headers (not allowed).  What pins this restatement instead:
 *   (1) oracle/torch_restatement.py calls, line by line, the same libtorch/ATen CPU
 *       operators the reference calls (conv2d, batch_norm(training), instance_norm,
 *       avg_pool2d, interpolate, softmax, max_pool2d, nonzero, grid_sample, argsort,
 *       normalize) and tests/test_oracle.py checks this C code against it, live, stage by stage -- since
 *       round 5 over eleven weight families x eight image families (test_campaign_oracle_vs_aten);
 *   (2) golden vectors produced by (1) are committed under tests/golden/ -- since round 6 for the whole campaign: 46 extraction cases (every
 *       weight family at VGA x 2 image families and at 720p, nfeatures = 1000 with the TUM lapping areas) and 9 matches (five on extracted
 *       descriptor blocks); the same files check the HIP path on the GPU (tests/test_gpu_extract.py, tests/test_gpu_match.py);
 *   (3) exp() of the softmax and the sigmoid is libtorch's own vector kernel (Sleef expf_u10, FMA form), restated
 *       as xfo_expf and checked bit for bit against torch.sigmoid / F.softmax (test_exp_is_atens_vector_exp).
 * None of this is an execution of the reference: parity stays "partial".
 * ORBmatcher::match has no compiled definition in the reference at all (dead code,
 * src/ORBmatcher.cc:340-405); DescriptorDistance calls cv::norm from OpenCV 4.5.4
 * (un-vendored).  Both are restated from the call sites.
 *
 * Tensor ids below are shared with the HIP library's debug accessor (XFH_T_* in
 * include/xfeat_hip.h) so that tests can diff every intermediate.  All image-like
 * tensors are NHWC float32.
 */
#ifndef XFEAT_ORACLE_H
#define XFEAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xfo_ctx xfo_ctx;

/* mirrors cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} xfo_keypoint;

enum {
    XFO_T_X = 0,        /* [H][W]        image/255 after the resize to multiples of 32 */
    XFO_T_XSTAT = 1,    /* [2]           InstanceNorm beta = -(mean * rstd), alpha = rstd */
    XFO_T_SKIP_POOL = 2,/* [H/4][W/4]    AvgPool4x4 of the normalised image            */
    XFO_T_XUNFOLD = 3,  /* [H/8][W/8][64] unfold2d(xhat, 8)                             */
    XFO_T_B2IN = 4,     /* [H/4][W/4][24] x1 + skip1(x)                                 */
    XFO_T_FUSE_IN = 5,  /* [H/8][W/8][64] x3 + up(x4) + up(x5)                          */
    XFO_T_FEATS = 6,    /* [H/8][W/8][64] block_fusion output                           */
    XFO_T_M1N = 7,      /* [H/8][W/8][64] F::normalize(feats, dim=1)                    */
    XFO_T_H1 = 8,       /* [H/8][W/8]    heatmap (sigmoid)                             */
    XFO_T_K1H = 9,      /* [H][W]        keypoint heatmap after softmax+depth-to-space  */
    XFO_T_LOGITS = 10,  /* [H/8][W/8][65] keypoint logits (oracle only)                 */
    XFO_T_RAW0 = 16,    /* +i: raw conv output of BasicLayer i (before BN), i = 0..22   */
    XFO_T_STAT0 = 48,   /* +i: [2][C]  beta[C] = -(mean * rstd) then alpha[C] = rstd of BasicLayer i (applied as fma(x, alpha, beta)) */
    XFO_T_SEL = 80,     /* [N][3]      x, y, score of the top-N list, descending score  */
    XFO_T_CAND = 81     /* [C][3]      x, y, score of every NMS candidate, row-major    */
};

#define XFO_NUM_LAYERS 23

xfo_ctx* xfo_create(const void* weight_blob, size_t nbytes);
void xfo_destroy(xfo_ctx* c);
/* 0 = BatchNorm on batch statistics (what the reference does), 1 = on the running statistics stored in
 * the blob (upstream-XFeat eval() semantics); returns -1 if the blob carries no running statistics */
/* 0 (default) = the reference's Long-typed keypoint rescale, a no-op; 1 = float rescale to input-image
 * coordinates as upstream XFeat does (optional) */
int xfo_set_rescale(xfo_ctx* c, int on);
int xfo_set_bn_mode(xfo_ctx* c, int mode);
void xfo_set_threads(int n); /* OpenMP threads for the heavy loops (<=0: library default) */
int xfo_get_threads(void);

/* XFextractor::operator() (reference src/XFextractor.cc:250-356).
 * gray: dense H x W u8.  kps: nfeatures records, desc: nfeatures x 64 floats.
 * returns 0, or -1 for an empty image (H*W == 0), -2 for an unsupported size. */
int xfo_extract(xfo_ctx* c, const uint8_t* gray, int H, int W, int nfeatures, int lap0, int lap1,
                xfo_keypoint* kps, float* desc, int* n_valid, int* mono_index);

/* intermediate of the last xfo_extract call; pointer stays valid until the next call */
int xfo_get_tensor(xfo_ctx* c, int id, const float** ptr, int64_t* count);

/* ORBmatcher::match, intended semantics of the dead code (src/ORBmatcher.cc:340-405). */
int xfo_match_mnn(const float* d1, int n1, const float* d2, int n2, float min_cossim,
                  int* idx1, int* idx2, float* dist, int* n_matches);

/* dense form of ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2242-2250) */
int xfo_distance_i32(const float* d1, int n1, const float* d2, int n2, int32_t* out);
int xfo_descriptor_distance(const float* a, const float* b);
/* exp() of libtorch's CPU softmax / sigmoid kernels (Sleef u10, FMA form), see xfeat_oracle.c */
float xfo_expf(float d);
void xfo_expf_array(const float* x, float* y, int64_t n);

/* best / second-best integer distance over per-query candidate lists: the inner loop of
 * ORBmatcher::SearchByProjection and friends (src/ORBmatcher.cc:75-119) */
int xfo_best2_csr(const float* q, int nq, const float* tg, const int* offsets, const int* indices, int init_dist,
                  int* best_idx, int* best_dist, int* second_idx, int* second_dist);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403) over CSR groups of descriptor rows:
 * position inside the group of the descriptor with the least median distance to the others, and that median */
int xfo_distinctive_csr(const float* table, const int* offsets, const int* indices, int n_groups, int* best_pos, int* best_median);

#ifdef __cplusplus
}
#endif
#endif
