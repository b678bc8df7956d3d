/*
 * XFextractor.h -- C++ drop-in for the reference's ORB_SLAM3::XFextractor
 * (include/XFextractor.h:32-67, src/XFextractor.cc:75-356) on top of the C ABI in
 * include/xfeat_hip.h.  Header-only; link with -lxfeat_hip.
 *
 * Same constructor signature, same operator() contract (exactly nfeatures keypoints,
 * default cv::KeyPoint() / zero descriptor rows as padding, lapping-area placement from the
 * front / back, return value = monoIndex, -1 for an empty image), same six scale getters and
 * the public, never-filled mvImagePyramid.  With OpenCV present the cv:: types are used
 * as in the reference; without it (this repository's CI image has no OpenCV) minimal POD
 * mirrors in namespace xfeat::cvlite stand in for them so that the wrapper still compiles
 * and is tested.
 *
 * Differences a maintainer should know:
 *   - weights come from a flat blob (tools/convert_weights.py) instead of a libtorch
 *     archive; default path = <dir of this header>/../../weights/xfeat.xfhw or $XFH_WEIGHTS;
 *   - the object is not copyable (it owns a GPU context).
 */
#ifndef XFEAT_XFEXTRACTOR_H
#define XFEAT_XFEXTRACTOR_H

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../xfeat_hip.h"

#if defined(XFEAT_USE_OPENCV) || (__has_include(<opencv2/core/core.hpp>) && !defined(XFEAT_NO_OPENCV))
#include <opencv2/core/core.hpp>
#define XFEAT_HAVE_OPENCV 1
namespace xfeat { namespace cvx = ::cv; }
#else
#define XFEAT_HAVE_OPENCV 0
namespace xfeat {
namespace cvlite {
struct Point2f { float x = 0.f, y = 0.f; };
// field-for-field cv::KeyPoint (28 bytes)
struct KeyPoint {
    Point2f pt; float size = 0.f, angle = -1.f, response = 0.f; int octave = 0, class_id = -1;
    KeyPoint() = default;
    KeyPoint(float x, float y, float s, float a = -1.f, float r = 0.f, int o = 0, int c = -1)
        : size(s), angle(a), response(r), octave(o), class_id(c) { pt.x = x; pt.y = y; }
};
// just enough of cv::Mat for a dense CV_8UC1 input and a CV_32F output
struct Mat {
    int rows = 0, cols = 0, elem = 1; size_t step = 0; unsigned char* data = nullptr;
    std::vector<unsigned char> store;
    Mat() = default;
    Mat(int r, int c, int elem_bytes) { create(r, c, elem_bytes); }
    Mat(const Mat& o) { *this = o; }
    Mat& operator=(const Mat& o) {                       // deep copy; `data` must follow `store`
        if (this == &o) return *this;
        rows = o.rows; cols = o.cols; elem = o.elem; step = o.step; store = o.store;
        data = o.data ? (o.store.empty() ? o.data : store.data()) : nullptr;
        return *this;
    }
    void create(int r, int c, int elem_bytes) {
        rows = r; cols = c; elem = elem_bytes; step = (size_t)c * elem_bytes;
        store.assign((size_t)r * step, 0); data = store.data();
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    void release() { store.clear(); data = nullptr; rows = cols = 0; step = 0; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
};
}  // namespace cvlite
namespace cvx = cvlite;
}  // namespace xfeat
#endif

namespace ORB_SLAM3 {

static_assert(sizeof(xfh_keypoint) == 28, "xfh_keypoint must mirror cv::KeyPoint");

class XFextractor {
public:
    using KeyPoint = xfeat::cvx::KeyPoint;
    using Mat = xfeat::cvx::Mat;

    // reference: XFextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    XFextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST,
                int max_height = 1088, int max_width = 1920, int device = 0, const char* weights_path = nullptr, int flags = 0, int max_batch = 1)
        : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
        // scale tables, XFextractor.cc:80-96
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mvImagePyramid.resize(nlevels);                      // :98, never written
        // per-level quota, :100-111 (computed by the reference, unused by XFeat)
        mnFeaturesPerLevel.resize(nlevels);
        float factor = 1.0f / (float)scaleFactor;
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = (int)std::lrint(nDesired);
            sum += mnFeaturesPerLevel[level];
            nDesired *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

        xfh_config cfg; xfh_config_default(&cfg);
        cfg.device = device; cfg.max_height = max_height; cfg.max_width = max_width; cfg.nfeatures = nfeatures; cfg.max_batch = max_batch; cfg.flags = flags;
        int rc = xfh_create(&cfg, &ctx);
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFextractor: xfh_create: ") + xfh_strerror(rc));
        std::string path = weights_path ? weights_path : getModelWeightsPath("weights/xfeat.xfhw");
        rc = xfh_load_weights_file(ctx, path.c_str());
        if (rc != XFH_OK) { xfh_destroy(ctx); ctx = nullptr; throw std::runtime_error("XFextractor: cannot load weights from " + path + ": " + xfh_strerror(rc)); }
        kbuf.resize(nfeatures);
    }
    ~XFextractor() { if (ctx) xfh_destroy(ctx); }
    XFextractor(const XFextractor&) = delete;
    XFextractor& operator=(const XFextractor&) = delete;

    // reference: int operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>&,
    //                           cv::OutputArray descriptors, std::vector<int>& vLappingArea)   (include/XFextractor.h:41-43)
#if XFEAT_HAVE_OPENCV
    // With OpenCV the signature is the reference's own, so every call site compiles unchanged whatever array type it passes
    // (Frame.cc:611-618 passes cv::Mat, cv::Mat(), std::vector<cv::KeyPoint>, cv::Mat).  Written against the OpenCV 4.5 API (_InputArray::getMat /
    // empty, _OutputArray::create / getMat / release).  This repository's image has no OpenCV: the branch is compiled and run by
    // tests/cpp/cv_branch_test.cpp against tests/stubs/opencv_api -- an API-shaped stand-in for exactly these members -- never against the real library.
    int operator()(cv::InputArray _image, cv::InputArray /*_mask: ignored, as in the reference*/, std::vector<cv::KeyPoint>& _keypoints,
                   cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
        if (_image.empty()) return -1;                       // :253-254
        const cv::Mat image = _image.getMat();
        if (image.type() != CV_8UC1) throw std::invalid_argument("Unsupported number of channels in the input image.");  // :179, :257
        if ((size_t)image.step < (size_t)image.cols) throw std::invalid_argument("image rows overlap (step < cols)");
        _descriptors.create(nfeatures, 64, CV_32F);          // :347
        cv::Mat desc = _descriptors.getMat();
        // create() keeps a pre-sized array of the right shape as it is -- also a non-continuous one (an ROI, a padded row step): the C ABI
        // writes nfeatures * 64 CONTIGUOUS floats, so such a destination gets them through a temporary
        cv::Mat dense = desc.isContinuous() ? desc : cv::Mat(nfeatures, 64, CV_32F);
        int n_valid = 0;
        const int mono = run(image.data, image.rows, image.cols, (int)image.step, dense.ptr<float>(0), _keypoints, vLappingArea, &n_valid);
        if (mono >= 0 && n_valid == 0) { _descriptors.release(); return mono; }   // :350-353
        if (dense.data != desc.data) dense.copyTo(desc);
        return mono;
    }
#else
    int operator()(const Mat& image, const Mat& /*mask (ignored, as in the reference)*/, std::vector<KeyPoint>& _keypoints,
                   Mat& _descriptors, std::vector<int>& vLappingArea) {
        if (image.empty()) return -1;                        // :253-254
        if (image.elem != 1) throw std::invalid_argument("Unsupported number of channels in the input image.");
        _descriptors.create(nfeatures, 64, 4);
        int n_valid = 0;
        const int mono = run(image.data, image.rows, image.cols, (int)image.step, _descriptors.template ptr<float>(0), _keypoints, vLappingArea, &n_valid);
        if (mono >= 0 && n_valid == 0) _descriptors.release();   // :350-353
        return mono;
    }
#endif

private:
    // the part of operator() that does not depend on the array types: C ABI call, keypoint conversion
    int run(const unsigned char* data, int rows, int cols, int step, float* desc_out, std::vector<KeyPoint>& _keypoints,
            std::vector<int>& vLappingArea, int* n_valid) {
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        int mono = 0;
        const int rc = xfh_extract(ctx, data, rows, cols, step, lap0, lap1, kbuf.data(), desc_out, n_valid, &mono);
        if (rc == XFH_ERR_EMPTY_IMAGE) return -1;
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFextractor: ") + xfh_strerror(rc) + " " + xfh_last_hip_error(ctx));
        _keypoints.assign(nfeatures, KeyPoint());            // vector<KeyPoint>(nfeatures), :310
        for (int i = 0; i < nfeatures; ++i) {
            const xfh_keypoint& k = kbuf[i];
            KeyPoint& o = _keypoints[i];
            o.pt.x = k.x; o.pt.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
        }
        return mono;
    }
public:

    // Split form (not in the reference): submit() enqueues the frame and returns, collect() waits and fills the outputs
    // like operator().  Lets the Tracking thread keep both cameras of a stereo rig (two XFextractor objects) or the next
    // frame in flight without a second CPU thread.
    void submit(const Mat& image, std::vector<int>& vLappingArea) {
        pending_empty = image.empty();
        if (pending_empty) return;
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        const int rc = xfh_extract_submit(ctx, image.data, image.rows, image.cols, (int)image.step, lap0, lap1);
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFextractor::submit: ") + xfh_strerror(rc) + " " + xfh_last_hip_error(ctx));
    }
    int collect(std::vector<KeyPoint>& _keypoints, Mat& _descriptors) {
        if (pending_empty) return -1;
#if XFEAT_HAVE_OPENCV
        _descriptors.create(nfeatures, 64, CV_32F);
#else
        _descriptors.create(nfeatures, 64, 4);
#endif
        int n_valid = 0, mono = 0;
        const int rc = xfh_extract_collect(ctx, kbuf.data(), _descriptors.template ptr<float>(0), &n_valid, &mono);
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFextractor::collect: ") + xfh_strerror(rc) + " " + xfh_last_hip_error(ctx));
        _keypoints.assign(nfeatures, KeyPoint());
        for (int i = 0; i < nfeatures; ++i) {
            const xfh_keypoint& k = kbuf[i];
            KeyPoint& o = _keypoints[i];
            o.pt.x = k.x; o.pt.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
        }
        if (n_valid == 0) _descriptors.release();
        return mono;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<Mat> mvImagePyramid;

    xfh_ctx* context() { return ctx; }                       // for ORBmatcher::match on the same GPU

    // ---- batch / multi-GPU use (not in the reference): frames and records stay in HBM ------------------------
    // B frames [B][H][W] u8 at d_gray -> B records (xfh_record_bytes(nfeatures) each) at d_records, asynchronous on the object's
    // stream; with d_images each frame's descriptors also come out as the matcher's prepared image (xfh_match_image_bytes(nfeatures)
    // each) for XFmatcher::matchPrepared.  B <= the max_batch given to the constructor.
    void extractBatchDevice(const uint8_t* d_gray, int B, int H, int W, int lap0, int lap1, void* d_records, void* d_images = nullptr) {
        check(d_images ? xfh_extract_batch_device_images(ctx, d_gray, B, H, W, lap0, lap1, d_records, d_images)
                       : xfh_extract_batch_device(ctx, d_gray, B, H, W, lap0, lap1, d_records), "extractBatchDevice");
    }
    void synchronize() { check(xfh_synchronize(ctx), "synchronize"); }
    // frame i of a global batch is extracted on rank i % world; the records travel to the SLAM rank through RCCL, called by the
    // library on this object's communication stream (the next extraction overlaps the exchange).  id: XFH_UNIQUE_ID_BYTES from
    // commUniqueId() on rank 0, shipped to the other ranks by the caller.
    static void commUniqueId(void* id) { if (xfh_comm_unique_id(id) != XFH_OK) throw std::runtime_error("XFextractor: xfh_comm_unique_id failed (librccl missing?)"); }
    void commCreate(const void* id, int rank, int world) { check(xfh_comm_create(ctx, id, rank, world), "commCreate"); }
    void commFence(int gen) { check(xfh_comm_fence(ctx, gen), "commFence"); }                 // before overwriting record buffer `gen`
    void allgatherRecords(const void* d_records, int B, void* d_all, int gen) { check(xfh_allgather_records(ctx, d_records, B, d_all, gen), "allgatherRecords"); }
    void gatherRecordsRoot(const void* d_records, int B, void* d_all, int root, int gen) { check(xfh_gather_records_root(ctx, d_records, B, d_all, root, gen), "gatherRecordsRoot"); }
    void commSynchronize() { check(xfh_comm_synchronize(ctx), "commSynchronize"); }

protected:
    void check(int rc, const char* what) {
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFextractor::") + what + ": " + xfh_strerror(rc) + " " + xfh_last_hip_error(ctx));
    }
    // reference getModelWeightsPath (:151-159): relative to this source file; $XFH_WEIGHTS overrides
    std::string getModelWeightsPath(std::string weights) {
        if (const char* e = std::getenv("XFH_WEIGHTS")) return e;
        std::string f = __FILE__;
        const size_t p = f.find_last_of('/');
        const std::string dir = p == std::string::npos ? "." : f.substr(0, p);
        return dir + "/../../" + weights;
    }

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    xfh_ctx* ctx = nullptr;
    std::vector<xfh_keypoint> kbuf;
    bool pending_empty = false;
};

}  // namespace ORB_SLAM3
#endif
