/*
 * ORBmatcher_xfeat.h -- the XFeat half of the reference's ORB_SLAM3::ORBmatcher
 * (include/ORBmatcher.h:43,77) on top of include/xfeat_hip.h.
 *
 *   static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b)   -- ORBmatcher.cc:2242-2250
 *   void match(cv::Mat d1, cv::Mat d2, std::vector<cv::DMatch>& out)    -- declared :77, the
 *        definition is commented out in the reference (ORBmatcher.cc:340-405); this supplies it.
 *   TH_LOW / TH_HIGH                                                    -- ORBmatcher.cc:34-35
 *   best2 / distinctive: the batched inner loops of SearchBy* (ORBmatcher.cc:75-119) and of
 *        MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403)
 *
 * Drop the two member definitions into src/ORBmatcher.cc or use this
 * class directly.  Without OpenCV the cvlite mirrors of XFextractor.h are used.
 */
#ifndef XFEAT_ORBMATCHER_XFEAT_H
#define XFEAT_ORBMATCHER_XFEAT_H

#include "XFextractor.h"

namespace xfeat {
#if !XFEAT_HAVE_OPENCV
namespace cvlite {
struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 3.402823466e+38f;
    DMatch() = default;
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {}
};
}  // namespace cvlite
#endif
}  // namespace xfeat

namespace ORB_SLAM3 {

class XFmatcher {
public:
    using Mat = xfeat::cvx::Mat;
    using DMatch = xfeat::cvx::DMatch;

    static constexpr int TH_HIGH = 1000;   // ORBmatcher.cc:34, USE_ORB unset
    static constexpr int TH_LOW = 100;     // ORBmatcher.cc:35

    explicit XFmatcher(xfh_ctx* shared_ctx, float nnratio = 0.6f, bool checkOri = true)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), ctx(shared_ctx) {}

    // stateless and thread-safe like the reference's static member
    static int DescriptorDistance(const Mat& a, const Mat& b) {
        return xfh_descriptor_distance(a.template ptr<float>(0), b.template ptr<float>(0));
    }

    // mutual-nearest-neighbour cosine matching on the GPU
    void match(const Mat& _frame1_desc, const Mat& _frame2_desc, std::vector<DMatch>& _matches, float min_cossim = -1.f) {
        const int n1 = _frame1_desc.rows, n2 = _frame2_desc.rows;
        _matches.clear();
        if (n1 == 0 || n2 == 0) return;
        const int nm = n1 < n2 ? n1 : n2;
        i1.resize(nm); i2.resize(nm); d.resize(nm);
        int n = 0;
        const int rc = xfh_match_mnn(ctx, _frame1_desc.template ptr<float>(0), n1, _frame2_desc.template ptr<float>(0), n2,
                                     min_cossim, i1.data(), i2.data(), d.data(), &n);
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFmatcher::match: ") + xfh_strerror(rc));
        _matches.reserve(n);
        for (int k = 0; k < n; ++k) _matches.emplace_back(DMatch(i1[k], i2[k], d[k]));   // :401
    }

    // The same match on two PREPARED images in device memory (XFextractor::extractBatchDevice with d_images, or
    // xfh_match_prepare_device): the tracker's frame-to-frame match without a normalisation pass, two kernel launches.  n1 / n2 =
    // rows the images were made from (nfeatures for images written by the extraction).  Blocks until the result is on the host.
    void matchPrepared(const void* d_image1, int n1, const void* d_image2, int n2, std::vector<DMatch>& _matches, float min_cossim = -1.f) {
        _matches.clear();
        if (n1 <= 0 || n2 <= 0) return;
        const int nm = n1 < n2 ? n1 : n2;
        const size_t bytes = (size_t)nm * 12 + 16;
        if (bytes > d_out_bytes) {
            if (d_out) xfh_dev_free(d_out);
            d_out = nullptr; d_out_bytes = 0;
            if (xfh_dev_alloc(&d_out, bytes) != XFH_OK) throw std::runtime_error("XFmatcher::matchPrepared: out of device memory");
            d_out_bytes = bytes;
        }
        char* o = (char*)d_out;
        int rc = xfh_match_mnn_prepared_device(ctx, d_image1, n1, d_image2, n2, min_cossim, (int*)o, (int*)(o + 4 * (size_t)nm), (float*)(o + 8 * (size_t)nm),
                                               (int*)(o + 12 * (size_t)nm));
        if (rc == XFH_OK) rc = xfh_synchronize(ctx);
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFmatcher::matchPrepared: ") + xfh_strerror(rc));
        int n = 0;
        xfh_memcpy_d2h(&n, o + 12 * (size_t)nm, 4);
        if (n < 0 || n > nm) throw std::runtime_error("XFmatcher::matchPrepared: the device reported a collector time-out (n_matches < 0)");
        i1.resize(nm); i2.resize(nm); d.resize(nm);
        if (n > 0) { xfh_memcpy_d2h(i1.data(), o, 4 * (size_t)n); xfh_memcpy_d2h(i2.data(), o + 4 * (size_t)nm, 4 * (size_t)n); xfh_memcpy_d2h(d.data(), o + 8 * (size_t)nm, 4 * (size_t)n); }
        _matches.reserve(n);
        for (int k = 0; k < n; ++k) _matches.emplace_back(DMatch(i1[k], i2[k], d[k]));
    }
    // One frame against several partners in ONE call (the tracker's frame against previous frame / key frames / loop candidates; the reference
    // calls match() once per pair, ORBmatcher.cc:358-372): xfh_match_mnn_prepared_batch_device -- one persistent GEMM launch over the tiles of
    // all pairs + one launch for the mutual check and the output.  _matches[p] = what matchPrepared(d_image1, n1, d_images2[p], n2[p]) gives.
    void matchPreparedMany(const void* d_image1, int n1, const std::vector<const void*>& d_images2, const std::vector<int>& n2,
                           std::vector<std::vector<DMatch>>& _matches, float min_cossim = -1.f) {
        const int P = (int)d_images2.size();
        _matches.assign(P, std::vector<DMatch>());
        if (P == 0 || n1 <= 0) return;
        std::vector<size_t> off(P + 1, 0);
        std::vector<int> nm(P);
        for (int p = 0; p < P; ++p) { nm[p] = n2[p] <= 0 ? 1 : (n1 < n2[p] ? n1 : n2[p]); off[p + 1] = off[p] + (((size_t)nm[p] * 12 + 63) & ~(size_t)63); }
        const size_t bytes = off[P] + (size_t)P * 4 + 64;
        if (bytes > d_out_bytes) {
            if (d_out) xfh_dev_free(d_out);
            d_out = nullptr; d_out_bytes = 0;
            if (xfh_dev_alloc(&d_out, bytes) != XFH_OK) throw std::runtime_error("XFmatcher::matchPreparedMany: out of device memory");
            d_out_bytes = bytes;
        }
        char* o = (char*)d_out;
        std::vector<const void*> a1(P, d_image1);
        std::vector<int> vn1(P, n1);
        std::vector<int*> p1(P), p2(P); std::vector<float*> pd(P);
        for (int p = 0; p < P; ++p) { p1[p] = (int*)(o + off[p]); p2[p] = (int*)(o + off[p] + 4 * (size_t)nm[p]); pd[p] = (float*)(o + off[p] + 8 * (size_t)nm[p]); }
        int* d_cnt = (int*)(o + off[P]);
        int rc = xfh_match_mnn_prepared_batch_device(ctx, P, a1.data(), vn1.data(), d_images2.data(), n2.data(), min_cossim, p1.data(), p2.data(), pd.data(), d_cnt);
        if (rc == XFH_OK) rc = xfh_synchronize(ctx);
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFmatcher::matchPreparedMany: ") + xfh_strerror(rc));
        std::vector<int> cnt(P, 0);
        xfh_memcpy_d2h(cnt.data(), d_cnt, (size_t)P * 4);
        for (int p = 0; p < P; ++p) {
            const int n = cnt[p];
            if (n < 0 || n > nm[p]) throw std::runtime_error("XFmatcher::matchPreparedMany: the device reported a collector time-out (n_matches < 0)");
            i1.resize(n); i2.resize(n); d.resize(n);
            if (n > 0) { xfh_memcpy_d2h(i1.data(), p1[p], 4 * (size_t)n); xfh_memcpy_d2h(i2.data(), p2[p], 4 * (size_t)n); xfh_memcpy_d2h(d.data(), pd[p], 4 * (size_t)n); }
            _matches[p].reserve(n);
            for (int k = 0; k < n; ++k) _matches[p].emplace_back(DMatch(i1[k], i2[k], d[k]));
        }
    }
    ~XFmatcher() { if (d_out) xfh_dev_free(d_out); }
    XFmatcher(const XFmatcher&) = delete;
    XFmatcher& operator=(const XFmatcher&) = delete;

    // Guided matching: the inner loop of SearchByProjection / SearchByBoW / SearchForTriangulation / Fuse
    // (ORBmatcher.cc:75-119): best and second-best DescriptorDistance over per-query candidate lists
    // (CSR: offsets[nq+1], indices[]) with the reference's initial value 256 for both.
    void best2(const Mat& queries, const Mat& targets, const std::vector<int>& offsets, const std::vector<int>& indices,
               std::vector<int>& bestIdx, std::vector<int>& bestDist, std::vector<int>& secondIdx, std::vector<int>& secondDist,
               int initDist = 256) {
        const int nq = queries.rows;
        bestIdx.assign(nq, -1); bestDist.assign(nq, initDist); secondIdx.assign(nq, -1); secondDist.assign(nq, initDist);
        if (nq == 0) return;
        const int rc = xfh_best2_csr(ctx, queries.template ptr<float>(0), nq, targets.template ptr<float>(0), targets.rows,
                                     offsets.data(), indices.data(), initDist, bestIdx.data(), bestDist.data(), secondIdx.data(), secondDist.data());
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFmatcher::best2: ") + xfh_strerror(rc));
    }

    // MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403), batched over map points: group g observes the
    // rows indices[offsets[g] .. offsets[g+1]) of `table`; bestPos[g] = position in the group of the descriptor with
    // the least median DescriptorDistance to the others (-1 for an empty group), bestMedian[g] = that median.
    void distinctive(const Mat& table, const std::vector<int>& offsets, const std::vector<int>& indices,
                     std::vector<int>& bestPos, std::vector<int>& bestMedian) {
        const int ng = offsets.empty() ? 0 : (int)offsets.size() - 1;
        bestPos.assign(ng, -1); bestMedian.assign(ng, 0x7fffffff);
        if (ng == 0) return;
        const int rc = xfh_distinctive_csr(ctx, table.template ptr<float>(0), table.rows, offsets.data(), indices.data(), ng,
                                           bestPos.data(), bestMedian.data());
        if (rc != XFH_OK) throw std::runtime_error(std::string("XFmatcher::distinctive: ") + xfh_strerror(rc));
    }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    xfh_ctx* ctx;
    std::vector<int> i1, i2;
    std::vector<float> d;
    void* d_out = nullptr; size_t d_out_bytes = 0;          // device result buffer of matchPrepared
};

}  // namespace ORB_SLAM3
#endif
