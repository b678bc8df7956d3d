// dropin_test.cpp -- exercises the C++ drop-in classes (include/xfeat/*.h) the way
// Frame::ExtractXF does (reference src/Frame.cc:611-618): (*mpXFextractor)(im, Mat(), keys, desc, lapping).
// usage: dropin_test weights.xfhw image.raw H W nfeatures lap0 lap1 out.bin
// out.bin: int32 ret, int32 nkeys, int32 desc_rows, int32 n_matches, keypoints[nkeys*28B],
//          desc[desc_rows*64 f32], then n_matches x (int32 q, int32 t, f32 dist) of desc vs itself.
#define XFEAT_NO_OPENCV 1
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "xfeat/XFextractor.h"
#include "xfeat/ORBmatcher_xfeat.h"

using namespace ORB_SLAM3;

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage\n"); return 2; }
    const int H = atoi(argv[3]), W = atoi(argv[4]), nf = atoi(argv[5]);
    std::vector<int> lap = {atoi(argv[6]), atoi(argv[7])};
    XFextractor::Mat im(H, W, 1);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(im.data, 1, (size_t)H * W, f) != (size_t)H * W) { fprintf(stderr, "bad image\n"); return 2; }
    fclose(f);
    try {
        XFextractor ex(nf, 1.2f, 8, 20, 7, H, W, 0, argv[1]);
        if (ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8 || ex.mvImagePyramid.size() != 8) return 3;
        std::vector<XFextractor::KeyPoint> keys;
        XFextractor::Mat desc;
        XFextractor::Mat empty;
        if (ex(empty, XFextractor::Mat(), keys, desc, lap) != -1) return 4;          // empty image -> -1
        const int ret = ex(im, XFextractor::Mat(), keys, desc, lap);
        std::vector<XFmatcher::DMatch> m;
        XFmatcher matcher(ex.context());
        if (!desc.empty()) matcher.match(desc, desc, m);
        FILE* o = fopen(argv[8], "wb");
        int hdr[4] = {ret, (int)keys.size(), desc.rows, (int)m.size()};
        fwrite(hdr, 4, 4, o);
        static_assert(sizeof(XFextractor::KeyPoint) == 28, "KeyPoint layout");
        fwrite(keys.data(), 28, keys.size(), o);
        if (!desc.empty()) fwrite(desc.data, 256, desc.rows, o);
        for (auto& x : m) { fwrite(&x.queryIdx, 4, 1, o); fwrite(&x.trainIdx, 4, 1, o); fwrite(&x.distance, 4, 1, o); }
        fclose(o);
        if (!desc.empty()) {
            const int d = XFmatcher::DescriptorDistance(desc, desc);
            if (d != 0) return 5;
            // the batched map-point primitives compile and answer through the wrapper: one group {0, 0, 1} -> the
            // duplicated row 0 has median distance 0 and comes first
            std::vector<int> off = {0, 3}, ind = {0, 0, 1}, pos, med;
            matcher.distinctive(desc, off, ind, pos, med);
            if (pos.size() != 1 || pos[0] != 0 || med[0] != 0) return 6;
        }
        // ---- batch / hand-off / multi-GPU methods of the wrappers: two device-resident frames, prepared images, the frame-to-frame
        // match on the images (must equal match() on the host descriptors), and the record exchange with a communicator of one rank
        {
            XFextractor bx(nf, 1.2f, 8, 20, 7, H, W, 0, argv[1], 0, /*max_batch*/ 2);
            const size_t fb = (size_t)H * W, rb = xfh_record_bytes(nf), ib = xfh_match_image_bytes(nf);
            void *d_gray = nullptr, *d_rec = nullptr, *d_img = nullptr, *d_all = nullptr;
            if (xfh_dev_alloc(&d_gray, 2 * fb) || xfh_dev_alloc(&d_rec, 2 * rb) || xfh_dev_alloc(&d_img, 2 * ib) || xfh_dev_alloc(&d_all, 2 * rb)) return 7;
            std::vector<unsigned char> two(2 * fb);
            memcpy(two.data(), im.data, fb);
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) two[fb + (size_t)y * W + x] = im.data[(size_t)y * W + (x + 2) % W];   // frame 1: shifted copy
            xfh_memcpy_h2d(d_gray, two.data(), 2 * fb);
            bx.extractBatchDevice((const uint8_t*)d_gray, 2, H, W, lap[0], lap[1], d_rec, d_img);
            bx.synchronize();
            std::vector<unsigned char> recs(2 * rb), all(2 * rb);
            xfh_memcpy_d2h(recs.data(), d_rec, 2 * rb);
            XFmatcher bm(bx.context());
            std::vector<XFmatcher::DMatch> mp, mh;
            bm.matchPrepared(d_img, nf, (char*)d_img + ib, nf, mp);
            XFextractor::Mat da(nf, 64, 4), db(nf, 64, 4);
            memcpy(da.data, recs.data() + xfh_record_desc_offset(nf), (size_t)nf * 256);
            memcpy(db.data, recs.data() + rb + xfh_record_desc_offset(nf), (size_t)nf * 256);
            bm.match(da, db, mh);
            if (mp.size() != mh.size() || mp.empty()) return 8;
            for (size_t k = 0; k < mp.size(); ++k) if (mp[k].queryIdx != mh[k].queryIdx || mp[k].trainIdx != mh[k].trainIdx) return 8;
            {   // one frame against several partners in one call == the pair-by-pair calls (here: frame 0 against frame 1, itself, frame 1 again)
                std::vector<std::vector<XFmatcher::DMatch>> many;
                std::vector<XFmatcher::DMatch> self;
                bm.matchPreparedMany(d_img, nf, {(const void*)((char*)d_img + ib), (const void*)d_img, (const void*)((char*)d_img + ib)}, {nf, nf, nf}, many);
                bm.matchPrepared(d_img, nf, d_img, nf, self);
                if (many.size() != 3 || many[0].size() != mp.size() || many[2].size() != mp.size() || many[1].size() != self.size()) return 10;
                for (size_t k = 0; k < mp.size(); ++k)
                    if (many[0][k].queryIdx != mp[k].queryIdx || many[0][k].trainIdx != mp[k].trainIdx || many[0][k].distance != mp[k].distance ||
                        many[2][k].trainIdx != mp[k].trainIdx) return 10;
                for (size_t k = 0; k < self.size(); ++k) if (many[1][k].queryIdx != self[k].queryIdx || many[1][k].trainIdx != self[k].trainIdx) return 10;
            }
            char id[XFH_UNIQUE_ID_BYTES];
            XFextractor::commUniqueId(id);
            bx.commCreate(id, 0, 1);
            bx.commFence(0);
            bx.gatherRecordsRoot(d_rec, 2, d_all, 0, 0);
            bx.commSynchronize();
            xfh_memcpy_d2h(all.data(), d_all, 2 * rb);
            if (memcmp(all.data(), recs.data(), 2 * rb) != 0) return 9;
            xfh_dev_free(d_gray); xfh_dev_free(d_rec); xfh_dev_free(d_img); xfh_dev_free(d_all);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
