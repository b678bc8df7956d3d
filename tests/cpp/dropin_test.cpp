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
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
