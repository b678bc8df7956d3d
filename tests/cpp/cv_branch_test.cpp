// cv_branch_test.cpp -- the `#if XFEAT_HAVE_OPENCV` branches of include/xfeat/XFextractor.h / ORBmatcher_xfeat.h through a compiler and (on a GPU) through
// a run: the reference's own call shape, Frame::ExtractXF (reference src/Frame.cc:611-618):  (*mpXFextractor)(im, cv::Mat(), keys, desc, lapping)
// with cv::InputArray / cv::OutputArray parameters (reference include/XFextractor.h:41-43).  Built against tests/stubs/opencv_api (an API-shaped
// stand-in, NOT OpenCV: see its header) because this image has no OpenCV; with real OpenCV on the include path the same file builds against it.
// usage: cv_branch_test weights.xfhw image.raw H W nfeatures lap0 lap1     exit code 0 = every check passed
#define XFEAT_USE_OPENCV 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "xfeat/XFextractor.h"
#include "xfeat/ORBmatcher_xfeat.h"
static_assert(XFEAT_HAVE_OPENCV == 1, "this test is about the cv:: branch");

using namespace ORB_SLAM3;

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage\n"); return 2; }
    const int H = atoi(argv[3]), W = atoi(argv[4]), nf = atoi(argv[5]);
    std::vector<int> lap = {atoi(argv[6]), atoi(argv[7])};
    cv::Mat im(H, W, CV_8UC1);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(im.data, 1, (size_t)H * W, f) != (size_t)H * W) { fprintf(stderr, "bad image\n"); return 2; }
    fclose(f);
    try {
        XFextractor ex(nf, 1.2f, 8, 20, 7, H, W, 0, argv[1]);
        std::vector<cv::KeyPoint> keys;
        cv::Mat desc;
        if (ex(cv::Mat(), cv::Mat(), keys, desc, lap) != -1) return 3;                       // empty image -> -1 (XFextractor.cc:253-254)
        const int mono = ex(im, cv::Mat(), keys, desc, lap);                                 // Frame.cc:611-618
        if ((int)keys.size() != nf || desc.rows != nf || desc.cols != 64 || desc.type() != CV_32F || !desc.isContinuous()) return 4;
        // the same frame through the C ABI
        std::vector<xfh_keypoint> k0(nf); std::vector<float> d0((size_t)nf * 64);
        int nv = 0, m0 = 0;
        if (xfh_extract(ex.context(), im.data, H, W, (int)im.step, lap[0], lap[1], k0.data(), d0.data(), &nv, &m0) != XFH_OK) return 5;
        if (m0 != mono || memcmp(d0.data(), desc.data, d0.size() * 4) != 0) return 6;
        for (int i = 0; i < nf; ++i)
            if (keys[i].pt.x != k0[i].x || keys[i].pt.y != k0[i].y || keys[i].size != k0[i].size || keys[i].class_id != k0[i].class_id) return 7;
        // a three-channel image is refused like the reference does (XFextractor.cc:179)
        {
            cv::Mat rgb(H, W, CV_8UC3); std::vector<cv::KeyPoint> kk; cv::Mat dd; bool threw = false;
            try { ex(rgb, cv::Mat(), kk, dd, lap); } catch (const std::invalid_argument&) { threw = true; }
            if (!threw) return 8;
        }
        // an image with padded rows (step > cols) goes through with its step
        {
            std::vector<unsigned char> padded((size_t)H * (W + 32), 7);
            for (int y = 0; y < H; ++y) memcpy(padded.data() + (size_t)y * (W + 32), im.data + (size_t)y * W, (size_t)W);
            cv::Mat view(H, W, CV_8UC1, padded.data(), (size_t)W + 32);
            std::vector<cv::KeyPoint> kk; cv::Mat dd;
            if (ex(view, cv::Mat(), kk, dd, lap) != mono || memcmp(dd.data, desc.data, (size_t)nf * 256) != 0) return 9;
        }
        // a pre-sized, NON-continuous destination (rows of 64 floats inside a wider buffer): create() keeps it, the wrapper must fill its rows
        // and nothing beyond them (the C ABI writes nf * 64 contiguous floats: through a temporary)
        {
            const size_t pitch = 64 + 16;
            std::vector<float> wide((size_t)nf * pitch, -5.f);
            cv::Mat roi(nf, 64, CV_32F, wide.data(), pitch * sizeof(float));
            if (roi.isContinuous()) return 10;
            std::vector<cv::KeyPoint> kk;
            if (ex(im, cv::Mat(), kk, roi, lap) != mono) return 11;
            for (int r = 0; r < nf; ++r) {
                if (memcmp(wide.data() + (size_t)r * pitch, d0.data() + (size_t)r * 64, 256) != 0) return 12;
                for (size_t c = 64; c < pitch; ++c) if (wide[(size_t)r * pitch + c] != -5.f) return 13;
            }
        }
        // submit / collect and the matcher on cv types
        {
            std::vector<cv::KeyPoint> kk; cv::Mat dd;
            ex.submit(im, lap);
            if (ex.collect(kk, dd) != mono || memcmp(dd.data, desc.data, (size_t)nf * 256) != 0) return 14;
            XFmatcher matcher(ex.context());
            std::vector<cv::DMatch> m;
            matcher.match(desc, dd, m);
            if ((int)m.size() != nv) return 15;                                              // every valid row matches itself
            for (auto& x : m) if (x.queryIdx != x.trainIdx) return 16;
            if (XFmatcher::DescriptorDistance(desc, dd) != 0) return 17;
        }
    } catch (const std::exception& e) { fprintf(stderr, "exception: %s\n", e.what()); return 20; }
    printf("cv branch ok\n");
    return 0;
}
