// frontend_replay.cpp -- sequence replay of the feature front end, shaped after the reference's
// examples/RGB-D/rgbd_tum.cc:75-143 (load image i, hand it to the tracker, record the time, print
// median / mean tracking time), minus everything that needs the SLAM stack: per frame it runs
// XFextractor::operator() and ORBmatcher::match against the previous frame's descriptors, and reports
// an inlier statistic per frame pair: the matches whose displacement lies within 3 px of the median
// displacement of the pair (a pure-translation consensus; there is no camera model without the SLAM stack).
//
//   frontend_replay weights.xfhw <associations.txt> <sequence_dir> [--dump file] [--rgb 0|1]
//        TUM-style list "ts rgb/x.png ts depth/x.png" (examples/RGB-D/associations/fr1_desk.txt); 8-bit PNG or binary PGM;
//        colour frames are converted as the reference does with Camera.RGB = 1 (include/xfeat/image_io.h)
//   frontend_replay weights.xfhw --synthetic N H W [--dump file]     N synthetic frames (texture drifting 2 px per frame)
//   --dump: per frame n_valid, keypoint (x, y) pairs, then per pair the (idx1, idx2, dist) match list, as little-endian
//        records -- tests/test_gpu_dropin_cpp.py recomputes them with the oracle.
//
// Build: g++ -std=c++17 -O2 -Iinclude examples/frontend_replay.cpp -Lxfeatslam_amd -lxfeat_hip -lz -o frontend_replay
#define XFEAT_NO_OPENCV 1
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "xfeat/XFextractor.h"
#include "xfeat/ORBmatcher_xfeat.h"
#include "xfeat/image_io.h"

using namespace ORB_SLAM3;
using Mat = XFextractor::Mat;

static int g_rgb = 1;                                                       // TUM1.yaml Camera.RGB
static bool load_frame(const std::string& path, Mat& im) {
    xfeat::Image8 raw;
    std::vector<unsigned char> gray;
    if (!xfeat::load_image(path, raw)) return false;
    xfeat::to_gray(raw, g_rgb, gray);
    im.create(raw.rows, raw.cols, 1);
    memcpy(im.data, gray.data(), gray.size());
    return true;
}

static void synth_frame(Mat& im, int H, int W, int t) {        // smooth texture drifting 2 px per frame
    im.create(H, W, 1);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int xs = x + 2 * t;
            unsigned v = (unsigned)(xs * 2654435761u) ^ (unsigned)(y * 40503u) ^ (unsigned)((xs / 8) * 97u + (y / 8) * 31u) * 2246822519u;
            v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
            im.data[(size_t)y * W + x] = (unsigned char)(((v >> 8) & 0xff) / 2 + (((xs / 16 + y / 16) & 1) ? 96 : 32));
        }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: see the header of examples/frontend_replay.cpp\n"); return 2; }
    FILE* dump = nullptr;
    for (int i = 4; i + 1 < argc; ++i) {
        if (std::string(argv[i]) == "--dump") dump = fopen(argv[i + 1], "wb");
        if (std::string(argv[i]) == "--rgb") g_rgb = atoi(argv[i + 1]);
    }
    const int nfeatures = getenv("XFH_NFEATURES") ? atoi(getenv("XFH_NFEATURES")) : 1000;     // TUM1.yaml: ORBextractor.nFeatures 1000
    std::vector<std::string> files;
    int nsyn = 0, H = 480, W = 640;
    if (std::string(argv[2]) == "--synthetic") { nsyn = atoi(argv[3]); if (argc > 5 && argv[4][0] != '-') { H = atoi(argv[4]); W = atoi(argv[5]); } }
    else {
        std::ifstream fa(argv[2]); std::string line;                                           // rgbd_tum.cc LoadImages (:152-179)
        while (std::getline(fa, line)) { std::stringstream ss(line); std::string t, rgb; if (ss >> t >> rgb) files.push_back(std::string(argv[3]) + "/" + rgb); }
        if (files.empty()) { fprintf(stderr, "no images in %s\n", argv[2]); return 2; }
    }
    const int n = nsyn ? nsyn : (int)files.size();
    Mat im;
    if (nsyn) synth_frame(im, H, W, 0); else if (!load_frame(files[0], im)) { fprintf(stderr, "cannot read %s (8-bit PNG or binary PGM expected)\n", files[0].c_str()); return 2; }
    XFextractor extractor(nfeatures, 1.2f, 8, 20, 7, im.rows, im.cols, 0, argv[1]);           // Tracking.cc:597
    XFmatcher matcher(extractor.context());
    std::vector<XFextractor::KeyPoint> keys, prev_keys;
    Mat desc, prev;
    std::vector<int> lap = {0, 0};                                                             // Frame.cc:311 (RGB-D)
    std::vector<XFmatcher::DMatch> matches;
    std::vector<double> vTimesTrack(n);
    long total_matches = 0, total_valid = 0, total_inliers = 0;
    auto put = [&](const void* p, size_t n) { if (dump) fwrite(p, 1, n, dump); };
    for (int ni = 0; ni < n; ++ni) {
        if (nsyn) synth_frame(im, H, W, ni); else if (!load_frame(files[ni], im)) { fprintf(stderr, "cannot read %s\n", files[ni].c_str()); return 2; }
        const auto t1 = std::chrono::steady_clock::now();
        const int ret = extractor(im, Mat(), keys, desc, lap);
        if (ret < 0) { fprintf(stderr, "empty image at %d\n", ni); return 1; }
        matches.clear();
        if (!prev.empty() && !desc.empty()) { matcher.match(prev, desc, matches); total_matches += (long)matches.size(); }
        const auto t2 = std::chrono::steady_clock::now();
        vTimesTrack[ni] = std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
        int nv = 0;
        for (auto& k : keys) nv += k.size > 0 ? 1 : 0;
        total_valid += nv;
        // inlier statistic: matches between two REAL keypoints whose displacement is within 3 px of the pair's median displacement
        int inl = 0; float mdx = 0.f, mdy = 0.f;
        {
            std::vector<float> dx, dy;
            for (auto& m : matches)
                if (prev_keys[m.queryIdx].size > 0 && keys[m.trainIdx].size > 0) { dx.push_back(keys[m.trainIdx].pt.x - prev_keys[m.queryIdx].pt.x); dy.push_back(keys[m.trainIdx].pt.y - prev_keys[m.queryIdx].pt.y); }
            if (!dx.empty()) {
                std::vector<float> sx = dx, sy = dy;
                std::nth_element(sx.begin(), sx.begin() + sx.size() / 2, sx.end()); std::nth_element(sy.begin(), sy.begin() + sy.size() / 2, sy.end());
                mdx = sx[sx.size() / 2]; mdy = sy[sy.size() / 2];
                for (size_t q = 0; q < dx.size(); ++q) inl += (std::fabs(dx[q] - mdx) <= 3.f && std::fabs(dy[q] - mdy) <= 3.f) ? 1 : 0;
            }
        }
        total_inliers += inl;
        if (ni > 0) printf("pair %d-%d: matches %zu inliers %d median displacement (%.1f, %.1f)\n", ni - 1, ni, matches.size(), inl, mdx, mdy);
        if (dump) {
            const int nk = (int)keys.size(), nm = (int)matches.size();
            put(&nv, 4); put(&nk, 4);
            for (auto& k : keys) { put(&k.pt.x, 4); put(&k.pt.y, 4); put(&k.size, 4); }
            put(&nm, 4);
            for (auto& m : matches) { put(&m.queryIdx, 4); put(&m.trainIdx, 4); put(&m.distance, 4); }
        }
        prev = desc; prev_keys = keys;
    }
    std::sort(vTimesTrack.begin(), vTimesTrack.end());                                         // rgbd_tum.cc:128-139
    double tot = 0; for (double t : vTimesTrack) tot += t;
    if (dump) fclose(dump);
    printf("-------\n\nframes: %d  keypoints/frame: %.1f  mutual matches/frame pair: %.1f  inliers/frame pair: %.1f\n", n, (double)total_valid / n,
           n > 1 ? (double)total_matches / (n - 1) : 0.0, n > 1 ? (double)total_inliers / (n - 1) : 0.0);
    printf("median front-end time: %f\nmean front-end time: %f\n", vTimesTrack[n / 2], tot / n);
    return 0;
}
