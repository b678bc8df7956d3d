// frontend_replay.cpp -- sequence replay of the feature front end, shaped after the reference's
// examples/RGB-D/rgbd_tum.cc:75-143 (load image i, hand it to the tracker, record the time, print
// median / mean tracking time), minus everything that needs the SLAM stack: per frame it runs
// XFextractor::operator() and ORBmatcher::match against the previous frame's descriptors.
//
//   frontend_replay weights.xfhw <associations.txt> <sequence_dir>   TUM-style list "ts rgb/x.pgm ts depth/x.png";
//                                                                    8-bit binary PGM (P5) images only
//   frontend_replay weights.xfhw --synthetic N H W                   N synthetic frames (drifting texture)
//
// Build: g++ -std=c++17 -O2 -Iinclude examples/frontend_replay.cpp -Lxfeatslam_amd -lxfeat_hip -o frontend_replay
#define XFEAT_NO_OPENCV 1
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "xfeat/XFextractor.h"
#include "xfeat/ORBmatcher_xfeat.h"

using namespace ORB_SLAM3;
using Mat = XFextractor::Mat;

static bool load_pgm(const std::string& path, Mat& im) {
    std::ifstream f(path, std::ios::binary);
    std::string magic; int w = 0, h = 0, maxv = 0;
    if (!(f >> magic) || magic != "P5") return false;
    auto skip = [&]() { while (f.peek() == '#' || isspace(f.peek())) { if (f.peek() == '#') { std::string l; std::getline(f, l); } else f.get(); } };
    skip(); f >> w; skip(); f >> h; skip(); f >> maxv; f.get();
    if (w <= 0 || h <= 0 || maxv != 255) return false;
    im.create(h, w, 1);
    f.read((char*)im.data, (std::streamsize)w * h);
    return (bool)f;
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
    const int nfeatures = getenv("XFH_NFEATURES") ? atoi(getenv("XFH_NFEATURES")) : 1000;     // TUM1.yaml: ORBextractor.nFeatures 1000
    std::vector<std::string> files;
    int nsyn = 0, H = 480, W = 640;
    if (std::string(argv[2]) == "--synthetic") { nsyn = atoi(argv[3]); if (argc > 5) { H = atoi(argv[4]); W = atoi(argv[5]); } }
    else {
        std::ifstream fa(argv[2]); std::string line;                                           // rgbd_tum.cc LoadImages (:152-179)
        while (std::getline(fa, line)) { std::stringstream ss(line); std::string t, rgb; if (ss >> t >> rgb) files.push_back(std::string(argv[3]) + "/" + rgb); }
        if (files.empty()) { fprintf(stderr, "no images in %s\n", argv[2]); return 2; }
    }
    const int n = nsyn ? nsyn : (int)files.size();
    Mat im;
    if (nsyn) synth_frame(im, H, W, 0); else if (!load_pgm(files[0], im)) { fprintf(stderr, "cannot read %s (binary PGM expected)\n", files[0].c_str()); return 2; }
    XFextractor extractor(nfeatures, 1.2f, 8, 20, 7, im.rows, im.cols, 0, argv[1]);           // Tracking.cc:597
    XFmatcher matcher(extractor.context());
    std::vector<XFextractor::KeyPoint> keys;
    Mat desc, prev;
    std::vector<int> lap = {0, 0};                                                             // Frame.cc:311 (RGB-D)
    std::vector<XFmatcher::DMatch> matches;
    std::vector<double> vTimesTrack(n);
    long total_matches = 0, total_valid = 0;
    for (int ni = 0; ni < n; ++ni) {
        if (nsyn) synth_frame(im, H, W, ni); else if (!load_pgm(files[ni], im)) { fprintf(stderr, "cannot read %s\n", files[ni].c_str()); return 2; }
        const auto t1 = std::chrono::steady_clock::now();
        const int ret = extractor(im, Mat(), keys, desc, lap);
        if (ret < 0) { fprintf(stderr, "empty image at %d\n", ni); return 1; }
        if (!prev.empty() && !desc.empty()) { matcher.match(prev, desc, matches); total_matches += (long)matches.size(); }
        const auto t2 = std::chrono::steady_clock::now();
        vTimesTrack[ni] = std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
        for (auto& k : keys) total_valid += k.size > 0 ? 1 : 0;
        prev = desc;
    }
    std::sort(vTimesTrack.begin(), vTimesTrack.end());                                         // rgbd_tum.cc:128-139
    double tot = 0; for (double t : vTimesTrack) tot += t;
    printf("-------\n\nframes: %d  keypoints/frame: %.1f  mutual matches/frame pair: %.1f\n", n, (double)total_valid / n, n > 1 ? (double)total_matches / (n - 1) : 0.0);
    printf("median front-end time: %f\nmean front-end time: %f\n", vTimesTrack[n / 2], tot / n);
    return 0;
}
