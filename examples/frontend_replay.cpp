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
//   --fast: the library's device-resident path through the C ABI instead of the drop-in classes: the frame goes up once,
//        xfh_extract_batch_device_images leaves the record AND the matcher's prepared image of its descriptors in HBM, the match against the
//        previous frame is xfh_match_mnn_prepared_device on the two images (two launches, no descriptor ever crosses PCIe, no normalisation
//        pass), and only the keypoints (header + 28 B per row) and the match list come back.  Same keypoints and the same match lists as the
//        default mode, bit for bit.  --valid-only (with --fast): xfh_match_records_device, i.e. pairs that touch a padding slot are dropped
//        (NOT what the reference's match() returns).
//   --window K (with --fast): every frame is matched against its K predecessors (the tracker's previous frame and the key frames / loop candidates
//        behind it: one ORBmatcher::match per pair in the reference, src/ORBmatcher.cc:358-372) in ONE call, xfh_match_mnn_prepared_batch_device.  The
//        pair (t - 1, t) is what the default mode reports; --dump-window file: per frame the number of partners, then per partner its frame index and
//        the (idx1 = partner's slot, idx2 = this frame's slot, dist) list.
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
    bool fast = false, valid_only = false;
    int window = 1; FILE* dumpw = nullptr;
    for (int i = 4; i < argc; ++i) {
        if (std::string(argv[i]) == "--dump" && i + 1 < argc) dump = fopen(argv[i + 1], "wb");
        if (std::string(argv[i]) == "--rgb" && i + 1 < argc) g_rgb = atoi(argv[i + 1]);
        if (std::string(argv[i]) == "--fast") fast = true;
        if (std::string(argv[i]) == "--valid-only") valid_only = true;
        if (std::string(argv[i]) == "--window" && i + 1 < argc) window = atoi(argv[i + 1]);
        if (std::string(argv[i]) == "--dump-window" && i + 1 < argc) dumpw = fopen(argv[i + 1], "wb");
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
    // --fast: device buffers of the C ABI path (two generations: frame t and frame t - 1)
    xfh_ctx* ctx = extractor.context();
    const size_t rec_bytes = xfh_record_bytes(nfeatures), img_bytes = xfh_match_image_bytes(nfeatures);
    if (window < 1 || window > 15) { fprintf(stderr, "--window 1 .. 15\n"); return 2; }
    const int NG = window + 1;                                                                  // generations kept in HBM: this frame and its `window` predecessors
    void* d_gray = nullptr; std::vector<void*> d_rec(NG, nullptr), d_img(NG, nullptr); void* d_out = nullptr;
    std::vector<int> gen_valid(NG, 0);
    std::vector<unsigned char> h_head(xfh_record_desc_offset(nfeatures));
    std::vector<int> h_i1(nfeatures), h_i2(nfeatures); std::vector<float> h_d(nfeatures);
    if (fast) {
        bool ok = xfh_dev_alloc(&d_gray, (size_t)im.rows * im.cols) == XFH_OK && xfh_dev_alloc(&d_out, (size_t)window * ((size_t)nfeatures * 12 + 64)) == XFH_OK;
        for (int g = 0; g < NG; ++g) ok = ok && xfh_dev_alloc(&d_rec[g], rec_bytes) == XFH_OK && xfh_dev_alloc(&d_img[g], img_bytes) == XFH_OK;
        if (!ok) { fprintf(stderr, "out of device memory\n"); return 1; }
    }
    int prev_valid = 0;
    for (int ni = 0; ni < n; ++ni) {
        if (nsyn) synth_frame(im, H, W, ni); else if (!load_frame(files[ni], im)) { fprintf(stderr, "cannot read %s\n", files[ni].c_str()); return 2; }
        const auto t1 = std::chrono::steady_clock::now();
        matches.clear();
        if (!fast) {
            const int ret = extractor(im, Mat(), keys, desc, lap);
            if (ret < 0) { fprintf(stderr, "empty image at %d\n", ni); return 1; }
            if (!prev.empty() && !desc.empty()) matcher.match(prev, desc, matches);
        } else {
            const int g = ni % NG, gp = (ni + NG - 1) % NG;          // this frame's generation, the previous frame's
            char* o = (char*)d_out;
            const size_t ostride = (size_t)nfeatures * 12 + 64;      // per partner: n_matches, then idx1 / idx2 / dist
            int rc = xfh_memcpy_h2d(d_gray, im.data, (size_t)im.rows * im.cols);
            if (rc == XFH_OK) rc = xfh_extract_batch_device_images(ctx, (const uint8_t*)d_gray, 1, im.rows, im.cols, lap[0], lap[1], d_rec[g], d_img[g]);
            const bool have_pair = ni > 0 && prev_valid > 0;          // (the reference releases the descriptors of a frame without keypoints: nothing to match)
            std::vector<int> partners;                                // frames ni - 1, ni - 2, ... that have keypoints (the first one is the default mode's pair)
            for (int k = 1; k <= window && k <= ni; ++k) if (gen_valid[(ni + NG - k) % NG] > 0) partners.push_back(ni - k);
            if (rc == XFH_OK && window > 1 && !partners.empty()) {
                // one call for all partners: pair p = (partner's image, this frame's image)
                const int P = (int)partners.size();
                std::vector<const void*> a1(P), a2(P, d_img[g]); std::vector<int> nn(P, nfeatures);
                std::vector<int*> p1(P), p2(P); std::vector<float*> pd(P);
                for (int p = 0; p < P; ++p) {
                    a1[p] = d_img[partners[p] % NG];
                    p1[p] = (int*)(o + p * ostride + 64); p2[p] = (int*)(o + p * ostride + 64 + 4 * (size_t)nfeatures); pd[p] = (float*)(o + p * ostride + 64 + 8 * (size_t)nfeatures);
                }
                std::vector<int> cnt(P, 0);
                void* d_cnt = nullptr;
                if (xfh_dev_alloc(&d_cnt, (size_t)P * 4 + 16) != XFH_OK) { fprintf(stderr, "out of device memory\n"); return 1; }
                rc = xfh_match_mnn_prepared_batch_device(ctx, P, a1.data(), nn.data(), a2.data(), nn.data(), -1.f, p1.data(), p2.data(), pd.data(), (int*)d_cnt);
                if (rc == XFH_OK) rc = xfh_synchronize(ctx);
                if (rc == XFH_OK) rc = xfh_memcpy_d2h(cnt.data(), d_cnt, (size_t)P * 4);
                xfh_dev_free(d_cnt);
                if (rc == XFH_OK) for (int p = 0; p < P; ++p) xfh_memcpy_h2d(o + p * ostride, &cnt[p], 4);     // n_matches where the one-pair path puts it
                if (rc == XFH_OK && dumpw) {
                    fwrite(&P, 4, 1, dumpw);
                    for (int p = 0; p < P; ++p) {
                        const int nm = cnt[p];
                        if (nm < 0 || nm > nfeatures) { fprintf(stderr, "fast path: the matcher reported a time-out\n"); return 1; }
                        std::vector<int> w1(nm), w2(nm); std::vector<float> wd(nm);
                        if (nm > 0) { xfh_memcpy_d2h(w1.data(), p1[p], 4 * (size_t)nm); xfh_memcpy_d2h(w2.data(), p2[p], 4 * (size_t)nm); xfh_memcpy_d2h(wd.data(), pd[p], 4 * (size_t)nm); }
                        fwrite(&partners[p], 4, 1, dumpw); fwrite(&nm, 4, 1, dumpw);
                        for (int q = 0; q < nm; ++q) { fwrite(&w1[q], 4, 1, dumpw); fwrite(&w2[q], 4, 1, dumpw); fwrite(&wd[q], 4, 1, dumpw); }
                    }
                }
                if (!(have_pair && partners[0] == ni - 1)) { const int zero = 0; xfh_memcpy_h2d(o, &zero, 4); }
            } else if (rc == XFH_OK && dumpw && window > 1) { const int P = 0; fwrite(&P, 4, 1, dumpw); }
            if (rc == XFH_OK && have_pair && window == 1) {
                if (valid_only) rc = xfh_match_records_device(ctx, d_rec[gp], d_img[gp], d_rec[g], d_img[g], -1.f, (int*)(o + 64), (int*)(o + 64 + 4 * (size_t)nfeatures),
                                                              (float*)(o + 64 + 8 * (size_t)nfeatures), (int*)o);
                else rc = xfh_match_mnn_prepared_device(ctx, d_img[gp], nfeatures, d_img[g], nfeatures, -1.f, (int*)(o + 64), (int*)(o + 64 + 4 * (size_t)nfeatures),
                                                        (float*)(o + 64 + 8 * (size_t)nfeatures), (int*)o);
            }
            if (rc == XFH_OK) rc = xfh_synchronize(ctx);
            if (rc == XFH_OK) rc = xfh_memcpy_d2h(h_head.data(), d_rec[g], h_head.size());       // header + keypoints: 28 B per row, no descriptors
            if (rc != XFH_OK) { fprintf(stderr, "fast path: %s (%s)\n", xfh_strerror(rc), xfh_last_hip_error(ctx)); return 1; }
            const int* hdr = (const int*)h_head.data();
            const xfh_keypoint* kp = (const xfh_keypoint*)(h_head.data() + xfh_record_kps_offset());
            keys.resize(nfeatures);
            for (int q = 0; q < nfeatures; ++q) { keys[q].pt.x = kp[q].x; keys[q].pt.y = kp[q].y; keys[q].size = kp[q].size; keys[q].angle = kp[q].angle; keys[q].response = kp[q].response; keys[q].octave = kp[q].octave; keys[q].class_id = kp[q].class_id; }
            int nm = 0;
            if (have_pair && hdr[0] > 0) {
                xfh_memcpy_d2h(&nm, o, 4);
                if (nm < 0 || nm > nfeatures) { fprintf(stderr, "fast path: the matcher reported a time-out\n"); return 1; }
                if (nm > 0) { xfh_memcpy_d2h(h_i1.data(), o + 64, 4 * (size_t)nm); xfh_memcpy_d2h(h_i2.data(), o + 64 + 4 * (size_t)nfeatures, 4 * (size_t)nm); xfh_memcpy_d2h(h_d.data(), o + 64 + 8 * (size_t)nfeatures, 4 * (size_t)nm); }
                for (int q = 0; q < nm; ++q) matches.emplace_back(XFmatcher::DMatch(h_i1[q], h_i2[q], h_d[q]));
            }
            prev_valid = hdr[0];
            gen_valid[g] = hdr[0];
        }
        total_matches += (long)matches.size();
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
    if (dumpw) fclose(dumpw);
    if (fast) { xfh_dev_free(d_gray); xfh_dev_free(d_out); for (int g = 0; g < NG; ++g) { xfh_dev_free(d_rec[g]); xfh_dev_free(d_img[g]); } }
    printf("-------\n\nframes: %d  keypoints/frame: %.1f  mutual matches/frame pair: %.1f  inliers/frame pair: %.1f\n", n, (double)total_valid / n,
           n > 1 ? (double)total_matches / (n - 1) : 0.0, n > 1 ? (double)total_inliers / (n - 1) : 0.0);
    printf("median front-end time: %f\nmean front-end time: %f\n", vTimesTrack[n / 2], tot / n);
    return 0;
}
