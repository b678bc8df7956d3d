// image_io.h -- minimal image input for the replay harness (examples/frontend_replay.cpp) where OpenCV is absent:
// binary PGM (P5) and non-interlaced 8-bit PNG (gray, gray+alpha, RGB, RGBA; zlib inflate + the five PNG filters),
// converted to the CV_8UC1 image the extractor takes.
//
// The reference reads frames with cv::imread(..., IMREAD_UNCHANGED) (examples/RGB-D/rgbd_tum.cc:78), which returns colour
// images in B,G,R memory order, and Tracking::GrabImageRGBD converts with cv::COLOR_RGB2GRAY when Camera.RGB is 1 (TUM1.yaml:29,
// src/Tracking.cc:1534-1537) -- i.e. it weights the BLUE channel of a TUM PNG with the red coefficient.  to_gray(rgb_flag)
// reproduces exactly that: OpenCV's 8-bit fixed-point formula (Y = (c0*R2Y + c1*G2Y + c2*B2Y + 2^13) >> 14 with R2Y 4899,
// G2Y 9617, B2Y 1868 -- OpenCV 4.5.4 imgproc color_yuv, restated; the library is not vendored in the reference tree) applied
// to the memory order imread would have produced.
#ifndef XFEAT_IMAGE_IO_H
#define XFEAT_IMAGE_IO_H
#include <cctype>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <zlib.h>

namespace xfeat {

struct Image8 { int rows = 0, cols = 0, channels = 0; std::vector<unsigned char> data; };   // channels in FILE order (R,G,B[,A])

static const int kMaxSide = 16384;      // largest image side a file header may claim (TUM / EuRoC / KITTI frames are < 2k)

inline bool load_pgm(const std::string& path, Image8& im) {
    std::ifstream f(path, std::ios::binary);
    std::string magic; int w = 0, h = 0, maxv = 0;
    if (!(f >> magic) || magic != "P5") return false;
    auto skip = [&]() { while (f.peek() == '#' || isspace(f.peek())) { if (f.peek() == '#') { std::string l; std::getline(f, l); } else f.get(); } };
    skip(); f >> w; skip(); f >> h; skip(); f >> maxv; f.get();
    if (w <= 0 || h <= 0 || w > kMaxSide || h > kMaxSide || maxv != 255) return false;       // bounded before anything is allocated
    im.rows = h; im.cols = w; im.channels = 1; im.data.resize((size_t)w * h);
    f.read((char*)im.data.data(), (std::streamsize)w * h);
    return (bool)f;
}

inline bool load_png(const std::string& path, Image8& im) {
    // width / height come from the file: they are bounded by kMaxSide before any size is computed from them
    std::ifstream f(path, std::ios::binary);
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 33 || memcmp(buf.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)buf[o] << 24) | ((uint32_t)buf[o + 1] << 16) | ((uint32_t)buf[o + 2] << 8) | (uint32_t)buf[o + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat;
    for (size_t o = 8; o + 12 <= buf.size();) {
        const uint32_t len = be32(o);
        if (o + 12 + (size_t)len > buf.size()) return false;
        const char* ty = (const char*)&buf[o + 4];
        if (!memcmp(ty, "IHDR", 4) && len >= 13 && ctype < 0) { w = be32(o + 8);       /* only the first IHDR counts */ h = be32(o + 12); depth = buf[o + 16]; ctype = buf[o + 17]; interlace = buf[o + 20]; }
        else if (!memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), buf.begin() + o + 8, buf.begin() + o + 8 + len);
        else if (!memcmp(ty, "IEND", 4)) break;
        o += 12 + (size_t)len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!w || !h || depth != 8 || !ch || interlace) return false;          // palette / 16-bit / Adam7 are not needed for TUM or EuRoC frames
    if (w > (uint32_t)kMaxSide || h > (uint32_t)kMaxSide) return false;    // a crafted header must not size the allocations below
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    im.rows = (int)h; im.cols = (int)w; im.channels = ch; im.data.assign(stride * h, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = &im.data[stride * y];
        const unsigned char* up = y ? cur - stride : nullptr;
        const int ft = in[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: return false;
            }
            cur[x] = (unsigned char)(in[1 + x] + pred);
        }
    }
    return true;
}

inline bool load_image(const std::string& path, Image8& im) {
    const size_t d = path.find_last_of('.');
    std::string ext = d == std::string::npos ? "" : path.substr(d + 1);
    for (auto& c : ext) c = (char)tolower(c);
    return ext == "png" ? load_png(path, im) : load_pgm(path, im);
}

// CV_8UC1 image as Tracking::GrabImage* would hand it to the extractor (see the header comment); rgb_flag = Camera.RGB
inline void to_gray(const Image8& im, int rgb_flag, std::vector<unsigned char>& gray) {
    gray.resize((size_t)im.rows * im.cols);
    if (im.channels == 1) { gray = im.data; return; }
    const int R2Y = 4899, G2Y = 9617, B2Y = 1868;
    for (size_t p = 0; p < gray.size(); ++p) {
        const unsigned char* px = &im.data[p * im.channels];
        if (im.channels == 2) { gray[p] = px[0]; continue; }            // gray + alpha: IMREAD_UNCHANGED keeps 2 channels; take luminance
        const int fr = px[0], fg = px[1], fb = px[2];                   // file order R, G, B  ->  imread memory order B, G, R
        const int c0 = fb, c1 = fg, c2 = fr;                            // memory channels 0, 1, 2
        // COLOR_RGB2GRAY reads memory channel 0 as R; COLOR_BGR2GRAY reads it as B
        const int y = rgb_flag ? (c0 * R2Y + c1 * G2Y + c2 * B2Y) : (c0 * B2Y + c1 * G2Y + c2 * R2Y);
        gray[p] = (unsigned char)((y + (1 << 13)) >> 14);
    }
}

}  // namespace xfeat
#endif
