// tests/stubs/opencv_api/opencv2/core/core.hpp -- NOT OpenCV.  TEST INFRASTRUCTURE ONLY.
//
// An API-shaped stand-in for the handful of OpenCV 4.x members that the `#if XFEAT_HAVE_OPENCV` branches of include/xfeat/XFextractor.h and
// include/xfeat/ORBmatcher_xfeat.h use (cv::Mat, cv::KeyPoint, cv::DMatch, cv::InputArray / cv::OutputArray), with the documented behaviour of
// exactly those members (Mat headers share their buffer; _OutputArray::create keeps an array that already has the requested shape and type --
// also a non-continuous one; copyTo honours the row step).  The image of this repository has no OpenCV, so without this header those branches
// never meet a compiler: tests/cpp/cv_branch_test.cpp compiles and runs them against it (tests/test_abi_and_host.py: compile, tests/
// test_gpu_dropin_cpp.py: run).  It type-checks the wrappers' calls and exercises their control flow; it says nothing about the real library
// (no parity claim rests on it), and nothing outside tests/ may include it.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {
typedef unsigned char uchar;

struct Point2f { float x = 0.f, y = 0.f; Point2f() = default; Point2f(float x_, float y_) : x(x_), y(y_) {} };

class KeyPoint {
public:
    KeyPoint() = default;
    KeyPoint(float x, float y, float s, float a = -1.f, float r = 0.f, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
    Point2f pt; float size = 0.f, angle = -1.f, response = 0.f; int octave = 0, class_id = -1;
};

class DMatch {
public:
    DMatch() = default;
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 3.402823466e+38f;
};

class _InputArray;
class _OutputArray;
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

class Mat {
public:
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    // a header over caller-owned memory with a row step in bytes (OpenCV: Mat(rows, cols, type, data, step))
    Mat(int r, int c, int type, void* d, size_t step_bytes) : rows(r), cols(c), data((uchar*)d), step(step_bytes ? step_bytes : (size_t)c * esz(type)), type_(type) {}
    static size_t esz(int type) { const int depth = type & 7, cn = (type >> CV_CN_SHIFT) + 1; return (size_t)(depth == CV_32F ? 4 : 1) * cn; }
    void create(int r, int c, int type) {
        if (data && rows == r && cols == c && type_ == type) return;              // cv::Mat::create: nothing to do for the same shape and type
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        store = std::shared_ptr<uchar>(new uchar[(size_t)r * step + 16], std::default_delete<uchar[]>());
        data = store.get();
    }
    void release() { store.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const { return esz(type_); }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    inline void copyTo(OutputArray dst) const;
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;                                        // (OpenCV: MatStep, converts to size_t)
private:
    int type_ = 0;
    std::shared_ptr<uchar> store;                           // headers share the buffer, as cv::Mat's reference count does
};

class _InputArray {
public:
    _InputArray() = default;
    _InputArray(const Mat& m) : mat(const_cast<Mat*>(&m)) {}
    Mat getMat(int = -1) const { return mat ? *mat : Mat(); }
    bool empty() const { return !mat || mat->empty(); }
protected:
    Mat* mat = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() = default;
    _OutputArray(Mat& m) : _InputArray(m) {}
    void create(int r, int c, int type) const { if (mat) mat->create(r, c, type); }
    void release() const { if (mat) mat->release(); }
};
inline void Mat::copyTo(OutputArray dst) const {
    dst.create(rows, cols, type_);
    Mat d = dst.getMat();
    for (int r = 0; r < rows; ++r) std::memcpy(d.data + (size_t)r * d.step, data + (size_t)r * step, (size_t)cols * elemSize());
}
}  // namespace cv
