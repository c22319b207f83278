// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A minimal stand-in for the OpenCV types and calls that the reference's
// own ORB extractor sources use (src/PLPSLAM/feature/orb_extractor{,_node}.cc, orb_params.cc, util/trigonometric.h,
// match/base.h, match/angle_checker.h), so that THOSE FILES compile unmodified, from where they lie under
// /root/reference, into oracle/_ref/libplpref.so (recipe: oracle/ref_build.sh).  OpenCV itself is not available in the
// build container; the four image primitives (cv::resize, cv::FAST, cv::GaussianBlur, cv::fastAtan2) forward to the
// restatement in oracle/cv_restated.hpp.  What oracle/_ref therefore pins is every line the REFERENCE owns on this
// path: cell / ROI / border logic, the threshold fallback, masks, the std::list quadtree with its pointer-ordered
// pool, orientation, the rBRIEF rotation and its trigonometric polynomials, scale correction, tables, Hamming
// distances and the angle histogram.  It does not pin the OpenCV primitives (see cv_restated.hpp's header).
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../cv_restated.hpp"

typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_PI 3.1415926535897932384626433832795
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

inline int cvRound(double v) { return oracle::cv_round(v); }
inline int cvRound(float v) { return oracle::cv_round(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { return oracle::cv_floor(v); }
inline int cvFloor(float v) { return oracle::cv_floor(v); }
inline int cvCeil(double v) { return oracle::cv_ceil(v); }
inline int cvCeil(float v) { return oracle::cv_ceil((double)v); }

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
    Point_& operator*=(T s) { x *= s; y *= s; return *this; }
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
template <typename T> inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x + b.x, a.y + b.y); }
template <typename T> inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }
template <typename T, typename S> inline Point_<T> operator*(const Point_<T>& a, S s) { return Point_<T>((T)(a.x * s), (T)(a.y * s)); }

struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Scalar { double val[4]; Scalar(double v0 = 0) : val{v0, 0, 0, 0} {} };
struct Range { int start, end; Range(int s, int e) : start(s), end(e) {} };

struct KeyPoint {   // 28 bytes, the layout of cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

struct MatZeros { int rows, cols, type; };

// single-channel 8-bit matrix header with shared ownership and row/column views
class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;
    Mat() = default;
    Mat(int r, int c, int /*type*/) { create(r, c, 0); }
    Mat(int r, int c, int /*type*/, const Scalar& s) { create(r, c, 0); std::memset(data, (int)s.val[0], (size_t)r * c); }
    Mat(int r, int c, int /*type*/, void* ext, size_t st = 0) : rows(r), cols(c), data((uchar*)ext), step(st ? st : (size_t)c) {}
    void create(int r, int c, int /*type*/) {
        if (data && r == rows && c == cols) return;
        buf_ = std::make_shared<std::vector<uchar>>((size_t)r * c);
        rows = r; cols = c; step = (size_t)c; data = buf_->data();
    }
    void release() { buf_.reset(); rows = cols = 0; data = nullptr; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    size_t step1() const { return step; }
    Mat clone() const {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, 0);
        for (int y = 0; y < rows; ++y) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, cols);
        return m;
    }
    Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t)a * step; m.rows = b - a; return m; }
    Mat colRange(int a, int b) const { Mat m = *this; m.data = data + a; m.cols = b - a; return m; }
    template <typename T> T& at(int y, int x) { return *reinterpret_cast<T*>(data + (size_t)y * step + x * sizeof(T)); }
    template <typename T> const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + (size_t)y * step + x * sizeof(T)); }
    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T* ptr(unsigned y) { return ptr<T>((int)y); }
    template <typename T> const T* ptr(unsigned y) const { return ptr<T>((int)y); }
    template <typename T> T* ptr(size_t y) { return ptr<T>((int)y); }
    template <typename T> T* ptr(int y = 0) { return reinterpret_cast<T*>(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + (size_t)y * step); }
    void copyTo(const class _OutputArray& dst) const;
    static MatZeros zeros(int r, int c, int type) { return MatZeros{r, c, type}; }
    // Mat = Mat::zeros(...) evaluates the expression INTO an existing matrix of the same size (cv::MatExpr semantics): a
    // row-range view keeps pointing into its parent
    Mat& operator=(const MatZeros& z) {
        if (!(data && rows == z.rows && cols == z.cols)) create(z.rows, z.cols, 0);
        for (int y = 0; y < rows; ++y) std::memset(data + (size_t)y * step, 0, cols);
        return *this;
    }
    Mat(const MatZeros& z) { create(z.rows, z.cols, 0); std::memset(data, 0, (size_t)rows * cols); }

private:
    std::shared_ptr<std::vector<uchar>> buf_;
};

class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
protected:
    Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) { m_ = &m; }
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
};
inline void Mat::copyTo(const _OutputArray& dst) const {
    dst.create(rows, cols, 0);
    Mat d = dst.getMat();
    for (int y = 0; y < rows; ++y) std::memcpy(d.data + (size_t)y * d.step, data + (size_t)y * step, cols);
}
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

enum { INTER_LINEAR = 1 };
enum { BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4 };
enum { LINE_AA = 16 };

inline oracle::Image to_image(const Mat& m) {
    oracle::Image im(m.rows, m.cols);
    for (int y = 0; y < m.rows; ++y) std::memcpy(im.row(y), m.data + (size_t)y * m.step, m.cols);
    return im;
}
inline void from_image(const oracle::Image& im, Mat& m) {
    m.create(im.rows, im.cols, 0);
    for (int y = 0; y < im.rows; ++y) std::memcpy(m.data + (size_t)y * m.step, im.row(y), im.cols);
}

// ---- the four OpenCV primitives on the path: forwarded to the restatement (cv_restated.hpp)
inline void resize(const Mat& src, Mat& dst, Size dsize, double, double, int) {
    Mat out;   // src and dst may be the same object
    from_image(oracle::resize_linear_u8(to_image(src), dsize.width, dsize.height), out);
    dst = out;
}
inline void FAST(const Mat& roi, std::vector<KeyPoint>& kps, int threshold, bool nonmax) {
    assert(nonmax);
    std::vector<oracle::FastPoint> pts;
    oracle::fast9_16_nms(roi.data, (int)roi.step, roi.cols, roi.rows, threshold, pts);
    kps.clear();
    for (const auto& p : pts) kps.emplace_back((float)p.x, (float)p.y, 7.f, -1.f, (float)p.score);
}
inline void GaussianBlur(const Mat& src, Mat& dst, Size k, double sx, double, int) {
    Mat out;
    from_image(oracle::gaussian_blur_u8(to_image(src), k.width, sx), out);
    if (dst.data && dst.rows == out.rows && dst.cols == out.cols)
        for (int y = 0; y < out.rows; ++y) std::memcpy(dst.data + (size_t)y * dst.step, out.data + (size_t)y * out.step, out.cols);
    else dst = out;
}
inline float fastAtan2(float y, float x) { return oracle::fast_atan2f_deg(y, x); }
// filled axis-aligned rectangle, both corners inclusive (thickness -1; LINE_AA does not soften a filled rectangle)
inline void rectangle(Mat& img, Point2i a, Point2i b, const Scalar& color, int thickness, int) {
    assert(thickness < 0);
    const int x0 = std::max(0, std::min(a.x, b.x)), x1 = std::min(img.cols - 1, std::max(a.x, b.x));
    const int y0 = std::max(0, std::min(a.y, b.y)), y1 = std::min(img.rows - 1, std::max(a.y, b.y));
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) img.at<uchar>(y, x) = (uchar)color.val[0];
}

}  // namespace cv
