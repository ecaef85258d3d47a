// oracle/cvshim/cvshim.hpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// A minimal stand-in for the slice of the OpenCV API that the reference's
// src/ORBextractor.cc + include/ORBextractor.h use, so that THE REFERENCE FILE
// COMPILES UNMODIFIED from where it lies under /root/reference (oracle/Makefile,
// output oracle/_ref/liborbref.so).  All ORB-SLAM2-owned logic (cell grid,
// quadtree, orientation, descriptor, pyramid orchestration) is then literally
// the reference's; only the OpenCV primitives are supplied from oracle/prims.h.
//
// Behaviours the reference relies on and that are reproduced here on purpose
// (SURVEY.md Appendix E):
//   * Mat::create() is a no-op when shape/type already match, so writing through
//     an ROI header (resize into mvImagePyramid[level], copyMakeBorder into `temp`)
//     lands in the parent buffer (src/ORBextractor.cc:1687-1701, 1728-1730);
//   * `m = Mat::zeros(r,c,t)` zero-fills IN PLACE when m already has that shape
//     (src/ORBextractor.cc:1531 writing through the rowRange view of :1638);
//   * cv::Point is two packed ints (cast from int[] at :560), cv::KeyPoint is the
//     28-byte {pt, size, angle, response, octave, class_id}.
#ifndef ORB_ORACLE_CVSHIM_HPP
#define ORB_ORACLE_CVSHIM_HPP

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../prims.h"

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

typedef unsigned char uchar;

static inline int cvRound(double v) { return op_round_d(v); }
static inline int cvRound(float v) { return op_round_f(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) { return op_floor_d(v); }
static inline int cvFloor(float v) { return op_floor_d(v); }
static inline int cvCeil(double v) { return op_ceil_d(v); }
static inline int cvCeil(float v) { return op_ceil_d(v); }

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
    template <typename U> Point_(const Point_<U> &p) : x((T)p.x), y((T)p.y) {}
    Point_ &operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {}
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};
static_assert(sizeof(Point) == 8, "cv::Point must be two packed ints");
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint must be 28 bytes");

enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4,
       BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };

struct MatZeros { int rows, cols, type; };

// 8-bit single-channel matrix header with a malloc-owned, ref-counted buffer.
// (malloc, not operator new: pyramid buffers outlive the per-frame arena that
// oracle/ref_wrap.cc installs for operator new.)
class Mat {
public:
    int rows, cols;
    size_t step;
    uchar *data;

    Mat() : rows(0), cols(0), step(0), data(0), rc_(0) {}
    Mat(int r, int c, int t) : rows(0), cols(0), step(0), data(0), rc_(0) { create(r, c, t); }
    Mat(Size s, int t) : rows(0), cols(0), step(0), data(0), rc_(0) { create(s.height, s.width, t); }
    // external (non-owned) buffer
    Mat(int r, int c, int t, void *ext, size_t st) : rows(r), cols(c), step(st), data((uchar *)ext), rc_(0) { assert(t == CV_8UC1); }
    Mat(const Mat &m) : rows(m.rows), cols(m.cols), step(m.step), data(m.data), rc_(m.rc_) { if (rc_) ++*rc_; }
    ~Mat() { release(); }
    Mat &operator=(const Mat &m)
    {
        if (this != &m) {
            if (m.rc_) ++*m.rc_;
            release();
            rows = m.rows; cols = m.cols; step = m.step; data = m.data; rc_ = m.rc_;
        }
        return *this;
    }
    Mat &operator=(const MatZeros &z)
    {
        create(z.rows, z.cols, z.type);
        for (int i = 0; i < rows; i++) memset(data + (size_t)i * step, 0, (size_t)cols);
        return *this;
    }
    void release()
    {
        if (rc_ && --*rc_ == 0) free(rc_);
        rc_ = 0; data = 0; rows = cols = 0; step = 0;
    }
    void create(int r, int c, int t)
    {
        assert(t == CV_8UC1);
        if (data && rows == r && cols == c) return;   // OpenCV: no-op when shape matches
        release();
        rows = r; cols = c; step = (size_t)c;
        size_t bytes = (size_t)r * c;
        int *blk = (int *)malloc(64 + (bytes ? bytes : 1));
        *blk = 1;
        rc_ = blk;
        data = (uchar *)blk + 64;
    }
    void create(Size s, int t) { create(s.height, s.width, t); }
    static MatZeros zeros(int r, int c, int t) { MatZeros z = {r, c, t}; return z; }
    int type() const { return CV_8UC1; }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    size_t step1() const { return step; }
    bool isContinuous() const { return step == (size_t)cols; }
    Mat operator()(const Rect &r) const { return roi(r.y, r.y + r.height, r.x, r.x + r.width); }
    Mat rowRange(int a, int b) const { return roi(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return roi(0, rows, a, b); }
    Mat clone() const
    {
        Mat m(rows, cols, CV_8UC1);
        for (int i = 0; i < rows; i++) memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols);
        return m;
    }
    template <typename T> T &at(int r, int c) { return *(T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const { return *(const T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    uchar *ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }

private:
    Mat roi(int r0, int r1, int c0, int c1) const
    {
        assert(0 <= r0 && r0 <= r1 && r1 <= rows && 0 <= c0 && c0 <= c1 && c1 <= cols);
        Mat m(*this);
        m.data = data + (size_t)r0 * step + c0;
        m.rows = r1 - r0; m.cols = c1 - c0;
        return m;
    }
    int *rc_;
};

// InputArray / OutputArray: thin proxies around a Mat.
class _InputArray {
public:
    _InputArray() : m_(0) {}
    _InputArray(const Mat &m) : m_(const_cast<Mat *>(&m)) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
protected:
    Mat *m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat &m) : _InputArray(m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void create(Size s, int t) const { m_->create(s, t); }
    void release() const { m_->release(); }
    Mat &getMatRef() const { return *m_; }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

inline float fastAtan2(float y, float x) { return op_fast_atan2(y, x); }

// cv::FAST(image, keypoints, threshold, nonmaxSuppression) == TYPE_9_16.
inline void FAST(InputArray _img, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true)
{
    Mat img = _img.getMat();
    keypoints.clear();
    assert(nonmaxSuppression);
    int cap = img.rows * img.cols;
    std::vector<op_fast_kp> tmp((size_t)(cap > 0 ? cap : 1));
    int n = op_fast9_nms(img.data, img.rows, img.cols, img.step, threshold, tmp.data(), cap);
    for (int i = 0; i < n; i++)
        keypoints.push_back(KeyPoint((float)tmp[i].x, (float)tmp[i].y, 7.f, -1, (float)tmp[i].score));
}

inline void resize(InputArray _src, OutputArray _dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR)
{
    (void)fx; (void)fy;
    assert(interpolation == INTER_LINEAR && dsize.width > 0 && dsize.height > 0);
    Mat src = _src.getMat();
    _dst.create(dsize, CV_8UC1);
    Mat &dst = _dst.getMatRef();
    assert(src.data != dst.data);
    op_resize_linear_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step);
}

inline void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType)
{
    Mat src = _src.getMat();
    int bt = borderType & ~BORDER_ISOLATED;
    assert(bt == BORDER_REFLECT_101);
    (void)bt;
    _dst.create(src.rows + top + bottom, src.cols + left + right, CV_8UC1);
    Mat &dst = _dst.getMatRef();
    // rows of the interior may alias src (src is an ROI of dst in the reference): build
    // each output row from the source row into a scratch line first.
    std::vector<uchar> line((size_t)dst.cols);
    std::vector<uchar> srccopy((size_t)src.rows * src.cols);
    for (int y = 0; y < src.rows; y++) memcpy(&srccopy[(size_t)y * src.cols], src.ptr(y), (size_t)src.cols);
    for (int y = 0; y < dst.rows; y++) {
        int sy = op_reflect101(y - top, src.rows);
        const uchar *s = &srccopy[(size_t)sy * src.cols];
        for (int x = 0; x < dst.cols; x++) line[(size_t)x] = s[op_reflect101(x - left, src.cols)];
        memcpy(dst.ptr(y), line.data(), (size_t)dst.cols);
    }
}

inline void GaussianBlur(InputArray _src, OutputArray _dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT)
{
    assert(ksize.width == 7 && ksize.height == 7 && sigmaX == 2 && sigmaY == 2 && borderType == BORDER_REFLECT_101);
    (void)ksize; (void)sigmaX; (void)sigmaY; (void)borderType;
    Mat src = _src.getMat();
    _dst.create(src.rows, src.cols, CV_8UC1);
    Mat &dst = _dst.getMatRef();
    op_gauss7_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.step, NULL);
}

struct KeyPointsFilter {
    // only referenced from the dead ComputeKeyPointsOld (src/ORBextractor.cc:1203-1514)
    static void retainBest(std::vector<KeyPoint> &, int) { abort(); }
};

}  // namespace cv

#endif
