// oracle/cvshim/cvshim.hpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// A minimal stand-in for the slice of the OpenCV API that the reference's hot-path
// sources use, so that THE REFERENCE FILES COMPILE UNMODIFIED from where they lie under
// /root/reference (oracle/Makefile):
//   oracle/_ref/liborbref.so      src/ORBextractor.cc
//   oracle/_ref/liborbslam.so     + src/ORBmatcher.cc, Frame.cc, KeyFrame.cc, MapPoint.cc,
//                                   Map.cc, KeyFrameDatabase.cc and the vendored DBoW2
// All ORB-SLAM2-owned logic is then literally the reference's; only the OpenCV primitives
// are supplied from here / oracle/prims.h.
//
// Behaviours the reference relies on and that are reproduced here on purpose
// (SURVEY.md Appendix E):
//   * Mat::create() is a no-op when shape/type already match, so writing through
//     an ROI header (resize into mvImagePyramid[level], copyMakeBorder into `temp`,
//     Rwc.copyTo(Twc.rowRange(0,3).colRange(0,3))) lands in the parent buffer
//     (src/ORBextractor.cc:1687-1701, 1728-1730; src/KeyFrame.cc:98-106);
//   * assigning a matrix EXPRESSION (`m = Mat::zeros(r,c,t)`, `m = a - b`) evaluates IN
//     PLACE when m already has that shape (src/ORBextractor.cc:1531 writing through the
//     rowRange view of :1638);
//   * cv::Point is two packed ints (cast from int[] at :560), cv::KeyPoint is the
//     28-byte {pt, size, angle, response, octave, class_id}.
// Float matrix algebra (operator*, dot, norm) accumulates in double like OpenCV's generic
// gemm / dotProd / norm kernels for CV_32F; the hot path only needs exact-integer cases
// (L1 norm of 11x11 pixel windows, src/Frame.cc:1292).
#ifndef ORB_ORACLE_CVSHIM_HPP
#define ORB_ORACLE_CVSHIM_HPP

#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../prims.h"

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_MAKETYPE(d, cn) (CV_MAT_DEPTH(d) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

typedef unsigned char uchar;

static inline int cvRound(double v) { return op_round_d(v); }
static inline int cvRound(float v) { return op_round_f(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) { return op_floor_d(v); }
static inline int cvFloor(float v) { return op_floor_d(v); }
static inline int cvCeil(double v) { return op_ceil_d(v); }
static inline int cvCeil(float v) { return op_ceil_d(v); }

namespace cv {

typedef std::string String;

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
typedef Point_<double> Point2d;
template <typename T> struct Point3_ {
    T x, y, z;
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T _x, T _y, T _z) : x(_x), y(_y), z(_z) {}
};
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;

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

struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
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
enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4 };

static inline size_t cvshim_depth_size(int depth)
{
    static const size_t s[8] = {1, 1, 2, 2, 4, 4, 8, 0};
    return s[depth & 7];
}

class MatExpr;

// Dense 2-D matrix header with a malloc-owned, ref-counted buffer.
// (malloc, not operator new: pyramid buffers outlive the per-frame arena that the
// wrappers install for operator new.)
class Mat {
public:
    int rows, cols;
    size_t step;   // bytes per row
    uchar *data;

    Mat() : rows(0), cols(0), step(0), data(0), rc_(0), type_(0) {}
    Mat(int r, int c, int t) : rows(0), cols(0), step(0), data(0), rc_(0), type_(0) { create(r, c, t); }
    Mat(Size s, int t) : rows(0), cols(0), step(0), data(0), rc_(0), type_(0) { create(s.height, s.width, t); }
    Mat(int r, int c, int t, const Scalar &s) : rows(0), cols(0), step(0), data(0), rc_(0), type_(0) { create(r, c, t); setTo(s.val[0]); }
    // external (non-owned) buffer; st == 0 means tight rows
    Mat(int r, int c, int t, void *ext, size_t st = 0) : rows(r), cols(c), step(st), data((uchar *)ext), rc_(0), type_(t)
    {
        if (!step) step = (size_t)c * elemSize();
    }
    Mat(const Mat &m) : rows(m.rows), cols(m.cols), step(m.step), data(m.data), rc_(m.rc_), type_(m.type_) { if (rc_) __atomic_add_fetch(rc_, 1, __ATOMIC_RELAXED); }
    ~Mat() { release(); }
#ifdef CVSHIM_CALLER_DECLS
    // COMPILE-ONLY declarations for the callers of the drop-in surface (src/Tracking.cc, src/LocalMapping.cc: `make callers`,
    // -fsyntax-only); never linked, never executed
    explicit Mat(const Point3_<float> &);
    void resize(size_t);
    void convertTo(Mat &dst, int rtype, double alpha) const;
#endif
    Mat &operator=(const Mat &m)
    {
        if (this != &m) {
            if (m.rc_) __atomic_add_fetch(m.rc_, 1, __ATOMIC_RELAXED);
            release();
            rows = m.rows; cols = m.cols; step = m.step; data = m.data; rc_ = m.rc_; type_ = m.type_;
        }
        return *this;
    }
    // OpenCV: MatExpr assignment evaluates into *this; create() keeps an existing buffer of
    // the same shape, i.e. views are written through.
    inline Mat &operator=(const MatExpr &e);
    Mat &operator=(const Scalar &s) { setTo(s.val[0]); return *this; }

    void release()
    {
        if (rc_ && __atomic_sub_fetch(rc_, 1, __ATOMIC_ACQ_REL) == 0) free(rc_);
        rc_ = 0; data = 0; rows = cols = 0; step = 0;
    }
    void create(int r, int c, int t)
    {
        if (data && rows == r && cols == c && type_ == t) return;   // OpenCV: no-op when shape matches
        release();
        type_ = t;
        rows = r; cols = c; step = (size_t)c * elemSize();
        size_t bytes = (size_t)r * step;
        int *blk = (int *)malloc(64 + (bytes ? bytes : 1));
        *blk = 1;
        rc_ = blk;
        data = (uchar *)blk + 64;
    }
    void create(Size s, int t) { create(s.height, s.width, t); }
    static inline MatExpr zeros(int r, int c, int t);
    static inline MatExpr ones(int r, int c, int t);
    static inline MatExpr eye(int r, int c, int t);
    static inline MatExpr zeros(Size s, int t);

    int type() const { return type_; }
    int depth() const { return CV_MAT_DEPTH(type_); }
    int channels() const { return CV_MAT_CN(type_); }
    size_t elemSize() const { return cvshim_depth_size(depth()) * (size_t)channels(); }
    size_t elemSize1() const { return cvshim_depth_size(depth()); }
    size_t total() const { return (size_t)rows * cols; }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    size_t step1() const { return step / elemSize1(); }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    Mat operator()(const Rect &r) const { return roi(r.y, r.y + r.height, r.x, r.x + r.width); }
    Mat rowRange(int a, int b) const { return roi(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return roi(0, rows, a, b); }
    Mat row(int i) const { return roi(i, i + 1, 0, cols); }
    Mat col(int j) const { return roi(0, rows, j, j + 1); }
    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    void copyTo(Mat &dst) const
    {
        if (dst.data == data && dst.rows == rows && dst.cols == cols && dst.step == step) return;
        dst.create(rows, cols, type_);
        size_t rb = (size_t)cols * elemSize();
        for (int i = 0; i < rows; i++) memmove(dst.data + (size_t)i * dst.step, data + (size_t)i * step, rb);
    }
    // copyTo into a temporary view header (Rwc.copyTo(Twc.rowRange(0,3).colRange(0,3)))
    void copyTo(const Mat &dstview) const { Mat d(dstview); copyTo(d); }
    void setTo(double v)
    {
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols * channels(); j++) put(i, j, v);
    }
    void convertTo(Mat &dst, int rtype) const
    {
        int t = CV_MAKETYPE(rtype, channels());
        Mat out(rows, cols, t);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols * channels(); j++) out.put(i, j, get(i, j));
        if (dst.data && dst.rows == rows && dst.cols == cols && dst.type_ == t && dst.data != data) out.copyTo(dst);
        else dst = out;
    }
    Mat reshape(int cn, int nrows = 0) const
    {
        assert(nrows == 0 && isContinuous());
        (void)nrows;
        Mat m(*this);
        int total_ch = cols * channels();
        assert(total_ch % cn == 0);
        m.cols = total_ch / cn;
        m.type_ = CV_MAKETYPE(depth(), cn);
        return m;
    }
    inline MatExpr t() const;
    inline MatExpr inv() const;
    inline MatExpr mul(const Mat &b) const;
    double dot(const Mat &b) const
    {
        assert(rows == b.rows && cols == b.cols && channels() == 1);
        double s = 0;
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) s += get(i, j) * b.get(i, j);
        return s;
    }

    template <typename T> T &at(int r, int c) { return *(T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const { return *(const T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    // single index: element i of a row or column vector (OpenCV Mat::at(int i0))
    template <typename T> T &at(int i) { return (rows == 1 || isContinuous()) ? ((T *)data)[i] : *(T *)(data + (size_t)i * step); }
    template <typename T> const T &at(int i) const { return (rows == 1 || isContinuous()) ? ((const T *)data)[i] : *(const T *)(data + (size_t)i * step); }
    uchar *ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar *ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }

    // generic element access in double (j counts scalars, i.e. col*channels + ch)
    double get(int i, int j) const
    {
        const uchar *p = data + (size_t)i * step;
        switch (depth()) {
        case CV_8U: return ((const uchar *)p)[j];
        case CV_8S: return ((const signed char *)p)[j];
        case CV_16U: return ((const unsigned short *)p)[j];
        case CV_16S: return ((const short *)p)[j];
        case CV_32S: return ((const int *)p)[j];
        case CV_32F: return ((const float *)p)[j];
        default: return ((const double *)p)[j];
        }
    }
    void put(int i, int j, double v)
    {
        uchar *p = data + (size_t)i * step;
        switch (depth()) {
        case CV_8U: { int r = cvRound(v); ((uchar *)p)[j] = (uchar)(r < 0 ? 0 : r > 255 ? 255 : r); break; }
        case CV_8S: { int r = cvRound(v); ((signed char *)p)[j] = (signed char)(r < -128 ? -128 : r > 127 ? 127 : r); break; }
        case CV_16U: { int r = cvRound(v); ((unsigned short *)p)[j] = (unsigned short)(r < 0 ? 0 : r > 65535 ? 65535 : r); break; }
        case CV_16S: { int r = cvRound(v); ((short *)p)[j] = (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r); break; }
        case CV_32S: ((int *)p)[j] = cvRound(v); break;
        case CV_32F: ((float *)p)[j] = (float)v; break;
        default: ((double *)p)[j] = v; break;
        }
    }

protected:
    Mat roi(int r0, int r1, int c0, int c1) const
    {
        assert(0 <= r0 && r0 <= r1 && r1 <= rows && 0 <= c0 && c0 <= c1 && c1 <= cols);
        Mat m(*this);
        m.data = data + (size_t)r0 * step + (size_t)c0 * elemSize();
        m.rows = r1 - r0; m.cols = c1 - c0;
        return m;
    }
    int *rc_;
    int type_;
};

// An evaluated matrix expression.  Only its assignment semantics differ from Mat.
class MatExpr : public Mat {
public:
    MatExpr() {}
    explicit MatExpr(const Mat &m) : Mat(m) {}
};

inline Mat &Mat::operator=(const MatExpr &e)
{
    if (data && rows == e.rows && cols == e.cols && type_ == e.type()) e.copyTo(*this);   // in place (views!)
    else *this = static_cast<const Mat &>(e);
    return *this;
}
inline MatExpr Mat::zeros(int r, int c, int t) { Mat m(r, c, t); m.setTo(0); return MatExpr(m); }
inline MatExpr Mat::zeros(Size s, int t) { return zeros(s.height, s.width, t); }
inline MatExpr Mat::ones(int r, int c, int t) { Mat m(r, c, t); m.setTo(1); return MatExpr(m); }
inline MatExpr Mat::eye(int r, int c, int t)
{
    Mat m(r, c, t);
    m.setTo(0);
    for (int i = 0; i < std::min(r, c); i++) m.put(i, i, 1);
    return MatExpr(m);
}
inline MatExpr Mat::t() const
{
    assert(channels() == 1);
    Mat m(cols, rows, type_);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) m.put(j, i, get(i, j));
    return MatExpr(m);
}
inline MatExpr Mat::mul(const Mat &b) const
{
    assert(rows == b.rows && cols == b.cols && type_ == b.type_);
    Mat m(rows, cols, type_);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) m.put(i, j, get(i, j) * b.get(i, j));
    return MatExpr(m);
}
inline MatExpr Mat::inv() const
{
    // Gauss-Jordan with partial pivoting in double (not on the hot path)
    assert(rows == cols && channels() == 1);
    int n = rows;
    std::vector<double> a((size_t)n * 2 * n, 0.0);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) a[(size_t)i * 2 * n + j] = get(i, j);
        a[(size_t)i * 2 * n + n + i] = 1.0;
    }
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(a[(size_t)r * 2 * n + c]) > std::fabs(a[(size_t)p * 2 * n + c])) p = r;
        if (p != c) for (int j = 0; j < 2 * n; j++) std::swap(a[(size_t)p * 2 * n + j], a[(size_t)c * 2 * n + j]);
        double d = a[(size_t)c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) a[(size_t)c * 2 * n + j] /= d;
        for (int r = 0; r < n; r++) if (r != c) {
            double f = a[(size_t)r * 2 * n + c];
            if (f != 0) for (int j = 0; j < 2 * n; j++) a[(size_t)r * 2 * n + j] -= f * a[(size_t)c * 2 * n + j];
        }
    }
    Mat m(n, n, type_);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) m.put(i, j, a[(size_t)i * 2 * n + n + j]);
    return MatExpr(m);
}

static inline MatExpr cvshim_binop(const Mat &a, const Mat &b, int op)
{
    assert(a.rows == b.rows && a.cols == b.cols && a.type() == b.type());
    Mat m(a.rows, a.cols, a.type());
    int n = a.cols * a.channels();
    if (a.depth() == CV_32F) {   // keep float arithmetic in float
        for (int i = 0; i < a.rows; i++) {
            const float *pa = a.ptr<float>(i), *pb = b.ptr<float>(i);
            float *pm = m.ptr<float>(i);
            for (int j = 0; j < n; j++) pm[j] = op == 0 ? pa[j] + pb[j] : pa[j] - pb[j];
        }
    } else {
        for (int i = 0; i < a.rows; i++)
            for (int j = 0; j < n; j++) m.put(i, j, op == 0 ? a.get(i, j) + b.get(i, j) : a.get(i, j) - b.get(i, j));
    }
    return MatExpr(m);
}
static inline MatExpr cvshim_scale(const Mat &a, double s)
{
    Mat m(a.rows, a.cols, a.type());
    int n = a.cols * a.channels();
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < n; j++) m.put(i, j, a.get(i, j) * s);
    return MatExpr(m);
}
inline MatExpr operator+(const Mat &a, const Mat &b) { return cvshim_binop(a, b, 0); }
inline MatExpr operator-(const Mat &a, const Mat &b) { return cvshim_binop(a, b, 1); }
inline MatExpr operator-(const Mat &a) { return cvshim_scale(a, -1.0); }
inline MatExpr operator*(const Mat &a, double s) { return cvshim_scale(a, s); }
inline MatExpr operator*(double s, const Mat &a) { return cvshim_scale(a, s); }
inline MatExpr operator/(const Mat &a, double s)
{
    Mat m(a.rows, a.cols, a.type());
    int n = a.cols * a.channels();
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < n; j++) m.put(i, j, a.get(i, j) / s);
    return MatExpr(m);
}
inline MatExpr operator*(const Mat &a, const Mat &b)
{
    assert(a.cols == b.rows && a.type() == b.type() && a.channels() == 1);
    Mat m(a.rows, b.cols, a.type());
    if (a.depth() == CV_32F && a.cols >= 2 && a.cols <= 4) {
        // OpenCV's gemm has a dedicated path for inner dimensions 2..4 that evaluates
        // a0*b0 + a1*b1 (+ a2*b2 (+ a3*b3)) in the element type, left to right (matmul.dispatch.cpp,
        // "if( flags == 0 && 2 <= len && len <= 4 ...") - the 3x3 * 3x1 products of the projection code
        // (src/ORBmatcher.cc:1609) take it.  Restated from memory: parity unpinned at the OpenCV level.
        for (int i = 0; i < a.rows; i++)
            for (int j = 0; j < b.cols; j++) {
                float s = a.at<float>(i, 0) * b.at<float>(0, j);
                for (int k = 1; k < a.cols; k++) s = s + a.at<float>(i, k) * b.at<float>(k, j);
                m.at<float>(i, j) = s;
            }
        return MatExpr(m);
    }
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += a.get(i, k) * b.get(k, j);
            m.put(i, j, s);
        }
    return MatExpr(m);
}

template <typename T> struct cvshim_depth_of;
template <> struct cvshim_depth_of<uchar> { enum { value = CV_8U }; };
template <> struct cvshim_depth_of<int> { enum { value = CV_32S }; };
template <> struct cvshim_depth_of<float> { enum { value = CV_32F }; };
template <> struct cvshim_depth_of<double> { enum { value = CV_64F }; };

template <typename T> class MatCommaInitializer_;
template <typename T> class Mat_ : public Mat {
public:
    Mat_() {}
    Mat_(int r, int c) : Mat(r, c, cvshim_depth_of<T>::value) {}
    Mat_(const Mat &m) : Mat(m) { assert(m.empty() || m.depth() == (int)cvshim_depth_of<T>::value); }
    T &operator()(int r, int c) { return this->template at<T>(r, c); }
    const T &operator()(int r, int c) const { return this->template at<T>(r, c); }
};
template <typename T> class MatCommaInitializer_ {
public:
    MatCommaInitializer_(const Mat_<T> &m) : m_(m), i_(0) {}
    template <typename U> MatCommaInitializer_ &operator,(U v)
    {
        m_.template at<T>(i_ / m_.cols, i_ % m_.cols) = (T)v;
        ++i_;
        return *this;
    }
    operator Mat_<T>() const { return m_; }
    operator Mat() const { return m_; }
    Mat_<T> m_;
    int i_;
};
template <typename T, typename U> inline MatCommaInitializer_<T> operator<<(const Mat_<T> &m, U v)
{
    MatCommaInitializer_<T> ci(m);
    return (ci, v);
}

inline double norm(const Mat &a, int normType = NORM_L2)
{
    double s = 0;
    int n = a.cols * a.channels();
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < n; j++) {
            double v = a.get(i, j);
            if (normType == NORM_L2) s += v * v;
            else if (normType == NORM_L1) s += std::fabs(v);
            else s = std::max(s, std::fabs(v));
        }
    return normType == NORM_L2 ? std::sqrt(s) : s;
}
inline double norm(const Mat &a, const Mat &b, int normType = NORM_L2)
{
    assert(a.rows == b.rows && a.cols == b.cols && a.type() == b.type());
    double s = 0;
    int n = a.cols * a.channels();
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < n; j++) {
            double v = a.get(i, j) - b.get(i, j);
            if (normType == NORM_L2) s += v * v;
            else if (normType == NORM_L1) s += std::fabs(v);
            else s = std::max(s, std::fabs(v));
        }
    return normType == NORM_L2 ? std::sqrt(s) : s;
}

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
    assert(img.type() == CV_8UC1);
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
    assert(src.type() == CV_8UC1);
    _dst.create(dsize, CV_8UC1);
    Mat &dst = _dst.getMatRef();
    assert(src.data != dst.data);
    op_resize_linear_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step);
}

inline void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType)
{
    Mat src = _src.getMat();
    assert(src.type() == CV_8UC1);
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
    assert(src.type() == CV_8UC1);
    _dst.create(src.rows, src.cols, CV_8UC1);
    Mat &dst = _dst.getMatRef();
    op_gauss7_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.step, NULL);
}

// Only reached with non-zero distortion coefficients (src/Frame.cc:905-912 returns early
// otherwise); the synthetic cameras of the tests are distortion free.
// cv::undistortPoints for the one form the reference uses (src/Frame.cc:925-931, 972): N x 1
// CV_32FC2 points, 3x3 K, 4/5/8 distortion coefficients, empty R, P = 3x3; dst may be src.
inline void undistortPoints(InputArray _src, OutputArray _dst, InputArray _K, InputArray _dist, InputArray _R = _InputArray(),
                            InputArray _P = _InputArray())
{
    Mat src = _src.getMat(), Km = _K.getMat(), dist = _dist.getMat(), P = _P.getMat();
    assert(src.type() == CV_32FC2 && src.isContinuous() && (src.cols == 1 || src.rows == 1));
    assert(_R.empty() && Km.rows == 3 && Km.cols == 3);
    double K[9], RR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K[3 * i + j] = Km.get(i, j);
    int iters = 0;
    if (!dist.empty()) {
        const int nd = dist.rows * dist.cols;
        assert((dist.rows == 1 || dist.cols == 1) && (nd == 4 || nd == 5 || nd == 8));
        for (int i = 0; i < nd; i++) k[i] = dist.rows == 1 ? dist.get(0, i) : dist.get(i, 0);
        iters = 5;
    }
    if (!P.empty()) {
        // cvMatMul(PP, RR, RR) with RR = identity: every entry is PP[i][j]*1 plus exact zeros
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RR[3 * i + j] = P.get(i, j);
    }
    const int n = src.rows * src.cols;
    std::vector<float> out((size_t)2 * n);
    op_undistort_points((const float *)src.data, out.data(), n, K, k, RR, iters);
    Mat &dst = _dst.getMatRef();
    if (!(dst.data && dst.type() == CV_32FC2 && dst.rows == src.rows && dst.cols == src.cols)) dst.create(src.rows, src.cols, CV_32FC2);
    memcpy(dst.data, out.data(), out.size() * sizeof(float));
}

struct KeyPointsFilter {
    // only referenced from the dead ComputeKeyPointsOld (src/ORBextractor.cc:1203-1514)
    static void retainBest(std::vector<KeyPoint> &, int) { abort(); }
};

// cv::FileStorage: the YAML vocabulary / settings I/O is never executed by the checkers
// (vocabularies are built in memory); the types exist so that the virtual save()/load() of
// DBoW2::TemplatedVocabulary (TemplatedVocabulary.h:1476-1640) compile.
class FileNode {
public:
    FileNode operator[](const std::string &) const { abort(); }
    FileNode operator[](const char *) const { abort(); }
    FileNode operator[](int) const { abort(); }
    size_t size() const { abort(); }
    bool empty() const { return true; }
    operator int() const { abort(); }
    operator float() const { abort(); }
    operator double() const { abort(); }
    operator std::string() const { abort(); }
};
template <typename T> inline void operator>>(const FileNode &, T &) { abort(); }
class FileStorage {
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string &, int) {}
    bool isOpened() const { return false; }
    void release() {}
    FileNode operator[](const std::string &) const { abort(); }
    FileNode operator[](const char *) const { abort(); }
};
template <typename T> inline FileStorage &operator<<(FileStorage &fs, const T &) { abort(); return fs; }

#ifdef CVSHIM_CALLER_DECLS
// compile-only (see class Mat): colour conversion of the input images (src/Tracking.cc:256-383), cv::SVD of the essential-matrix
// check (src/LocalMapping.cc:481)
enum { CV_RGB2GRAY_ = 7 };
void cvtColor(InputArray, OutputArray, int);
struct SVD {
    enum { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 };
    static void compute(InputArray, OutputArray, OutputArray, OutputArray, int flags = 0);
};
#endif

}  // namespace cv

#ifdef CVSHIM_CALLER_DECLS
#define CV_BGR2GRAY 6
#define CV_RGB2GRAY 7
#define CV_BGRA2GRAY 10
#define CV_RGBA2GRAY 11
struct CvMat;      // include/PnPsolver.h names the C API type in member declarations only
#endif

#endif
