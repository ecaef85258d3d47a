// oracle/eigenshim/eigenshim.hpp -- TEST INFRASTRUCTURE ONLY.
//
// A stand-in for the slice of the Eigen 3 API that the reference's vendored g2o
// (/root/reference/Thirdparty/g2o) and src/Optimizer.cc, src/Converter.cc use, so that those files
// compile UNMODIFIED in a container that has no Eigen (same idea as oracle/cvshim for OpenCV).
// oracle/Makefile builds them into oracle/_ref/liboptref.so, the checker for the optimizer rows.
//
// Design: every dense object (Matrix, Map, and the views returned by block / transpose / col /
// segment / diagonal ...) is "pointer + rows + cols + row stride + column stride" over column-major
// storage; MatrixBase<Derived> implements the whole API on top of those five accessors.  There are
// no expression templates: every arithmetic operator evaluates eagerly into a plain Matrix, so
// aliasing is never an issue.  Only the API is Eigen's; the arithmetic is straightforward loops
// (sums run in index order, no vectorisation, no FMA with -ffp-contract=off).  Results therefore
// agree with a stock Eigen build up to floating-point rounding of the dense kernels, which is what
// "g2o semantics" means for the 1e-5 parity bar.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <algorithm>
#include <iostream>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 2
#define EIGEN_MINOR_VERSION 10
#define EIGEN_VERSION_AT_LEAST(x, y, z) (EIGEN_WORLD_VERSION > x || (EIGEN_WORLD_VERSION >= x && (EIGEN_MAJOR_VERSION > y || (EIGEN_MAJOR_VERSION >= y && EIGEN_MINOR_VERSION >= z))))

namespace Eigen {

typedef std::ptrdiff_t Index;
enum { Dynamic = -1 };
enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum { Unaligned = 0, Aligned = 1 };
enum { AlignedBit = 0x80 };
enum { Lower = 1, Upper = 2 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
enum { ComputeEigenvectors = 0x80, EigenvaluesOnly = 0x40 };
enum TransformTraits { Isometry = 1, Affine = 2, AffineCompact = 0x10 | Affine, Projective = 0x20 };

inline void initParallel() {}

template <class T> struct aligned_allocator : public std::allocator<T> {
    template <class U> struct rebind { typedef aligned_allocator<U> other; };
    aligned_allocator() {}
    aligned_allocator(const aligned_allocator &o) : std::allocator<T>(o) {}
    template <class U> aligned_allocator(const aligned_allocator<U> &) {}
};

template <class S, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix;
template <class S, int R, int C> class View;
template <class MT, int MapOpt = Unaligned, class Stride = void> class Map;
template <class MT> class LLT;
template <class MT> class LDLT;
template <class MT> class PartialPivLU;
template <class MT> class SelfAdjointEigenSolver;

namespace internal {
template <class T> struct traits;
template <class S, int R, int C, int O, int MR, int MC> struct traits<Matrix<S, R, C, O, MR, MC> > {
    typedef S Scalar; enum { Rows = R, Cols = C };
};
template <class S, int R, int C> struct traits<View<S, R, C> > { typedef S Scalar; enum { Rows = R, Cols = C }; };
template <class MT, int O, class St> struct traits<Map<MT, O, St> > {
    typedef typename traits<typename std::remove_const<MT>::type>::Scalar Scalar;
    enum { Rows = traits<typename std::remove_const<MT>::type>::Rows, Cols = traits<typename std::remove_const<MT>::type>::Cols };
};
template <int A, int B> struct pick_dim { enum { value = (A == Dynamic) ? B : A }; };
template <int A, int B> struct mul_dim { enum { value = (A == Dynamic || B == Dynamic) ? Dynamic : A * B }; };
}  // namespace internal

template <class Derived> class ArrayWrapper;

// ---------------------------------------------------------------------------------------------
// MatrixBase: the API, on top of data()/rows()/cols()/rowStride()/colStride() of the derived class
// ---------------------------------------------------------------------------------------------
template <class Derived> class MatrixBase {
public:
    typedef typename internal::traits<Derived>::Scalar Scalar;
    enum {
        RowsAtCompileTime = internal::traits<Derived>::Rows,
        ColsAtCompileTime = internal::traits<Derived>::Cols,
        SizeAtCompileTime = internal::mul_dim<RowsAtCompileTime, ColsAtCompileTime>::value,
        IsVectorAtCompileTime = (RowsAtCompileTime == 1 || ColsAtCompileTime == 1),
        Flags = 0
    };
    typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
    typedef Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> TransposedPlain;
    typedef View<Scalar, RowsAtCompileTime, ColsAtCompileTime> SelfView;

    Derived &derived() { return *static_cast<Derived *>(this); }
    const Derived &derived() const { return *static_cast<const Derived *>(this); }

    Index rows() const { return derived().rows(); }
    Index cols() const { return derived().cols(); }
    Index size() const { return rows() * cols(); }
    Scalar *ptr_() const { return const_cast<Scalar *>(derived().data()); }
    Index rs_() const { return derived().rowStride(); }
    Index cs_() const { return derived().colStride(); }

    Scalar &coeffRef(Index i, Index j) { return ptr_()[i * rs_() + j * cs_()]; }
    const Scalar &coeff(Index i, Index j) const { return ptr_()[i * rs_() + j * cs_()]; }
    Scalar &operator()(Index i, Index j) { assert(i >= 0 && i < rows() && j >= 0 && j < cols()); return coeffRef(i, j); }
    const Scalar &operator()(Index i, Index j) const { assert(i >= 0 && i < rows() && j >= 0 && j < cols()); return coeff(i, j); }
    // vector access (column or row vectors)
    Scalar &operator()(Index i) { return vec_(i); }
    const Scalar &operator()(Index i) const { return const_cast<MatrixBase *>(this)->vec_(i); }
    Scalar &operator[](Index i) { return vec_(i); }
    const Scalar &operator[](Index i) const { return const_cast<MatrixBase *>(this)->vec_(i); }
    Scalar &x() { return vec_(0); } Scalar &y() { return vec_(1); } Scalar &z() { return vec_(2); } Scalar &w() { return vec_(3); }
    const Scalar &x() const { return (*this)[0]; } const Scalar &y() const { return (*this)[1]; }
    const Scalar &z() const { return (*this)[2]; } const Scalar &w() const { return (*this)[3]; }

    // ---- assignment-like -------------------------------------------------------------------
    template <class O> Derived &assign_(const MatrixBase<O> &o) {
        derived().resizeLike_(o.rows(), o.cols());
        assert(rows() == o.rows() && cols() == o.cols());
        if ((const void *)o.ptr_() == (const void *)ptr_() && (o.rs_() != rs_() || o.cs_() != cs_())) { PlainObject t(o); return assign_(t); }
        for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = (Scalar)o.coeff(i, j);
        return derived();
    }
    template <class O> Derived &operator+=(const MatrixBase<O> &o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) += o.coeff(i, j);
        return derived();
    }
    template <class O> Derived &operator-=(const MatrixBase<O> &o) {
        assert(rows() == o.rows() && cols() == o.cols());
        for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) -= o.coeff(i, j);
        return derived();
    }
    Derived &operator*=(Scalar s) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) *= s; return derived(); }
    Derived &operator/=(Scalar s) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) /= s; return derived(); }
    template <class O> Derived &operator*=(const MatrixBase<O> &o) { PlainObject t = (*this) * o; return assign_(t); }
    Derived &noalias() { return derived(); }
    const PlainObject eval() const { return PlainObject(*this); }

    Derived &setZero() { return setConstant(Scalar(0)); }
    Derived &setOnes() { return setConstant(Scalar(1)); }
    Derived &setConstant(Scalar v) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = v; return derived(); }
    void fill(Scalar v) { setConstant(v); }
    Derived &setIdentity() { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0); return derived(); }

    // ---- comma initialiser: row-major fill ---------------------------------------------------
    struct CommaInitializer {
        MatrixBase &m; Index k;
        CommaInitializer(MatrixBase &m_, Scalar v) : m(m_), k(0) { put(v); }
        void put(Scalar v) { Index c = m.cols(); assert(k < m.size()); m.coeffRef(k / c, k % c) = v; ++k; }
        CommaInitializer &operator,(Scalar v) { put(v); return *this; }
    };
    CommaInitializer operator<<(const Scalar &v) { return CommaInitializer(*this, v); }

    // ---- views -----------------------------------------------------------------------------
    View<Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() const {
        return View<Scalar, ColsAtCompileTime, RowsAtCompileTime>(ptr_(), cols(), rows(), cs_(), rs_());
    }
    View<Scalar, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) const {
        assert(i >= 0 && j >= 0 && i + r <= rows() && j + c <= cols());
        return View<Scalar, Dynamic, Dynamic>(ptr_() + i * rs_() + j * cs_(), r, c, rs_(), cs_());
    }
    template <int BR, int BC> View<Scalar, BR, BC> block(Index i, Index j) const {
        assert(i >= 0 && j >= 0 && i + BR <= rows() && j + BC <= cols());
        return View<Scalar, BR, BC>(ptr_() + i * rs_() + j * cs_(), BR, BC, rs_(), cs_());
    }
    template <int BR, int BC> View<Scalar, BR, BC> topLeftCorner() const { return block<BR, BC>(0, 0); }
    View<Scalar, Dynamic, Dynamic> topLeftCorner(Index r, Index c) const { return block(0, 0, r, c); }
    View<Scalar, RowsAtCompileTime, 1> col(Index j) const { return View<Scalar, RowsAtCompileTime, 1>(ptr_() + j * cs_(), rows(), 1, rs_(), cs_()); }
    View<Scalar, 1, ColsAtCompileTime> row(Index i) const { return View<Scalar, 1, ColsAtCompileTime>(ptr_() + i * rs_(), 1, cols(), rs_(), cs_()); }
    // vector segments: the step of a vector is whichever stride walks it
    Index vstep_() const { return cols() == 1 ? rs_() : cs_(); }
    View<Scalar, Dynamic, 1> segment(Index i, Index n) const { assert(i >= 0 && i + n <= size()); return View<Scalar, Dynamic, 1>(ptr_() + i * vstep_(), n, 1, vstep_(), 0); }
    template <int N> View<Scalar, N, 1> segment(Index i) const { assert(i >= 0 && i + N <= size()); return View<Scalar, N, 1>(ptr_() + i * vstep_(), N, 1, vstep_(), 0); }
    template <int N> View<Scalar, N, 1> segment(Index i, Index) const { return segment<N>(i); }
    View<Scalar, Dynamic, 1> head(Index n) const { return segment(0, n); }
    template <int N> View<Scalar, N, 1> head() const { return segment<N>(0); }
    View<Scalar, Dynamic, 1> tail(Index n) const { return segment(size() - n, n); }
    template <int N> View<Scalar, N, 1> tail() const { return segment<N>(size() - N); }
    View<Scalar, Dynamic, 1> diagonal() const { Index n = std::min(rows(), cols()); return View<Scalar, Dynamic, 1>(ptr_(), n, 1, rs_() + cs_(), 0); }
    ArrayWrapper<Derived> array() { return ArrayWrapper<Derived>(derived()); }
    const Derived &matrix() const { return derived(); }

    // ---- reductions -------------------------------------------------------------------------
    Scalar sum() const { Scalar s = 0; for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) s += coeff(i, j); return s; }
    Scalar trace() const { Scalar s = 0; for (Index i = 0; i < std::min(rows(), cols()); ++i) s += coeff(i, i); return s; }
    Scalar squaredNorm() const { Scalar s = 0; for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) s += coeff(i, j) * coeff(i, j); return s; }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    Scalar maxCoeff() const { Scalar m = coeff(0, 0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) m = std::max(m, coeff(i, j)); return m; }
    Scalar minCoeff() const { Scalar m = coeff(0, 0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) m = std::min(m, coeff(i, j)); return m; }
    PlainObject cwiseAbs() const { PlainObject r(*this); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = std::abs(coeff(i, j)); return r; }
    template <class O> PlainObject cwiseProduct(const MatrixBase<O> &o) const { PlainObject r(*this); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) *= o.coeff(i, j); return r; }
    void normalize() { Scalar n = norm(); if (n > Scalar(0)) (*this) /= n; }
    PlainObject normalized() const { PlainObject r(*this); r.normalize(); return r; }
    template <class O> Scalar dot(const MatrixBase<O> &o) const {
        assert(size() == o.size()); Scalar s = 0;
        for (Index k = 0; k < size(); ++k) s += (*this)[k] * o[k];
        return s;
    }
    template <class O> Matrix<Scalar, 3, 1> cross(const MatrixBase<O> &o) const {
        const MatrixBase &a = *this; Matrix<Scalar, 3, 1> r;
        r[0] = a[1] * o[2] - a[2] * o[1]; r[1] = a[2] * o[0] - a[0] * o[2]; r[2] = a[0] * o[1] - a[1] * o[0];
        return r;
    }
    bool allFinite() const { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (!std::isfinite(coeff(i, j))) return false; return true; }
    template <class O> bool isApprox(const MatrixBase<O> &o, Scalar prec = Scalar(1e-12)) const {
        PlainObject d(*this); d -= o; return d.squaredNorm() <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
    }

    // ---- dense decompositions -----------------------------------------------------------------
    Scalar determinant() const;
    PlainObject inverse() const;
    LLT<PlainObject> llt() const;
    LDLT<PlainObject> ldlt() const;
    PartialPivLU<PlainObject> lu() const;
    PartialPivLU<PlainObject> partialPivLu() const;

    // ---- statics ------------------------------------------------------------------------------
    static PlainObject Zero() { PlainObject m; m.setZero(); return m; }
    static PlainObject Zero(Index r, Index c) { PlainObject m(r, c); m.setZero(); return m; }
    static PlainObject Zero(Index n) { PlainObject m(n); m.setZero(); return m; }
    static PlainObject Ones() { PlainObject m; m.setOnes(); return m; }
    static PlainObject Constant(Scalar v) { PlainObject m; m.setConstant(v); return m; }
    static PlainObject Identity() { PlainObject m; m.setIdentity(); return m; }
    static PlainObject Identity(Index r, Index c) { PlainObject m(r, c); m.setIdentity(); return m; }

private:
    Scalar &vec_(Index i) { assert(i >= 0 && i < size()); return ptr_()[i * vstep_()]; }
};

template <class Derived> class ArrayWrapper {
    Derived &m;
public:
    typedef typename internal::traits<Derived>::Scalar Scalar;
    explicit ArrayWrapper(Derived &m_) : m(m_) {}
    ArrayWrapper &operator+=(Scalar s) { for (Index j = 0; j < m.cols(); ++j) for (Index i = 0; i < m.rows(); ++i) m.coeffRef(i, j) += s; return *this; }
    ArrayWrapper &operator-=(Scalar s) { return (*this) += -s; }
    ArrayWrapper &operator*=(Scalar s) { m *= s; return *this; }
};

// ---------------------------------------------------------------------------------------------
// storage
// ---------------------------------------------------------------------------------------------
namespace internal {
template <class S, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct Storage;
template <class S, int R, int C> struct Storage<S, R, C, false> {
    S a[R * C > 0 ? R * C : 1];
    Storage() {}
    Index rows() const { return R; } Index cols() const { return C; }
    S *data() { return a; } const S *data() const { return a; }
    void resize(Index r, Index c) { assert(r == R && c == C); (void)r; (void)c; }
};
template <class S, int R, int C> struct Storage<S, R, C, true> {
    std::vector<S> v; Index r_, c_;
    Storage() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
    Index rows() const { return r_; } Index cols() const { return c_; }
    S *data() { return v.empty() ? 0 : &v[0]; } const S *data() const { return v.empty() ? 0 : &v[0]; }
    void resize(Index r, Index c) {
        assert((R == Dynamic || r == R) && (C == Dynamic || c == C));
        if (r != r_ || c != c_ || (Index)v.size() != r * c) { v.assign((size_t)(r * c), S(0)); r_ = r; c_ = c; }
    }
};
}  // namespace internal

template <class S, int R, int C, int Opt, int MR, int MC>
class Matrix : public MatrixBase<Matrix<S, R, C, Opt, MR, MC> > {
    internal::Storage<S, R, C> st;
    typedef MatrixBase<Matrix> Base;
public:
    typedef S Scalar;
    typedef Map<Matrix, Unaligned> MapType;
    typedef Map<const Matrix, Unaligned> ConstMapType;
    typedef Map<Matrix, Aligned> AlignedMapType;
    typedef Map<const Matrix, Aligned> ConstAlignedMapType;
    enum { IsDyn = (R == Dynamic || C == Dynamic), IsVec = (R == 1 || C == 1), FixedSize = IsDyn ? 0 : R * C };

    Matrix() {}
    Matrix(const Matrix &o) : Base(), st(o.st) {}
    template <class O> Matrix(const MatrixBase<O> &o) { this->assign_(o); }
    // one integer: a size (dynamic vectors) / nothing to do (fixed sizes)
    template <class T> explicit Matrix(const T &n, typename std::enable_if<std::is_integral<T>::value, void *>::type = 0) {
        if (IsDyn) { if (R == 1 && C == Dynamic) st.resize(1, (Index)n); else st.resize(R == Dynamic ? (Index)n : R, C == Dynamic ? 1 : C); }
    }
    // two values: sizes, or the coefficients of a fixed 2-vector
    template <class A, class B> Matrix(const A &a, const B &b, typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value, void *>::type = 0) {
        if (FixedSize == 2) { st.data()[0] = (S)a; st.data()[1 % (FixedSize ? FixedSize : 1)] = (S)b; }
        else st.resize((Index)a, (Index)b);
    }
    Matrix(const S &a, const S &b, const S &c) { assert(this->size() == 3); S *d = st.data(); d[0] = a; d[1] = b; d[2] = c; }
    Matrix(const S &a, const S &b, const S &c, const S &e) { assert(this->size() == 4); S *d = st.data(); d[0] = a; d[1] = b; d[2] = c; d[3] = e; }
    explicit Matrix(const S *p) { std::memcpy(st.data(), p, sizeof(S) * (size_t)(st.rows() * st.cols())); }

    Matrix &operator=(const Matrix &o) { st = o.st; return *this; }
    template <class O> Matrix &operator=(const MatrixBase<O> &o) { return this->assign_(o); }

    Index rows() const { return st.rows(); }
    Index cols() const { return st.cols(); }
    S *data() { return st.data(); }
    const S *data() const { return st.data(); }
    Index rowStride() const { return 1; }
    Index colStride() const { return st.rows(); }
    Index innerStride() const { return 1; }
    Index outerStride() const { return st.rows(); }
    void resize(Index r, Index c) { st.resize(r, c); }
    void resize(Index n) { if (C == 1) st.resize(n, 1); else if (R == 1) st.resize(1, n); else st.resize(n, 1); }
    void conservativeResize(Index n) { Matrix t(*this); resize(n); for (Index i = 0; i < std::min<Index>(n, t.size()); ++i) (*this)[i] = t[i]; }
    void resizeLike_(Index r, Index c) { if (IsDyn) st.resize(r, c); }
};

// strided view: what block()/transpose()/col()/segment()/diagonal() return
template <class S, int R, int C> class View : public MatrixBase<View<S, R, C> > {
    S *p; Index r_, c_, rs, cs;
public:
    typedef S Scalar;
    View(S *p_, Index r, Index c, Index rs_, Index cs_) : p(p_), r_(r), c_(c), rs(rs_), cs(cs_) {}
    View(const View &o) : MatrixBase<View>(), p(o.p), r_(o.r_), c_(o.c_), rs(o.rs), cs(o.cs) {}
    View &operator=(const View &o) { return this->assign_(o); }
    template <class O> View &operator=(const MatrixBase<O> &o) { return this->assign_(o); }
    Index rows() const { return r_; } Index cols() const { return c_; }
    S *data() const { return p; }
    Index rowStride() const { return rs; } Index colStride() const { return cs; }
    void resizeLike_(Index, Index) {}
};

// Map: a view over caller memory with the plain type's shape
template <class MT, int MapOpt, class Stride>
class Map : public MatrixBase<Map<MT, MapOpt, Stride> > {
    typedef typename std::remove_const<MT>::type Plain;
public:
    typedef typename internal::traits<Plain>::Scalar Scalar;
    enum { R = internal::traits<Plain>::Rows, C = internal::traits<Plain>::Cols };
private:
    Scalar *p; Index r_, c_;
public:
    Map(const Scalar *d) : p(const_cast<Scalar *>(d)), r_(R), c_(C) { assert(R != Dynamic && C != Dynamic); }
    Map(const Scalar *d, Index n) : p(const_cast<Scalar *>(d)), r_(C == 1 ? n : (R == Dynamic ? n : R)), c_(C == 1 ? 1 : (R == 1 ? n : (C == Dynamic ? 1 : C))) {}
    Map(const Scalar *d, Index r, Index c) : p(const_cast<Scalar *>(d)), r_(r), c_(c) {}
    Map(const Map &o) : MatrixBase<Map>(), p(o.p), r_(o.r_), c_(o.c_) {}
    Map &operator=(const Map &o) { return this->assign_(o); }
    template <class O> Map &operator=(const MatrixBase<O> &o) { return this->assign_(o); }
    Index rows() const { return r_; } Index cols() const { return c_; }
    Scalar *data() const { return p; }
    Index rowStride() const { return 1; } Index colStride() const { return r_; }
    void resizeLike_(Index, Index) {}
};

// ---------------------------------------------------------------------------------------------
// arithmetic (eager)
// ---------------------------------------------------------------------------------------------
#define ES_PLAIN2(A, B) Matrix<typename MatrixBase<A>::Scalar, internal::pick_dim<MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::RowsAtCompileTime>::value, internal::pick_dim<MatrixBase<A>::ColsAtCompileTime, MatrixBase<B>::ColsAtCompileTime>::value>

template <class A, class B> ES_PLAIN2(A, B) operator+(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    ES_PLAIN2(A, B) r(a); r += b; return r;
}
template <class A, class B> ES_PLAIN2(A, B) operator-(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    ES_PLAIN2(A, B) r(a); r -= b; return r;
}
template <class A> typename MatrixBase<A>::PlainObject operator-(const MatrixBase<A> &a) {
    typename MatrixBase<A>::PlainObject r(a);
    for (Index j = 0; j < r.cols(); ++j) for (Index i = 0; i < r.rows(); ++i) r.coeffRef(i, j) = -r.coeffRef(i, j);
    return r;
}
template <class A, class T> typename std::enable_if<std::is_arithmetic<T>::value, typename MatrixBase<A>::PlainObject>::type
operator*(const MatrixBase<A> &a, const T &s) { typename MatrixBase<A>::PlainObject r(a); r *= (typename MatrixBase<A>::Scalar)s; return r; }
template <class A, class T> typename std::enable_if<std::is_arithmetic<T>::value, typename MatrixBase<A>::PlainObject>::type
operator*(const T &s, const MatrixBase<A> &a) {
    typename MatrixBase<A>::PlainObject r(a); typedef typename MatrixBase<A>::Scalar S;
    for (Index j = 0; j < r.cols(); ++j) for (Index i = 0; i < r.rows(); ++i) r.coeffRef(i, j) = (S)s * r.coeffRef(i, j);
    return r;
}
template <class A, class T> typename std::enable_if<std::is_arithmetic<T>::value, typename MatrixBase<A>::PlainObject>::type
operator/(const MatrixBase<A> &a, const T &s) { typename MatrixBase<A>::PlainObject r(a); r /= (typename MatrixBase<A>::Scalar)s; return r; }

template <class A, class B>
Matrix<typename MatrixBase<A>::Scalar, MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::ColsAtCompileTime>
operator*(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    typedef typename MatrixBase<A>::Scalar S;
    assert(a.cols() == b.rows());
    Matrix<S, MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::ColsAtCompileTime> r;
    r.resizeLike_(a.rows(), b.cols());
    const Index n = a.rows(), m = b.cols(), kk = a.cols();
    for (Index j = 0; j < m; ++j)
        for (Index i = 0; i < n; ++i) {
            S s = S(0);
            for (Index k = 0; k < kk; ++k) s += a.coeff(i, k) * b.coeff(k, j);
            r.coeffRef(i, j) = s;
        }
    return r;
}

template <class A, class B> bool operator==(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
    for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) if (!(a.coeff(i, j) == b.coeff(i, j))) return false;
    return true;
}
template <class A, class B> bool operator!=(const MatrixBase<A> &a, const MatrixBase<B> &b) { return !(a == b); }

template <class A> std::ostream &operator<<(std::ostream &os, const MatrixBase<A> &m) {
    for (Index i = 0; i < m.rows(); ++i) {
        for (Index j = 0; j < m.cols(); ++j) os << (j ? " " : "") << m.coeff(i, j);
        if (i + 1 < m.rows()) os << "\n";
    }
    return os;
}

// ---------------------------------------------------------------------------------------------
// dense decompositions (plain textbook algorithms on a dynamic copy)
// ---------------------------------------------------------------------------------------------
template <class MT> class PartialPivLU {
    typedef typename internal::traits<MT>::Scalar S;
    Matrix<S, Dynamic, Dynamic> lu_; std::vector<Index> perm; int sign; bool singular;
public:
    PartialPivLU() : sign(1), singular(false) {}
    template <class O> explicit PartialPivLU(const MatrixBase<O> &a) { compute(a); }
    template <class O> PartialPivLU &compute(const MatrixBase<O> &a) {
        lu_ = a; const Index n = lu_.rows(); assert(n == lu_.cols());
        perm.resize((size_t)n); for (Index i = 0; i < n; ++i) perm[(size_t)i] = i; sign = 1; singular = false;
        for (Index k = 0; k < n; ++k) {
            Index piv = k; S best = std::abs(lu_(k, k));
            for (Index i = k + 1; i < n; ++i) if (std::abs(lu_(i, k)) > best) { best = std::abs(lu_(i, k)); piv = i; }
            if (best == S(0)) { singular = true; continue; }
            if (piv != k) { for (Index j = 0; j < n; ++j) std::swap(lu_(k, j), lu_(piv, j)); std::swap(perm[(size_t)k], perm[(size_t)piv]); sign = -sign; }
            for (Index i = k + 1; i < n; ++i) {
                lu_(i, k) /= lu_(k, k);
                const S f = lu_(i, k);
                for (Index j = k + 1; j < n; ++j) lu_(i, j) -= f * lu_(k, j);
            }
        }
        return *this;
    }
    S determinant() const { S d = S(sign); for (Index i = 0; i < lu_.rows(); ++i) d *= lu_(i, i); return d; }
    template <class B> typename MatrixBase<B>::PlainObject solve(const MatrixBase<B> &b) const {
        typename MatrixBase<B>::PlainObject x(b); const Index n = lu_.rows();
        for (Index c = 0; c < b.cols(); ++c) {
            for (Index i = 0; i < n; ++i) x.coeffRef(i, c) = b.coeff(perm[(size_t)i], c);
            for (Index i = 0; i < n; ++i) { S s = x.coeffRef(i, c); for (Index k = 0; k < i; ++k) s -= lu_(i, k) * x.coeffRef(k, c); x.coeffRef(i, c) = s; }
            for (Index i = n - 1; i >= 0; --i) { S s = x.coeffRef(i, c); for (Index k = i + 1; k < n; ++k) s -= lu_(i, k) * x.coeffRef(k, c); x.coeffRef(i, c) = s / lu_(i, i); }
        }
        return x;
    }
    MT inverse() const { MT id; id.resizeLike_(lu_.rows(), lu_.rows()); id.setIdentity(); return solve(id); }
};

template <class MT> class LLT {
    typedef typename internal::traits<MT>::Scalar S;
    Matrix<S, Dynamic, Dynamic> l_; ComputationInfo info_;
public:
    LLT() : info_(Success) {}
    template <class O> explicit LLT(const MatrixBase<O> &a) { compute(a); }
    template <class O> LLT &compute(const MatrixBase<O> &a) {
        l_ = a; const Index n = l_.rows(); info_ = Success;
        for (Index j = 0; j < n; ++j) {
            S d = l_(j, j); for (Index k = 0; k < j; ++k) d -= l_(j, k) * l_(j, k);
            if (!(d > S(0))) { info_ = NumericalIssue; return *this; }
            d = std::sqrt(d); l_(j, j) = d;
            for (Index i = j + 1; i < n; ++i) { S s = l_(i, j); for (Index k = 0; k < j; ++k) s -= l_(i, k) * l_(j, k); l_(i, j) = s / d; }
        }
        return *this;
    }
    ComputationInfo info() const { return info_; }
    template <class B> typename MatrixBase<B>::PlainObject solve(const MatrixBase<B> &b) const {
        typename MatrixBase<B>::PlainObject x(b); const Index n = l_.rows();
        for (Index c = 0; c < b.cols(); ++c) {
            for (Index i = 0; i < n; ++i) { S s = x.coeffRef(i, c); for (Index k = 0; k < i; ++k) s -= l_(i, k) * x.coeffRef(k, c); x.coeffRef(i, c) = s / l_(i, i); }
            for (Index i = n - 1; i >= 0; --i) { S s = x.coeffRef(i, c); for (Index k = i + 1; k < n; ++k) s -= l_(k, i) * x.coeffRef(k, c); x.coeffRef(i, c) = s / l_(i, i); }
        }
        return x;
    }
    Matrix<S, Dynamic, Dynamic> matrixL() const { Matrix<S, Dynamic, Dynamic> L(l_); for (Index j = 0; j < L.cols(); ++j) for (Index i = 0; i < j; ++i) L(i, j) = 0; return L; }
};

// LDL^T of a symmetric matrix (lower triangle read).  Eigen's dense LDLT pivots on the largest remaining
// diagonal entry; so does this one, which keeps isPositive()/solve() meaningful for semi-definite input.
template <class MT> class LDLT {
    typedef typename internal::traits<MT>::Scalar S;
    Matrix<S, Dynamic, Dynamic> m_; std::vector<Index> tr_; int sign_; ComputationInfo info_;
public:
    LDLT() : sign_(0), info_(Success) {}
    template <class O> explicit LDLT(const MatrixBase<O> &a) { compute(a); }
    template <class O> LDLT &compute(const MatrixBase<O> &a) {
        m_ = a; const Index n = m_.rows(); tr_.assign((size_t)n, 0); sign_ = 0; info_ = Success;
        // work on the full symmetric matrix (both triangles kept consistent)
        for (Index j = 0; j < n; ++j) for (Index i = 0; i < j; ++i) m_(i, j) = m_(j, i);
        bool pos = true, neg = true;
        for (Index k = 0; k < n; ++k) {
            Index piv = k; S best = std::abs(m_(k, k));
            for (Index i = k + 1; i < n; ++i) if (std::abs(m_(i, i)) > best) { best = std::abs(m_(i, i)); piv = i; }
            tr_[(size_t)k] = piv;
            if (piv != k) {
                for (Index j = 0; j < n; ++j) std::swap(m_(k, j), m_(piv, j));
                for (Index i = 0; i < n; ++i) std::swap(m_(i, k), m_(i, piv));
            }
            const S d = m_(k, k);
            if (d > S(0)) neg = false; else if (d < S(0)) pos = false;
            if (best == S(0)) {   // the rest is zero: stop (positive / negative semi-definite so far)
                for (Index i = k + 1; i < n; ++i) tr_[(size_t)i] = i;
                break;
            }
            for (Index i = k + 1; i < n; ++i) m_(i, k) /= d;
            for (Index j = k + 1; j < n; ++j) {
                const S f = m_(j, k) * d;
                for (Index i = j; i < n; ++i) m_(i, j) -= m_(i, k) * f;
            }
            for (Index j = k + 1; j < n; ++j) for (Index i = k + 1; i < j; ++i) m_(i, j) = m_(j, i);
        }
        sign_ = pos ? 1 : (neg ? -1 : 0);
        return *this;
    }
    bool isPositive() const { return sign_ == 1; }
    bool isNegative() const { return sign_ == -1; }
    ComputationInfo info() const { return info_; }
    template <class B> typename MatrixBase<B>::PlainObject solve(const MatrixBase<B> &b) const {
        typename MatrixBase<B>::PlainObject x(b); const Index n = m_.rows();
        for (Index c = 0; c < b.cols(); ++c) {
            for (Index i = 0; i < n; ++i) std::swap(x.coeffRef(i, c), x.coeffRef(tr_[(size_t)i], c));
            for (Index i = 0; i < n; ++i) { S s = x.coeffRef(i, c); for (Index k = 0; k < i; ++k) s -= m_(i, k) * x.coeffRef(k, c); x.coeffRef(i, c) = s; }
            for (Index i = 0; i < n; ++i) { const S d = m_(i, i); x.coeffRef(i, c) = (std::abs(d) > std::numeric_limits<S>::min()) ? x.coeffRef(i, c) / d : S(0); }
            for (Index i = n - 1; i >= 0; --i) { S s = x.coeffRef(i, c); for (Index k = i + 1; k < n; ++k) s -= m_(k, i) * x.coeffRef(k, c); x.coeffRef(i, c) = s; }
            for (Index i = n - 1; i >= 0; --i) std::swap(x.coeffRef(i, c), x.coeffRef(tr_[(size_t)i], c));
        }
        return x;
    }
};

template <class D> typename MatrixBase<D>::Scalar MatrixBase<D>::determinant() const {
    const Index n = rows(); assert(n == cols());
    if (n == 1) return coeff(0, 0);
    if (n == 2) return coeff(0, 0) * coeff(1, 1) - coeff(0, 1) * coeff(1, 0);
    if (n == 3)
        return coeff(0, 0) * (coeff(1, 1) * coeff(2, 2) - coeff(1, 2) * coeff(2, 1)) - coeff(0, 1) * (coeff(1, 0) * coeff(2, 2) - coeff(1, 2) * coeff(2, 0)) +
               coeff(0, 2) * (coeff(1, 0) * coeff(2, 1) - coeff(1, 1) * coeff(2, 0));
    return PartialPivLU<PlainObject>(*this).determinant();
}
template <class D> typename MatrixBase<D>::PlainObject MatrixBase<D>::inverse() const {
    const Index n = rows(); assert(n == cols());
    PlainObject r; r.resizeLike_(n, n);
    if (n == 1) { r.coeffRef(0, 0) = Scalar(1) / coeff(0, 0); return r; }
    if (n == 2) {
        const Scalar invdet = Scalar(1) / determinant();
        r.coeffRef(0, 0) = coeff(1, 1) * invdet; r.coeffRef(1, 0) = -coeff(1, 0) * invdet;
        r.coeffRef(0, 1) = -coeff(0, 1) * invdet; r.coeffRef(1, 1) = coeff(0, 0) * invdet;
        return r;
    }
    if (n == 3) {   // cofactors, as Eigen does for fixed 3x3
        const MatrixBase &m = *this;
        const Scalar c00 = m.coeff(1, 1) * m.coeff(2, 2) - m.coeff(1, 2) * m.coeff(2, 1);
        const Scalar c10 = m.coeff(1, 2) * m.coeff(2, 0) - m.coeff(1, 0) * m.coeff(2, 2);
        const Scalar c20 = m.coeff(1, 0) * m.coeff(2, 1) - m.coeff(1, 1) * m.coeff(2, 0);
        const Scalar invdet = Scalar(1) / (c00 * m.coeff(0, 0) + c10 * m.coeff(0, 1) + c20 * m.coeff(0, 2));
        r.coeffRef(0, 0) = c00 * invdet; r.coeffRef(1, 0) = c10 * invdet; r.coeffRef(2, 0) = c20 * invdet;
        r.coeffRef(0, 1) = (m.coeff(0, 2) * m.coeff(2, 1) - m.coeff(0, 1) * m.coeff(2, 2)) * invdet;
        r.coeffRef(1, 1) = (m.coeff(0, 0) * m.coeff(2, 2) - m.coeff(0, 2) * m.coeff(2, 0)) * invdet;
        r.coeffRef(2, 1) = (m.coeff(0, 1) * m.coeff(2, 0) - m.coeff(0, 0) * m.coeff(2, 1)) * invdet;
        r.coeffRef(0, 2) = (m.coeff(0, 1) * m.coeff(1, 2) - m.coeff(0, 2) * m.coeff(1, 1)) * invdet;
        r.coeffRef(1, 2) = (m.coeff(0, 2) * m.coeff(1, 0) - m.coeff(0, 0) * m.coeff(1, 2)) * invdet;
        r.coeffRef(2, 2) = (m.coeff(0, 0) * m.coeff(1, 1) - m.coeff(0, 1) * m.coeff(1, 0)) * invdet;
        return r;
    }
    return PartialPivLU<PlainObject>(*this).inverse();
}
template <class D> LLT<typename MatrixBase<D>::PlainObject> MatrixBase<D>::llt() const { return LLT<PlainObject>(*this); }
template <class D> LDLT<typename MatrixBase<D>::PlainObject> MatrixBase<D>::ldlt() const { return LDLT<PlainObject>(*this); }
template <class D> PartialPivLU<typename MatrixBase<D>::PlainObject> MatrixBase<D>::lu() const { return PartialPivLU<PlainObject>(*this); }
template <class D> PartialPivLU<typename MatrixBase<D>::PlainObject> MatrixBase<D>::partialPivLu() const { return PartialPivLU<PlainObject>(*this); }

// eigenvalues (ascending) / eigenvectors of a symmetric matrix by cyclic Jacobi rotations
template <class MT> class SelfAdjointEigenSolver {
    typedef typename internal::traits<MT>::Scalar S;
    Matrix<S, Dynamic, 1> w_; Matrix<S, Dynamic, Dynamic> v_; ComputationInfo info_;
public:
    SelfAdjointEigenSolver() : info_(Success) {}
    template <class O> explicit SelfAdjointEigenSolver(const MatrixBase<O> &a, int opt = ComputeEigenvectors) { compute(a, opt); }
    template <class O> SelfAdjointEigenSolver &compute(const MatrixBase<O> &a0, int = ComputeEigenvectors) {
        Matrix<S, Dynamic, Dynamic> a(a0); const Index n = a.rows();
        v_.resize(n, n); v_.setIdentity();
        for (int sweep = 0; sweep < 100; ++sweep) {
            S off = 0; for (Index j = 0; j < n; ++j) for (Index i = 0; i < j; ++i) off += a(i, j) * a(i, j);
            if (off <= std::numeric_limits<S>::min()) break;
            for (Index p = 0; p < n; ++p) for (Index q = p + 1; q < n; ++q) {
                if (a(p, q) == S(0)) continue;
                const S theta = (a(q, q) - a(p, p)) / (S(2) * a(p, q));
                const S t = (theta >= 0 ? S(1) : S(-1)) / (std::abs(theta) + std::sqrt(theta * theta + S(1)));
                const S c = S(1) / std::sqrt(t * t + S(1)), s = t * c;
                for (Index k = 0; k < n; ++k) { const S akp = a(k, p), akq = a(k, q); a(k, p) = c * akp - s * akq; a(k, q) = s * akp + c * akq; }
                for (Index k = 0; k < n; ++k) { const S apk = a(p, k), aqk = a(q, k); a(p, k) = c * apk - s * aqk; a(q, k) = s * apk + c * aqk; }
                for (Index k = 0; k < n; ++k) { const S vkp = v_(k, p), vkq = v_(k, q); v_(k, p) = c * vkp - s * vkq; v_(k, q) = s * vkp + c * vkq; }
            }
        }
        w_.resize(n); for (Index i = 0; i < n; ++i) w_[i] = a(i, i);
        for (Index i = 0; i < n; ++i) {   // selection sort, ascending
            Index m = i; for (Index k = i + 1; k < n; ++k) if (w_[k] < w_[m]) m = k;
            if (m != i) { std::swap(w_[i], w_[m]); for (Index k = 0; k < n; ++k) std::swap(v_(k, i), v_(k, m)); }
        }
        return *this;
    }
    const Matrix<S, Dynamic, 1> &eigenvalues() const { return w_; }
    const Matrix<S, Dynamic, Dynamic> &eigenvectors() const { return v_; }
    ComputationInfo info() const { return info_; }
};

// ---------------------------------------------------------------------------------------------
// typedefs
// ---------------------------------------------------------------------------------------------
#define ES_TYPEDEFS(S, sfx)                                                               \
    typedef Matrix<S, 2, 2> Matrix2##sfx; typedef Matrix<S, 3, 3> Matrix3##sfx;           \
    typedef Matrix<S, 4, 4> Matrix4##sfx; typedef Matrix<S, Dynamic, Dynamic> MatrixX##sfx; \
    typedef Matrix<S, 2, 1> Vector2##sfx; typedef Matrix<S, 3, 1> Vector3##sfx;           \
    typedef Matrix<S, 4, 1> Vector4##sfx; typedef Matrix<S, Dynamic, 1> VectorX##sfx;     \
    typedef Matrix<S, 1, 2> RowVector2##sfx; typedef Matrix<S, 1, 3> RowVector3##sfx;     \
    typedef Matrix<S, 1, 4> RowVector4##sfx; typedef Matrix<S, 1, Dynamic> RowVectorX##sfx;
ES_TYPEDEFS(double, d)
ES_TYPEDEFS(float, f)
ES_TYPEDEFS(int, i)
#undef ES_TYPEDEFS

// ---------------------------------------------------------------------------------------------
// Geometry: Quaternion (coefficients stored x, y, z, w as in Eigen), Transform
// ---------------------------------------------------------------------------------------------
template <class S> class Quaternion {
    Matrix<S, 4, 1> c_;   // x y z w
public:
    typedef S Scalar;
    typedef Matrix<S, 3, 1> Vector3; typedef Matrix<S, 3, 3> Matrix3; typedef Matrix<S, 4, 1> Coefficients;
    Quaternion() {}
    Quaternion(const S &w, const S &x, const S &y, const S &z) { c_[0] = x; c_[1] = y; c_[2] = z; c_[3] = w; }
    Quaternion(const Quaternion &o) : c_(o.c_) {}
    template <class D> explicit Quaternion(const MatrixBase<D> &m) { *this = m; }
    Quaternion &operator=(const Quaternion &o) { c_ = o.c_; return *this; }
    // rotation matrix -> quaternion (the branches of Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>);
    // a 4-vector is taken as coefficients
    template <class D> Quaternion &operator=(const MatrixBase<D> &mat) {
        if (mat.rows() == 4 && mat.cols() == 1) { for (int i = 0; i < 4; ++i) c_[i] = mat[i]; return *this; }
        assert(mat.rows() == 3 && mat.cols() == 3);
        S t = mat.coeff(0, 0) + mat.coeff(1, 1) + mat.coeff(2, 2);
        if (t > S(0)) {
            t = std::sqrt(t + S(1.0));
            w() = S(0.5) * t;
            t = S(0.5) / t;
            x() = (mat.coeff(2, 1) - mat.coeff(1, 2)) * t;
            y() = (mat.coeff(0, 2) - mat.coeff(2, 0)) * t;
            z() = (mat.coeff(1, 0) - mat.coeff(0, 1)) * t;
        } else {
            Index i = 0;
            if (mat.coeff(1, 1) > mat.coeff(0, 0)) i = 1;
            if (mat.coeff(2, 2) > mat.coeff(i, i)) i = 2;
            Index j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(mat.coeff(i, i) - mat.coeff(j, j) - mat.coeff(k, k) + S(1.0));
            c_[i] = S(0.5) * t;
            t = S(0.5) / t;
            w() = (mat.coeff(k, j) - mat.coeff(j, k)) * t;
            c_[j] = (mat.coeff(j, i) + mat.coeff(i, j)) * t;
            c_[k] = (mat.coeff(k, i) + mat.coeff(i, k)) * t;
        }
        return *this;
    }
    S &x() { return c_[0]; } S &y() { return c_[1]; } S &z() { return c_[2]; } S &w() { return c_[3]; }
    const S &x() const { return c_[0]; } const S &y() const { return c_[1]; } const S &z() const { return c_[2]; } const S &w() const { return c_[3]; }
    Coefficients &coeffs() { return c_; }
    const Coefficients &coeffs() const { return c_; }
    View<S, 3, 1> vec() const { return c_.template head<3>(); }
    Quaternion &setIdentity() { c_[0] = c_[1] = c_[2] = S(0); c_[3] = S(1); return *this; }
    static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
    S squaredNorm() const { return c_.squaredNorm(); }
    S norm() const { return c_.norm(); }
    void normalize() { c_.normalize(); }
    Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
    Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
    Quaternion inverse() const {
        S n2 = squaredNorm();
        if (n2 > S(0)) { Quaternion q = conjugate(); q.c_ /= n2; return q; }
        Quaternion q; q.c_.setZero(); return q;
    }
    S dot(const Quaternion &o) const { return c_.dot(o.c_); }
    Quaternion operator*(const Quaternion &b) const {
        const Quaternion &a = *this;
        return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                          a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                          a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                          a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    Quaternion &operator*=(const Quaternion &b) { *this = (*this) * b; return *this; }
    // rotate a vector: v + w*uv + vec x uv with uv = 2 vec x v (Eigen's _transformVector)
    template <class D> Vector3 operator*(const MatrixBase<D> &v) const {
        Vector3 q; q[0] = x(); q[1] = y(); q[2] = z();
        Vector3 uv = q.cross(v); uv += uv;
        Vector3 r(v); r += w() * uv; r += q.cross(uv);
        return r;
    }
    Vector3 _transformVector(const Vector3 &v) const { return (*this) * v; }
    Matrix3 toRotationMatrix() const {
        Matrix3 res;
        const S tx = S(2) * x(), ty = S(2) * y(), tz = S(2) * z();
        const S twx = tx * w(), twy = ty * w(), twz = tz * w();
        const S txx = tx * x(), txy = ty * x(), txz = tz * x();
        const S tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        res.coeffRef(0, 0) = S(1) - (tyy + tzz); res.coeffRef(0, 1) = txy - twz; res.coeffRef(0, 2) = txz + twy;
        res.coeffRef(1, 0) = txy + twz; res.coeffRef(1, 1) = S(1) - (txx + tzz); res.coeffRef(1, 2) = tyz - twx;
        res.coeffRef(2, 0) = txz - twy; res.coeffRef(2, 1) = tyz + twx; res.coeffRef(2, 2) = S(1) - (txx + tyy);
        return res;
    }
    Matrix3 matrix() const { return toRotationMatrix(); }
};
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;

// Transform: a (Dim+1)x(Dim+1) homogeneous matrix
template <class S, int Dim, int Mode, int Opt = 0> class Transform {
    Matrix<S, Dim + 1, Dim + 1> m_;
public:
    typedef Matrix<S, Dim + 1, Dim + 1> MatrixType;
    typedef Matrix<S, Dim, 1> VectorType;
    Transform() { m_.setIdentity(); }
    Transform(const Transform &o) : m_(o.m_) {}
    Transform(const Quaternion<S> &q) { m_.setIdentity(); m_.template block<Dim, Dim>(0, 0) = q.toRotationMatrix(); }
    template <class D> explicit Transform(const MatrixBase<D> &m) {
        m_.setIdentity();
        if (m.rows() == Dim + 1) m_ = m; else m_.template block<Dim, Dim>(0, 0) = m;
    }
    Transform &operator=(const Transform &o) { m_ = o.m_; return *this; }
    Transform &operator=(const Quaternion<S> &q) { m_.setIdentity(); m_.template block<Dim, Dim>(0, 0) = q.toRotationMatrix(); return *this; }
    template <class D> Transform &operator=(const MatrixBase<D> &m) { Transform t(m); m_ = t.m_; return *this; }
    static Transform Identity() { return Transform(); }
    void setIdentity() { m_.setIdentity(); }
    MatrixType &matrix() { return m_; }
    const MatrixType &matrix() const { return m_; }
    View<S, Dim, Dim> linear() const { return m_.template block<Dim, Dim>(0, 0); }
    View<S, Dim, Dim> rotation() const { return linear(); }
    View<S, Dim, 1> translation() const { return View<S, Dim, 1>(const_cast<S *>(m_.data()) + Dim * (Dim + 1), Dim, 1, 1, 0); }
    S &operator()(Index i, Index j) { return m_(i, j); }
    const S &operator()(Index i, Index j) const { return m_(i, j); }
    Transform operator*(const Transform &o) const { Transform r; r.m_ = m_ * o.m_; return r; }
    VectorType operator*(const VectorType &v) const { VectorType r = linear() * v; r += translation(); return r; }
    Transform inverse() const {
        Transform r;
        if (Mode == Isometry) { r.m_.template block<Dim, Dim>(0, 0) = linear().transpose(); }
        else { r.m_.template block<Dim, Dim>(0, 0) = linear().inverse(); }
        VectorType t = r.linear() * translation(); t *= S(-1);
        r.translation() = t;
        return r;
    }
};
typedef Transform<double, 3, Isometry> Isometry3d;
typedef Transform<double, 2, Isometry> Isometry2d;
typedef Transform<double, 3, Affine> Affine3d;
typedef Transform<double, 2, Affine> Affine2d;

template <class S> class AngleAxis {
    Matrix<S, 3, 1> axis_; S angle_;
public:
    AngleAxis() : angle_(0) {}
    template <class D> AngleAxis(const S &a, const MatrixBase<D> &ax) : axis_(ax), angle_(a) {}
    S angle() const { return angle_; } const Matrix<S, 3, 1> &axis() const { return axis_; }
    Matrix<S, 3, 3> toRotationMatrix() const {
        Matrix<S, 3, 3> res; const S s = std::sin(angle_), c = std::cos(angle_);
        Matrix<S, 3, 1> cc = axis_ * (S(1) - c), ss = axis_ * s;
        S tmp = cc[0] * axis_[1]; res(0, 1) = tmp - ss[2]; res(1, 0) = tmp + ss[2];
        tmp = cc[0] * axis_[2]; res(0, 2) = tmp + ss[1]; res(2, 0) = tmp - ss[1];
        tmp = cc[1] * axis_[2]; res(1, 2) = tmp - ss[0]; res(2, 1) = tmp + ss[0];
        res(0, 0) = cc[0] * axis_[0] + c; res(1, 1) = cc[1] * axis_[1] + c; res(2, 2) = cc[2] * axis_[2] + c;
        return res;
    }
};
typedef AngleAxis<double> AngleAxisd;

// ---------------------------------------------------------------------------------------------
// Sparse: column-compressed SparseMatrix, Triplet, PermutationMatrix, SimplicialLDLT
// ---------------------------------------------------------------------------------------------
template <class S, class I = int> class Triplet {
    I r_, c_; S v_;
public:
    Triplet() : r_(0), c_(0), v_(0) {}
    Triplet(const I &r, const I &c, const S &v = S(0)) : r_(r), c_(c), v_(v) {}
    const I &row() const { return r_; } const I &col() const { return c_; } const S &value() const { return v_; }
};

template <int SizeR = Dynamic, int SizeC = SizeR> class PermutationMatrix {
    Matrix<int, Dynamic, 1> idx_;
public:
    PermutationMatrix() {}
    explicit PermutationMatrix(Index n) { resize(n); }
    void resize(Index n) { idx_.resize(n); for (Index i = 0; i < n; ++i) idx_[i] = (int)i; }
    Index size() const { return idx_.size(); } Index rows() const { return size(); } Index cols() const { return size(); }
    Matrix<int, Dynamic, 1> &indices() { return idx_; }
    const Matrix<int, Dynamic, 1> &indices() const { return idx_; }
    void setIdentity(Index n) { resize(n); }
    PermutationMatrix inverse() const { PermutationMatrix p(size()); for (Index i = 0; i < size(); ++i) p.idx_[idx_[i]] = (int)i; return p; }
};

template <class SM, int UpLo> class SparseSelfAdjointView;
template <class SM, int UpLo> struct SparseSymmetricPermutationProduct { const SM &m; const PermutationMatrix<> &p; };

template <class S, int Opt = ColMajor, class I = int> class SparseMatrix {
public:
    typedef S Scalar; typedef I StorageIndex;
    std::vector<I> outer_, inner_; std::vector<S> val_; Index r_, c_;
    SparseMatrix() : outer_(1, 0), r_(0), c_(0) {}
    SparseMatrix(Index r, Index c) : outer_((size_t)c + 1, 0), r_(r), c_(c) {}
    void resize(Index r, Index c) { r_ = r; c_ = c; outer_.assign((size_t)c + 1, 0); inner_.clear(); val_.clear(); }
    Index rows() const { return r_; } Index cols() const { return c_; }
    Index nonZeros() const { return (Index)val_.size(); }
    S *valuePtr() { return val_.empty() ? 0 : &val_[0]; } const S *valuePtr() const { return val_.empty() ? 0 : &val_[0]; }
    I *innerIndexPtr() { return inner_.empty() ? 0 : &inner_[0]; } const I *innerIndexPtr() const { return inner_.empty() ? 0 : &inner_[0]; }
    I *outerIndexPtr() { return &outer_[0]; } const I *outerIndexPtr() const { return &outer_[0]; }
    void setZero() { outer_.assign((size_t)c_ + 1, 0); inner_.clear(); val_.clear(); }
    // duplicates are summed; entries end up sorted by (column, row)
    template <class It> void setFromTriplets(It b, It e) {
        std::vector<std::pair<std::pair<I, I>, S> > t;
        for (It it = b; it != e; ++it) t.push_back(std::make_pair(std::make_pair((I)it->col(), (I)it->row()), (S)it->value()));
        std::stable_sort(t.begin(), t.end(), [](const std::pair<std::pair<I, I>, S> &a, const std::pair<std::pair<I, I>, S> &b2) { return a.first < b2.first; });
        outer_.assign((size_t)c_ + 1, 0); inner_.clear(); val_.clear();
        for (size_t k = 0; k < t.size(); ++k) {
            if (k && t[k].first == t[k - 1].first) { val_.back() += t[k].second; continue; }
            inner_.push_back(t[k].first.second); val_.push_back(t[k].second); outer_[(size_t)t[k].first.first + 1]++;
        }
        for (Index j = 0; j < c_; ++j) outer_[(size_t)j + 1] += outer_[(size_t)j];
    }
    S coeff(Index i, Index j) const {
        for (I k = outer_[(size_t)j]; k < outer_[(size_t)j + 1]; ++k) if (inner_[(size_t)k] == (I)i) return val_[(size_t)k];
        return S(0);
    }
    template <int UpLo> SparseSelfAdjointView<SparseMatrix, UpLo> selfadjointView() { return SparseSelfAdjointView<SparseMatrix, UpLo>(*this); }
    template <int UpLo> SparseSelfAdjointView<const SparseMatrix, UpLo> selfadjointView() const { return SparseSelfAdjointView<const SparseMatrix, UpLo>(*this); }
    template <class SM2, int U2> SparseMatrix &operator=(const SparseSelfAdjointView<SM2, U2> &v);   // full symmetric matrix from one triangle
};

template <class SM, int UpLo> class SparseSelfAdjointView {
public:
    SM &m;
    explicit SparseSelfAdjointView(SM &m_) : m(m_) {}
    SparseSymmetricPermutationProduct<SM, UpLo> twistedBy(const PermutationMatrix<> &p) const { SparseSymmetricPermutationProduct<SM, UpLo> r = {m, p}; return r; }
    // dest(upper) = P * src(sym) * P^-1 : entry (i,j) goes to (p[i], p[j]) (Eigen's permute_symm_to_symm convention)
    template <class SM2, int U2> SparseSelfAdjointView &operator=(const SparseSymmetricPermutationProduct<SM2, U2> &prod) {
        typedef typename std::remove_const<SM>::type Plain; typedef typename Plain::Scalar S;
        std::vector<Triplet<S> > t;
        const SM2 &a = prod.m;
        for (Index j = 0; j < a.cols(); ++j)
            for (int k = a.outer_[(size_t)j]; k < a.outer_[(size_t)j + 1]; ++k) {
                Index i = a.inner_[(size_t)k];
                if ((U2 == Upper && i > j) || (U2 == Lower && i < j)) continue;
                Index pi = prod.p.size() ? prod.p.indices()[i] : i, pj = prod.p.size() ? prod.p.indices()[j] : j;
                if ((UpLo == Upper && pi > pj) || (UpLo == Lower && pi < pj)) std::swap(pi, pj);
                t.push_back(Triplet<S>((int)pi, (int)pj, a.val_[(size_t)k]));
            }
        m.resize(a.rows(), a.cols());
        m.setFromTriplets(t.begin(), t.end());
        return *this;
    }
};
template <class S, int O, class I> template <class SM2, int U2>
SparseMatrix<S, O, I> &SparseMatrix<S, O, I>::operator=(const SparseSelfAdjointView<SM2, U2> &v) {
    std::vector<Triplet<S> > t; const SM2 &a = v.m;
    for (Index j = 0; j < a.cols(); ++j)
        for (int k = a.outer_[(size_t)j]; k < a.outer_[(size_t)j + 1]; ++k) {
            Index i = a.inner_[(size_t)k];
            if ((U2 == Upper && i > j) || (U2 == Lower && i < j)) continue;
            t.push_back(Triplet<S>((int)i, (int)j, a.val_[(size_t)k]));
            if (i != j) t.push_back(Triplet<S>((int)j, (int)i, a.val_[(size_t)k]));
        }
    resize(a.rows(), a.cols());
    setFromTriplets(t.begin(), t.end());
    return *this;
}

namespace internal {
// Fill-reducing ordering stand-in: the natural order.  (Eigen runs AMD here; an ordering changes the
// fill-in and the rounding of the factorisation, not the solution.)
template <class SM, class P> void minimum_degree_ordering(SM &C, P &perm) { perm.resize(C.cols()); }
}  // namespace internal

// Sparse LDL^T without pivoting of the matrix whose UpLo triangle is stored: the up-looking algorithm of
// T. Davis' LDL package, which is also what Eigen's SimplicialCholesky implements.  D == 0 -> NumericalIssue.
template <class SM, int UpLo_ = Lower> class SimplicialLDLT {
public:
    typedef typename SM::Scalar Scalar;
    typedef SM CholMatrixType;
    typedef SM MatrixType;
    enum { UpLo = UpLo_ };
protected:
    ComputationInfo m_info; bool m_analysisIsOk;
    PermutationMatrix<> m_P, m_Pinv;      // m_P: new -> old ("inverse" in g2o's naming), m_Pinv: old -> new
    Index n_;
    std::vector<int> parent_, lp_, li_, lnz_; std::vector<Scalar> lx_, d_;
    struct LView { const SimplicialLDLT *s; struct Nested { Index nz; Index nonZeros() const { return nz; } }; Nested nestedExpression() const { Nested n = {(Index)s->lx_.size()}; return n; } };

    // upper triangle (column j holds rows <= j) of P A P^T as triplet-free CCS
    void permutedUpper(const SM &a, SM &ap) const {
        std::vector<Triplet<Scalar> > t;
        for (Index j = 0; j < a.cols(); ++j)
            for (int k = a.outer_[(size_t)j]; k < a.outer_[(size_t)j + 1]; ++k) {
                Index i = a.inner_[(size_t)k];
                if ((UpLo == Upper && i > j) || (UpLo == Lower && i < j)) continue;
                Index pi = m_Pinv.size() ? m_Pinv.indices()[i] : i, pj = m_Pinv.size() ? m_Pinv.indices()[j] : j;
                if (pi > pj) std::swap(pi, pj);
                t.push_back(Triplet<Scalar>((int)pi, (int)pj, a.val_[(size_t)k]));
            }
        ap.resize(a.rows(), a.cols());
        ap.setFromTriplets(t.begin(), t.end());
    }
    void symbolic(const SM &ap) {
        n_ = ap.cols(); const Index n = n_;
        parent_.assign((size_t)n, -1); lnz_.assign((size_t)n, 0); lp_.assign((size_t)n + 1, 0);
        std::vector<int> flag((size_t)n, -1);
        for (Index k = 0; k < n; ++k) {
            flag[(size_t)k] = (int)k;
            for (int p = ap.outer_[(size_t)k]; p < ap.outer_[(size_t)k + 1]; ++p) {
                int i = ap.inner_[(size_t)p];
                if (i < k)
                    for (; flag[(size_t)i] != k; i = parent_[(size_t)i]) {
                        if (parent_[(size_t)i] == -1) parent_[(size_t)i] = (int)k;
                        lnz_[(size_t)i]++; flag[(size_t)i] = (int)k;
                    }
            }
        }
        for (Index k = 0; k < n; ++k) lp_[(size_t)k + 1] = lp_[(size_t)k] + lnz_[(size_t)k];
        li_.assign((size_t)lp_[(size_t)n], 0); lx_.assign((size_t)lp_[(size_t)n], Scalar(0)); d_.assign((size_t)n, Scalar(0));
        m_analysisIsOk = true;
    }
    void numeric(const SM &ap) {
        const Index n = n_;
        std::vector<Scalar> y((size_t)n, Scalar(0)); std::vector<int> pattern((size_t)n, 0), flag((size_t)n, -1);
        std::fill(lnz_.begin(), lnz_.end(), 0);
        m_info = Success;
        for (Index k = 0; k < n; ++k) {
            Index top = n; flag[(size_t)k] = (int)k;
            for (int p = ap.outer_[(size_t)k]; p < ap.outer_[(size_t)k + 1]; ++p) {
                int i = ap.inner_[(size_t)p];
                if (i <= k) {
                    y[(size_t)i] += ap.val_[(size_t)p];
                    Index len = 0;
                    for (; flag[(size_t)i] != k; i = parent_[(size_t)i]) { pattern[(size_t)len++] = i; flag[(size_t)i] = (int)k; }
                    while (len > 0) pattern[(size_t)(--top)] = pattern[(size_t)(--len)];
                }
            }
            Scalar dk = y[(size_t)k]; y[(size_t)k] = Scalar(0);
            for (; top < n; ++top) {
                const int i = pattern[(size_t)top];
                const Scalar yi = y[(size_t)i]; y[(size_t)i] = Scalar(0);
                const int p2 = lp_[(size_t)i] + lnz_[(size_t)i];
                for (int p = lp_[(size_t)i]; p < p2; ++p) y[(size_t)li_[(size_t)p]] -= lx_[(size_t)p] * yi;
                const Scalar lki = yi / d_[(size_t)i];
                dk -= lki * yi;
                li_[(size_t)p2] = (int)k; lx_[(size_t)p2] = lki; lnz_[(size_t)i]++;
            }
            d_[(size_t)k] = dk;
            if (dk == Scalar(0)) { m_info = NumericalIssue; return; }
        }
    }
public:
    SimplicialLDLT() : m_info(Success), m_analysisIsOk(false), n_(0) {}
    ComputationInfo info() const { return m_info; }
    void analyzePattern(const SM &a) {
        SM c; c.resize(a.cols(), a.cols());
        internal::minimum_degree_ordering(c, m_P);
        m_Pinv = m_P.inverse();
        SM ap; permutedUpper(a, ap); symbolic(ap);
    }
    void analyzePattern_preordered(const SM &ap, bool /*doLDLT*/) { symbolic(ap); }
    void factorize(const SM &a) { assert(m_analysisIsOk); SM ap; permutedUpper(a, ap); numeric(ap); }
    void compute(const SM &a) { analyzePattern(a); factorize(a); }
    LView matrixL() const { LView v = {this}; return v; }
    template <class B> Matrix<Scalar, Dynamic, 1> solve(const MatrixBase<B> &b) const {
        const Index n = n_; Matrix<Scalar, Dynamic, 1> x(n), r(n);
        for (Index i = 0; i < n; ++i) x[m_Pinv.size() ? m_Pinv.indices()[i] : i] = b[i];
        for (Index j = 0; j < n; ++j) for (int p = lp_[(size_t)j]; p < lp_[(size_t)j] + lnz_[(size_t)j]; ++p) x[li_[(size_t)p]] -= lx_[(size_t)p] * x[j];
        for (Index j = 0; j < n; ++j) x[j] /= d_[(size_t)j];
        for (Index j = n - 1; j >= 0; --j) for (int p = lp_[(size_t)j]; p < lp_[(size_t)j] + lnz_[(size_t)j]; ++p) x[j] -= lx_[(size_t)p] * x[li_[(size_t)p]];
        for (Index i = 0; i < n; ++i) r[i] = x[m_Pinv.size() ? m_Pinv.indices()[i] : i];
        return r;
    }
};

}  // namespace Eigen
