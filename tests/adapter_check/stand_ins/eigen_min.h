// TEST-ONLY STAND-IN.  Not Eigen, not part of the product, not an oracle: the smallest set of declarations with Eigen's
// names and storage conventions (column-major; fixed-size matrices are plain arrays; Isometry3d = 4x4 matrix) that
// include/calibba_adapter.hpp touches, so that the adapter can be type-checked and exercised in an image that has no
// Eigen.  It pins nothing about the reference; in the reference's tree the adapter is compiled against the real Eigen.
#pragma once
#include <cmath>
#include <cstddef>
#include <type_traits>
#include <utility>
#include <vector>

namespace Eigen {
using Index = std::ptrdiff_t;
constexpr int Dynamic = -1;

template <class M> class Map;

namespace stand_in {
template <class T, int R, int C>
struct FixedStore {  // exactly R*C scalars, like the real fixed-size types (PlanarObservation relies on it)
    T a[R * C];
    FixedStore() : a{} {}
    static constexpr Index rows() { return R; }
    static constexpr Index cols() { return C; }
    void resize(Index, Index) {}
    T* data() { return a; }
    const T* data() const { return a; }
};
template <class T>
struct DynStore {
    std::vector<T> a;
    Index r = 0, c = 0;
    Index rows() const { return r; }
    Index cols() const { return c; }
    void resize(Index rr, Index cc) { r = rr; c = cc; a.assign(static_cast<size_t>(rr * cc), T(0)); }
    T* data() { return a.data(); }
    const T* data() const { return a.data(); }
};
}  // namespace stand_in

template <class T, int R, int C>
class Matrix {
  public:
    using Scalar = T;
    static constexpr bool kFixed = R != Dynamic && C != Dynamic;
    Matrix() = default;
    Matrix(Index r, Index c) { s_.resize(r, c); }
    explicit Matrix(Index n) { s_.resize(C == 1 ? n : 1, C == 1 ? 1 : n); }
    template <int RR = R, int CC = C, class = std::enable_if_t<RR * CC == 2>>
    Matrix(T a, T b) { s_.data()[0] = a; s_.data()[1] = b; }
    template <int RR = R, int CC = C, class = std::enable_if_t<RR * CC == 3>>
    Matrix(T a, T b, T c) { s_.data()[0] = a; s_.data()[1] = b; s_.data()[2] = c; }
    template <class M> Matrix(const Map<M>& m) { *this = m; }
    template <class M> Matrix& operator=(const Map<M>& m) {
        s_.resize(m.rows(), m.cols());
        for (Index i = 0; i < size(); ++i) data()[i] = m.data()[i];
        return *this;
    }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() { Matrix m; for (Index i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
    Index rows() const { return s_.rows(); }
    Index cols() const { return s_.cols(); }
    Index size() const { return rows() * cols(); }
    T* data() { return s_.data(); }
    const T* data() const { return s_.data(); }
    T& operator()(Index r, Index c) { return data()[r + c * rows()]; }
    const T& operator()(Index r, Index c) const { return data()[r + c * rows()]; }
    T& operator()(Index i) { return data()[i]; }
    const T& operator()(Index i) const { return data()[i]; }
    T& operator[](Index i) { return data()[i]; }
    const T& operator[](Index i) const { return data()[i]; }
    T& x() { return data()[0]; }
    const T& x() const { return data()[0]; }
    T& y() { return data()[1]; }
    const T& y() const { return data()[1]; }
    Matrix<T, Dynamic, 1> diagonal() const {
        const Index n = rows() < cols() ? rows() : cols();
        Matrix<T, Dynamic, 1> d(n);
        for (Index i = 0; i < n; ++i) d(i) = (*this)(i, i);
        return d;
    }
    Matrix cwiseAbs() const {
        Matrix m = *this;
        for (Index i = 0; i < size(); ++i) m(i) = std::abs(m(i));
        return m;
    }
    T maxCoeff() const {
        T b = size() ? data()[0] : T(0);
        for (Index i = 1; i < size(); ++i) b = data()[i] > b ? data()[i] : b;
        return b;
    }

  private:
    std::conditional_t<kFixed, stand_in::FixedStore<T, (kFixed ? R : 1), (kFixed ? C : 1)>, stand_in::DynStore<T>> s_;
};

template <class M>
class Map {
  public:
    using T = typename M::Scalar;
    Map(const T* p, Index r, Index c) : p_(p), rows_(r), cols_(c) {}
    Map(const T* p, Index n) : p_(p), rows_(n), cols_(1) {}
    const T* data() const { return p_; }
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }

  private:
    const T* p_;
    Index rows_, cols_;
};

using MatrixXd = Matrix<double, Dynamic, Dynamic>;
using VectorXd = Matrix<double, Dynamic, 1>;
using Vector2d = Matrix<double, 2, 1>;
using Vector3d = Matrix<double, 3, 1>;
using Matrix3d = Matrix<double, 3, 3>;
using Matrix4d = Matrix<double, 4, 4>;

enum TransformTraits { Isometry = 1 };
template <class T, int Dim, int Mode>
class Transform {
  public:
    Transform() = default;
    static Transform Identity() {
        Transform t;
        t.m_ = Matrix<T, Dim + 1, Dim + 1>::Identity();
        return t;
    }
    T* data() { return m_.data(); }
    const T* data() const { return m_.data(); }
    Matrix<T, Dim, Dim> linear() const {
        Matrix<T, Dim, Dim> r;
        for (int i = 0; i < Dim; ++i)
            for (int j = 0; j < Dim; ++j) r(i, j) = m_(i, j);
        return r;
    }
    Matrix<T, Dim, 1> translation() const {
        Matrix<T, Dim, 1> t;
        for (int i = 0; i < Dim; ++i) t(i) = m_(i, Dim);
        return t;
    }
    Matrix<T, Dim + 1, Dim + 1>& matrix() { return m_; }
    const Matrix<T, Dim + 1, Dim + 1>& matrix() const { return m_; }

  private:
    Matrix<T, Dim + 1, Dim + 1> m_;
};
using Isometry3d = Transform<double, 3, Isometry>;
}  // namespace Eigen
