// oracle/jet.hpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Minimal forward-mode dual number, used so the oracle differentiates the
// reference's residual formulas exactly the way the reference does: by running
// the *same scalar-templated formula* on a value+gradient type.  The reference
// gets this from ceres::Jet via ceres::AutoDiffCostFunction
// (src/estimation/residuals/intrinsicresidual.h:44-46,
//  extrinsicsresidual.h:55-57, bundleresidual.h:64-66, handeyeresidual.h:52).
// Ceres itself is not in /root/reference (un-vendored, version-unpinned
// dependency: cmake/Dependencies.cmake:3); the rules below are the standard
// first-order dual-number rules that ceres::Jet publishes.
#pragma once
#include <cmath>

namespace orc {

template <int N>
struct Jet {
    double a;
    double v[N];

    Jet() : a(0.0) {
        for (int i = 0; i < N; ++i) v[i] = 0.0;
    }
    Jet(double s) : a(s) {  // NOLINT: implicit on purpose, mirrors T(x)
        for (int i = 0; i < N; ++i) v[i] = 0.0;
    }
    Jet(double s, int k) : a(s) {
        for (int i = 0; i < N; ++i) v[i] = 0.0;
        v[k] = 1.0;
    }
};

template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) {
    Jet<N> r; r.a = x.a + y.a;
    for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i];
    return r;
}
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) {
    Jet<N> r; r.a = x.a - y.a;
    for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i];
    return r;
}
template <int N> inline Jet<N> operator-(const Jet<N>& x) {
    Jet<N> r; r.a = -x.a;
    for (int i = 0; i < N; ++i) r.v[i] = -x.v[i];
    return r;
}
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) {
    Jet<N> r; r.a = x.a * y.a;
    for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a;
    return r;
}
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
    // (x/y)' = (x' - (x/y) y') / y
    Jet<N> r;
    const double inv = 1.0 / y.a;
    r.a = x.a * inv;
    for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv;
    return r;
}
template <int N> inline Jet<N> operator+(const Jet<N>& x, double s) { Jet<N> r = x; r.a += s; return r; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& x) { Jet<N> r = x; r.a += s; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, double s) { Jet<N> r = x; r.a -= s; return r; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& x) { Jet<N> r = -x; r.a += s; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, double s) {
    Jet<N> r; r.a = x.a * s;
    for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s;
    return r;
}
template <int N> inline Jet<N> operator*(double s, const Jet<N>& x) { return x * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, double s) { return x * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& y) { return Jet<N>(s) / y; }
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline Jet<N>& operator-=(Jet<N>& x, const Jet<N>& y) { x = x - y; return x; }
template <int N> inline Jet<N>& operator*=(Jet<N>& x, const Jet<N>& y) { x = x * y; return x; }
template <int N> inline Jet<N>& operator+=(Jet<N>& x, double y) { x.a += y; return x; }

// comparisons act on the scalar part (as ceres::Jet does)
template <int N> inline bool operator<(const Jet<N>& x, const Jet<N>& y) { return x.a < y.a; }
template <int N> inline bool operator>(const Jet<N>& x, const Jet<N>& y) { return x.a > y.a; }
template <int N> inline bool operator<(const Jet<N>& x, double y) { return x.a < y; }
template <int N> inline bool operator>(const Jet<N>& x, double y) { return x.a > y; }
template <int N> inline bool operator!=(const Jet<N>& x, double y) { return x.a != y; }

template <int N> inline Jet<N> sin(const Jet<N>& x) {
    Jet<N> r; r.a = std::sin(x.a); const double c = std::cos(x.a);
    for (int i = 0; i < N; ++i) r.v[i] = c * x.v[i];
    return r;
}
template <int N> inline Jet<N> cos(const Jet<N>& x) {
    Jet<N> r; r.a = std::cos(x.a); const double s = -std::sin(x.a);
    for (int i = 0; i < N; ++i) r.v[i] = s * x.v[i];
    return r;
}
template <int N> inline Jet<N> sqrt(const Jet<N>& x) {
    Jet<N> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a;
    for (int i = 0; i < N; ++i) r.v[i] = d * x.v[i];
    return r;
}
template <int N> inline Jet<N> abs(const Jet<N>& x) { return x.a < 0.0 ? -x : x; }
template <int N> inline Jet<N> atan2(const Jet<N>& y, const Jet<N>& x) {
    // d atan2(y,x) = (x dy - y dx) / (x^2 + y^2)
    Jet<N> r; r.a = std::atan2(y.a, x.a);
    const double d = 1.0 / (x.a * x.a + y.a * y.a);
    for (int i = 0; i < N; ++i) r.v[i] = (x.a * y.v[i] - y.a * x.v[i]) * d;
    return r;
}

inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double abs(double x) { return std::fabs(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

inline double scalar_of(double x) { return x; }
template <int N> inline double scalar_of(const Jet<N>& x) { return x.a; }

}  // namespace orc
