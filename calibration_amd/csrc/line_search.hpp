// line_search.hpp — the projected Armijo line search of bounds-constrained problems (host code of the LM driver).
//
// ceres::Solver runs a line search on every trust-region step of a problem that has parameter bounds
// (TrustRegionMinimizer::DoLineSearch, Ceres 2.x; fx, fy >= 0 make every problem with variable intrinsics one:
// intrinsics.cpp:81-82, extrinsics.cpp:143-144, bundle.cpp:121-123): an ARMIJO search along the step from the current point,
// through the projecting Plus, first trial step size 1, sufficient decrease 1e-4, cubic interpolation of the sampled
// values and directional derivatives, contraction of the step size into [1e-3, 0.6] of the previous one, at most 20
// iterations, minimum step size 1e-9 (Solver::Options defaults; the reference sets none of them, ceresutils.h:28-35).
// The step is then SCALED by the step size found; its model-cost change keeps the value of the full step.
// A full step that already satisfies the Armijo condition — every step the gain-ratio test would accept does — is left as it
// is, so the search only ever runs on steps that would otherwise be rejected.  Ceres is not in /root/reference: restated.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace cba {

struct LineSample {
    double step = 0, value = 0, slope = 0;  // slope: directional derivative along the search direction
    bool has_value = false, has_slope = false;
};

namespace ls_detail {

inline double horner(const std::vector<double>& c, double x) {  // c[0] x^d + ... + c[d]
    double v = 0.0;
    for (double ci : c) v = v * x + ci;
    return v;
}

// coefficients (highest power first) of the polynomial through every value and slope the samples carry
inline std::vector<double> fit(const std::vector<LineSample>& samples) {
    int m = 0;
    for (const LineSample& s : samples) m += int(s.has_value) + int(s.has_slope);
    const int d = m - 1;
    std::vector<std::vector<double>> A(m, std::vector<double>(m + 1, 0.0));
    int r = 0;
    for (const LineSample& s : samples) {
        if (s.has_value) {
            for (int j = 0; j <= d; ++j) A[r][j] = std::pow(s.step, d - j);
            A[r++][m] = s.value;
        }
        if (s.has_slope) {
            for (int j = 0; j < d; ++j) A[r][j] = (d - j) * std::pow(s.step, d - j - 1);
            A[r++][m] = s.slope;
        }
    }
    std::vector<int> col(m);
    for (int i = 0; i < m; ++i) col[i] = i;
    for (int k = 0; k < m; ++k) {  // complete pivoting
        int pi = k, pj = k;
        for (int i = k; i < m; ++i)
            for (int j = k; j < m; ++j)
                if (std::fabs(A[i][j]) > std::fabs(A[pi][pj])) { pi = i; pj = j; }
        if (A[pi][pj] == 0.0) break;
        std::swap(A[pi], A[k]);
        if (pj != k) {
            for (int i = 0; i < m; ++i) std::swap(A[i][pj], A[i][k]);
            std::swap(col[pj], col[k]);
        }
        for (int i = k + 1; i < m; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j <= m; ++j) A[i][j] -= f * A[k][j];
        }
    }
    std::vector<double> y(m, 0.0), c(m, 0.0);
    for (int i = m - 1; i >= 0; --i) {
        double v = A[i][m];
        for (int j = i + 1; j < m; ++j) v -= A[i][j] * y[j];
        y[i] = A[i][i] != 0.0 ? v / A[i][i] : 0.0;
    }
    for (int i = 0; i < m; ++i) c[col[i]] = y[i];
    return c;
}

// real parts of the roots of c[0] x^d + ... (Ceres takes the real parts of ALL roots of the derivative, complex ones included)
inline std::vector<double> root_real_parts(std::vector<double> c) {
    while (!c.empty() && c.front() == 0.0) c.erase(c.begin());
    const int d = static_cast<int>(c.size()) - 1;
    std::vector<double> r;
    if (d < 1) return r;
    if (d == 1) return {-c[1] / c[0]};
    if (d == 2) {
        const double disc = c[1] * c[1] - 4.0 * c[0] * c[2], s = std::sqrt(std::fabs(disc));
        if (disc < 0.0) return {-c[1] / (2.0 * c[0]), -c[1] / (2.0 * c[0])};
        const double q = c[1] >= 0.0 ? -c[1] - s : -c[1] + s;  // cancellation-free
        return c[1] >= 0.0 ? std::vector<double>{q / (2.0 * c[0]), 2.0 * c[2] / q} : std::vector<double>{2.0 * c[2] / q, q / (2.0 * c[0])};
    }
    std::vector<std::complex<double>> z(d);
    double radius = 0.0;
    for (int i = 1; i <= d; ++i) radius = std::max(radius, std::fabs(c[i] / c[0]));
    radius = 0.5 * (1.0 + radius);
    for (int i = 0; i < d; ++i) z[i] = std::polar(radius, 0.4 + 6.283185307179586 * i / d);
    for (int sweep = 0; sweep < 500; ++sweep) {  // simultaneous (Weierstrass) iteration
        double moved = 0.0;
        for (int i = 0; i < d; ++i) {
            std::complex<double> num = c[0], den = c[0];
            for (int k = 1; k <= d; ++k) num = num * z[i] + c[k];
            for (int j = 0; j < d; ++j)
                if (j != i) den *= z[i] - z[j];
            const std::complex<double> step = num / den;
            z[i] -= step;
            moved = std::max(moved, std::abs(step));
        }
        if (moved <= 2e-15 * radius) break;
    }
    for (const auto& zi : z) r.push_back(zi.real());
    return r;
}

// the abscissa in [lo, hi] where the interpolating polynomial is smallest: interval ends, stationary points, sample abscissae
inline double argmin_on(const std::vector<LineSample>& samples, double lo, double hi) {
    const std::vector<double> c = fit(samples);
    const int d = static_cast<int>(c.size()) - 1;
    double best = 0.5 * (lo + hi), best_v = horner(c, best);
    auto consider = [&](double x) {
        if (x < lo || x > hi) return;
        const double v = horner(c, x);
        if (v < best_v) { best_v = v; best = x; }
    };
    std::vector<double> dc;
    for (int j = 0; j < d; ++j) dc.push_back((d - j) * c[j]);
    for (double x : root_real_parts(dc)) consider(x);
    {   // the better of the two ends replaces an interior candidate only when strictly lower
        const double vl = horner(c, lo), vh = horner(c, hi);
        const double xe = vl < vh ? lo : hi, ve = std::min(vl, vh);
        if (ve < best_v) { best_v = ve; best = xe; }
    }
    for (const LineSample& s : samples) consider(s.step);
    return best;
}

}  // namespace ls_detail

// The search.  sample(step, with_slope) evaluates the objective (and, when asked, its directional derivative) at
// Plus(x, step * direction).  Returns the step size to scale the direction with; <= 0 means the search failed and the step is
// left as it is.  evaluations: number of sample() calls made.  The step size returned is that of the LAST sample() call.
template <class Sample>
double armijo_line_search(double value0, double slope0, double direction_max_abs, Sample&& sample, int* evaluations,
                          const LineSample* first = nullptr) {
    constexpr double kDecrease = 1e-4, kMaxContraction = 1e-3, kMinContraction = 0.6, kMinStep = 1e-9;
    constexpr int kMaxIterations = 20;
    LineSample start;
    start.step = 0.0; start.value = value0; start.slope = slope0; start.has_value = start.has_slope = true;
    // first: the sample at step size 1 when the caller has it already (the LM's trial point IS that sample: no second evaluation)
    LineSample prev, cur = first ? *first : sample(1.0, false);
    *evaluations = first ? 0 : 1;
    for (int it = 0; !cur.has_value || cur.value > value0 + kDecrease * slope0 * cur.step;) {
        if (++it >= kMaxIterations) return -1.0;
        if (cur.has_value && !cur.has_slope) { cur = sample(cur.step, true); ++*evaluations; }  // the cubic wants the slope here
        const double lo = kMaxContraction * cur.step, hi = kMinContraction * cur.step;
        double next;
        if (!cur.has_value) {
            next = std::min(std::max(0.5 * cur.step, lo), hi);
        } else {
            std::vector<LineSample> pts{start, cur};
            if (prev.has_value) pts.push_back(prev);
            next = ls_detail::argmin_on(pts, lo, hi);
        }
        if (next * direction_max_abs < kMinStep) return -1.0;
        prev = cur;
        cur = sample(next, false);
        ++*evaluations;
    }
    return cur.step;
}

}  // namespace cba
