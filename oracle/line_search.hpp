// oracle/line_search.hpp — TEST INFRASTRUCTURE ONLY.
//
// Restatement of the projected Armijo line search Ceres' trust-region minimiser runs on every step of a bounds-constrained
// problem (trust_region_minimizer.cc DoLineSearch -> line_search.cc ArmijoLineSearch::DoSearch, polynomial.cc), with the
// Solver::Options defaults the reference leaves untouched (ceresutils.h:28-35 sets only tolerances, iterations, threads):
//   line_search_interpolation_type CUBIC, line_search_sufficient_function_decrease 1e-4, max_line_search_step_contraction 1e-3,
//   min_line_search_step_contraction 0.6, max_num_line_search_step_size_iterations 20, min_line_search_step_size 1e-9.
// Ceres is a third-party dependency that is not in /root/reference: restated from its published sources, PARITY UNPINNED (no
// fixture of the reference records a Ceres iteration).  Reference call sites that make the problems constrained:
// intrinsics.cpp:81-82, extrinsics.cpp:143-144, bundle.cpp:121-123 (fx, fy >= 0).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>
#include <vector>

namespace orc {

struct LsSample {  // ceres::FunctionSample
    double x = 0, value = 0, gradient = 0;
    bool value_valid = false, gradient_valid = false;
};

inline double poly_eval(const std::vector<double>& p, double x) {  // highest power first (polynomial.h EvaluatePolynomial)
    double v = 0;
    for (double c : p) v = v * x + c;
    return v;
}

// real parts of all roots of p (highest power first), as FindPolynomialRoots returns them (closed forms up to degree 2; above,
// Ceres takes the eigenvalues of the balanced companion matrix — here a Durand-Kerner iteration on the monic polynomial)
inline std::vector<double> poly_roots_real(std::vector<double> p) {
    while (!p.empty() && p.front() == 0.0) p.erase(p.begin());
    const int deg = static_cast<int>(p.size()) - 1;
    std::vector<double> out;
    if (deg < 1) return out;
    if (deg == 1) { out.push_back(-p[1] / p[0]); return out; }
    if (deg == 2) {  // FindQuadraticPolynomialRoots
        const double a = p[0], b = p[1], c = p[2], D = b * b - 4 * a * c, sD = std::sqrt(std::fabs(D));
        if (D >= 0) {
            if (b >= 0) { out.push_back((-b - sD) / (2.0 * a)); out.push_back((2.0 * c) / (-b - sD)); }
            else { out.push_back((2.0 * c) / (-b + sD)); out.push_back((-b + sD) / (2.0 * a)); }
        } else {
            out.push_back(-b / (2.0 * a)); out.push_back(-b / (2.0 * a));
        }
        return out;
    }
    using cd = std::complex<double>;
    std::vector<cd> z(deg);
    double bound = 0;  // Cauchy bound for the start radius
    for (int i = 1; i <= deg; ++i) bound = std::max(bound, std::fabs(p[i] / p[0]));
    for (int i = 0; i < deg; ++i) z[i] = std::polar(0.5 * (1.0 + bound), 2.0 * M_PI * i / deg + 0.4);
    for (int it = 0; it < 500; ++it) {
        double move = 0;
        for (int i = 0; i < deg; ++i) {
            cd v = p[0];
            for (int k = 1; k <= deg; ++k) v = v * z[i] + p[k];
            cd den = p[0];
            for (int j = 0; j < deg; ++j)
                if (j != i) den *= (z[i] - z[j]);
            const cd d = v / den;
            z[i] -= d;
            move = std::max(move, std::abs(d));
        }
        if (move <= 1e-15 * (1.0 + bound)) break;
    }
    for (const cd& r : z) out.push_back(r.real());
    return out;
}

// FindInterpolatingPolynomial: every valid value and gradient of the samples is one linear constraint
inline std::vector<double> interpolating_polynomial(const std::vector<LsSample>& s) {
    int nc = 0;
    for (const auto& a : s) nc += (a.value_valid ? 1 : 0) + (a.gradient_valid ? 1 : 0);
    const int deg = nc - 1;
    std::vector<double> A(static_cast<size_t>(nc) * nc, 0.0), b(nc, 0.0);
    int row = 0;
    for (const auto& a : s) {
        if (a.value_valid) {
            for (int j = 0; j <= deg; ++j) A[row * nc + j] = std::pow(a.x, deg - j);
            b[row++] = a.value;
        }
        if (a.gradient_valid) {
            for (int j = 0; j < deg; ++j) A[row * nc + j] = (deg - j) * std::pow(a.x, deg - j - 1);
            b[row++] = a.gradient;
        }
    }
    // full-pivot LU (Eigen::FullPivLU in Ceres)
    std::vector<int> colperm(nc);
    for (int i = 0; i < nc; ++i) colperm[i] = i;
    for (int k = 0; k < nc; ++k) {
        int pr = k, pc = k;
        double best = -1;
        for (int i = k; i < nc; ++i)
            for (int j = k; j < nc; ++j)
                if (std::fabs(A[i * nc + j]) > best) { best = std::fabs(A[i * nc + j]); pr = i; pc = j; }
        if (best <= 0) break;
        if (pr != k) { for (int j = 0; j < nc; ++j) std::swap(A[pr * nc + j], A[k * nc + j]); std::swap(b[pr], b[k]); }
        if (pc != k) { for (int i = 0; i < nc; ++i) std::swap(A[i * nc + pc], A[i * nc + k]); std::swap(colperm[pc], colperm[k]); }
        for (int i = k + 1; i < nc; ++i) {
            const double f = A[i * nc + k] / A[k * nc + k];
            if (f == 0.0) continue;
            for (int j = k; j < nc; ++j) A[i * nc + j] -= f * A[k * nc + j];
            b[i] -= f * b[k];
        }
    }
    std::vector<double> y(nc, 0.0), coeff(nc, 0.0);
    for (int i = nc - 1; i >= 0; --i) {
        double v = b[i];
        for (int j = i + 1; j < nc; ++j) v -= A[i * nc + j] * y[j];
        y[i] = A[i * nc + i] != 0.0 ? v / A[i * nc + i] : 0.0;
    }
    for (int i = 0; i < nc; ++i) coeff[colperm[i]] = y[i];
    return coeff;
}

// MinimizeInterpolatingPolynomial (polynomial.cc): end points, the real parts of the derivative's roots inside the interval,
// and the sample abscissae inside the interval
inline double minimize_interpolating_polynomial(const std::vector<LsSample>& s, double x_min, double x_max) {
    const std::vector<double> p = interpolating_polynomial(s);
    double best_x = 0.5 * (x_min + x_max), best_v = poly_eval(p, best_x);
    const double vmin = poly_eval(p, x_min), vmax = poly_eval(p, x_max);
    const double end_x = vmin < vmax ? x_min : x_max, end_v = std::min(vmin, vmax);
    std::vector<double> d;
    const int deg = static_cast<int>(p.size()) - 1;
    for (int j = 0; j < deg; ++j) d.push_back((deg - j) * p[j]);
    for (double r : poly_roots_real(d)) {
        if (r < x_min || r > x_max) continue;
        const double v = poly_eval(p, r);
        if (v < best_v) { best_v = v; best_x = r; }
    }
    if (end_v < best_v) { best_v = end_v; best_x = end_x; }
    for (const auto& a : s) {
        if (a.x < x_min || a.x > x_max) continue;
        const double v = poly_eval(p, a.x);
        if (v < best_v) { best_v = v; best_x = a.x; }
    }
    return best_x;
}

// ArmijoLineSearch::DoSearch with CUBIC interpolation.  eval(step) evaluates cost and directional derivative at
// Plus(x, step * direction) (LineSearchFunction::Evaluate; Plus projects onto the bounds).  Returns the accepted step size, or a
// negative number when the search fails (Ceres then leaves the step as it is).
// (Ceres evaluates the gradient with every sample; the decision at a sample needs the value only, so eval(step, false) is
// asked first and the gradient is evaluated — eval(step, true) — only for samples that fail the condition and feed the cubic.)
inline double armijo_search(double cost0, double dir_grad0, double dir_inf_norm, const std::function<LsSample(double, bool)>& eval,
                            int* n_evals = nullptr) {
    const double sufficient_decrease = 1e-4, max_contraction = 1e-3, min_contraction = 0.6, min_step_size = 1e-9;
    const int max_iterations = 20;
    LsSample initial;
    initial.x = 0; initial.value = cost0; initial.gradient = dir_grad0; initial.value_valid = initial.gradient_valid = true;
    LsSample previous, current = eval(1.0, false);
    int iters = 0, evals = 1;
    while (!current.value_valid || current.value > cost0 + sufficient_decrease * dir_grad0 * current.x) {
        if (++iters >= max_iterations) { if (n_evals) *n_evals = evals; return -1.0; }
        if (current.value_valid && !current.gradient_valid) { current = eval(current.x, true); ++evals; }
        double step;
        const double lo = max_contraction * current.x, hi = min_contraction * current.x;
        if (!current.value_valid) {
            step = std::min(std::max(current.x * 0.5, lo), hi);
        } else {
            std::vector<LsSample> samples{initial, current};
            if (previous.value_valid) samples.push_back(previous);
            step = minimize_interpolating_polynomial(samples, lo, hi);
        }
        if (step * dir_inf_norm < min_step_size) { if (n_evals) *n_evals = evals; return -1.0; }
        previous = current;
        current = eval(step, false);
        ++evals;
    }
    if (n_evals) *n_evals = evals;
    return current.x;
}

}  // namespace orc
