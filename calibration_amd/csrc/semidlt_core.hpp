// semidlt_core.hpp — host side of optimize_intrinsics_semidlt (src/estimation/optim/intrinsicssemidlt.cpp:155-191):
// the Levenberg-Marquardt driver over [kappa(5) | 6 tangent unknowns per view] and the linear algebra of its
// variable-projection normal equations.  The O(#observations) work is the evaluator's (HIP kernels in semidlt.hip;
// a single-thread evaluator in tests/cpu_backend); everything here is O(#views).
//
// One evaluation returns alpha and per-view sums (semidlt_math.hpp).  With rho' = w the Huber weight of the ONE
// residual block (intrinsicssemidlt.cpp:112-113) and N = A^T A = L L^T:
//     H = w (W^T W - Bt^T Bt + Dt^T Dt),   Bt = L^-1 A^T W,   Dt = L^-1 dA^T r,       g = w W^T r.
// W^T W is block-arrow (kappa block + one 6x6 block per view); the damped system (H + D_lm) delta = -g is solved by
// eliminating the view blocks of the arrow part and a Woodbury correction of rank 2m — O(#views) per step, never a
// dense (5 + 6V)^2 matrix.  The dense tangent matrix is formed only for the covariance, whose output (the
// reference's dense (5 + 7V)^2 ambient matrix, ceresutils.h:69-126) is quadratic in V anyway.
// Solver semantics: the restated Ceres rules of lm_core.hpp, incl. box bounds on kappa by projection
// (IntrinsicsOptimOptions::bounds, intrinsicssemidlt.cpp:121-135) and SubsetManifold for a fixed skew (:136-140).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

#include "../../include/calibba.h"
#include "dense.hpp"
#include "reproj_math.hpp"
#include "schur_math.hpp"

namespace cba {

struct SemiDltEval {
    int V = 0, nr = 2;
    int64_t n_obs = 0;
    virtual ~SemiDltEval() = default;
    int m() const { return nr + 2; }
    int n2() const { return 78 + 22 * m(); }
    // pass 1 summed over views: N (m x m, full symmetric) and A^T b at (kappa, poses)
    virtual void normal(const double* kappa5, const double* poses7, double* N, double* rhs) = 0;
    // whole evaluation: pass 1 -> alpha = N^-1 A^T b -> pass 2; per_view[V][n2()].  false if N is not positive definite.
    virtual bool evaluate(const double* kappa5, const double* poses7, double* N, double* rhs, double* alpha, double* per_view) = 0;
    // |A alpha - b|^2 per view for a given alpha
    virtual void resid(const double* kappa5, const double* poses7, const double* alpha, double* s_view) = 0;
};

struct SemiDltBounds {  // CalibrationBounds; lo > hi disables
    double lo[5], hi[5];
    bool enabled = false;
};

// One linearisation, assembled from the per-view sums.
struct SemiDltSystem {
    int V = 0, m = 0, nk = 0, n = 0;
    int kidx[5];                     // tangent column -> kappa index (skew dropped when fixed)
    std::vector<double> Kkk, Kkp, Kpp;  // w W^T W: nk*nk | V*nk*6 | V*36
    std::vector<double> g;              // n
    std::vector<double> U;              // n x 2m, columns sqrt(w) [Bt^T | Dt^T]
    std::vector<double> alpha;
    double s = 0.0, w = 1.0, cost = 0.0;
    bool ok = false;

    double diag(int i) const {
        double d;
        if (i < nk) d = Kkk[static_cast<size_t>(i) * nk + i];
        else { const int v = (i - nk) / 6, a = (i - nk) % 6; d = Kpp[static_cast<size_t>(v) * 36 + a * 6 + a]; }
        for (int c = 0; c < 2 * m; ++c) {
            const double u = U[static_cast<size_t>(i) * 2 * m + c];
            d += (c < m ? -u * u : u * u);
        }
        return d;
    }
    // y = H x
    void apply(const double* x, double* y) const {
        for (int i = 0; i < n; ++i) y[i] = 0.0;
        for (int i = 0; i < nk; ++i)
            for (int j = 0; j < nk; ++j) y[i] += Kkk[static_cast<size_t>(i) * nk + j] * x[j];
        for (int v = 0; v < V; ++v) {
            const double* E = &Kkp[static_cast<size_t>(v) * nk * 6];
            const double* P = &Kpp[static_cast<size_t>(v) * 36];
            const double* xp = x + nk + 6 * v;
            double* yp = y + nk + 6 * v;
            for (int i = 0; i < nk; ++i)
                for (int a = 0; a < 6; ++a) { y[i] += E[i * 6 + a] * xp[a]; yp[a] += E[i * 6 + a] * x[i]; }
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) yp[a] += P[a * 6 + b] * xp[b];
        }
        for (int c = 0; c < 2 * m; ++c) {
            double t = 0.0;
            for (int i = 0; i < n; ++i) t += U[static_cast<size_t>(i) * 2 * m + c] * x[i];
            const double sg = c < m ? -t : t;
            for (int i = 0; i < n; ++i) y[i] += U[static_cast<size_t>(i) * 2 * m + c] * sg;
        }
    }
    // (arrow + diag(dlm)) X = RHS for nrhs right-hand sides stored as rows of length n (in place); false if not PD
    bool arrow_solve(const std::vector<double>& dlm, std::vector<double>& rhs, int nrhs) const {
        std::vector<double> Lp(static_cast<size_t>(V) * 36), F(static_cast<size_t>(V) * nk * 6);  // F_v = L_v^-1 E_v^T  (6 x nk)
        std::vector<double> S(static_cast<size_t>(nk) * nk);
        for (int i = 0; i < nk; ++i)
            for (int j = 0; j < nk; ++j) S[static_cast<size_t>(i) * nk + j] = Kkk[static_cast<size_t>(i) * nk + j] + (i == j ? dlm[i] : 0.0);
        for (int v = 0; v < V; ++v) {
            double* L = &Lp[static_cast<size_t>(v) * 36];
            for (int a = 0; a < 36; ++a) L[a] = Kpp[static_cast<size_t>(v) * 36 + a];
            for (int a = 0; a < 6; ++a) L[a * 6 + a] += dlm[nk + 6 * v + a];
            if (!chol6(L)) return false;
            const double* E = &Kkp[static_cast<size_t>(v) * nk * 6];
            double* Fv = &F[static_cast<size_t>(v) * nk * 6];
            for (int i = 0; i < nk; ++i) {
                double col[6];
                for (int a = 0; a < 6; ++a) col[a] = E[i * 6 + a];
                fwd6(L, col);
                for (int a = 0; a < 6; ++a) Fv[a * nk + i] = col[a];
            }
            for (int i = 0; i < nk; ++i)
                for (int j = 0; j < nk; ++j) {
                    double t = 0.0;
                    for (int a = 0; a < 6; ++a) t += Fv[a * nk + i] * Fv[a * nk + j];
                    S[static_cast<size_t>(i) * nk + j] -= t;
                }
        }
        if (nk > 0 && !chol_inplace(S, nk)) return false;
        std::vector<double> bk(nk);
        for (int r = 0; r < nrhs; ++r) {
            double* b = &rhs[static_cast<size_t>(r) * n];
            for (int i = 0; i < nk; ++i) bk[i] = b[i];
            for (int v = 0; v < V; ++v) {  // y_v = L_v^-1 b_pv (kept in place), b_k -= F_v^T y_v
                double* bp = b + nk + 6 * v;
                fwd6(&Lp[static_cast<size_t>(v) * 36], bp);
                const double* Fv = &F[static_cast<size_t>(v) * nk * 6];
                for (int i = 0; i < nk; ++i)
                    for (int a = 0; a < 6; ++a) bk[i] -= Fv[a * nk + i] * bp[a];
            }
            if (nk > 0) chol_solve(S, nk, bk.data());
            for (int i = 0; i < nk; ++i) b[i] = bk[i];
            for (int v = 0; v < V; ++v) {  // x_pv = L_v^-T (y_v - F_v x_k)
                double* bp = b + nk + 6 * v;
                const double* Fv = &F[static_cast<size_t>(v) * nk * 6];
                for (int a = 0; a < 6; ++a)
                    for (int i = 0; i < nk; ++i) bp[a] -= Fv[a * nk + i] * bk[i];
                bwd6(&Lp[static_cast<size_t>(v) * 36], bp);
            }
        }
        return true;
    }
    // delta = -(H + diag(dlm))^-1 g by arrow elimination + Woodbury; false if the step is invalid
    bool solve(const std::vector<double>& dlm, std::vector<double>& delta) const {
        const int k2 = 2 * m;
        std::vector<double> rhs(static_cast<size_t>(k2 + 1) * n);
        for (int i = 0; i < n; ++i) rhs[i] = g[i];
        for (int c = 0; c < k2; ++c)
            for (int i = 0; i < n; ++i) rhs[static_cast<size_t>(c + 1) * n + i] = U[static_cast<size_t>(i) * k2 + c];
        if (!arrow_solve(dlm, rhs, k2 + 1)) return false;
        // T = Sg + U^T Y (Sg = diag(-1.., +1..)), z = T^-1 U^T y0
        std::vector<double> T(static_cast<size_t>(k2) * k2), z(k2);
        for (int a = 0; a < k2; ++a) {
            double t0 = 0.0;
            for (int i = 0; i < n; ++i) t0 += U[static_cast<size_t>(i) * k2 + a] * rhs[i];
            z[a] = t0;
            for (int c = 0; c < k2; ++c) {
                double t = 0.0;
                for (int i = 0; i < n; ++i) t += U[static_cast<size_t>(i) * k2 + a] * rhs[static_cast<size_t>(c + 1) * n + i];
                T[static_cast<size_t>(a) * k2 + c] = t + (a == c ? (a < m ? -1.0 : 1.0) : 0.0);
            }
        }
        for (int c = 0; c < k2; ++c) {  // Gaussian elimination with partial pivoting (T is symmetric indefinite)
            int p = c;
            for (int r = c + 1; r < k2; ++r)
                if (std::fabs(T[static_cast<size_t>(r) * k2 + c]) > std::fabs(T[static_cast<size_t>(p) * k2 + c])) p = r;
            if (!(std::fabs(T[static_cast<size_t>(p) * k2 + c]) > 0.0)) return false;
            if (p != c) {
                for (int j = 0; j < k2; ++j) std::swap(T[static_cast<size_t>(p) * k2 + j], T[static_cast<size_t>(c) * k2 + j]);
                std::swap(z[p], z[c]);
            }
            for (int r = c + 1; r < k2; ++r) {
                const double f = T[static_cast<size_t>(r) * k2 + c] / T[static_cast<size_t>(c) * k2 + c];
                for (int j = c; j < k2; ++j) T[static_cast<size_t>(r) * k2 + j] -= f * T[static_cast<size_t>(c) * k2 + j];
                z[r] -= f * z[c];
            }
        }
        for (int c = k2 - 1; c >= 0; --c) {
            double t = z[c];
            for (int j = c + 1; j < k2; ++j) t -= T[static_cast<size_t>(c) * k2 + j] * z[j];
            z[c] = t / T[static_cast<size_t>(c) * k2 + c];
        }
        delta.assign(n, 0.0);
        for (int i = 0; i < n; ++i) {
            double t = rhs[i];
            for (int c = 0; c < k2; ++c) t -= rhs[static_cast<size_t>(c + 1) * n + i] * z[c];
            delta[i] = -t;
        }
        return true;
    }
    void dense(std::vector<double>& H) const {
        H.assign(static_cast<size_t>(n) * n, 0.0);
        std::vector<double> e(n, 0.0), y(n);
        for (int c = 0; c < n; ++c) {
            e[c] = 1.0;
            apply(e.data(), y.data());
            e[c] = 0.0;
            for (int r = 0; r < n; ++r) H[static_cast<size_t>(r) * n + c] = y[r];
        }
    }
};

struct SemiDltResult {
    std::vector<double> alpha;        // fitted distortion [k1..k_nr, p1, p2] (solve_full, with the fixed entries)
    std::vector<double> view_errors;  // sqrt(sum r^2 / (2 N_v)) per view (intrinsicssemidlt.cpp:137-153)
    std::vector<double> cov;          // (5 + 7V)^2 or empty
};

class SemiDltDriver {
  public:
    SemiDltEval& ev;
    const cba_options o;
    SemiDltBounds bounds;
    std::vector<double> kappa, poses;  // current point
    explicit SemiDltDriver(SemiDltEval& e, const cba_options& opt) : ev(e), o(opt) {}

    bool linearise(const std::vector<double>& kap, const std::vector<double>& pos, SemiDltSystem& S) {
        const int V = ev.V, m = ev.m(), n2 = ev.n2();
        S.V = V; S.m = m; S.ok = false;
        S.nk = 0;
        for (int k = 0; k < 5; ++k)
            if (k != 4 || o.optimize_skew) S.kidx[S.nk++] = k;
        S.n = S.nk + 6 * V;
        if (ev.n_obs < 8) return false;  // fit_distortion_full: k_min_observations (distortion.h:235-238)
        std::vector<double> N(static_cast<size_t>(m) * m), rhs(m);
        S.alpha.assign(m, 0.0);
        pv_.resize(static_cast<size_t>(V) * n2);
        if (!ev.evaluate(kap.data(), pos.data(), N.data(), rhs.data(), S.alpha.data(), pv_.data())) return false;
        std::vector<double> L = N;
        if (!chol_inplace(L, m)) return false;
        const int OFF_G = 66, OFF_B = 77, OFF_D = 77 + 11 * m, OFF_S = 77 + 22 * m;
        S.s = 0.0;
        for (int v = 0; v < V; ++v) S.s += pv_[static_cast<size_t>(v) * n2 + OFF_S];
        double rho;
        huber(S.s, o.huber_delta, &rho, &S.w);
        S.cost = 0.5 * rho;
        if (!std::isfinite(S.cost)) return false;
        const double w = S.w, sw = std::sqrt(w);
        const int nk = S.nk, n = S.n, k2 = 2 * m;
        S.Kkk.assign(static_cast<size_t>(nk) * nk, 0.0);
        S.Kkp.assign(static_cast<size_t>(V) * nk * 6, 0.0);
        S.Kpp.assign(static_cast<size_t>(V) * 36, 0.0);
        S.g.assign(n, 0.0);
        S.U.assign(static_cast<size_t>(n) * k2, 0.0);
        auto loc = [&](int t, int v) { return t < nk ? S.kidx[t] : 5 + (t - nk - 6 * v); };  // tangent column -> local 0..10
        std::vector<double> col(m);
        for (int v = 0; v < V; ++v) {
            const double* a = &pv_[static_cast<size_t>(v) * n2];
            auto hw = [&](int p, int q) { if (p > q) std::swap(p, q); return a[hidx(11, p, q)]; };
            for (int i = 0; i < nk; ++i) {
                for (int j = 0; j < nk; ++j) S.Kkk[static_cast<size_t>(i) * nk + j] += w * hw(S.kidx[i], S.kidx[j]);
                for (int b = 0; b < 6; ++b) S.Kkp[(static_cast<size_t>(v) * nk + i) * 6 + b] = w * hw(S.kidx[i], 5 + b);
                S.g[i] += w * a[OFF_G + S.kidx[i]];
            }
            for (int p = 0; p < 6; ++p) {
                for (int q = 0; q < 6; ++q) S.Kpp[static_cast<size_t>(v) * 36 + p * 6 + q] = w * hw(5 + p, 5 + q);
                S.g[nk + 6 * v + p] = w * a[OFF_G + 5 + p];
            }
            // B and D columns: kappa columns accumulate over views, pose columns are the view's own
            for (int t = 0; t < nk + 6; ++t) {
                const int gi = t < nk ? t : nk + 6 * v + (t - nk);
                const int lc = t < nk ? S.kidx[t] : 5 + (t - nk);
                for (int c = 0; c < m; ++c) {
                    S.U[static_cast<size_t>(gi) * k2 + c] += a[OFF_B + c * 11 + lc];
                    S.U[static_cast<size_t>(gi) * k2 + m + c] += a[OFF_D + c * 11 + lc];
                }
            }
        }
        (void)loc;
        for (int i = 0; i < n; ++i)  // U_i <- sqrt(w) L^-1 [B_i | D_i]
            for (int half = 0; half < 2; ++half) {
                double* u = &S.U[static_cast<size_t>(i) * k2 + half * m];
                for (int r = 0; r < m; ++r) {
                    double t = u[r];
                    for (int k = 0; k < r; ++k) t -= L[static_cast<size_t>(r) * m + k] * u[k];
                    u[r] = t / L[static_cast<size_t>(r) * m + r];
                }
                for (int r = 0; r < m; ++r) u[r] *= sw;
            }
        S.ok = true;
        return true;
    }

    // Plus with bounds projection: kappa Euclidean (skew kept when fixed), quaternion manifold per view
    void plus(const std::vector<double>& kap, const std::vector<double>& pos, const SemiDltSystem& S, const double* delta,
              std::vector<double>& kap2, std::vector<double>& pos2) const {
        kap2 = kap; pos2 = pos;
        for (int i = 0; i < S.nk; ++i) kap2[S.kidx[i]] = kap[S.kidx[i]] + delta[i];
        if (bounds.enabled)
            for (int k = 0; k < 5; ++k) kap2[k] = std::min(std::max(kap2[k], bounds.lo[k]), bounds.hi[k]);
        for (int v = 0; v < S.V; ++v) {
            const double* d = delta + S.nk + 6 * v;
            quat_plus(&pos[7 * static_cast<size_t>(v)], d, &pos2[7 * static_cast<size_t>(v)]);
            for (int k = 0; k < 3; ++k) pos2[7 * static_cast<size_t>(v) + 4 + k] = pos[7 * static_cast<size_t>(v) + 4 + k] + d[3 + k];
        }
    }

    double grad_max(const SemiDltSystem& S) const {
        double mx = 0.0;
        if (!bounds.enabled) {
            for (int i = 0; i < S.n; ++i) mx = std::max(mx, std::fabs(S.g[i]));
            return mx;
        }
        std::vector<double> ng(S.n), k2, p2;  // |Plus(x, -g) - x|_inf over the ambient vector (projected gradient)
        for (int i = 0; i < S.n; ++i) ng[i] = -S.g[i];
        plus(kappa, poses, S, ng.data(), k2, p2);
        for (int k = 0; k < 5; ++k) mx = std::max(mx, std::fabs(k2[k] - kappa[k]));
        for (size_t i = 0; i < poses.size(); ++i) mx = std::max(mx, std::fabs(p2[i] - poses[i]));
        return mx;
    }

    void solve(double* kappa5, double* poses7, cba_summary* out) {
        const auto t0 = std::chrono::steady_clock::now();
        const int V = ev.V;
        const double eps = o.epsilon;
        kappa.assign(kappa5, kappa5 + 5);
        poses.assign(poses7, poses7 + 7 * static_cast<size_t>(V));
        if (bounds.enabled)  // Ceres projects the start point onto the feasible set
            for (int k = 0; k < 5; ++k) kappa[k] = std::min(std::max(kappa[k], bounds.lo[k]), bounds.hi[k]);
        int iter = 0, invalid = 0, successful = 0, term = CBA_TERM_FAILURE;
        const char* msg = "Residual and Jacobian evaluation failed.";
        double initial_cost = 0.0, cost = 0.0;
        if (linearise(kappa, poses, sys)) {
            SemiDltSystem cs;
            cost = initial_cost = sys.cost;
            const int n = sys.n;
            std::vector<double> scale2(n), dlm(n), delta, Hd(n), ck, cp;
            for (int i = 0; i < n; ++i) { const double sc = 1.0 / (1.0 + std::sqrt(std::max(0.0, sys.diag(i)))); scale2[i] = sc * sc; }
            double gmax = grad_max(sys), radius = 1e4, decrease_factor = 2.0;
            if (gmax <= eps) { term = CBA_TERM_CONVERGENCE; msg = "Gradient tolerance reached."; }
            else while (true) {
                if (iter >= o.max_iterations) { term = CBA_TERM_NO_CONVERGENCE; msg = "Maximum number of iterations reached."; break; }
                if (gmax <= eps) { term = CBA_TERM_CONVERGENCE; msg = "Gradient tolerance reached."; break; }
                if (radius <= 1e-32) { term = CBA_TERM_CONVERGENCE; msg = "Minimum trust region radius reached."; break; }
                ++iter;
                for (int i = 0; i < n; ++i) dlm[i] = lm_diag(sys.diag(i), scale2[i], radius);
                bool valid = sys.solve(dlm, delta);
                double model_change = 0.0;
                if (valid) {
                    sys.apply(delta.data(), Hd.data());
                    double dg = 0.0, dHd = 0.0;
                    for (int i = 0; i < n; ++i) { dg += delta[i] * sys.g[i]; dHd += delta[i] * Hd[i]; if (!std::isfinite(delta[i])) valid = false; }
                    model_change = -dg - 0.5 * dHd;
                    if (!(model_change > 0.0)) valid = false;
                }
                if (!valid) {
                    if (++invalid >= 5) { term = CBA_TERM_FAILURE; msg = "Number of consecutive invalid steps more than max."; break; }
                    radius *= 0.5;
                    continue;
                }
                invalid = 0;
                plus(kappa, poses, sys, delta.data(), ck, cp);
                double cand_cost = std::numeric_limits<double>::max();
                const bool cok = linearise(ck, cp, cs);
                if (cok) cand_cost = cs.cost;
                double sn = 0.0, xn = 0.0;
                for (int k = 0; k < sys.nk; ++k) { const int j = sys.kidx[k]; sn += (kappa[j] - ck[j]) * (kappa[j] - ck[j]); xn += kappa[j] * kappa[j]; }
                if (!o.optimize_skew) xn += kappa[4] * kappa[4];
                for (size_t i = 0; i < poses.size(); ++i) { sn += (poses[i] - cp[i]) * (poses[i] - cp[i]); xn += poses[i] * poses[i]; }
                if (std::sqrt(sn) <= eps * (std::sqrt(xn) + eps)) { term = CBA_TERM_CONVERGENCE; msg = "Parameter tolerance reached."; break; }
                const double cost_change = cost - cand_cost;
                if (std::fabs(cost_change) <= eps * cost) { term = CBA_TERM_CONVERGENCE; msg = "Function tolerance reached."; break; }
                const double rel = cost_change / model_change;
                if (o.verbose) std::printf("[cba semidlt] it %3d cost %.12e cand %.12e rel %.3e radius %.3e\n", iter, cost, cand_cost, rel, radius);
                if (rel > 1e-3 && cok) {
                    kappa = ck; poses = cp; cost = cand_cost; ++successful;
                    std::swap(sys, cs);
                    gmax = grad_max(sys);
                    radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
                    decrease_factor = 2.0;
                } else {
                    radius /= decrease_factor;
                    decrease_factor *= 2.0;
                }
            }
        }
        std::memcpy(kappa5, kappa.data(), sizeof(double) * 5);
        std::memcpy(poses7, poses.data(), sizeof(double) * poses.size());
        out->termination = term; out->success = term == CBA_TERM_CONVERGENCE;
        out->iterations = iter; out->successful_steps = successful;
        out->initial_cost = initial_cost; out->final_cost = cost;
        out->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::snprintf(out->report, sizeof(out->report), "calibba(semi-DLT VP LM, %d views, %lld obs): %s iters=%d cost %.6e -> %.6e", V,
                      static_cast<long long>(ev.n_obs), msg, iter, initial_cost, cost);
    }

    // solve_full (intrinsicssemidlt.cpp:74-90) at the final point: alpha with the requested entries held fixed
    // (distortion.h:296-363), per-view errors and ssr.  Throws runtime_error like the reference when the fit fails.
    void finish(const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const int64_t* view_offset, SemiDltResult& res,
                double* ssr_out) {
        const int V = ev.V, m = ev.m();
        if (ev.n_obs < 8) throw std::runtime_error("Failed to compute distortion parameters");
        std::vector<double> N(static_cast<size_t>(m) * m), rhs(m), alpha(m, 0.0);
        ev.normal(kappa.data(), poses.data(), N.data(), rhs.data());
        std::vector<char> fixed(m, 0);
        for (int i = 0; i < n_fixed; ++i) {
            const int idx = fixed_idx[i];
            if (idx < 0 || idx >= m) throw std::invalid_argument("Fixed distortion index out of range");
            if (!fixed[idx]) { fixed[idx] = 1; alpha[idx] = fixed_val ? fixed_val[i] : 0.0; }  // first occurrence wins after the stable sort + unique
        }
        std::vector<int> fr;
        for (int a = 0; a < m; ++a) if (!fixed[a]) fr.push_back(a);
        const int nf = static_cast<int>(fr.size());
        if (nf > 0) {
            std::vector<double> Nf(static_cast<size_t>(nf) * nf), bf(nf);
            for (int i = 0; i < nf; ++i) {
                double t = rhs[fr[i]];
                for (int a = 0; a < m; ++a) if (fixed[a]) t -= N[static_cast<size_t>(fr[i]) * m + a] * alpha[a];
                bf[i] = t;
                for (int j = 0; j < nf; ++j) Nf[static_cast<size_t>(i) * nf + j] = N[static_cast<size_t>(fr[i]) * m + fr[j]];
            }
            if (!chol_inplace(Nf, nf)) throw std::runtime_error("Failed to compute distortion parameters");
            chol_solve(Nf, nf, bf.data());
            for (int i = 0; i < nf; ++i) alpha[fr[i]] = bf[i];
        }
        std::vector<double> sv(V);
        ev.resid(kappa.data(), poses.data(), alpha.data(), sv.data());
        res.alpha = alpha;
        res.view_errors.resize(V);
        double ssr = 0.0;
        for (int v = 0; v < V; ++v) {
            ssr += sv[v];
            res.view_errors[v] = std::sqrt(sv[v] / (2.0 * static_cast<double>(view_offset[v + 1] - view_offset[v])));
        }
        *ssr_out = ssr;
    }

    // ceres::Covariance over [intr(5), quats(4 each), trans(3 each)] in ambient sizes, scaled by ssr / max(1, 2N - (5 + 7V))
    // (intrinsicssemidlt.cpp:184-188, ceresutils.h:69-126).  Returns false (matrix left empty) if rank deficient.
    bool covariance(double ssr, std::vector<double>& cov) {
        cov.clear();
        if (!sys.ok) return false;
        const int V = ev.V, n = sys.n, nk = sys.nk;
        std::vector<double> H;
        sys.dense(H);
        std::vector<double> L = H;
        if (!chol_inplace(L, n)) return false;
        double cmax = 0.0, dmin = 1e300;
        for (int i = 0; i < n; ++i) { cmax = std::max(cmax, std::sqrt(std::max(0.0, H[static_cast<size_t>(i) * n + i]))); dmin = std::min(dmin, L[static_cast<size_t>(i) * n + i]); }
        const double m_rows = 2.0 * static_cast<double>(ev.n_obs);
        if (dmin <= 20.0 * (m_rows + n) * 2.220446049250313e-16 * cmax) return false;  // SuiteSparseQR default rank tolerance (restated)
        std::vector<double> Sig;
        chol_inverse(L, n, Sig);
        const int dim = 5 + 7 * V;
        const int total_params = dim;
        const double dof = std::max(1.0, m_rows - total_params);
        const double vf = ssr / dof;
        // lift matrix P (dim x n): kappa identity on its free columns; quaternion rows = PlusJacobian; translation identity
        std::vector<double> P(static_cast<size_t>(dim) * n, 0.0);
        for (int i = 0; i < nk; ++i) P[static_cast<size_t>(sys.kidx[i]) * n + i] = 1.0;
        for (int v = 0; v < V; ++v) {
            const double* q = &poses[7 * static_cast<size_t>(v)];
            const double PJ[12] = {-q[1], -q[2], -q[3], q[0], q[3], -q[2], -q[3], q[0], q[1], q[2], -q[1], q[0]};
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 3; ++c) P[static_cast<size_t>(5 + 4 * v + r) * n + nk + 6 * v + c] = PJ[r * 3 + c];
            for (int k = 0; k < 3; ++k) P[static_cast<size_t>(5 + 4 * V + 3 * v + k) * n + nk + 6 * v + 3 + k] = 1.0;
        }
        // cov = P Sig P^T (P has <= 3 non-zeros per row)
        std::vector<double> PS(static_cast<size_t>(dim) * n, 0.0);
        for (int i = 0; i < dim; ++i)
            for (int a = 0; a < n; ++a) {
                const double p = P[static_cast<size_t>(i) * n + a];
                if (p == 0.0) continue;
                for (int b = 0; b < n; ++b) PS[static_cast<size_t>(i) * n + b] += p * Sig[static_cast<size_t>(a) * n + b];
            }
        cov.assign(static_cast<size_t>(dim) * dim, 0.0);
        for (int j = 0; j < dim; ++j)
            for (int b = 0; b < n; ++b) {
                const double p = P[static_cast<size_t>(j) * n + b];
                if (p == 0.0) continue;
                for (int i = 0; i < dim; ++i) cov[static_cast<size_t>(i) * dim + j] += PS[static_cast<size_t>(i) * n + b] * p * vf;
            }
        return true;
    }

    SemiDltSystem sys;

  private:
    std::vector<double> pv_;
};

}  // namespace cba
