// oracle/lm.hpp — TEST INFRASTRUCTURE ONLY.
//
// Dense restatement of what the reference asks Ceres to do in
// src/estimation/detail/ceresutils.h:27-43 (solve_problem) and :69-126
// (compute_covariance): Levenberg-Marquardt trust region, per-residual-block
// Huber loss, QuaternionManifold / SubsetManifold / constant blocks / lower
// bounds, termination rules, and tangent->ambient covariance lifting.
//
// Ceres Solver is a third-party dependency that is NOT in /root/reference
// (cmake/Dependencies.cmake:3, version unpinned; CI uses libceres-dev 2.2).
// The algorithm below restates Ceres 2.x's published trust-region minimiser
// (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc, corrector.cc,
// loss_function.cc) with its documented defaults.  PARITY WITH CERES' OWN
// ITERATION NUMERICS IS UNPINNED: the reference stores no Ceres outputs; what
// pins this file are the reference's ground-truth-recovery tests
// (tests/unit/*_test.cpp tolerances, see tests/test_oracle_kat.py).
//
// On bounds-constrained problems (fx, fy >= 0 make every problem with variable
// intrinsics "constrained") Ceres runs a projected Armijo line search on every
// trust-region step before it evaluates the candidate: line_search.hpp.
//
// This is a *dense* solver (forms the full tangent-space J^T J), usable up to
// a few thousand tangent dimensions.  It deliberately shares no code with the
// product's Schur-complement solver.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "line_search.hpp"
#include "models.hpp"

namespace orc {

enum BlockKind { BLK_EUCLID = 0, BLK_QUAT = 1 };

struct ParamBlock {
    double* x = nullptr;
    int size = 0;
    BlockKind kind = BLK_EUCLID;
    bool constant = false;
    int subset_const = -1;          // ceres::SubsetManifold({idx}); -1 = none
    std::vector<int> lower_idx;     // indices with lower bound
    std::vector<double> lower_val;
    std::vector<int> upper_idx;     // indices with upper bound (SetParameterUpperBound)
    std::vector<double> upper_val;
    int toff = -1;                  // tangent offset (-1 if constant)
    int aoff = -1;                  // offset in the packed ambient state vector
    int tsize() const {
        if (constant) return 0;
        if (kind == BLK_QUAT) return 3;
        return subset_const >= 0 ? size - 1 : size;
    }
};

// A residual block evaluates r (nres) and ambient Jacobians (row-major
// nres x size per parameter block), like ceres::CostFunction::Evaluate.
struct ResidualBlock {
    int nres = 0;
    std::vector<int> pb;
    virtual ~ResidualBlock() = default;
    virtual void evaluate(const double* const* x, double* r, double** J) const = 0;
};

struct LMOptions {
    double huber_delta = 1.0;   // <= 0: no loss (intrinsics.cpp:70-71 etc.)
    double epsilon = 1e-9;      // function/gradient/parameter tolerance, ceresutils.h:32-34
    int max_iterations = 1000;  // optimize.h:26
    int verbose = 0;
    int num_threads = 1;
    int line_search = 1;        // the projected Armijo search of bounds-constrained problems (line_search.hpp); 0 = off (A/B tests)
};

enum Termination { TERM_CONVERGENCE = 0, TERM_NO_CONVERGENCE = 1, TERM_FAILURE = 2 };

struct LMSummary {
    int termination = TERM_FAILURE;
    int iterations = 0;
    int successful_steps = 0;
    double initial_cost = 0, final_cost = 0;
    int line_search_steps = 0;        // trust-region steps the line search shortened
    int line_search_evaluations = 0;  // cost + gradient evaluations spent in line searches
    char message[160] = {0};
};

inline bool cholesky_inplace(std::vector<double>& A, int n) {
    // lower Cholesky, row-major, in place; returns false if not PD
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return true;
}
inline void cholesky_solve(const std::vector<double>& L, int n, std::vector<double>& b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

class Problem {
  public:
    std::vector<ParamBlock> params;
    std::vector<std::unique_ptr<ResidualBlock>> residuals;

    int add_param(double* x, int size, BlockKind kind = BLK_EUCLID) {
        ParamBlock p; p.x = x; p.size = size; p.kind = kind;
        params.push_back(p);
        return static_cast<int>(params.size()) - 1;
    }
    void set_lower(int id, int idx, double v) {
        params[id].lower_idx.push_back(idx);
        params[id].lower_val.push_back(v);
    }
    void set_upper(int id, int idx, double v) {
        params[id].upper_idx.push_back(idx);
        params[id].upper_val.push_back(v);
    }

    // ---- layout -----------------------------------------------------------
    int ntan = 0, namb = 0;
    bool constrained = false;
    void finalize_layout() {
        ntan = 0; namb = 0; constrained = false;
        // Ceres drops parameter blocks that are constant or unused by any
        // residual from the reduced program.
        std::vector<char> used(params.size(), 0);
        for (auto& rb : residuals) for (int id : rb->pb) used[id] = 1;
        for (size_t i = 0; i < params.size(); ++i) {
            auto& p = params[i];
            if (p.constant || !used[i]) { p.toff = -1; p.aoff = -1; continue; }
            p.toff = ntan; ntan += p.tsize();
            p.aoff = namb; namb += p.size;
            if (!p.lower_idx.empty() || !p.upper_idx.empty()) constrained = true;
        }
    }

    // tangent Jacobian of one parameter block from its ambient Jacobian
    static void to_tangent(const ParamBlock& p, const double* x, const double* Ja, int nres, double* Jt) {
        const int ts = p.tsize();
        if (p.kind == BLK_QUAT) {
            double PJ[12];
            quat_plus_jacobian(x, PJ);
            for (int r = 0; r < nres; ++r)
                for (int c = 0; c < 3; ++c) {
                    double s = 0;
                    for (int k = 0; k < 4; ++k) s += Ja[r * 4 + k] * PJ[k * 3 + c];
                    Jt[r * ts + c] = s;
                }
        } else {
            for (int r = 0; r < nres; ++r) {
                int c = 0;
                for (int k = 0; k < p.size; ++k) {
                    if (k == p.subset_const) continue;
                    Jt[r * ts + c++] = Ja[r * p.size + k];
                }
            }
        }
    }

    // Plus with bounds projection (ceres ParameterBlock::Plus)
    void plus(const std::vector<double>& x, const std::vector<double>& delta, std::vector<double>& out) const {
        out = x;
        for (const auto& p : params) {
            if (p.toff < 0) continue;
            const double* xs = &x[p.aoff];
            const double* d = &delta[p.toff];
            double* o = &out[p.aoff];
            if (p.kind == BLK_QUAT) {
                quat_plus(xs, d, o);
            } else {
                int c = 0;
                for (int k = 0; k < p.size; ++k) {
                    if (k == p.subset_const) { o[k] = xs[k]; continue; }
                    o[k] = xs[k] + d[c++];
                }
            }
            for (size_t b = 0; b < p.lower_idx.size(); ++b)
                o[p.lower_idx[b]] = std::max(o[p.lower_idx[b]], p.lower_val[b]);
            for (size_t b = 0; b < p.upper_idx.size(); ++b)
                o[p.upper_idx[b]] = std::min(o[p.upper_idx[b]], p.upper_val[b]);
        }
    }

    void gather(std::vector<double>& x) const {
        x.assign(namb, 0.0);
        for (const auto& p : params)
            if (p.aoff >= 0) std::memcpy(&x[p.aoff], p.x, sizeof(double) * p.size);
    }
    void scatter(const std::vector<double>& x) {
        for (auto& p : params)
            if (p.aoff >= 0) std::memcpy(p.x, &x[p.aoff], sizeof(double) * p.size);
    }

    // ---- evaluation -------------------------------------------------------
    // cost = 1/2 sum_blocks rho(|r_b|^2); optionally H = J~^T J~, g = J~^T r~
    // with the Huber corrector applied per block (corrector.cc: rho'' <= 0 =>
    // both r and J scaled by sqrt(rho')).
    struct Accum {
        double cost = 0;
        std::vector<double> H, g;
    };
    void eval_range(const std::vector<double>& x, size_t b0, size_t b1, double huber, bool want_jac, Accum& acc) const {
        std::vector<double> r, Jbuf, Jt;
        std::vector<const double*> xp;
        std::vector<double*> Jp;
        std::vector<int> toffs, tsizes;
        for (size_t b = b0; b < b1; ++b) {
            const ResidualBlock& rb = *residuals[b];
            const int nres = rb.nres;
            r.assign(nres, 0.0);
            xp.clear(); Jp.clear();
            size_t jtot = 0;
            for (int id : rb.pb) jtot += static_cast<size_t>(nres) * params[id].size;
            if (want_jac) Jbuf.assign(jtot, 0.0);
            size_t off = 0;
            for (int id : rb.pb) {
                const ParamBlock& p = params[id];
                xp.push_back(p.aoff >= 0 ? &x[p.aoff] : p.x);
                Jp.push_back(want_jac ? &Jbuf[off] : nullptr);
                off += static_cast<size_t>(nres) * p.size;
            }
            rb.evaluate(xp.data(), r.data(), want_jac ? Jp.data() : nullptr);
            double s = 0;
            for (int i = 0; i < nres; ++i) s += r[i] * r[i];
            double rho0 = s, rho1 = 1.0;
            if (huber > 0) {  // ceres::HuberLoss::Evaluate
                const double b2 = huber * huber;
                if (s > b2) {
                    const double rt = std::sqrt(s);
                    rho0 = 2.0 * huber * rt - b2;
                    rho1 = std::max(std::numeric_limits<double>::min(), huber / rt);
                }
            }
            acc.cost += 0.5 * rho0;
            if (!want_jac) continue;
            const double sq = std::sqrt(rho1);
            // tangent J of every non-constant block, concatenated
            toffs.clear(); tsizes.clear();
            int tcols = 0;
            for (int id : rb.pb) tcols += params[id].tsize();
            Jt.assign(static_cast<size_t>(nres) * std::max(tcols, 1), 0.0);
            std::vector<double> Jblk;
            int col = 0;
            for (size_t k = 0; k < rb.pb.size(); ++k) {
                const ParamBlock& p = params[rb.pb[k]];
                const int ts = p.tsize();
                if (ts == 0 || p.toff < 0) continue;
                Jblk.assign(static_cast<size_t>(nres) * ts, 0.0);
                to_tangent(p, xp[k], Jp[k], nres, Jblk.data());
                for (int rr = 0; rr < nres; ++rr)
                    for (int c = 0; c < ts; ++c) Jt[static_cast<size_t>(rr) * tcols + col + c] = sq * Jblk[rr * ts + c];
                toffs.push_back(p.toff); tsizes.push_back(ts);
                col += ts;
            }
            // global column index of each local column
            std::vector<int> gcol;
            for (size_t k = 0; k < toffs.size(); ++k)
                for (int c = 0; c < tsizes[k]; ++c) gcol.push_back(toffs[k] + c);
            const int nc = static_cast<int>(gcol.size());
            for (int rr = 0; rr < nres; ++rr) {
                const double* row = &Jt[static_cast<size_t>(rr) * tcols];
                const double rs = sq * r[rr];
                for (int a = 0; a < nc; ++a) {
                    const double ja = row[a];
                    if (ja == 0.0) continue;
                    acc.g[gcol[a]] += ja * rs;
                    double* Hrow = &acc.H[static_cast<size_t>(gcol[a]) * ntan];
                    for (int c = 0; c < nc; ++c) Hrow[gcol[c]] += ja * row[c];
                }
            }
        }
    }
    void evaluate(const std::vector<double>& x, const LMOptions& o, bool want_jac, double* cost,
                  std::vector<double>* H, std::vector<double>* g) const {
        const int nt = std::max(1, std::min<int>(o.num_threads, static_cast<int>(residuals.size())));
        std::vector<Accum> accs(nt);
        for (auto& a : accs)
            if (want_jac) { a.H.assign(static_cast<size_t>(ntan) * ntan, 0.0); a.g.assign(ntan, 0.0); }
        const size_t nb = residuals.size();
        if (nt == 1) {
            eval_range(x, 0, nb, o.huber_delta, want_jac, accs[0]);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) {
                const size_t b0 = nb * t / nt, b1 = nb * (t + 1) / nt;
                th.emplace_back([&, t, b0, b1] { eval_range(x, b0, b1, o.huber_delta, want_jac, accs[t]); });
            }
            for (auto& t : th) t.join();
        }
        *cost = 0;
        for (auto& a : accs) *cost += a.cost;
        if (want_jac) {
            *H = accs[0].H; *g = accs[0].g;
            for (int t = 1; t < nt; ++t) {
                for (size_t i = 0; i < H->size(); ++i) (*H)[i] += accs[t].H[i];
                for (int i = 0; i < ntan; ++i) (*g)[i] += accs[t].g[i];
            }
        }
    }

    // ---- Levenberg-Marquardt (Ceres TrustRegionMinimizer + LM strategy) ----
    void solve(const LMOptions& o, LMSummary* sum) {
        finalize_layout();
        const int n = ntan;
        std::vector<double> x, cand, H, g, delta(n), scale(n, 1.0);
        gather(x);
        {   // Ceres projects the start point onto the feasible set
            std::vector<double> zero(n, 0.0), x0;
            plus(x, zero, x0);
            x = x0;
        }
        double cost = 0;
        evaluate(x, o, true, &cost, &H, &g);
        sum->initial_cost = cost;
        // jacobi_scaling = true: 1 / (1 + ||J_col||), computed once at x0
        for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H[static_cast<size_t>(i) * n + i]));

        auto grad_max_norm = [&]() {
            double m = 0;
            if (!constrained) {
                for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(g[i]));
            } else {
                std::vector<double> ng(n), xp;
                for (int i = 0; i < n; ++i) ng[i] = -g[i];
                plus(x, ng, xp);
                for (int i = 0; i < namb; ++i) m = std::max(m, std::fabs(xp[i] - x[i]));
            }
            return m;
        };
        auto norm2 = [](const std::vector<double>& v) {
            double s = 0; for (double e : v) s += e * e; return std::sqrt(s);
        };

        double radius = 1e4, decrease_factor = 2.0;
        const double min_radius = 1e-32, max_radius = 1e16;
        const double min_diag = 1e-6, max_diag = 1e32, min_rel_decrease = 1e-3;
        int iter = 0, invalid = 0;
        double gmax = grad_max_norm();
        auto finish = [&](int term, const char* msg) {
            sum->termination = term; sum->iterations = iter; sum->final_cost = cost;
            std::snprintf(sum->message, sizeof(sum->message), "%s", msg);
            scatter(x);
        };
        if (n == 0) { finish(TERM_CONVERGENCE, "no free parameters"); return; }
        if (gmax <= o.epsilon) { finish(TERM_CONVERGENCE, "Gradient tolerance reached."); return; }

        std::vector<double> A;
        while (true) {
            if (iter >= o.max_iterations) { finish(TERM_NO_CONVERGENCE, "Maximum number of iterations reached."); return; }
            if (gmax <= o.epsilon) { finish(TERM_CONVERGENCE, "Gradient tolerance reached."); return; }
            if (radius <= min_radius) { finish(TERM_CONVERGENCE, "Minimum trust region radius reached."); return; }
            ++iter;
            // (J^T J + D^T D) step = -g, D^2 = clamp(diag(Js^T Js))/radius in scaled coordinates
            A = H;
            for (int i = 0; i < n; ++i) {
                const double s2 = scale[i] * scale[i];
                const double ds = std::min(std::max(H[static_cast<size_t>(i) * n + i] * s2, min_diag), max_diag);
                A[static_cast<size_t>(i) * n + i] += ds / radius / s2;
            }
            bool valid = cholesky_inplace(A, n);
            double model_change = 0;
            if (valid) {
                for (int i = 0; i < n; ++i) delta[i] = -g[i];
                cholesky_solve(A, n, delta);
                // model_cost_change = -(J d)^T (r + J d / 2) = -d^T g - d^T H d / 2
                double dg = 0, dHd = 0;
                for (int i = 0; i < n; ++i) {
                    dg += delta[i] * g[i];
                    double s = 0;
                    const double* Hr = &H[static_cast<size_t>(i) * n];
                    for (int j = 0; j < n; ++j) s += Hr[j] * delta[j];
                    dHd += delta[i] * s;
                    if (!std::isfinite(delta[i])) valid = false;
                }
                model_change = -dg - 0.5 * dHd;
                if (!(model_change > 0.0)) valid = false;
            }
            if (!valid) {
                if (++invalid >= 5) { finish(TERM_FAILURE, "Number of consecutive invalid steps more than max."); return; }
                radius *= 0.5;
                continue;
            }
            invalid = 0;
            if (constrained && o.line_search) {
                // TrustRegionMinimizer::DoLineSearch: search along delta from x with the projected Plus; on success delta is scaled
                // by the step size found (1 whenever the full step already satisfies the Armijo condition).  model_change keeps the
                // value of the FULL step, as in Ceres.
                double dg0 = 0, dinf = 0;
                for (int i = 0; i < n; ++i) { dg0 += g[i] * delta[i]; dinf = std::max(dinf, std::fabs(delta[i])); }
                const std::vector<double> dir = delta;
                auto eval_at = [&](double a, bool want_gradient) {
                    LsSample sm;
                    sm.x = a;
                    std::vector<double> sd(n), xa, Ha, ga;
                    for (int i = 0; i < n; ++i) sd[i] = a * dir[i];
                    plus(x, sd, xa);
                    double ca = 0;
                    evaluate(xa, o, want_gradient, &ca, want_gradient ? &Ha : nullptr, want_gradient ? &ga : nullptr);
                    if (!std::isfinite(ca)) return sm;
                    if (!want_gradient) { sm.value = ca; sm.value_valid = true; return sm; }
                    sm.value = ca; sm.value_valid = true;
                    double dgr = 0;
                    for (int i = 0; i < n; ++i) dgr += dir[i] * ga[i];
                    if (std::isfinite(dgr)) { sm.gradient = dgr; sm.gradient_valid = true; }
                    return sm;
                };
                int ne = 0;
                const double a = armijo_search(cost, dg0, dinf, eval_at, &ne);
                sum->line_search_evaluations += ne;
                if (a > 0.0 && a != 1.0) {
                    ++sum->line_search_steps;
                    for (int i = 0; i < n; ++i) delta[i] *= a;
                }
            }
            plus(x, delta, cand);
            double cand_cost = 0;
            evaluate(cand, o, false, &cand_cost, nullptr, nullptr);
            if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
            // parameter tolerance
            double sn = 0;
            for (int i = 0; i < namb; ++i) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
            sn = std::sqrt(sn);
            const double xn = norm2(x);
            if (sn <= o.epsilon * (xn + o.epsilon)) { finish(TERM_CONVERGENCE, "Parameter tolerance reached."); return; }
            // function tolerance
            const double cost_change = cost - cand_cost;
            if (std::fabs(cost_change) <= o.epsilon * cost) { finish(TERM_CONVERGENCE, "Function tolerance reached."); return; }
            const double rel = cost_change / model_change;
            if (o.verbose)
                std::printf("[orc] it %3d cost %.12e cand %.12e rel %.3e radius %.3e |g| %.3e\n", iter, cost,
                            cand_cost, rel, radius, gmax);
            if (rel > min_rel_decrease) {
                x = cand; cost = cand_cost;
                ++sum->successful_steps;
                double c2;
                evaluate(x, o, true, &c2, &H, &g);
                gmax = grad_max_norm();
                radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
                radius = std::min(max_radius, radius);
                decrease_factor = 2.0;
            } else {
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
            }
        }
    }

    // ---- covariance (ceresutils.h:69-126) ----------------------------------
    // Dense symmetric matrix over `order` (a list of parameter-block ids) in
    // AMBIENT sizes; constant blocks give zero rows/cols.  Returns false if
    // J^T J is rank deficient (ceres::Covariance::Compute fails => nullopt).
    bool covariance(const LMOptions& o, const std::vector<int>& order, std::vector<double>* cov, int* dim) {
        finalize_layout();
        {   // ceres::Covariance keeps every NON-CONSTANT parameter block it is asked about: one that no residual block
            // uses contributes all-zero Jacobian columns, i.e. a rank-deficient Jacobian, and Compute() fails.
            std::vector<char> used(params.size(), 0);
            for (auto& rb : residuals) for (int id : rb->pb) used[id] = 1;
            for (int id : order)
                if (!params[id].constant && !used[id]) return false;
        }
        std::vector<double> x, H, g;
        gather(x);
        double cost;
        evaluate(x, o, true, &cost, &H, &g);
        const int n = ntan;
        std::vector<double> L = H;
        if (n > 0 && !cholesky_inplace(L, n)) return false;
        // ceres::Covariance defaults to SPARSE_QR; SuiteSparseQR declares a column dependent when its
        // R diagonal is below 20 (m + n) eps max_j |J_j| (third-party default, restated).
        {
            size_t m = 0;
            for (auto& rb : residuals) m += static_cast<size_t>(rb->nres);
            double cmax = 0, dmin = 1e300;
            for (int i = 0; i < n; ++i) {
                cmax = std::max(cmax, std::sqrt(H[static_cast<size_t>(i) * n + i]));
                dmin = std::min(dmin, L[static_cast<size_t>(i) * n + i]);
            }
            if (n > 0 && dmin <= 20.0 * static_cast<double>(m + n) * 2.220446049250313e-16 * cmax) return false;
        }
        std::vector<double> Sig(static_cast<size_t>(n) * n, 0.0), e(n);
        for (int c = 0; c < n; ++c) {
            std::fill(e.begin(), e.end(), 0.0); e[c] = 1.0;
            cholesky_solve(L, n, e);
            for (int r = 0; r < n; ++r) Sig[static_cast<size_t>(r) * n + c] = e[r];
        }
        int tot = 0;
        std::vector<int> offs;
        for (int id : order) { offs.push_back(tot); tot += params[id].size; }
        *dim = tot;
        cov->assign(static_cast<size_t>(tot) * tot, 0.0);
        // lift: Sigma_amb(i,j) = P_i Sigma_tan(i,j) P_j^T
        auto lift = [&](const ParamBlock& p, std::vector<double>& P) {
            const int ts = p.tsize();
            P.assign(static_cast<size_t>(p.size) * std::max(ts, 1), 0.0);
            if (p.kind == BLK_QUAT) {
                double PJ[12]; quat_plus_jacobian(p.x, PJ);
                for (int i = 0; i < 12; ++i) P[i] = PJ[i];
            } else {
                int c = 0;
                for (int k = 0; k < p.size; ++k) { if (k == p.subset_const) continue; P[static_cast<size_t>(k) * ts + c] = 1.0; ++c; }
            }
        };
        std::vector<double> Pi, Pj;
        for (size_t a = 0; a < order.size(); ++a) {
            const ParamBlock& pa = params[order[a]];
            if (pa.toff < 0) continue;
            lift(pa, Pi);
            const int ta = pa.tsize();
            for (size_t b = 0; b < order.size(); ++b) {
                const ParamBlock& pb = params[order[b]];
                if (pb.toff < 0) continue;
                lift(pb, Pj);
                const int tb = pb.tsize();
                for (int i = 0; i < pa.size; ++i)
                    for (int j = 0; j < pb.size; ++j) {
                        double s = 0;
                        for (int k = 0; k < ta; ++k)
                            for (int l = 0; l < tb; ++l)
                                s += Pi[static_cast<size_t>(i) * ta + k] * Sig[static_cast<size_t>(pa.toff + k) * n + pb.toff + l] * Pj[static_cast<size_t>(j) * tb + l];
                        (*cov)[static_cast<size_t>(offs[a] + i) * tot + offs[b] + j] = s;
                    }
            }
        }
        return true;
    }
};

}  // namespace orc
