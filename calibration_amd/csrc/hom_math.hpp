// hom_math.hpp — optimize_homography (src/estimation/optim/homography.cpp:144-175) as a small_lm problem.
//
// Reference functor HomographyResidual (homography.cpp:103-130): 8 parameters h = [H00 H01 H02 H10 H11 H12 H20 H21],
// H22 = 1; per correspondence  (x, y) -> (u^, v^) = hnormalized(H [x y 1]^T),  r = (u^ - u, v^ - v).
// build_problem (:132-142) adds ONE residual block PER POINT, each with its own HuberLoss(huber_delta): the loss
// weight is per correspondence here (unlike the per-view blocks of the reprojection chains).
// Analytic Jacobian (w = H20 x + H21 y + 1):
//     du^/dh = [x y 1 0 0 0 -u^ x -u^ y] / w        dv^/dh = [0 0 0 x y 1 -v^ x -v^ y] / w
#pragma once
#include "small_lm.hpp"

namespace cba {

struct HomAux {
    double ssr_raw;     // sum |r_i|^2
    double ssr_robust;  // sum rho'(s_i) |r_i|^2  — what Problem::Evaluate returns with apply_loss_function = true
};

struct HomProblem {
    using Aux = HomAux;
    int n;
    const double *X, *Y, *u, *v;
    double huber_delta;

    template <class Coop>
    CBA_HD bool evaluate(Coop& co, const double* h, bool want_jac, double* cost, double* H, double* g, Aux* aux) const {
        // packed upper triangle: the three 3x3-ish groups of the 8x8 normal matrix share the factor (x,y,1)(x,y,1)^T / w^2
        double acc[36 + 8 + 3];
        for (int a = 0; a < 47; ++a) acc[a] = 0.0;
        for (int i = co.lane(); i < n; i += co.width()) {
            const double x = X[i], y = Y[i];
            const double iw = 1.0 / (h[6] * x + h[7] * y + 1.0);
            const double uh = (h[0] * x + h[1] * y + h[2]) * iw, vh = (h[3] * x + h[4] * y + h[5]) * iw;
            const double ru = uh - u[i], rv = vh - v[i];
            const double s = ru * ru + rv * rv;
            double rho, w;
            huber(s, huber_delta, &rho, &w);
            acc[44] += 0.5 * rho;
            acc[45] += s;
            acc[46] += w * s;
            if (!want_jac) continue;
            const double Ju[8] = {x * iw, y * iw, iw, 0.0, 0.0, 0.0, -uh * x * iw, -uh * y * iw};
            const double Jv[8] = {0.0, 0.0, 0.0, x * iw, y * iw, iw, -vh * x * iw, -vh * y * iw};
            int k = 0;
            for (int a = 0; a < 8; ++a) {
                const double wa_u = w * Ju[a], wa_v = w * Jv[a];
                for (int c = a; c < 8; ++c, ++k) acc[k] += wa_u * Ju[c] + wa_v * Jv[c];
                acc[36 + a] += wa_u * ru + wa_v * rv;
            }
        }
        if (want_jac) {
            int k = 0;
            for (int a = 0; a < 8; ++a)
                for (int c = a; c < 8; ++c, ++k) {
                    const double t = co.sum(acc[k]);
                    H[a * 8 + c] = t;
                    H[c * 8 + a] = t;
                }
            for (int a = 0; a < 8; ++a) g[a] = co.sum(acc[36 + a]);
        }
        *cost = co.sum(acc[44]);
        aux->ssr_raw = co.sum(acc[45]);
        aux->ssr_robust = co.sum(acc[46]);
        return true;
    }
};

struct HomResult {
    double h[8];
    double cov[64];
    double initial_cost, final_cost, ssr_raw, ssr_robust;
    int iterations, successful_steps, termination, cov_ok;
};

// The whole refinement of one view; res.h holds the initial parameters on entry and the result on exit.
template <class Coop>
CBA_HD void hom_solve_view(const HomProblem& P, Coop& co, double eps, int max_iterations, bool want_cov, HomResult& res) {
    SmallLMState<8> st;
    for (int k = 0; k < 8; ++k) st.x[k] = res.h[k];
    HomAux aux{0.0, 0.0};
    small_lm_solve<8>(P, co, eps, max_iterations, st, aux);
    for (int k = 0; k < 8; ++k) res.h[k] = st.x[k];
    res.initial_cost = st.initial_cost; res.final_cost = st.cost;
    res.ssr_raw = aux.ssr_raw; res.ssr_robust = aux.ssr_robust;
    res.iterations = st.iterations; res.successful_steps = st.successful_steps; res.termination = st.termination;
    res.cov_ok = 0;
    for (int a = 0; a < 64; ++a) res.cov[a] = 0.0;
    if (want_cov && st.evaluated) {
        // homography.cpp:163-173: ssr from Problem::Evaluate (default EvaluateOptions apply the loss => robustified
        // residuals), n_res = 2N, scaled by ssr / max(1, 2N - 8) (ceresutils.h:117-123)
        const long long n_res = 2LL * P.n;
        const long long dof = n_res - 8 > 1 ? n_res - 8 : 1;
        res.cov_ok = small_covariance<8>(st.H, n_res, aux.ssr_robust / static_cast<double>(dof), res.cov) ? 1 : 0;
    }
}

}  // namespace cba
