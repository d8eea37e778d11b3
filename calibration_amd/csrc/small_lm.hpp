// small_lm.hpp — Levenberg-Marquardt for small dense problems (NP <= 8 unknowns, thousands of observations),
// solved ENTIRELY by one cooperative group: on the GPU one 64-lane wavefront per problem (lanes stride over the
// observations with unit-stride loads, sums cross the wave in DPP, every lane then takes the same tiny
// Cholesky step redundantly — wave-uniform control flow, no LDS, no host round trips); in the CPU test build a
// single thread (tests/cpu_backend).  A batch of independent problems (views) is one launch.
//
// Used by: optimize_homography (hom_math.hpp) and optimize_planar_pose (vp_math.hpp).
//
// Solver semantics: the same restated Ceres trust-region rules as lm_core.hpp (Ceres is a
// third-party dependency outside /root/reference; call site src/estimation/detail/ceresutils.h:27-43):
// unconstrained Euclidean block, Jacobi scaling fixed at x0, D^2 = clamp(diag)/radius, gain-ratio acceptance
// at 1e-3, radius / max(1/3, 1-(2 rho-1)^3) on success and /2, /4, ... on failure, parameter / function /
// gradient tolerance = epsilon, 5 consecutive invalid steps = FAILURE.
#pragma once
#include "../../include/calibba.h"
#include "schur_math.hpp"
#if defined(__HIPCC__)
#include "wave_reduce.hpp"
#endif

namespace cba {

// ---- cooperative groups --------------------------------------------------------------------------------------
struct SerialCoop {
    CBA_HD int lane() const { return 0; }
    CBA_HD int width() const { return 1; }
    CBA_HD double sum(double v) const { return v; }
    CBA_HD double max(double v) const { return v; }
};

#if defined(__HIPCC__)
// One wavefront.  sum() returns the wave total IN EVERY LANE (fixed DPP order, then a readlane broadcast from
// lane 63), so all data-dependent branches of the solver stay wave-uniform.
struct WaveCoop {
    __device__ __forceinline__ int lane() const { return static_cast<int>(threadIdx.x) & 63; }
    __device__ __forceinline__ int width() const { return 64; }
    __device__ __forceinline__ double sum(double v) const {
        v = wave_sum63(v);
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
        return __hiloint2double(hi, lo);
    }
};
#endif

// ---- NP x NP dense helpers (row-major full storage) ---------------------------------------------------------
template <int NP>
CBA_HD bool chol_n(double* A) {
    for (int j = 0; j < NP; ++j) {
        double d = A[j * NP + j];
        for (int k = 0; k < j; ++k) d -= A[j * NP + k] * A[j * NP + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        A[j * NP + j] = d;
        for (int i = j + 1; i < NP; ++i) {
            double s = A[i * NP + j];
            for (int k = 0; k < j; ++k) s -= A[i * NP + k] * A[j * NP + k];
            A[i * NP + j] = s / d;
        }
    }
    return true;
}
template <int NP>
CBA_HD void chol_solve_n(const double* L, double* b) {  // b <- (L L^T)^-1 b
    for (int i = 0; i < NP; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * NP + k] * b[k];
        b[i] = s / L[i * NP + i];
    }
    for (int i = NP - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < NP; ++k) s -= L[k * NP + i] * b[k];
        b[i] = s / L[i * NP + i];
    }
}

template <int NP>
struct SmallLMState {
    double x[NP];
    double H[NP * NP];  // robustified J~^T J~ at x (full symmetric storage)
    double g[NP];       // J~^T r~
    double cost, initial_cost;
    int iterations, successful_steps, termination;
    int evaluated;  // 0 when the functor failed at x0 (H, g undefined)
};

// Problem concept:
//   struct Aux;                                     side results of an evaluation (adopted with an accepted step)
//   bool evaluate(Coop&, const double* x, bool want_jac, double* cost, double* H, double* g, Aux* aux) const
//     cost = 1/2 sum rho(.), H / g already loss-corrected; every lane must receive identical values; returns
//     false when the reference's functor would fail to evaluate.
template <int NP, class Problem, class Coop>
CBA_HD void small_lm_solve(const Problem& P, Coop& co, double eps, int max_iterations, SmallLMState<NP>& st,
                           typename Problem::Aux& aux) {
    double cand[NP], delta[NP], scale2[NP], A[NP * NP];
    st.iterations = 0; st.successful_steps = 0; st.termination = CBA_TERM_FAILURE;
    st.cost = st.initial_cost = 0.0;
    st.evaluated = 0;
    if (!P.evaluate(co, st.x, true, &st.cost, st.H, st.g, &aux)) return;  // evaluation failure at x0: Ceres reports FAILURE
    {   // non-finite cost or gradient at x0 (NaN / Inf observations): Ceres rejects the initial evaluation -> FAILURE
        bool finite = st.cost == st.cost && fabs(st.cost) <= 1.7976931348623157e308;
        for (int i = 0; i < NP; ++i) finite = finite && st.g[i] == st.g[i] && fabs(st.g[i]) <= 1.7976931348623157e308;
        if (!finite) return;
    }
    st.evaluated = 1;
    st.initial_cost = st.cost;
    for (int i = 0; i < NP; ++i) { const double sc = 1.0 / (1.0 + sqrt(st.H[i * NP + i])); scale2[i] = sc * sc; }
    double gmax = 0.0;
    for (int i = 0; i < NP; ++i) gmax = fmax(gmax, fabs(st.g[i]));
    double radius = 1e4, decrease_factor = 2.0;
    int iter = 0, invalid = 0, term = CBA_TERM_FAILURE;
    if (gmax <= eps) term = CBA_TERM_CONVERGENCE;
    else while (true) {
        if (iter >= max_iterations) { term = CBA_TERM_NO_CONVERGENCE; break; }
        if (gmax <= eps) { term = CBA_TERM_CONVERGENCE; break; }
        if (radius <= 1e-32) { term = CBA_TERM_CONVERGENCE; break; }
        ++iter;
        for (int a = 0; a < NP * NP; ++a) A[a] = st.H[a];
        for (int i = 0; i < NP; ++i) A[i * NP + i] += lm_diag(st.H[i * NP + i], scale2[i], radius);
        bool valid = chol_n<NP>(A);
        double model_change = 0.0;
        if (valid) {
            for (int i = 0; i < NP; ++i) delta[i] = -st.g[i];
            chol_solve_n<NP>(A, delta);
            double dg = 0.0, dHd = 0.0;
            for (int i = 0; i < NP; ++i) {
                dg += delta[i] * st.g[i];
                double t = 0.0;
                for (int j = 0; j < NP; ++j) t += st.H[i * NP + j] * delta[j];
                dHd += delta[i] * t;
                if (!(delta[i] == delta[i]) || fabs(delta[i]) > 1e300) valid = false;
            }
            model_change = -dg - 0.5 * dHd;
            if (!(model_change > 0.0)) valid = false;
        }
        if (!valid) {
            if (++invalid >= 5) { term = CBA_TERM_FAILURE; break; }
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        for (int k = 0; k < NP; ++k) cand[k] = st.x[k] + delta[k];
        // one evaluation gives the candidate cost and, if the step is accepted, its linearisation
        double cH[NP * NP], cg[NP], cand_cost = 1.7976931348623157e308, cc = 0.0;
        typename Problem::Aux caux;
        if (P.evaluate(co, cand, true, &cc, cH, cg, &caux) && cc == cc && cc < 1.7976931348623157e308) cand_cost = cc;
        double sn = 0.0, xn = 0.0;
        for (int k = 0; k < NP; ++k) { sn += delta[k] * delta[k]; xn += st.x[k] * st.x[k]; }
        if (sqrt(sn) <= eps * (sqrt(xn) + eps)) { term = CBA_TERM_CONVERGENCE; break; }
        const double cost_change = st.cost - cand_cost;
        if (fabs(cost_change) <= eps * st.cost) { term = CBA_TERM_CONVERGENCE; break; }
        const double rel = cost_change / model_change;
        if (rel > 1e-3) {
            for (int k = 0; k < NP; ++k) { st.x[k] = cand[k]; st.g[k] = cg[k]; }
            for (int a = 0; a < NP * NP; ++a) st.H[a] = cH[a];
            st.cost = cand_cost;
            aux = caux;
            ++st.successful_steps;
            gmax = 0.0;
            for (int i = 0; i < NP; ++i) gmax = fmax(gmax, fabs(st.g[i]));
            const double t = 2.0 * rel - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            if (radius > 1e16) radius = 1e16;
            decrease_factor = 2.0;
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.0;
        }
    }
    st.iterations = iter;
    st.termination = term;
}

// (J~^T J~)^-1 * variance_factor (ceresutils.h:69-126 for a single Euclidean block).  n_res = number of scalar
// residuals (for SuiteSparseQR's default rank tolerance 20 (m+n) eps max|J_j|, third-party, restated).
// Returns false (cov zeroed) when rank deficient.
template <int NP>
CBA_HD bool small_covariance(const double* H, long long n_res, double variance_factor, double* cov) {
    double L[NP * NP];
    for (int a = 0; a < NP * NP; ++a) { L[a] = H[a]; cov[a] = 0.0; }
    if (!chol_n<NP>(L)) return false;
    double cmax = 0.0, dmin = 1e300;
    for (int i = 0; i < NP; ++i) { cmax = fmax(cmax, sqrt(H[i * NP + i])); dmin = fmin(dmin, L[i * NP + i]); }
    if (dmin <= 20.0 * static_cast<double>(n_res + NP) * 2.220446049250313e-16 * cmax) return false;
    for (int c = 0; c < NP; ++c) {
        double e[NP];
        for (int r = 0; r < NP; ++r) e[r] = 0.0;
        e[c] = 1.0;
        chol_solve_n<NP>(L, e);
        for (int r = 0; r < NP; ++r) cov[r * NP + c] = e[r] * variance_factor;
    }
    return true;
}

}  // namespace cba
