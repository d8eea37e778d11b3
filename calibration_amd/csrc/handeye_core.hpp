// handeye_core.hpp — host LM driver of the AX = XB refinement (optimize_handeye,
// src/estimation/optim/handeye.cpp:45-78): one quaternion block + one translation block, 6 tangent
// unknowns, a per-pair Huber loss.  Same restated Ceres trust-region semantics as lm_core.hpp
// (unconstrained case).  The O(#pairs) evaluation is the Backend's (HIP kernel k_axxb).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

#include "../../include/calibba.h"
#include "axxb_math.hpp"
#include "dense.hpp"

namespace cba {

// Multi-GPU AX = XB: rank r evaluates the pairs (i, j > i) whose FIRST pose i lies in [i0, i1); the ranges
// tile [0, n - 1) and are balanced by pair count (first pose i has n - 1 - i partners).
inline void axxb_rank_range(int n, int n_ranks, int rank, int* i0, int* i1) {
    const long long total = static_cast<long long>(n) * (n - 1) / 2;
    auto first_with = [&](long long want) {  // smallest k with #pairs(first pose < k) >= want
        int k = 0;
        long long c = 0;
        while (k < n - 1 && c < want) { c += n - 1 - k; ++k; }
        return k;
    };
    *i0 = first_with(total * rank / n_ranks);
    *i1 = rank == n_ranks - 1 ? std::max(0, n - 1) : first_with(total * (rank + 1) / n_ranks);
}

struct AxxbEval {
    virtual ~AxxbEval() = default;
    // acc = [H upper 21 | g 6 | cost | #pairs] at pose7
    virtual void eval(const double* pose7, double huber_delta, double* acc) = 0;
};

inline void axxb_unpack(const double* acc, std::vector<double>& H, std::vector<double>& g) {
    H.assign(36, 0.0); g.assign(6, 0.0);
    int e = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { H[a * 6 + b] = acc[e]; H[b * 6 + a] = acc[e]; ++e; }
    for (int a = 0; a < 6; ++a) g[a] = acc[21 + a];
}

inline void handeye_lm(AxxbEval& ev, double* pose7, const cba_options& o, cba_summary* out, double* cov77) {
    const auto t0 = std::chrono::steady_clock::now();
    const double eps = o.epsilon;
    double acc[AXXB_NACC];
    std::vector<double> x(pose7, pose7 + 7), cand(7), H, g, A, delta(6), scale2(6);
    ev.eval(x.data(), o.huber_delta, acc);
    if (acc[28] < 0.5)  // handeyedlt.cpp:76-79
        throw std::runtime_error("No valid motion pairs after filtering. Increase motion or relax thresholds.");
    double cost = acc[27];
    const double initial_cost = cost;
    bool finite0 = std::isfinite(cost);
    for (int e = 0; e < 27; ++e) finite0 = finite0 && std::isfinite(acc[e]);
    axxb_unpack(acc, H, g);
    for (int i = 0; i < 6; ++i) { const double s = 1.0 / (1.0 + std::sqrt(H[i * 6 + i])); scale2[i] = s * s; }
    auto gmax_of = [&]() { double m = 0; for (int i = 0; i < 6; ++i) m = std::max(m, std::fabs(g[i])); return m; };
    double gmax = gmax_of(), radius = 1e4, decrease_factor = 2.0;
    int iter = 0, invalid = 0, successful = 0, term = CBA_TERM_FAILURE;
    const char* msg = "";
    if (!finite0) { term = CBA_TERM_FAILURE; msg = "Residual and Jacobian evaluation failed (non-finite values)."; }
    else if (gmax <= eps) { term = CBA_TERM_CONVERGENCE; msg = "Gradient tolerance reached."; }
    else while (true) {
        if (iter >= o.max_iterations) { term = CBA_TERM_NO_CONVERGENCE; msg = "Maximum number of iterations reached."; break; }
        if (gmax <= eps) { term = CBA_TERM_CONVERGENCE; msg = "Gradient tolerance reached."; break; }
        if (radius <= 1e-32) { term = CBA_TERM_CONVERGENCE; msg = "Minimum trust region radius reached."; break; }
        ++iter;
        A = H;
        for (int i = 0; i < 6; ++i) {
            const double ds = std::min(std::max(H[i * 6 + i] * scale2[i], 1e-6), 1e32);
            A[i * 6 + i] += ds / radius / scale2[i];
        }
        bool valid = chol_inplace(A, 6);
        double model_change = 0;
        if (valid) {
            for (int i = 0; i < 6; ++i) delta[i] = -g[i];
            chol_solve(A, 6, delta.data());
            double dg = 0, dHd = 0;
            for (int i = 0; i < 6; ++i) {
                dg += delta[i] * g[i];
                double s = 0;
                for (int j = 0; j < 6; ++j) s += H[i * 6 + j] * delta[j];
                dHd += delta[i] * s;
                if (!std::isfinite(delta[i])) valid = false;
            }
            model_change = -dg - 0.5 * dHd;
            if (!(model_change > 0.0)) valid = false;
        }
        if (!valid) {
            if (++invalid >= 5) { term = CBA_TERM_FAILURE; msg = "Number of consecutive invalid steps more than max."; break; }
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        quat_plus(x.data(), delta.data(), cand.data());
        for (int k = 0; k < 3; ++k) cand[4 + k] = x[4 + k] + delta[3 + k];
        double cacc[AXXB_NACC];
        ev.eval(cand.data(), o.huber_delta, cacc);
        double cand_cost = cacc[27];
        if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
        double sn = 0, xn = 0;
        for (int k = 0; k < 7; ++k) { sn += (x[k] - cand[k]) * (x[k] - cand[k]); xn += x[k] * x[k]; }
        if (std::sqrt(sn) <= eps * (std::sqrt(xn) + eps)) { term = CBA_TERM_CONVERGENCE; msg = "Parameter tolerance reached."; break; }
        const double cost_change = cost - cand_cost;
        if (std::fabs(cost_change) <= eps * cost) { term = CBA_TERM_CONVERGENCE; msg = "Function tolerance reached."; break; }
        const double rel = cost_change / model_change;
        if (o.verbose) std::printf("[cba axxb] it %3d cost %.12e cand %.12e rel %.3e radius %.3e\n", iter, cost, cand_cost, rel, radius);
        if (rel > 1e-3) {
            x = cand; cost = cand_cost; ++successful;
            std::memcpy(acc, cacc, sizeof(acc));
            axxb_unpack(acc, H, g);
            gmax = gmax_of();
            radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            decrease_factor = 2.0;
        } else {
            radius /= decrease_factor;
            decrease_factor *= 2.0;
        }
    }
    std::memcpy(pose7, x.data(), sizeof(double) * 7);
    out->termination = term; out->success = term == CBA_TERM_CONVERGENCE;
    out->iterations = iter; out->successful_steps = successful;
    out->initial_cost = initial_cost; out->final_cost = cost;
    out->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::snprintf(out->report, sizeof(out->report), "calibba(AX=XB LM, %d pairs): %s iters=%d cost %.6e -> %.6e",
                  static_cast<int>(acc[28] + 0.5), msg, iter, initial_cost, cost);
    if (cov77) {  // ceresutils.h:69-126, blocks [quat(4), tran(3)]; left empty (zeros) if rank deficient
        std::memset(cov77, 0, sizeof(double) * 49);
        std::vector<double> L = H, Hi;
        if (chol_inplace(L, 6)) {
            chol_inverse(L, 6, Hi);
            const double* q = pose7;
            const double PJ[12] = {-q[1], -q[2], -q[3], q[0], q[3], -q[2], -q[3], q[0], q[1], q[2], -q[1], q[0]};
            double P[7][6] = {{0}};
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) P[r][c] = PJ[r * 3 + c];
            for (int k = 0; k < 3; ++k) P[4 + k][3 + k] = 1.0;
            for (int i = 0; i < 7; ++i)
                for (int j = 0; j < 7; ++j) {
                    double s = 0;
                    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) s += P[i][a] * Hi[a * 6 + b] * P[j][b];
                    cov77[i * 7 + j] = s;
                }
        }
    }
}

}  // namespace cba
