// tests/cpu_backend/homography_cpu.cpp — TEST-ONLY host run of hom_math.hpp / small_lm.hpp (the per-view homography
// refinement the GPU runs one wavefront per view) with the single-thread cooperative group.  Never linked into the product.
#include <cstring>

#include "../../calibration_amd/csrc/hom_math.hpp"

using namespace cba;

extern "C" {

// robustified cost, H = J~^T J~ (8x8), g = J~^T r~ of one view at h8
int hm_homography_eval(int n, const double* X, const double* Y, const double* u, const double* v, const double* h8, double huber_delta,
                       double* cost, double* H64, double* g8) {
    HomProblem P{n, X, Y, u, v, huber_delta};
    SerialCoop co;
    HomAux aux;
    return P.evaluate(co, h8, true, cost, H64, g8, &aux) ? 0 : 1;
}

// same outputs as orc_homography_solve
int hm_homography_solve(int n, const double* X, const double* Y, const double* u, const double* v, double* h9, const cba_options* o,
                        cba_summary* out, double* cov64) {
    if (n < 4) return 1;
    HomProblem P{n, X, Y, u, v, o->huber_delta};
    SerialCoop co;
    HomResult R;
    for (int k = 0; k < 8; ++k) R.h[k] = h9[k];
    hom_solve_view(P, co, o->epsilon, o->max_iterations, cov64 != nullptr, R);
    for (int k = 0; k < 8; ++k) h9[k] = R.h[k];
    h9[8] = 1.0;
    out->termination = R.termination; out->success = R.termination == CBA_TERM_CONVERGENCE;
    out->iterations = R.iterations; out->successful_steps = R.successful_steps;
    out->initial_cost = R.initial_cost; out->final_cost = R.final_cost;
    if (cov64) for (int k = 0; k < 64; ++k) cov64[k] = R.cov_ok ? R.cov[k] : 0.0;
    return 0;
}
}
