// tests/cpu_backend/planarpose_cpu.cpp — TEST-ONLY host run of vp_math.hpp (the per-view planar-pose VP solve
// the GPU runs one wavefront per view), here with the single-thread group.  Never linked into libcalibba.so.
#include <cmath>

#include "../../calibration_amd/csrc/vp_math.hpp"
#include "../../include/calibba.h"

using namespace cba;

// r (2N), J (2N x 6 row-major), alpha via the analytic Golub-Pereyra derivative; returns 0 on success
template <int NR>
static int vp_eval_rows(int n, const double* X, const double* Y, const double* u, const double* v, const double* kmtx5,
                        const double* pose6, double* r, double* J, double* alpha, double* H36, double* g6) {
    VPView V{n, X, Y, u, v, {kmtx5[0], kmtx5[1], kmtx5[2], kmtx5[3], kmtx5[4]}, NR};
    double al[VP_MAX_M], s, H[36], g[6];
    SerialCoop co;
    if (!vp_evaluate<NR>(V, co, pose6, true, al, &s, H, g)) return 1;
    constexpr int m = NR + 2;
    for (int a = 0; a < m; ++a) alpha[a] = al[a];
    for (int a = 0; a < 36; ++a) H36[a] = H[a];
    for (int a = 0; a < 6; ++a) g6[a] = g[a];
    // explicit rows for the test: recompute with the same formulas (finite set of points, serial)
    double M[VP_MAX_M * VP_MAX_M] = {0}, G[6][VP_MAX_M] = {{0}};
    VPRow R;
    for (int i = 0; i < n; ++i) {
        vp_row<NR>(V, pose6, i, false, R);
        for (int a = 0; a < m; ++a)
            for (int c = 0; c <= a; ++c) M[a * m + c] += R.Au[a] * R.Au[c] + R.Av[a] * R.Av[c];
    }
    vp_chol<m>(M);
    for (int i = 0; i < n; ++i) {
        vp_row<NR>(V, pose6, i, true, R);
        double ru = -R.bu, rv = -R.bv, qux = -R.bux, quy = -R.buy, qvx = -R.bvx, qvy = -R.bvy;
        for (int a = 0; a < m; ++a) {
            ru += R.Au[a] * al[a]; rv += R.Av[a] * al[a];
            qux += R.Aux[a] * al[a]; quy += R.Auy[a] * al[a]; qvx += R.Avx[a] * al[a]; qvy += R.Avy[a] * al[a];
        }
        r[2 * i] = ru; r[2 * i + 1] = rv;
        for (int k = 0; k < 6; ++k) {
            const double wu = qux * R.dx[k] + quy * R.dy[k], wv = qvx * R.dx[k] + qvy * R.dy[k];
            for (int a = 0; a < m; ++a)
                G[k][a] += R.Au[a] * wu + R.Av[a] * wv + (R.Aux[a] * R.dx[k] + R.Auy[a] * R.dy[k]) * ru + (R.Avx[a] * R.dx[k] + R.Avy[a] * R.dy[k]) * rv;
        }
    }
    for (int k = 0; k < 6; ++k) vp_chol_solve<m>(M, G[k]);
    for (int i = 0; i < n; ++i) {
        vp_row<NR>(V, pose6, i, true, R);
        double qux = -R.bux, quy = -R.buy, qvx = -R.bvx, qvy = -R.bvy;
        for (int a = 0; a < m; ++a) { qux += R.Aux[a] * al[a]; quy += R.Auy[a] * al[a]; qvx += R.Avx[a] * al[a]; qvy += R.Avy[a] * al[a]; }
        for (int k = 0; k < 6; ++k) {
            double ju = qux * R.dx[k] + quy * R.dy[k], jv = qvx * R.dx[k] + qvy * R.dy[k];
            for (int a = 0; a < m; ++a) { ju -= R.Au[a] * G[k][a]; jv -= R.Av[a] * G[k][a]; }
            J[(2 * i) * 6 + k] = ju; J[(2 * i + 1) * 6 + k] = jv;
        }
    }
    return 0;
}

template <int NR>
static void vp_solve(const VPView& V, const cba_options* o, bool want_cov, VPResult& R) {
    SerialCoop co;
    vp_solve_view<NR>(V, co, o->huber_delta, o->epsilon, o->max_iterations, want_cov, R);
}

extern "C" {

int hm_planar_vp_eval(int n, const double* X, const double* Y, const double* u, const double* v, const double* kmtx5,
                      int num_radial, const double* pose6, double* r, double* J, double* alpha, double* H36, double* g6) {
    switch (num_radial) {
        case 0: return vp_eval_rows<0>(n, X, Y, u, v, kmtx5, pose6, r, J, alpha, H36, g6);
        case 1: return vp_eval_rows<1>(n, X, Y, u, v, kmtx5, pose6, r, J, alpha, H36, g6);
        case 2: return vp_eval_rows<2>(n, X, Y, u, v, kmtx5, pose6, r, J, alpha, H36, g6);
        default: return vp_eval_rows<3>(n, X, Y, u, v, kmtx5, pose6, r, J, alpha, H36, g6);
    }
}

// full per-view solve with pose6 in/out (angle-axis + t), same outputs as orc_planar_pose_solve
int hm_planar_pose_solve(int n, const double* X, const double* Y, const double* u, const double* v, const double* kmtx5,
                         int num_radial, double* pose6, const cba_options* o, cba_summary* out, double* distortion, double* rms,
                         double* cov66) {
    VPView V{n, X, Y, u, v, {kmtx5[0], kmtx5[1], kmtx5[2], kmtx5[3], kmtx5[4]}, num_radial};
    VPResult R;
    for (int k = 0; k < 6; ++k) R.pose6[k] = pose6[k];
    switch (num_radial) {
        case 0: vp_solve<0>(V, o, cov66 != nullptr, R); break;
        case 1: vp_solve<1>(V, o, cov66 != nullptr, R); break;
        case 2: vp_solve<2>(V, o, cov66 != nullptr, R); break;
        default: vp_solve<3>(V, o, cov66 != nullptr, R); break;
    }
    for (int k = 0; k < 6; ++k) pose6[k] = R.pose6[k];
    out->termination = R.termination; out->success = R.termination == CBA_TERM_CONVERGENCE;
    out->iterations = R.iterations; out->successful_steps = R.successful_steps;
    out->initial_cost = R.initial_cost; out->final_cost = R.final_cost;
    if (distortion) for (int k = 0; k < num_radial + 2; ++k) distortion[k] = R.alpha[k];
    if (rms) *rms = R.rms;
    if (cov66) for (int k = 0; k < 36; ++k) cov66[k] = R.cov_ok ? R.cov[k] : 0.0;
    return 0;
}

void hm_quat_to_angle_axis(const double* q, double* aa) { quat_to_angle_axis_ceres(q, aa); }
void hm_angle_axis_to_quat(const double* aa, double* q) { angle_axis_to_quat_ceres(aa, q); }
}
