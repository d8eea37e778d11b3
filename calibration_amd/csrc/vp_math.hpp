// vp_math.hpp — planar-pose refinement by variable projection, one whole Levenberg-Marquardt solve per
// view as __host__ __device__ inline code (kernel: planarpose.hip, ONE WAVEFRONT per view through the cooperative
// solver of small_lm.hpp, batched over views; CPU test build: tests/cpu_backend with the single-thread group).
//
// Reference: optimize_planar_pose src/estimation/optim/planarpose.cpp:84-127 with the functor
// PlanarPoseVPResidual (:39-57): pose6 = [angle-axis(3), t(3)] (no manifold), each evaluation builds the
// 2N x (nr+2) distortion design matrix A(pose) and right-hand side b(pose)
// (include/calib/models/distortion.h:254-288, to_observation src/estimation/detail/observationutils.h:97-113),
// eliminates alpha = argmin |A alpha - b| and returns r = A alpha - b; one residual block, one Huber loss.
// The reference differentiates THROUGH the least-squares solve with Jets; the analytic equivalent is the
// full Golub-Pereyra derivative of the projected residual:
//     dr/dp_k = P_perp (dA_k alpha - db_k) - A (A^T A)^-1 dA_k^T r,     P_perp = I - A (A^T A)^-1 A^T
// With per-row scalars q = (dA_row/d(x,y)) alpha - db_row/d(x,y) this needs three passes over the points:
//   1. A^T A, A^T b -> alpha          2. r, |r|^2, G_k = A^T w_k + dA_k^T r          3. J rows -> J^T J, J^T r
// LM semantics: small_lm.hpp (the same restated Ceres trust-region rules as lm_core.hpp, unconstrained Euclidean block).
#pragma once
#include "../../include/calibba.h"
#include "reproj_math.hpp"
#include "schur_math.hpp"
#include "small_lm.hpp"

namespace cba {

constexpr int VP_MAX_M = 5;  // num_radial <= 3

struct VPView {
    int n;
    const double *X, *Y, *u, *v;
    double K[5];  // fx fy cx cy skew (CameraMatrix, camera_matrix.h:12-19)
    int num_radial;
};

struct VPResult {
    double pose6[6];
    double alpha[VP_MAX_M];
    double initial_cost, final_cost, rms;
    double cov[36];
    int iterations, successful_steps, termination, cov_ok;
};

// ceres::AngleAxisRotatePoint (third-party, restated) and d(result)/d(aa) = -R [p]x Jr(aa)
CBA_HD void aa_rotate(const double* aa, const double* p, double* out, double* dRda /*3x3 or null*/) {
    const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    double R[9];
    if (th2 > 2.220446049250313e-16) {
        const double th = sqrt(th2), c = cos(th), s = sin(th), ti = 1.0 / th;
        const double w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
        double wxp[3];
        cross3(w, p, wxp);
        const double tmp = (w[0] * p[0] + w[1] * p[1] + w[2] * p[2]) * (1.0 - c);
        for (int i = 0; i < 3; ++i) out[i] = p[i] * c + wxp[i] * s + w[i] * tmp;
        if (!dRda) return;
        const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double K2[9];
        mat3_mul(K, K, K2);
        for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + (1.0 - c) * K2[i];
        // right Jacobian Jr = I - (1-c)/th K + (th - s)/th K^2   (K = [w]x, w unit)
        double Jr[9];
        for (int i = 0; i < 9; ++i) Jr[i] = ((i % 4 == 0) ? 1.0 : 0.0) - (1.0 - c) * ti * K[i] + (th - s) * ti * K2[i];
        const double Px[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
        double T1[9], T2[9];
        mat3_mul(R, Px, T1);
        mat3_mul(T1, Jr, T2);
        for (int i = 0; i < 9; ++i) dRda[i] = -T2[i];
    } else {
        double wxp[3];
        cross3(aa, p, wxp);
        for (int i = 0; i < 3; ++i) out[i] = p[i] + wxp[i];
        if (!dRda) return;
        // d(p + aa x p)/d(aa) = -[p]x
        dRda[0] = 0; dRda[1] = p[2]; dRda[2] = -p[1];
        dRda[3] = -p[2]; dRda[4] = 0; dRda[5] = p[0];
        dRda[6] = p[1]; dRda[7] = -p[0]; dRda[8] = 0;
    }
}

// One observation: normalised (x, y), their derivatives w.r.t. pose6, the two design rows Au, Av (m),
// their x- and y-derivatives, b and its x/y derivatives.
struct VPRow {
    double Au[VP_MAX_M], Av[VP_MAX_M], Aux[VP_MAX_M], Auy[VP_MAX_M], Avx[VP_MAX_M], Avy[VP_MAX_M];
    double bu, bv, bux, buy, bvx, bvy;
    double dx[6], dy[6];
};

// The two design rows of one observation at normalised coordinates (x, y): Au, Av (m = NR + 2), b, and (deriv) their
// x- and y-derivatives (distortion.h:254-288).  K = [fx fy cx cy skew].
template <int NR>
CBA_HD void vp_design(const double* K, double x, double y, double uo, double vo, bool deriv, VPRow& R) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], skew = K[4];
    constexpr int nr = NR;
    const double r2 = x * x + y * y;
    const double g = fx * x + skew * y, h = fy * y;
    double rpow = r2, rprev = 1.0;  // rho^(j+1), rho^j
    for (int j = 0; j < nr; ++j) {
        R.Au[j] = g * rpow;
        R.Av[j] = h * rpow;
        if (deriv) {
            const double dr = (j + 1) * rprev;  // d rho^(j+1) / d rho
            R.Aux[j] = fx * rpow + g * dr * 2.0 * x;
            R.Auy[j] = skew * rpow + g * dr * 2.0 * y;
            R.Avx[j] = h * dr * 2.0 * x;
            R.Avy[j] = fy * rpow + h * dr * 2.0 * y;
        }
        rprev = rpow;
        rpow *= r2;
    }
    R.Au[nr] = fx * (2.0 * x * y) + skew * (r2 + 2.0 * y * y);
    R.Au[nr + 1] = fx * (r2 + 2.0 * x * x) + skew * (2.0 * x * y);
    R.Av[nr] = fy * (r2 + 2.0 * y * y);
    R.Av[nr + 1] = fy * (2.0 * x * y);
    R.bu = uo - (g + cx);
    R.bv = vo - (h + cy);
    if (!deriv) return;
    R.Aux[nr] = fx * 2.0 * y + skew * 2.0 * x;       R.Auy[nr] = fx * 2.0 * x + skew * 6.0 * y;
    R.Aux[nr + 1] = fx * 6.0 * x + skew * 2.0 * y;   R.Auy[nr + 1] = fx * 2.0 * y + skew * 2.0 * x;
    R.Avx[nr] = fy * 2.0 * x;                        R.Avy[nr] = fy * 6.0 * y;
    R.Avx[nr + 1] = fy * 2.0 * y;                    R.Avy[nr + 1] = fy * 2.0 * x;
    R.bux = -fx; R.buy = -skew; R.bvx = 0.0; R.bvy = -fy;
}

template <int NR>
CBA_HD void vp_row(const VPView& V, const double* pose6, int i, bool deriv, VPRow& R) {
    const double pt[3] = {V.X[i], V.Y[i], 0.0};
    double pc[3], dRda[9];
    aa_rotate(pose6, pt, pc, deriv ? dRda : nullptr);
    for (int k = 0; k < 3; ++k) pc[k] += pose6[3 + k];
    const double iz = 1.0 / pc[2];
    const double x = pc[0] * iz, y = pc[1] * iz;
    vp_design<NR>(V.K, x, y, V.u[i], V.v[i], deriv, R);
    if (!deriv) return;
    // d(x, y)/d pose6: d pc/d aa = dRda, d pc/d t = I;  d(x,y)/d pc = iz [1 0 -x; 0 1 -y]
    for (int k = 0; k < 3; ++k) {
        R.dx[k] = iz * (dRda[0 * 3 + k] - x * dRda[2 * 3 + k]);
        R.dy[k] = iz * (dRda[1 * 3 + k] - y * dRda[2 * 3 + k]);
    }
    R.dx[3] = iz; R.dx[4] = 0.0; R.dx[5] = -x * iz;
    R.dy[3] = 0.0; R.dy[4] = iz; R.dy[5] = -y * iz;
}

// small SPD solve (m <= 5), in place lower Cholesky of M (row-major m x m); false if not PD
template <int m>
CBA_HD bool vp_chol(double* M) {
    for (int j = 0; j < m; ++j) {
        double d = M[j * m + j];
        for (int k = 0; k < j; ++k) d -= M[j * m + k] * M[j * m + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        M[j * m + j] = d;
        for (int i = j + 1; i < m; ++i) {
            double s = M[i * m + j];
            for (int k = 0; k < j; ++k) s -= M[i * m + k] * M[j * m + k];
            M[i * m + j] = s / d;
        }
    }
    return true;
}
template <int m>
CBA_HD void vp_chol_solve(const double* L, double* b) {
    for (int i = 0; i < m; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * m + k] * b[k];
        b[i] = s / L[i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < m; ++k) s -= L[k * m + i] * b[k];
        b[i] = s / L[i * m + i];
    }
}

// Evaluate at pose6: alpha, s = |r|^2 and (if want_jac) the UNWEIGHTED H = J^T J (36, full), g = J^T r (6).
// Returns false when fit_distortion_full would fail (N < 8, distortion.h:236-239) or A^T A is singular.
// `co` is the cooperative group that owns the view (small_lm.hpp): its lanes stride over the points and the
// partial sums of each pass cross the group once; every lane leaves with identical results.
template <int NR, class Coop>
CBA_HD bool vp_evaluate(const VPView& V, Coop& co, const double* pose6, bool want_jac, double* alpha, double* s_out, double* H,
                        double* g) {
    constexpr int m = NR + 2;
    const int N = V.n;
    if (N < 8) return false;
    double M[VP_MAX_M * VP_MAX_M], rhs[VP_MAX_M];
    for (int a = 0; a < VP_MAX_M * VP_MAX_M; ++a) M[a] = 0.0;
    for (int a = 0; a < VP_MAX_M; ++a) rhs[a] = 0.0;
    VPRow R;
    for (int i = co.lane(); i < N; i += co.width()) {  // pass 1
        vp_row<NR>(V, pose6, i, false, R);
        for (int a = 0; a < m; ++a) {
            rhs[a] += R.Au[a] * R.bu + R.Av[a] * R.bv;
            for (int c = 0; c <= a; ++c) M[a * m + c] += R.Au[a] * R.Au[c] + R.Av[a] * R.Av[c];
        }
    }
    for (int a = 0; a < m; ++a) {
        rhs[a] = co.sum(rhs[a]);
        for (int c = 0; c <= a; ++c) M[a * m + c] = co.sum(M[a * m + c]);
    }
    if (!vp_chol<m>(M)) return false;
    for (int a = 0; a < m; ++a) alpha[a] = rhs[a];
    vp_chol_solve<m>(M, alpha);
    double s = 0.0;
    double G[6][VP_MAX_M];
    for (int k = 0; k < 6; ++k) for (int a = 0; a < VP_MAX_M; ++a) G[k][a] = 0.0;
    for (int i = co.lane(); i < N; i += co.width()) {  // pass 2
        vp_row<NR>(V, pose6, i, want_jac, R);
        double ru = -R.bu, rv = -R.bv;
        for (int a = 0; a < m; ++a) { ru += R.Au[a] * alpha[a]; rv += R.Av[a] * alpha[a]; }
        s += ru * ru + rv * rv;
        if (!want_jac) continue;
        double qux = -R.bux, quy = -R.buy, qvx = -R.bvx, qvy = -R.bvy;
        for (int a = 0; a < m; ++a) {
            qux += R.Aux[a] * alpha[a]; quy += R.Auy[a] * alpha[a];
            qvx += R.Avx[a] * alpha[a]; qvy += R.Avy[a] * alpha[a];
        }
        for (int k = 0; k < 6; ++k) {
            const double wu = qux * R.dx[k] + quy * R.dy[k], wv = qvx * R.dx[k] + qvy * R.dy[k];
            for (int a = 0; a < m; ++a)
                G[k][a] += R.Au[a] * wu + R.Av[a] * wv + (R.Aux[a] * R.dx[k] + R.Auy[a] * R.dy[k]) * ru +
                           (R.Avx[a] * R.dx[k] + R.Avy[a] * R.dy[k]) * rv;
        }
    }
    *s_out = co.sum(s);
    if (!want_jac) return true;
    for (int k = 0; k < 6; ++k) {
        for (int a = 0; a < m; ++a) G[k][a] = co.sum(G[k][a]);
        vp_chol_solve<m>(M, G[k]);  // c_k = (A^T A)^-1 G_k
    }
    double Hs[21], gs[6];  // upper triangle of H, row-major
    for (int a = 0; a < 21; ++a) Hs[a] = 0.0;
    for (int a = 0; a < 6; ++a) gs[a] = 0.0;
    for (int i = co.lane(); i < N; i += co.width()) {  // pass 3
        vp_row<NR>(V, pose6, i, true, R);
        double ru = -R.bu, rv = -R.bv;
        double qux = -R.bux, quy = -R.buy, qvx = -R.bvx, qvy = -R.bvy;
        for (int a = 0; a < m; ++a) {
            ru += R.Au[a] * alpha[a]; rv += R.Av[a] * alpha[a];
            qux += R.Aux[a] * alpha[a]; quy += R.Auy[a] * alpha[a];
            qvx += R.Avx[a] * alpha[a]; qvy += R.Avy[a] * alpha[a];
        }
        double Ju[6], Jv[6];
        for (int k = 0; k < 6; ++k) {
            double ju = qux * R.dx[k] + quy * R.dy[k], jv = qvx * R.dx[k] + qvy * R.dy[k];
            for (int a = 0; a < m; ++a) { ju -= R.Au[a] * G[k][a]; jv -= R.Av[a] * G[k][a]; }
            Ju[k] = ju; Jv[k] = jv;
        }
        int k = 0;
        for (int a = 0; a < 6; ++a) {
            gs[a] += Ju[a] * ru + Jv[a] * rv;
            for (int c = a; c < 6; ++c, ++k) Hs[k] += Ju[a] * Ju[c] + Jv[a] * Jv[c];
        }
    }
    int k = 0;
    for (int a = 0; a < 6; ++a) {
        g[a] = co.sum(gs[a]);
        for (int c = a; c < 6; ++c, ++k) {
            const double t = co.sum(Hs[k]);
            H[a * 6 + c] = t;
            H[c * 6 + a] = t;
        }
    }
    return true;
}

// optimize_planar_pose as a small_lm problem: ONE residual block (all 2N residuals) under one Huber loss
// (planarpose.cpp:99-102), so the loss weight is a scalar of the whole evaluation.
struct VPAux {
    double alpha[VP_MAX_M];
    double s;  // |r|^2, unweighted
};
template <int NR>
struct VPProblem {
    using Aux = VPAux;
    VPView V;
    double huber_delta;
    template <class Coop>
    CBA_HD bool evaluate(Coop& co, const double* pose6, bool want_jac, double* cost, double* H, double* g, Aux* aux) const {
        for (int a = 0; a < VP_MAX_M; ++a) aux->alpha[a] = 0.0;
        if (!vp_evaluate<NR>(V, co, pose6, want_jac, aux->alpha, &aux->s, H, g)) return false;
        double rho, w;
        huber(aux->s, huber_delta, &rho, &w);
        *cost = 0.5 * rho;
        if (want_jac) {
            for (int a = 0; a < 36; ++a) H[a] *= w;
            for (int a = 0; a < 6; ++a) g[a] *= w;
        }
        return true;
    }
};

// The whole solve for one view.  pose6 in `res.pose6` on entry (initial guess) and exit (result).
template <int NR, class Coop>
CBA_HD void vp_solve_view(const VPView& V, Coop& co, double huber_delta, double eps, int max_iterations, bool want_cov, VPResult& res) {
    VPProblem<NR> P{V, huber_delta};
    SmallLMState<6> st;
    for (int k = 0; k < 6; ++k) st.x[k] = res.pose6[k];
    VPAux aux;
    for (int a = 0; a < VP_MAX_M; ++a) aux.alpha[a] = 0.0;
    aux.s = 0.0;
    small_lm_solve<6>(P, co, eps, max_iterations, st, aux);
    for (int k = 0; k < 6; ++k) res.pose6[k] = st.x[k];
    for (int a = 0; a < VP_MAX_M; ++a) res.alpha[a] = aux.alpha[a];
    res.iterations = st.iterations; res.successful_steps = st.successful_steps; res.termination = st.termination;
    res.initial_cost = st.initial_cost; res.final_cost = st.cost;
    res.rms = st.evaluated ? sqrt(aux.s / (2.0 * V.n)) : 0.0;  // planarpose.cpp:112-114
    res.cov_ok = 0;
    for (int a = 0; a < 36; ++a) res.cov[a] = 0.0;
    if (want_cov && st.evaluated) {  // ceresutils.h:69-126 with (ssr, n_res): (J~^T J~)^-1 * ssr / max(1, 2N - 6)
        const long long n_res = 2LL * V.n;
        const long long dof = n_res - 6 > 1 ? n_res - 6 : 1;
        res.cov_ok = small_covariance<6>(st.H, n_res, aux.s / static_cast<double>(dof), res.cov) ? 1 : 0;
    }
}

// ceres::RotationMatrixToAngleAxis / QuaternionToAngleAxis (third-party, restated), from a unit quaternion
CBA_HD void quat_to_angle_axis_ceres(const double* q, double* aa) {
    const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2 > 0.0) {
        const double st = sqrt(s2), ct = q[0];
        const double two_theta = 2.0 * ((ct < 0.0) ? atan2(-st, -ct) : atan2(st, ct));
        const double k = two_theta / st;
        aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
    } else {
        aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
    }
}
// ceres::AngleAxisToQuaternion
CBA_HD void angle_axis_to_quat_ceres(const double* aa, double* q) {
    const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (th2 > 0.0) {
        const double th = sqrt(th2), half = th * 0.5, k = sin(half) / th;
        q[0] = cos(half); q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
    } else {
        q[0] = 1.0; q[1] = aa[0] * 0.5; q[2] = aa[1] * 0.5; q[3] = aa[2] * 0.5;
    }
}

}  // namespace cba
