// axxb_math.hpp — AX = XB hand-eye residual with ANALYTIC tangent Jacobian, and the motion-pair
// construction, as __host__ __device__ inline code (kernel: handeye.hip; CPU test build: tests/cpu_backend).
//
// Reference: AxXbResidual::operator() src/estimation/residuals/handeyeresidual.h:25-49 (6 residuals:
// rotation log of R_A R_X R_B^T R_X^T via Eigen::AngleAxis, translation (R_A - I) t_X - (R_X t_B - t_A));
// pairs: make_motion_pair / is_good_pair src/estimation/linear/handeyedlt.cpp:11-49 and log_so3
// include/calib/estimation/common/se3_utils.h:27-40.  The reference re-projects A, B onto SO(3) by SVD
// (se3_utils.h:10-19); products of rotation matrices built from unit quaternions are orthonormal to
// ~1e-16, so that projection is the identity to rounding and is not repeated here.
//
// Derivative (X perturbed on the left by exp([2d]x), ceres::QuaternionManifold):
//   R_S+ = exp([w]x) R_S,  w = 2 R_A (I - C) d,  C = R_X R_B^T R_X^T
//   d Log(R_S)/dd = 2 Jl^-1(phi) R_A (I - C),   Jl^-1 = I - 1/2 [phi]x + k [phi]x^2,
//   k = 1/th^2 - (1 + cos th) / (2 th sin th)   (-> 1/12 as th -> 0)
//   d r_t / dd = 2 [R_X t_B]x,   d r_t / dt = R_A - I.
// Exactly at phi = 0 the reference's autodiff returns a zero rotation Jacobian (Eigen's axis fallback);
// the analytic form is the true derivative there (SURVEY.md §7 "hard parts", last item).
#pragma once
#include "reproj_math.hpp"

namespace cba {

// Eigen: Quaternion(Matrix3) then AngleAxis(Quaternion) -> rotation vector angle*axis
CBA_HD void rotmat_log_eigen(const double* m, double* phi) {
    double q[4];
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[7] - m[5]) * t;
        q[2] = (m[2] - m[6]) * t;
        q[3] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
    double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n != 0.0) {
        const double angle = 2.0 * atan2(n, fabs(q[0]));
        if (q[0] < 0.0) n = -n;
        const double s = angle / n;
        phi[0] = q[1] * s; phi[1] = q[2] * s; phi[2] = q[3] * s;
    } else {
        phi[0] = phi[1] = phi[2] = 0.0;
    }
}

// se3_utils.h:27-40 (without the SVD projection)
CBA_HD double log_so3(const double* R, double* w) {
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double th = acos(c);
    if (th < 1e-12) { w[0] = w[1] = w[2] = 0.0; return 0.0; }
    const double k = 0.5 / sin(th) * th;
    w[0] = (R[7] - R[5]) * k; w[1] = (R[2] - R[6]) * k; w[2] = (R[3] - R[1]) * k;
    return sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
}

// Motion pair (handeyedlt.cpp:11-23) from poses given as R (row-major 9) + t (3):
//   A = (b_T_g,i)^-1 b_T_g,j      B = c_T_t,i (c_T_t,j)^-1.   Returns is_good_pair (handeyedlt.cpp:25-49).
CBA_HD bool motion_pair(const double* Rbi, const double* tbi, const double* Rbj, const double* tbj, const double* Rci,
                        const double* tci, const double* Rcj, const double* tcj, double min_angle, double axis_parallel_eps,
                        double* RA, double* RB, double* tA, double* tB) {
    double Rt[9], d[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = Rbi[3 * c + r];
    mat3_mul(Rt, Rbj, RA);
    for (int k = 0; k < 3; ++k) d[k] = tbj[k] - tbi[k];
    mat3_vec(Rt, d, tA);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = Rcj[3 * c + r];
    mat3_mul(Rci, Rt, RB);
    mat3_vec(RB, tcj, d);
    for (int k = 0; k < 3; ++k) tB[k] = tci[k] - d[k];
    double al[3], be[3];
    const double na = log_so3(RA, al), nb = log_so3(RB, be);
    if ((na < nb ? na : nb) < min_angle) return false;
    if (na >= 1e-9 && nb >= 1e-9) {
        for (int k = 0; k < 3; ++k) { al[k] /= na; be[k] /= nb; }
        double cr[3];
        cross3(al, be, cr);
        if (sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) < axis_parallel_eps) return false;
    }
    return true;
}

// residual r[6] and tangent Jacobian J[6][6] (row-major; columns [d(3) t(3)]) at X = (RX, tX)
CBA_HD void axxb_point(const double* RX, const double* tX, const double* RA, const double* RB, const double* tA,
                       const double* tB, double* r, double* J) {
    double RBt[9], RXt[9], M[9], C[9], RS[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { RBt[3 * a + b] = RB[3 * b + a]; RXt[3 * a + b] = RX[3 * b + a]; }
    mat3_mul(RX, RBt, M);
    mat3_mul(M, RXt, C);
    // the reference multiplies left to right: ((R_A R_X) R_B^T) R_X^T
    double T1[9], T2[9];
    mat3_mul(RA, RX, T1);
    mat3_mul(T1, RBt, T2);
    mat3_mul(T2, RXt, RS);
    double phi[3];
    rotmat_log_eigen(RS, phi);
    double v[3], e1[3];
    mat3_vec(RX, tB, v);
    const double AmI[9] = {RA[0] - 1.0, RA[1], RA[2], RA[3], RA[4] - 1.0, RA[5], RA[6], RA[7], RA[8] - 1.0};
    mat3_vec(AmI, tX, e1);
    for (int k = 0; k < 3; ++k) { r[k] = phi[k]; r[3 + k] = e1[k] - (v[k] - tA[k]); }
    // Jl^-1(phi)
    const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const double th = sqrt(th2);
    double kk;
    if (th < 1e-4) kk = 1.0 / 12.0 + th2 / 720.0;
    else kk = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
    const double K[9] = {0.0, -phi[2], phi[1], phi[2], 0.0, -phi[0], -phi[1], phi[0], 0.0};
    double K2[9], Jli[9], ImC[9], G[9], Jr[9];
    mat3_mul(K, K, K2);
    for (int a = 0; a < 9; ++a) Jli[a] = ((a % 4 == 0) ? 1.0 : 0.0) - 0.5 * K[a] + kk * K2[a];
    for (int a = 0; a < 9; ++a) ImC[a] = ((a % 4 == 0) ? 1.0 : 0.0) - C[a];
    mat3_mul(RA, ImC, G);
    mat3_mul(Jli, G, Jr);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            J[a * 6 + b] = 2.0 * Jr[3 * a + b];
            J[a * 6 + 3 + b] = 0.0;
            J[(3 + a) * 6 + 3 + b] = AmI[3 * a + b];
        }
    // 2 [v]x
    J[3 * 6 + 0] = 0.0;          J[3 * 6 + 1] = -2.0 * v[2];  J[3 * 6 + 2] = 2.0 * v[1];
    J[4 * 6 + 0] = 2.0 * v[2];   J[4 * 6 + 1] = 0.0;          J[4 * 6 + 2] = -2.0 * v[0];
    J[5 * 6 + 0] = -2.0 * v[1];  J[5 * 6 + 1] = 2.0 * v[0];   J[5 * 6 + 2] = 0.0;
}

// One pair's contribution to [H upper (21) | g (6) | cost (1) | count (1)] with per-pair Huber
// (handeye.cpp:48-55: every pair has its own HuberLoss).
// Tsai-Lenz all-pairs sums (estimate_rotation_allpairs_weighted / estimate_translation_allpairs_weighted,
// src/estimation/linear/handeyedlt.cpp:84-124) of ONE motion pair, into acc[0..5] (upper triangle of the 3x3 normal matrix),
// acc[6..8] (right-hand side), acc[9] (pair count).
//   mode 0 (rotation):    rows skew(alpha + beta), rhs beta - alpha        alpha = log A, beta = log B
//   mode 1 (translation): rows R_A - I,            rhs R_X t_B - t_A
CBA_HD void tsai_lenz_accumulate(int mode, const double* RA, const double* RB, const double* tA, const double* tB, const double* RX,
                                 double* acc) {
    double M[9], d[3];
    if (mode == 0) {
        double al[3], be[3];
        log_so3(RA, al);
        log_so3(RB, be);
        const double s[3] = {al[0] + be[0], al[1] + be[1], al[2] + be[2]};
        M[0] = 0; M[1] = -s[2]; M[2] = s[1]; M[3] = s[2]; M[4] = 0; M[5] = -s[0]; M[6] = -s[1]; M[7] = s[0]; M[8] = 0;
        for (int k = 0; k < 3; ++k) d[k] = be[k] - al[k];
    } else {
        for (int k = 0; k < 9; ++k) M[k] = RA[k] - ((k % 4 == 0) ? 1.0 : 0.0);
        double rtb[3];
        mat3_vec(RX, tB, rtb);
        for (int k = 0; k < 3; ++k) d[k] = rtb[k] - tA[k];
    }
    int e = 0;
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) acc[e++] += M[a] * M[b] + M[3 + a] * M[3 + b] + M[6 + a] * M[6 + b];
    for (int a = 0; a < 3; ++a) acc[6 + a] += M[a] * d[0] + M[3 + a] * d[1] + M[6 + a] * d[2];
    acc[9] += 1.0;
}

// ridge_llsq (se3_utils.h:57-63): (A^T A + lambda I) x = A^T b from the packed sums above (Eigen LDLT -> Cholesky, SPD)
CBA_HD bool tsai_lenz_solve(const double* acc, double lambda, double* x) {
    double A[9] = {acc[0] + lambda, acc[1], acc[2], acc[1], acc[3] + lambda, acc[4], acc[2], acc[4], acc[5] + lambda};
    if (!(A[0] > 0.0)) return false;
    // 3x3 Cholesky
    const double l00 = sqrt(A[0]), l10 = A[3] / l00, l20 = A[6] / l00;
    const double d1 = A[4] - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double l11 = sqrt(d1), l21 = (A[7] - l20 * l10) / l11;
    const double d2 = A[8] - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    const double l22 = sqrt(d2);
    const double y0 = acc[6] / l00, y1 = (acc[7] - l10 * y0) / l11, y2 = (acc[8] - l20 * y0 - l21 * y1) / l22;
    x[2] = y2 / l22;
    x[1] = (y1 - l21 * x[2]) / l11;
    x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
    return true;
}

// exp_so3 (se3_utils.h:42-51), row-major
CBA_HD void exp_so3(const double* w, double* R) {
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (th < 1e-12) return;
    const double a[3] = {w[0] / th, w[1] / th, w[2] / th};
    const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    double K2[9];
    mat3_mul(K, K, K2);
    const double s = sin(th), c = 1.0 - cos(th);
    for (int k = 0; k < 9; ++k) R[k] += s * K[k] + c * K2[k];
}

constexpr int AXXB_NACC = 29;
CBA_HD void axxb_accumulate(const double* r, const double* J, double huber_delta, double* acc) {
    double s = 0.0;
    for (int k = 0; k < 6; ++k) s += r[k] * r[k];
    double rho, w;
    huber(s, huber_delta, &rho, &w);
    int e = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            double h = 0.0;
            for (int k = 0; k < 6; ++k) h += J[k * 6 + a] * J[k * 6 + b];
            acc[e++] += w * h;
        }
    for (int a = 0; a < 6; ++a) {
        double g = 0.0;
        for (int k = 0; k < 6; ++k) g += J[k * 6 + a] * r[k];
        acc[21 + a] += w * g;
    }
    acc[27] += 0.5 * rho;
    acc[28] += 1.0;
}

}  // namespace cba
