// seed_math.hpp — the per-view pose seed estimate_planar_pose (src/estimation/linear/planarpose_linear.cpp:54-76)
// as __host__ __device__ code for one cooperative group (one wavefront per view on the GPU):
//   pixels -> normalised coordinates (camera_matrix.h:33-39) -> Hartley-normalised DLT homography
//   (src/estimation/linear/homographyestimator.cpp:17-87) -> pose_from_homography_normalized (planarpose_linear.cpp:17-52).
// at C2/C3 scale one 2N x 9 SVD per view on the host is what is left on the wall clock once the
// refinement runs on the GPU; here a view costs three strided passes over its points (centroids, mean distances, the
// 45 sums of A^T A) and O(1) work per lane.
//
// The reference takes the right singular vector of the 2N x 9 design matrix for its smallest singular value
// (Eigen::JacobiSVD).  That vector is the eigenvector of A^T A for its smallest eigenvalue; it is obtained here by
// inverse iteration on the 9 x 9 Gram matrix (shift 1e-14 trace, 8 steps): with Hartley normalisation the gap between
// the two smallest eigenvalues is O(1) against an O(noise^2) smallest one, so every step gains >= 4 digits.
// The 3x3 orthogonalisation U V^T of the reference's JacobiSVD is the polar factor R (R^T R)^-1/2, evaluated through a
// Jacobi eigen-decomposition of R^T R with the reference's det < 0 rule (flip the smallest singular direction).
// Results agree with an SVD-based restatement to ~1e-12 on well-posed views; degenerate views (< 4 points,
// rank-deficient point sets) give the identity / an arbitrary member of the null space, as in the reference.
#pragma once
#include "small_lm.hpp"

namespace cba {

// in-place LDL^T-free Cholesky solve of a 9x9 SPD system kept as a full row-major matrix
CBA_HD bool seed_chol9(double* A) { return chol_n<9>(A); }

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): A -> diagonal in d, eigenvectors in the columns of V
CBA_HD void seed_eig3(const double* S, double* d, double* V) {
    double A[9];
    for (int i = 0; i < 9; ++i) { A[i] = S[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off <= 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p * 3 + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    d[0] = A[0]; d[1] = A[4]; d[2] = A[8];
}

// pose_from_homography_normalized (planarpose_linear.cpp:17-52); H row-major; out: R row-major (9), t (3)
CBA_HD void seed_pose_from_h(const double* H, double* R, double* t) {
    const double h1[3] = {H[0], H[3], H[6]}, h2[3] = {H[1], H[4], H[7]}, h3[3] = {H[2], H[5], H[8]};
    double s = sqrt(sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]) * sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]));
    if (s < 1e-12) s = 1.0;
    const double r1[3] = {h1[0] / s, h1[1] / s, h1[2] / s}, r2[3] = {h2[0] / s, h2[1] / s, h2[2] / s};
    double r3[3];
    cross3(r1, r2, r3);
    const double Ri[9] = {r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]};
    // polar factor U V^T of Ri: eigen-decomposition of Ri^T Ri = V diag(sig^2) V^T, U V^T = Ri V diag(1/sig) V^T
    double S[9], d[3], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[i * 3 + j] = Ri[0 * 3 + i] * Ri[0 * 3 + j] + Ri[1 * 3 + i] * Ri[1 * 3 + j] + Ri[2 * 3 + i] * Ri[2 * 3 + j];
    seed_eig3(S, d, V);
    const double det = Ri[0] * (Ri[4] * Ri[8] - Ri[5] * Ri[7]) - Ri[1] * (Ri[3] * Ri[8] - Ri[5] * Ri[6]) + Ri[2] * (Ri[3] * Ri[7] - Ri[4] * Ri[6]);
    int kmin = 0;
    for (int k = 1; k < 3; ++k) if (d[k] < d[kmin]) kmin = k;
    double w[3];
    for (int k = 0; k < 3; ++k) w[k] = 1.0 / sqrt(d[k] > 1e-300 ? d[k] : 1e-300);
    if (det < 0.0) w[kmin] = -w[kmin];  // vmtx.col(2) *= -1 with singular values sorted descending (:36-40)
    double Q[9];  // V diag(w) V^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Q[i * 3 + j] = V[i * 3 + 0] * w[0] * V[j * 3 + 0] + V[i * 3 + 1] * w[1] * V[j * 3 + 1] + V[i * 3 + 2] * w[2] * V[j * 3 + 2];
    mat3_mul(Ri, Q, R);
    for (int k = 0; k < 3; ++k) t[k] = h3[k] / s;
    if (R[8] < 0.0) {
        for (int k = 0; k < 9; ++k) R[k] = -R[k];
        for (int k = 0; k < 3; ++k) t[k] = -t[k];
    }
}

// Eigen::Quaterniond(Matrix3d) (third-party, restated; same as cba_pose_from_matrix) from a row-major rotation
CBA_HD void seed_rotmat_to_quat(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}

// Hartley-normalised DLT homography of one view (HomographyEstimator::fit -> normalize_and_estimate_homography,
// src/estimation/linear/homographyestimator.cpp:17-87, 123-146) from target points (X, Y) to pixels normalised by
// K = [fx fy cx cy skew] (K = [1 1 0 0 0]: raw pixels, i.e. estimate_homography's DLT path, homography.cpp:31-43).
// H row-major = T_dst^-1 Hn T_src with Hn(2,2) = 1 (:70); like the reference, no final rescale of H itself.  Returns false (H untouched) for
// fewer than 4 points or a non-finite result (fit returns nullopt).
template <class Coop>
CBA_HD bool dlt_homography_view(int n, const double* X, const double* Y, const double* u, const double* v, const double* K, Coop& co,
                                double* H) {
    if (n < 4) return false;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], skew = K[4];
    // pass 1: centroids of the target points and of the normalised pixels
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int i = co.lane(); i < n; i += co.width()) {
        const double yn = (v[i] - cy) / fy, xn = (u[i] - cx - skew * yn) / fx;
        a0 += X[i]; a1 += Y[i]; a2 += xn; a3 += yn;
    }
    const double inv_n = 1.0 / static_cast<double>(n);
    const double csx = co.sum(a0) * inv_n, csy = co.sum(a1) * inv_n, cdx = co.sum(a2) * inv_n, cdy = co.sum(a3) * inv_n;
    // pass 2: mean distances to the centroids
    a0 = 0.0; a1 = 0.0;
    for (int i = co.lane(); i < n; i += co.width()) {
        const double yn = (v[i] - cy) / fy, xn = (u[i] - cx - skew * yn) / fx;
        a0 += sqrt((X[i] - csx) * (X[i] - csx) + (Y[i] - csy) * (Y[i] - csy));
        a1 += sqrt((xn - cdx) * (xn - cdx) + (yn - cdy) * (yn - cdy));
    }
    const double ms = co.sum(a0) * inv_n, md = co.sum(a1) * inv_n;
    const double ss = ms > 0.0 ? 1.4142135623730951 / ms : 1.0, sd = md > 0.0 ? 1.4142135623730951 / md : 1.0;
    // pass 3: G = A^T A of the normalised DLT rows  [-x -y -1 0 0 0 ux uy u], [0 0 0 -x -y -1 vx vy v]
    double G[45];
    for (int e = 0; e < 45; ++e) G[e] = 0.0;
    for (int i = co.lane(); i < n; i += co.width()) {
        const double yn = (v[i] - cy) / fy, xn = (u[i] - cx - skew * yn) / fx;
        const double x = ss * X[i] - ss * csx, y = ss * Y[i] - ss * csy, uu = sd * xn - sd * cdx, vv = sd * yn - sd * cdy;
        const double ru[9] = {-x, -y, -1.0, 0.0, 0.0, 0.0, uu * x, uu * y, uu};
        const double rv[9] = {0.0, 0.0, 0.0, -x, -y, -1.0, vv * x, vv * y, vv};
        int e = 0;
        for (int a = 0; a < 9; ++a)
            for (int b = a; b < 9; ++b, ++e) G[e] += ru[a] * ru[b] + rv[a] * rv[b];
    }
    double A[81], tr = 0.0;
    {
        int e = 0;
        for (int a = 0; a < 9; ++a)
            for (int b = a; b < 9; ++b, ++e) {
                const double t = co.sum(G[e]);
                A[a * 9 + b] = t;
                A[b * 9 + a] = t;
            }
        for (int a = 0; a < 9; ++a) tr += A[a * 9 + a];
    }
    // smallest eigenvector by inverse iteration on G + 1e-14 trace I
    const double shift = 1e-14 * tr + 1e-300;
    for (int a = 0; a < 9; ++a) A[a * 9 + a] += shift;
    if (!seed_chol9(A)) return false;
    double h[9] = {0.37, -0.61, 0.83, 0.29, 0.71, -0.43, 0.53, -0.19, 0.97};
    for (int it = 0; it < 8; ++it) {
        chol_solve_n<9>(A, h);
        double nn = 0.0;
        for (int a = 0; a < 9; ++a) nn += h[a] * h[a];
        nn = 1.0 / sqrt(nn);
        for (int a = 0; a < 9; ++a) h[a] *= nn;
    }
    // Hn / Hn(2,2) (homographyestimator.cpp:70), then H = T_dst^-1 Hn T_src (:79-87)
    double Hn[9];
    for (int a = 0; a < 9; ++a) Hn[a] = h[a] / h[8];
    const double Ts[9] = {ss, 0.0, -ss * csx, 0.0, ss, -ss * csy, 0.0, 0.0, 1.0};
    const double Tdi[9] = {1.0 / sd, 0.0, cdx, 0.0, 1.0 / sd, cdy, 0.0, 0.0, 1.0};
    double T1[9], Hf[9];
    mat3_mul(Hn, Ts, T1);
    mat3_mul(Tdi, T1, Hf);
    for (int a = 0; a < 9; ++a)
        if (!(Hf[a] == Hf[a]) || fabs(Hf[a]) > 1.7e308) return false;  // non-finite homography (fit returns nullopt)
    for (int a = 0; a < 9; ++a) H[a] = Hf[a];
    return true;
}

// The whole seed of one view.  pose7 = [qw qx qy qz tx ty tz] (identity for < 4 points, planarpose_linear.cpp:55-57).
template <class Coop>
CBA_HD void planar_seed_view(int n, const double* X, const double* Y, const double* u, const double* v, const double* K, Coop& co,
                             double* pose7) {
    pose7[0] = 1.0;
    for (int k = 1; k < 7; ++k) pose7[k] = 0.0;
    double H[9];
    if (!dlt_homography_view(n, X, Y, u, v, K, co, H)) return;
    if (fabs(H[8]) > 1e-15) {  // planarpose_linear.cpp:72-74
        const double inv = 1.0 / H[8];
        for (int a = 0; a < 9; ++a) H[a] *= inv;
    }
    double R[9], t[3];
    seed_pose_from_h(H, R, t);
    seed_rotmat_to_quat(R, pose7);
    for (int k = 0; k < 3; ++k) pose7[4 + k] = t[k];
}

}  // namespace cba
