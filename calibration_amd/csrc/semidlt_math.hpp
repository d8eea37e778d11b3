// semidlt_math.hpp — per-view bodies of optimize_intrinsics_semidlt (src/estimation/optim/intrinsicssemidlt.cpp:155-191,
// functor CalibVPResidual src/estimation/residuals/intrinsicsemidltresidual.h:19-73) as __host__ __device__ code.
//
// The reference's ONE residual block takes kappa = [fx fy cx cy skew] and a (quaternion, translation) pair per view,
// maps every target point to normalised coordinates (planar_observables_to_observables, observationutils.h:78-95),
// fits the Brown-Conrady coefficients alpha of ALL views at once by linear least squares
// (fit_distortion_full, include/calib/models/distortion.h:229-295) and returns r = A alpha - b; Ceres differentiates
// through the solve.  The analytic equivalent (Golub-Pereyra, full): with w_k = dA_k alpha - db_k (non-zero only on
// the rows that depend on parameter k), B = A^T W, D_k = dA_k^T r and N = A^T A,
//     J = W - A N^-1 (B + D)      =>      J^T J = W^T W - B^T N^-1 B + D^T N^-1 D,      J^T r = W^T r   (A^T r = 0).
// W^T W has the arrow structure of an intrinsics bundle (5 shared + 6 per view); the other two terms have rank <= m.
// So one evaluation is two passes over the observations with PER-VIEW sums only:
//   pass 1:  N_v, (A^T b)_v                       -> summed over views -> alpha
//   pass 2:  W^T W (11x11), W^T r (11), A^T W (m x 11), dA^T r (m x 11), |r|^2   per view
// Parameter order inside a view's 11: [fx fy cx cy skew | delta(3) t(3)], delta in the tangent space of
// ceres::QuaternionManifold (Plus(q, d) = q(d) * q), exactly as reproj_math.hpp.
#pragma once
#include "small_lm.hpp"
#include "vp_math.hpp"

namespace cba {

constexpr int SD_PV = 11;  // parameters one observation touches

template <int NR>
struct SDLayout {
    static constexpr int M = NR + 2;
    static constexpr int N1 = M * (M + 1) / 2 + M;  // pass 1: lower triangle of N (row-major) | A^T b
    static constexpr int OFF_G = SD_PV * (SD_PV + 1) / 2;  // 66
    static constexpr int OFF_B = OFF_G + SD_PV;            // B[a][k] at OFF_B + a * 11 + k
    static constexpr int OFF_D = OFF_B + SD_PV * M;
    static constexpr int OFF_S = OFF_D + SD_PV * M;
    static constexpr int N2 = OFF_S + 1;
};

struct SDView {
    int n;
    const double *X, *Y, *u, *v;
    double bc[BC_SIZE];  // block_consts<CH_INTRINSIC> of the view pose
};

// normalised coordinates of point i and (deriv) d(x, y)/d[delta(3) t(3)]
CBA_HD void sd_point(const SDView& V, int i, bool deriv, double* x, double* y, double* dx, double* dy) {
    const double X = V.X[i], Y = V.Y[i];
    const double* bc = V.bc;
    const double P0 = X * bc[BC_M1] + Y * bc[BC_M2] + bc[BC_P0];
    const double P1 = X * bc[BC_M1 + 1] + Y * bc[BC_M2 + 1] + bc[BC_P0 + 1];
    const double P2 = X * bc[BC_M1 + 2] + Y * bc[BC_M2 + 2] + bc[BC_P0 + 2];
    const double iz = 1.0 / P2;
    *x = P0 * iz; *y = P1 * iz;
    if (!deriv) return;
    const double gx[3] = {iz, 0.0, -(*x) * iz}, gy[3] = {0.0, iz, -(*y) * iz};
    const double c[3] = {X * bc[BC_A1] + Y * bc[BC_A2], X * bc[BC_A1 + 1] + Y * bc[BC_A2 + 1], X * bc[BC_A1 + 2] + Y * bc[BC_A2 + 2]};
    double cr[3];
    cross3(c, gx, cr); for (int k = 0; k < 3; ++k) { dx[k] = 2.0 * cr[k]; dx[3 + k] = gx[k]; }
    cross3(c, gy, cr); for (int k = 0; k < 3; ++k) { dy[k] = 2.0 * cr[k]; dy[3 + k] = gy[k]; }
}

// pass 1 of one view: out[N1] = [N lower | A^T b], written by lane 0 of the group
template <int NR, class Coop>
CBA_HD void sd_pass1(const SDView& V, const double* K5, Coop& co, double* out) {
    using L = SDLayout<NR>;
    constexpr int m = L::M;
    double acc[L::N1];
    for (int e = 0; e < L::N1; ++e) acc[e] = 0.0;
    VPRow R;
    for (int i = co.lane(); i < V.n; i += co.width()) {
        double x, y;
        sd_point(V, i, false, &x, &y, nullptr, nullptr);
        vp_design<NR>(K5, x, y, V.u[i], V.v[i], false, R);
        int e = 0;
        for (int a = 0; a < m; ++a)
            for (int c = 0; c <= a; ++c, ++e) acc[e] += R.Au[a] * R.Au[c] + R.Av[a] * R.Av[c];
        for (int a = 0; a < m; ++a) acc[e + a] += R.Au[a] * R.bu + R.Av[a] * R.bv;
    }
    for (int e = 0; e < L::N1; ++e) {
        const double t = co.sum(acc[e]);
        if (co.lane() == 0) out[e] = t;
    }
}

// residual-only pass: |A alpha - b|^2 of the view
template <int NR, class Coop>
CBA_HD double sd_resid(const SDView& V, const double* K5, const double* alpha, Coop& co) {
    constexpr int m = NR + 2;
    double s = 0.0;
    VPRow R;
    for (int i = co.lane(); i < V.n; i += co.width()) {
        double x, y;
        sd_point(V, i, false, &x, &y, nullptr, nullptr);
        vp_design<NR>(K5, x, y, V.u[i], V.v[i], false, R);
        double ru = -R.bu, rv = -R.bv;
        for (int a = 0; a < m; ++a) { ru += R.Au[a] * alpha[a]; rv += R.Av[a] * alpha[a]; }
        s += ru * ru + rv * rv;
    }
    return co.sum(s);
}

// pass 2 of one view, accumulator entries e with e % NPARTS == PART (a lane cannot hold all 78 + 22 m sums in
// registers: the kernel walks the view once per part and re-evaluates the cheap rows); out[e] written for those e.
template <int NR, int NPARTS, int PART, class Coop>
CBA_HD void sd_pass2_part(const SDView& V, const double* K5, const double* alpha, Coop& co, double* out) {
    using L = SDLayout<NR>;
    constexpr int m = L::M, PV = SD_PV;
    constexpr int NLOC = (L::N2 + NPARTS - 1) / NPARTS;
    double acc[NLOC];
    for (int e = 0; e < NLOC; ++e) acc[e] = 0.0;
    VPRow R;
    for (int i = co.lane(); i < V.n; i += co.width()) {
        double x, y, dx[6], dy[6];
        sd_point(V, i, true, &x, &y, dx, dy);
        vp_design<NR>(K5, x, y, V.u[i], V.v[i], true, R);
        double ru = -R.bu, rv = -R.bv, qux = -R.bux, quy = -R.buy, qvx = -R.bvx, qvy = -R.bvy;
        for (int a = 0; a < m; ++a) {
            ru += R.Au[a] * alpha[a]; rv += R.Av[a] * alpha[a];
            qux += R.Aux[a] * alpha[a]; quy += R.Auy[a] * alpha[a];
            qvx += R.Avx[a] * alpha[a]; qvy += R.Avy[a] * alpha[a];
        }
        // design rows per unit of fx / skew (u row) and fy (v row): Au = fx PX + skew PY, Av = fy PY
        double PX[m], PY[m], dxd = 0.0, dyd = 0.0;
        {
            const double r2 = x * x + y * y;
            double rpow = r2;
            for (int j = 0; j < NR; ++j) { PX[j] = x * rpow; PY[j] = y * rpow; rpow *= r2; }
            PX[NR] = 2.0 * x * y;          PY[NR] = r2 + 2.0 * y * y;
            PX[NR + 1] = r2 + 2.0 * x * x; PY[NR + 1] = 2.0 * x * y;
            for (int a = 0; a < m; ++a) { dxd += PX[a] * alpha[a]; dyd += PY[a] * alpha[a]; }
        }
        // w_k = dA_k alpha - db_k and dA_k (u row, v row) for the 11 parameters
        double wu[PV], wv[PV], dAu[PV][m], dAv[PV][m];
        wu[0] = dxd + x; wv[0] = 0.0;      // fx
        wu[1] = 0.0;     wv[1] = dyd + y;  // fy
        wu[2] = 1.0;     wv[2] = 0.0;      // cx
        wu[3] = 0.0;     wv[3] = 1.0;      // cy
        wu[4] = dyd + y; wv[4] = 0.0;      // skew
        for (int a = 0; a < m; ++a) {
            dAu[0][a] = PX[a]; dAv[0][a] = 0.0;
            dAu[1][a] = 0.0;   dAv[1][a] = PY[a];
            dAu[2][a] = 0.0;   dAv[2][a] = 0.0;
            dAu[3][a] = 0.0;   dAv[3][a] = 0.0;
            dAu[4][a] = PY[a]; dAv[4][a] = 0.0;
        }
        for (int k = 0; k < 6; ++k) {
            wu[5 + k] = qux * dx[k] + quy * dy[k];
            wv[5 + k] = qvx * dx[k] + qvy * dy[k];
            for (int a = 0; a < m; ++a) {
                dAu[5 + k][a] = R.Aux[a] * dx[k] + R.Auy[a] * dy[k];
                dAv[5 + k][a] = R.Avx[a] * dx[k] + R.Avy[a] * dy[k];
            }
        }
        int e = 0;
        for (int k = 0; k < PV; ++k)
            for (int l = k; l < PV; ++l, ++e)
                if (e % NPARTS == PART) acc[e / NPARTS] += wu[k] * wu[l] + wv[k] * wv[l];
        for (int k = 0; k < PV; ++k, ++e)
            if (e % NPARTS == PART) acc[e / NPARTS] += wu[k] * ru + wv[k] * rv;
        for (int a = 0; a < m; ++a)
            for (int k = 0; k < PV; ++k, ++e)
                if (e % NPARTS == PART) acc[e / NPARTS] += R.Au[a] * wu[k] + R.Av[a] * wv[k];
        for (int a = 0; a < m; ++a)
            for (int k = 0; k < PV; ++k, ++e)
                if (e % NPARTS == PART) acc[e / NPARTS] += dAu[k][a] * ru + dAv[k][a] * rv;
        if (e % NPARTS == PART) acc[e / NPARTS] += ru * ru + rv * rv;
    }
    for (int e = 0; e < L::N2; ++e)
        if (e % NPARTS == PART) {
            const double t = co.sum(acc[e / NPARTS]);
            if (co.lane() == 0) out[e] = t;
        }
}

}  // namespace cba
