// reproj_math.hpp — per-observation reprojection residual + ANALYTIC tangent-space Jacobian.
//
// This is the arithmetic of the hot path, written once as __host__ __device__ inline code.  The
// HIP kernels (kernels_reproj.hip) call it per lane; tests/cpu_backend compiles the very same
// header with g++ to check the analytic derivatives against the oracle's dual numbers without a
// GPU.  It is NOT a CPU fallback: the product library only instantiates it in device code.
//
// What it replaces in the reference (evaluated there through ceres::Jet autodiff):
//   rigid chains     src/estimation/residuals/intrinsicresidual.h:20-35,
//                    extrinsicsresidual.h:14-46, bundleresidual.h:15-56
//   pose helpers     src/estimation/detail/observationutils.h:20-41
//   pinhole + BC     include/calib/models/pinhole.h:102-107, distortion.h:91-116,
//                    camera_matrix.h:41-46
//   Scheimpflug      include/calib/models/scheimpflug.h:139-181
//   manifold         ceres::QuaternionManifold: x+ = q(delta) (x) x, i.e. R+ = exp([2 delta]x) R
//
// Local tangent column order of one observation (cba_local_columns()):
//   [ poseA: d(3) t(3) | poseB: d(3) t(3) (EXTRINSIC/BUNDLE only) | intrinsics (10 | 12) ]
//   INTRINSIC: A = c_T_t                      EXTRINSIC: A = r_T_t, B = c_T_r
//   BUNDLE:    A = b_T_t, B = g_T_c
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define CBA_HD __host__ __device__ __forceinline__
#else
#define CBA_HD inline
#endif

namespace cba {

enum { CH_INTRINSIC = 0, CH_EXTRINSIC = 1, CH_BUNDLE = 2 };
enum { CAM_PINHOLE_BC = 0, CAM_SCHEIMPFLUG = 1 };

template <int MODEL> struct IntrSize { static constexpr int value = MODEL == CAM_SCHEIMPFLUG ? 12 : 10; };
template <int CHAIN, int MODEL> struct LocalCols {
    static constexpr int value = (CHAIN == CH_INTRINSIC ? 6 : 12) + IntrSize<MODEL>::value;
};

// ---- per-block constants (computed once per residual block by k_block_consts) -----------------
// P = X*m1 + Y*m2 + p0 is the camera-frame point of target point (X, Y, 0).
//   a1, a2 : columns 0,1 of R_A (rotation of pose A)         c = X a1 + Y a2 = R_A (X,Y,0)
//   M      : dP/d(t_A)                                        (I | R_cr | R_gc^T R_bg^T)
//   N, tb  : pose-B helpers (EXTRINSIC: tb = t_cr; BUNDLE: N = R_gc^T)
constexpr int BC_M1 = 0, BC_M2 = 3, BC_P0 = 6, BC_A1 = 9, BC_A2 = 12, BC_M = 15, BC_N = 24, BC_TB = 33, BC_SIZE = 36;

CBA_HD void quat_to_rotmat(const double* q, double* R) {
    // Eigen::Quaternion::toRotationMatrix, no normalisation (observationutils.h:20-24)
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
CBA_HD void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
CBA_HD void mat3_vec(const double* A, const double* x, double* y) {
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
CBA_HD void mat3_tvec(const double* A, const double* x, double* y) {  // y = A^T x
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

// poseA / poseB: 7-vectors [qw qx qy qz tx ty tz]; bTg: rotation row-major(9) + translation(3)
template <int CHAIN>
CBA_HD void block_consts(const double* poseA, const double* poseB, const double* bTg, double* bc) {
    double RA[9];
    quat_to_rotmat(poseA, RA);
    const double* tA = poseA + 4;
    for (int i = 0; i < 3; ++i) { bc[BC_A1 + i] = RA[3 * i]; bc[BC_A2 + i] = RA[3 * i + 1]; }
    for (int i = 0; i < 9; ++i) bc[BC_N + i] = 0.0;
    for (int i = 0; i < 3; ++i) bc[BC_TB + i] = 0.0;
    if (CHAIN == CH_INTRINSIC) {
        for (int i = 0; i < 9; ++i) bc[BC_M + i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) { bc[BC_M1 + i] = RA[3 * i]; bc[BC_M2 + i] = RA[3 * i + 1]; bc[BC_P0 + i] = tA[i]; }
    } else if (CHAIN == CH_EXTRINSIC) {
        // c_T_t = c_T_r * r_T_t (extrinsicsresidual.h:14-20, product(): observationutils.h:34-41)
        double RB[9], R[9], t[3];
        quat_to_rotmat(poseB, RB);
        mat3_mul(RB, RA, R);
        mat3_vec(RB, tA, t);
        for (int i = 0; i < 9; ++i) bc[BC_M + i] = RB[i];
        for (int i = 0; i < 3; ++i) {
            bc[BC_M1 + i] = R[3 * i]; bc[BC_M2 + i] = R[3 * i + 1];
            bc[BC_P0 + i] = t[i] + poseB[4 + i];
            bc[BC_TB + i] = poseB[4 + i];
        }
    } else {
        // c_T_t = (g_T_c)^-1 * (b_T_g)^-1 * b_T_t (bundleresidual.h:15-27)
        double RB[9], Rcg[9], tcg[3], Rgb[9], tgb[3], Rcb[9], tcb[3], R[9], t[3], tmp[3];
        quat_to_rotmat(poseB, RB);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Rcg[3 * i + j] = RB[3 * j + i]; Rgb[3 * i + j] = bTg[3 * j + i]; }
        mat3_vec(Rcg, poseB + 4, tmp); for (int i = 0; i < 3; ++i) tcg[i] = -tmp[i];
        mat3_vec(Rgb, bTg + 9, tmp);   for (int i = 0; i < 3; ++i) tgb[i] = -tmp[i];
        mat3_mul(Rcg, Rgb, Rcb);
        mat3_vec(Rcg, tgb, tmp);       for (int i = 0; i < 3; ++i) tcb[i] = tmp[i] + tcg[i];
        mat3_mul(Rcb, RA, R);
        mat3_vec(Rcb, tA, tmp);        for (int i = 0; i < 3; ++i) t[i] = tmp[i] + tcb[i];
        for (int i = 0; i < 9; ++i) { bc[BC_M + i] = Rcb[i]; bc[BC_N + i] = Rcg[i]; }
        for (int i = 0; i < 3; ++i) { bc[BC_M1 + i] = R[3 * i]; bc[BC_M2 + i] = R[3 * i + 1]; bc[BC_P0 + i] = t[i]; }
    }
}

// ---- per-camera derived constants for the Scheimpflug model ------------------------------------
// Rs (9) | dRs/dtau_x (9) | dRs/dtau_y (9) | m0 (2) | dm0/dtau_x (2) | dm0/dtau_y (2)
constexpr int SD_RS = 0, SD_DX = 9, SD_DY = 18, SD_M0 = 27, SD_DM0X = 29, SD_DM0Y = 31, SD_SIZE = 36;

CBA_HD void scheimpflug_consts(const double* intr, double* sd) {
    const double ctx = cos(intr[10]), stx = sin(intr[10]), cty = cos(intr[11]), sty = sin(intr[11]);
    // rot_sensor rows, scheimpflug.h:150-152
    const double Rs[9] = {cty, stx * sty, ctx * sty, 0.0, ctx, -stx, -sty, stx * cty, ctx * cty};
    const double Dx[9] = {0.0, ctx * sty, -stx * sty, 0.0, -stx, -ctx, 0.0, ctx * cty, -stx * cty};
    const double Dy[9] = {-sty, stx * cty, ctx * cty, 0.0, 0.0, 0.0, -cty, -stx * sty, -ctx * sty};
    for (int i = 0; i < 9; ++i) { sd[SD_RS + i] = Rs[i]; sd[SD_DX + i] = Dx[i]; sd[SD_DY + i] = Dy[i]; }
    const double s0 = Rs[8];
    const double mx0 = Rs[6] / s0, my0 = Rs[7] / s0;  // scheimpflug.h:165-167
    sd[SD_M0] = mx0; sd[SD_M0 + 1] = my0;
    sd[SD_DM0X] = (Dx[6] - mx0 * Dx[8]) / s0; sd[SD_DM0X + 1] = (Dx[7] - my0 * Dx[8]) / s0;
    sd[SD_DM0Y] = (Dy[6] - mx0 * Dy[8]) / s0; sd[SD_DM0Y + 1] = (Dy[7] - my0 * Dy[8]) / s0;
}

CBA_HD void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// Residual only.  bc: block constants, intr: camera params, sd: Scheimpflug constants (unused for
// pinhole), all already in the arithmetic type T (double, or float for the fp32 study of BASELINE
// config 5).  Returns r = (u - u_obs, v - v_obs).
template <int MODEL, typename T>
CBA_HD void reproj_residual(const T* bc, const T* intr, const T* sd, T X, T Y, T uo, T vo, T* r) {
    const T P0 = X * bc[BC_M1] + Y * bc[BC_M2] + bc[BC_P0];
    const T P1 = X * bc[BC_M1 + 1] + Y * bc[BC_M2 + 1] + bc[BC_P0 + 1];
    const T P2 = X * bc[BC_M1 + 2] + Y * bc[BC_M2 + 2] + bc[BC_P0 + 2];
    T x, y, su = T(0), sv = T(0);
    if (MODEL == CAM_PINHOLE_BC) {
        const T iz = T(1) / P2;
        x = P0 * iz; y = P1 * iz;
    } else {
        const T* Rs = sd + SD_RS;
        const T is = T(1) / (Rs[2] * P0 + Rs[5] * P1 + Rs[8] * P2);
        x = (Rs[0] * P0 + Rs[3] * P1 + Rs[6] * P2) * is - sd[SD_M0];
        y = (Rs[1] * P0 + Rs[4] * P1 + Rs[7] * P2) * is - sd[SD_M0 + 1];
        su = intr[0] * sd[SD_M0] + intr[4] * sd[SD_M0 + 1];
        sv = intr[1] * sd[SD_M0 + 1];
    }
    const T r2 = x * x + y * y;
    const T rad = T(1) + r2 * (intr[5] + r2 * (intr[6] + r2 * intr[7]));
    const T xy = x * y;
    const T xd = x * rad + T(2) * intr[8] * xy + intr[9] * (r2 + T(2) * x * x);
    const T yd = y * rad + intr[8] * (r2 + T(2) * y * y) + T(2) * intr[9] * xy;
    r[0] = (intr[0] * xd + intr[4] * yd + intr[2] + su) - uo;
    r[1] = (intr[1] * yd + intr[3] + sv) - vo;
}

template <typename T>
CBA_HD void cross3t(const T* a, const T* b, T* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T>
CBA_HD void mat3_tvec_t(const T* A, const T* x, T* y) {  // y = A^T x
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

// Everything of one observation that does not depend on the pose chain: residual r, the camera-frame point P,
// du = d u / d P and dv = d v / d P (3 each), and the intrinsics columns Jui / Jvi (IntrSize<MODEL> each).
template <int MODEL, typename T>
CBA_HD void reproj_core(const T* bc, const T* intr, const T* sd, T X, T Y, T uo, T vo, T* r, T* P, T* du, T* dv, T* Jui, T* Jvi) {
    const T fx = intr[0], fy = intr[1], skew = intr[4];
    const T k1 = intr[5], k2 = intr[6], k3 = intr[7], p1 = intr[8], p2 = intr[9];

    P[0] = X * bc[BC_M1] + Y * bc[BC_M2] + bc[BC_P0];
    P[1] = X * bc[BC_M1 + 1] + Y * bc[BC_M2 + 1] + bc[BC_P0 + 1];
    P[2] = X * bc[BC_M1 + 2] + Y * bc[BC_M2 + 2] + bc[BC_P0 + 2];
    // (x, y) = normalised coordinates fed to Brown-Conrady and d(x,y)/dP rows gx, gy
    T x, y, gx[3], gy[3], m0x = T(0), m0y = T(0), mx = T(0), my = T(0), is = T(0);
    if (MODEL == CAM_PINHOLE_BC) {
        const T iz = T(1) / P[2];
        x = P[0] * iz; y = P[1] * iz;
        gx[0] = iz; gx[1] = T(0); gx[2] = -x * iz;
        gy[0] = T(0); gy[1] = iz; gy[2] = -y * iz;
    } else {
        const T* Rs = sd + SD_RS;
        is = T(1) / (Rs[2] * P[0] + Rs[5] * P[1] + Rs[8] * P[2]);
        mx = (Rs[0] * P[0] + Rs[3] * P[1] + Rs[6] * P[2]) * is;
        my = (Rs[1] * P[0] + Rs[4] * P[1] + Rs[7] * P[2]) * is;
        m0x = sd[SD_M0]; m0y = sd[SD_M0 + 1];
        x = mx - m0x; y = my - m0y;
        for (int i = 0; i < 3; ++i) {
            gx[i] = (Rs[3 * i] - mx * Rs[3 * i + 2]) * is;
            gy[i] = (Rs[3 * i + 1] - my * Rs[3 * i + 2]) * is;
        }
    }
    const T r2 = x * x + y * y, xx = x * x, yy = y * y, xy = x * y;
    const T rad = T(1) + r2 * (k1 + r2 * (k2 + r2 * k3));
    const T drad = k1 + r2 * (T(2) * k2 + T(3) * k3 * r2);
    const T xd = x * rad + T(2) * p1 * xy + p2 * (r2 + T(2) * xx);
    const T yd = y * rad + p1 * (r2 + T(2) * yy) + T(2) * p2 * xy;
    r[0] = (fx * xd + skew * yd + intr[2] + (fx * m0x + skew * m0y)) - uo;
    r[1] = (fy * yd + intr[3] + fy * m0y) - vo;

    // d(xd,yd)/d(x,y)
    const T dxdx = rad + T(2) * xx * drad + T(2) * p1 * y + T(6) * p2 * x;
    const T dxdy = T(2) * xy * drad + T(2) * p1 * x + T(2) * p2 * y;  // = dyd/dx
    const T dydy = rad + T(2) * yy * drad + T(6) * p1 * y + T(2) * p2 * x;
    // d(u,v)/d(x,y)
    const T ux = fx * dxdx + skew * dxdy, uy = fx * dxdy + skew * dydy;
    const T vx = fy * dxdy, vy = fy * dydy;
    // d(u,v)/dP
    for (int i = 0; i < 3; ++i) { du[i] = ux * gx[i] + uy * gy[i]; dv[i] = vx * gx[i] + vy * gy[i]; }

    // ---- intrinsics [fx fy cx cy skew k1 k2 k3 p1 p2 (tau_x tau_y)] ----
    const T r4 = r2 * r2, r6 = r4 * r2;
    const T t1x = T(2) * xy, t1y = r2 + T(2) * yy;  // d(xd,yd)/dp1
    const T t2x = r2 + T(2) * xx, t2y = T(2) * xy;  // d(xd,yd)/dp2
    Jui[0] = xd + m0x; Jvi[0] = T(0);
    Jui[1] = T(0);     Jvi[1] = yd + m0y;
    Jui[2] = T(1);     Jvi[2] = T(0);
    Jui[3] = T(0);     Jvi[3] = T(1);
    Jui[4] = yd + m0y; Jvi[4] = T(0);
    const T gu = fx * x + skew * y, gv = fy * y;
    Jui[5] = gu * r2;  Jvi[5] = gv * r2;
    Jui[6] = gu * r4;  Jvi[6] = gv * r4;
    Jui[7] = gu * r6;  Jvi[7] = gv * r6;
    Jui[8] = fx * t1x + skew * t1y; Jvi[8] = fy * t1y;
    Jui[9] = fx * t2x + skew * t2y; Jvi[9] = fy * t2y;
    if (MODEL == CAM_SCHEIMPFLUG) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const T* D = sd + (k == 0 ? SD_DX : SD_DY);
            const T* dm0 = sd + (k == 0 ? SD_DM0X : SD_DM0Y);
            const T dnP = D[2] * P[0] + D[5] * P[1] + D[8] * P[2];
            const T dmx = ((D[0] * P[0] + D[3] * P[1] + D[6] * P[2]) - mx * dnP) * is;
            const T dmy = ((D[1] * P[0] + D[4] * P[1] + D[7] * P[2]) - my * dnP) * is;
            const T dx = dmx - dm0[0], dy = dmy - dm0[1];
            Jui[10 + k] = ux * dx + uy * dy + fx * dm0[0] + skew * dm0[1];
            Jvi[10 + k] = vx * dx + vy * dy + fy * dm0[1];
        }
    }
}

// Residual + Jacobian rows.  Ju/Jv: LocalCols<CHAIN,MODEL>::value entries each, [pose A d(3) t(3) | pose B d(3) t(3) | intr].
template <int CHAIN, int MODEL, typename T>
CBA_HD void reproj_point(const T* bc, const T* intr, const T* sd, T X, T Y, T uo, T vo, T* r, T* Ju, T* Jv) {
    constexpr int OI = CHAIN == CH_INTRINSIC ? 6 : 12;  // offset of the intrinsics columns
    T P[3], du[3], dv[3];
    reproj_core<MODEL, T>(bc, intr, sd, X, Y, uo, vo, r, P, du, dv, Ju + OI, Jv + OI);

    // ---- pose A: dP/d(delta_A) = -2 M [c]x, dP/d(t_A) = M,  c = X a1 + Y a2 ----
    const T c[3] = {X * bc[BC_A1] + Y * bc[BC_A2], X * bc[BC_A1 + 1] + Y * bc[BC_A2 + 1],
                    X * bc[BC_A1 + 2] + Y * bc[BC_A2 + 2]};
    T eu[3], ev[3], cr[3];
    if (CHAIN == CH_INTRINSIC) {
        for (int i = 0; i < 3; ++i) { eu[i] = du[i]; ev[i] = dv[i]; }
    } else {
        mat3_tvec_t(bc + BC_M, du, eu);  // row vector du^T M
        mat3_tvec_t(bc + BC_M, dv, ev);
    }
    cross3t(c, eu, cr); for (int i = 0; i < 3; ++i) { Ju[i] = T(2) * cr[i]; Ju[3 + i] = eu[i]; }
    cross3t(c, ev, cr); for (int i = 0; i < 3; ++i) { Jv[i] = T(2) * cr[i]; Jv[3 + i] = ev[i]; }

    // ---- pose B ----
    if (CHAIN == CH_EXTRINSIC) {
        // dP/d(delta_B) = -2 [P - t_cr]x, dP/d(t_B) = I
        const T w[3] = {P[0] - bc[BC_TB], P[1] - bc[BC_TB + 1], P[2] - bc[BC_TB + 2]};
        cross3t(w, du, cr); for (int i = 0; i < 3; ++i) { Ju[6 + i] = T(2) * cr[i]; Ju[9 + i] = du[i]; }
        cross3t(w, dv, cr); for (int i = 0; i < 3; ++i) { Jv[6 + i] = T(2) * cr[i]; Jv[9 + i] = dv[i]; }
    } else if (CHAIN == CH_BUNDLE) {
        // dP/d(delta_B) = 2 [P]x N, dP/d(t_B) = -N,  N = R_gc^T
        T h[3], hn[3], dn[3];
        cross3t(du, P, h); mat3_tvec_t(bc + BC_N, h, hn); mat3_tvec_t(bc + BC_N, du, dn);
        for (int i = 0; i < 3; ++i) { Ju[6 + i] = T(2) * hn[i]; Ju[9 + i] = -dn[i]; }
        cross3t(dv, P, h); mat3_tvec_t(bc + BC_N, h, hn); mat3_tvec_t(bc + BC_N, dv, dn);
        for (int i = 0; i < 3; ++i) { Jv[6 + i] = T(2) * hn[i]; Jv[9 + i] = -dn[i]; }
    }
}

// The pose columns of one observation are d(u,v)/dP times a 3 x 12 matrix that is AFFINE in the target point:
//     [Ju_pose; Jv_pose] = [du; dv]^T (G0 + X G1 + Y G2),      G_a constant over the residual block.
// (pose A: dP/d(delta_A) = -2 M [X a1 + Y a2]x, dP/d(t_A) = M; pose B, EXTRINSIC: -2 [X m1 + Y m2 + p0 - t_B]x and I;
// BUNDLE: 2 [X m1 + Y m2 + p0]x N and -N.)  Mode B of the two-pose chains therefore accumulates MOMENTS of
// Q = du du^T + dv dv^T etc. instead of the 12-column blocks (k_normal_eq_mom), and this routine expands a block's
// moments into the packed [H | g | s] row.  G[a] is row-major 3 x 12.
// G[a] alone (a = 0: constant term, 1: X, 2: Y): the three are independent, three lanes of k_mom_expand fill one each
template <int CHAIN>
CBA_HD void pose_affine_G_part(const double* bc, int a, double* Ga /*3 x 12 row-major*/) {
    for (int i = 0; i < 36; ++i) Ga[i] = 0.0;
    const double* M = bc + BC_M;
    auto skew_cols = [](const double* v, double* S /*3x3 row-major [v]x*/) {
        S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
    };
    if (a >= 1) {  // pose A rotation: -2 M [a_k]x for X (a1) and Y (a2)
        double S[9], MS[9];
        skew_cols(bc + (a == 1 ? BC_A1 : BC_A2), S);
        mat3_mul(M, S, MS);
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) Ga[k * 12 + i] = -2.0 * MS[k * 3 + i];
    } else {  // pose A translation: M (constant)
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) Ga[k * 12 + 3 + i] = M[k * 3 + i];
    }
    // pose B
    const double* vec = bc + (a == 0 ? BC_P0 : (a == 1 ? BC_M1 : BC_M2));
    double v[3] = {vec[0], vec[1], vec[2]}, S[9];
    if (CHAIN == CH_EXTRINSIC) {
        if (a == 0) for (int k = 0; k < 3; ++k) v[k] -= bc[BC_TB + k];
        skew_cols(v, S);
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) Ga[k * 12 + 6 + i] = -2.0 * S[k * 3 + i];
    } else {
        double SN[9];
        skew_cols(v, S);
        mat3_mul(S, bc + BC_N, SN);
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) Ga[k * 12 + 6 + i] = 2.0 * SN[k * 3 + i];
    }
    if (a == 0)
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i)
                Ga[k * 12 + 9 + i] = CHAIN == CH_EXTRINSIC ? (k == i ? 1.0 : 0.0) : -bc[BC_N + k * 3 + i];
}
template <int CHAIN>
CBA_HD void pose_affine_G(const double* bc, double G[3][36]) {
    for (int a = 0; a < 3; ++a) pose_affine_G_part<CHAIN>(bc, a, G[a]);
}

// moment-row layout of one residual block (PI = intrinsics size):
//   [ Qm 6x6 | qm 3x3 | Em 3 x 3 x PI | Hii PI(PI+1)/2 | gi PI | s ]
// Qm[(a,b)][(k,l)] over the pairs (0,0),(0,1),(0,2),(1,1),(1,2),(2,2) of m = (1, X, Y) and of the 3x3 symmetric Q.
template <int PI>
struct MomLayout {
    static constexpr int OFF_Q = 0, OFF_q = 36, OFF_E = 45, OFF_H = 45 + 9 * PI, OFF_G = OFF_H + PI * (PI + 1) / 2, OFF_S = OFF_G + PI,
                         N = OFF_S + 1;
};
CBA_HD int sym3(int k, int l) {  // 00 01 02 11 12 22
    const int a = k < l ? k : l, b = k < l ? l : k;
    return a == 0 ? b : (a == 1 ? 2 + b : 5);
}

// entry e of the packed [H | g | s] row (width NACC, PL = 12 + PI columns) from the block's moments
template <int PI>
CBA_HD double mom_expand_entry(const double* mom, const double G[3][36], int e) {
    using L = MomLayout<PI>;
    constexpr int PL = 12 + PI, NH = PL * (PL + 1) / 2;
    if (e == NH + PL) return mom[L::OFF_S];
    if (e >= NH) {  // gradient
        const int i = e - NH;
        if (i >= 12) return mom[L::OFF_G + i - 12];
        double s = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 3; ++k) s += G[a][k * 12 + i] * mom[L::OFF_q + a * 3 + k];
        return s;
    }
    // (i, j) of the upper triangle, row-major
    int i = 0, rem = e;
    while (rem >= PL - i) { rem -= PL - i; ++i; }
    const int j = i + rem;
    if (i >= 12) {  // intrinsics-intrinsics
        const int a = i - 12, b = j - 12;
        return mom[L::OFF_H + a * PI - a * (a - 1) / 2 + (b - a)];
    }
    if (j >= 12) {  // pose-intrinsics
        double s = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 3; ++k) s += G[a][k * 12 + i] * mom[L::OFF_E + (a * 3 + k) * PI + (j - 12)];
        return s;
    }
    double s = 0.0;  // pose-pose
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const double* Q = mom + L::OFF_Q + 6 * sym3(a, b);
            for (int k = 0; k < 3; ++k) {
                const double gk = G[a][k * 12 + i];
                if (gk == 0.0) continue;
                for (int l = 0; l < 3; ++l) s += gk * Q[sym3(k, l)] * G[b][l * 12 + j];
            }
        }
    return s;
}

// The same expansion in two stages (k_mom_expand): with the stacked 9 x 12 matrix Gh[(a,k)][i] = G[a][k*12+i] and the 9 x 9
// symmetric Qh[(a,k)][(b,l)] = Qm[(a,b)][(k,l)] the pose-pose block is Gh^T Qh Gh.  T = Qh Gh first (108 entries of 9 products),
// then every pose-pose entry is 9 products instead of 81.
template <int PI>
CBA_HD double mom_expand_T(const double* mom, const double G[3][36], int ak, int j) {
    using L = MomLayout<PI>;
    const int a = ak / 3, k = ak % 3;
    double s = 0.0;
    for (int b = 0; b < 3; ++b) {
        const double* Q = mom + L::OFF_Q + 6 * sym3(a, b);
        for (int l = 0; l < 3; ++l) s += Q[sym3(k, l)] * G[b][l * 12 + j];
    }
    return s;
}
// (i, j), i <= j, of entry e of a row-major packed upper triangle of width PL
template <int PL>
struct UpperIndex {
    unsigned char i[PL * (PL + 1) / 2], j[PL * (PL + 1) / 2];
    constexpr UpperIndex() : i(), j() {
        int e = 0;
        for (int r = 0; r < PL; ++r)
            for (int c = r; c < PL; ++c) { i[e] = static_cast<unsigned char>(r); j[e] = static_cast<unsigned char>(c); ++e; }
    }
};
template <int PI>
CBA_HD double mom_expand_entry_T(const double* mom, const double G[3][36], const double* T /*9 x 12*/, int e, int i, int j) {
    using L = MomLayout<PI>;
    constexpr int PL = 12 + PI, NH = PL * (PL + 1) / 2;
    if (e == NH + PL) return mom[L::OFF_S];
    if (e >= NH) {  // gradient
        const int g = e - NH;
        if (g >= 12) return mom[L::OFF_G + g - 12];
        double s = 0.0;
        for (int ak = 0; ak < 9; ++ak) s += G[ak / 3][(ak % 3) * 12 + g] * mom[L::OFF_q + ak];
        return s;
    }
    if (i >= 12) {  // intrinsics-intrinsics
        const int a = i - 12, b = j - 12;
        return mom[L::OFF_H + a * PI - a * (a - 1) / 2 + (b - a)];
    }
    double s = 0.0;
    if (j >= 12) {  // pose-intrinsics
        for (int ak = 0; ak < 9; ++ak) s += G[ak / 3][(ak % 3) * 12 + i] * mom[L::OFF_E + ak * PI + (j - 12)];
        return s;
    }
    for (int ak = 0; ak < 9; ++ak) s += G[ak / 3][(ak % 3) * 12 + i] * T[ak * 12 + j];  // pose-pose
    return s;
}

// The per-observation quantities the moment sums are built from ("moment rows"), in the order
//   w = [ r_u, r_v | du[3] | dv[3] | the intrinsics entries of the u row that are not structural constants | those of the v row ]
// Of the intrinsics columns [fx fy cx cy skew ...] the u row has no fy / cy entry and d u / d cx = 1, the v row has no fx / cx /
// skew entry and d v / d cy = 1 (both camera models), and d v / d fy IS d u / d skew (the same expression, yd + m0y: reproj_core):
// 2 * PI - 8 live entries.  A wavefront that evaluated an observation hands exactly these MomRows<PI>::N numbers to the wavefronts
// that accumulate other parts of the moment row (kernels_modeb.hip); the receiver takes Jv[fy] from Ju[skew].
template <int PI>
struct MomRows {
    static constexpr int NU = PI - 3, NV = PI - 5, N = 8 + NU + NV;
    static constexpr bool u_live(int c) { return !(c == 1 || c == 2 || c == 3); }
    static constexpr bool v_live(int c) { return !(c == 0 || c == 1 || c == 2 || c == 3 || c == 4); }
};

// Which wavefront ("part") of a Mode B workgroup keeps which entry of the moment row, and in which of its accumulators.
// The entries come in families that share per-observation products: the 3 PI entries (a, k, j) of Em with the same k share
// e_kj = du_k Jui_j + dv_k Jvi_j, the 36 of Qm share the six q_kl, the 9 of qm the three du_k r_u + dv_k r_v.  Dealing the entries
// out round-robin (entry e to part e mod NP, the first form of this kernel) made every part form nearly all of those products:
// 380 accumulate instructions per observation over the four parts where 276 suffice.  Here whole families go to one part
// (Em by k, Qm, qm, gradient + |r|^2; the rows of the intrinsics block, which share nothing, fill the parts up), assigned
// largest-first to the least loaded part (by instruction count) that has room for it.  NP = 1 gives the identity.
template <int PI, int NP>
struct MomSplitTable {
    static constexpr int N = MomLayout<PI>::N, NATOM = 6 + PI;
    short part[N], slot[N], count[NP], entry[NP][N];
    static constexpr bool hu(int j) { return !(j == 1 || j == 3); }
    static constexpr bool hv(int j) { return !(j == 0 || j == 2 || j == 4); }
    // atom -> instruction count (products + accumulations), as mom_accumulate spends them
    static constexpr int atom_cost(int t) {
        if (t < 3) {  // Em, fixed k
            int c = 3 * PI;
            for (int j = 0; j < PI; ++j) c += (hu(j) && j != 2 ? 1 : 0) + (hv(j) && j != 3 ? 1 : 0);
            return c;
        }
        if (t == 3) return 12 + 3 + 36;  // Qm
        if (t == 4) return 15;           // qm
        if (t == 5) {                    // gradient, |r|^2
            int c = 2;
            for (int j = 0; j < PI; ++j) c += (hu(j) ? 1 : 0) + (hv(j) ? 1 : 0);
            return c;
        }
        const int a = t - 6;             // row a of the intrinsics block
        int c = 0;
        for (int b = a; b < PI; ++b) c += (hu(a) && hu(b) ? 1 : 0) + (hv(a) && hv(b) ? 1 : 0);
        return c;
    }
    static constexpr int CAP = (N + NP - 1) / NP + 1;
    static constexpr int atom_size(int t) { return t < 3 ? 3 * PI : (t == 3 ? 36 : (t == 4 ? 9 : (t == 5 ? PI + 1 : PI - (t - 6)))); }
    static constexpr int atom_of(int e) {
        using L = MomLayout<PI>;
        if (e < L::OFF_q) return 3;
        if (e < L::OFF_E) return 4;
        if (e < L::OFF_H) return ((e - L::OFF_E) / PI) % 3;  // (a * 3 + k) * PI + j
        if (e >= L::OFF_G) return 5;
        int a = 0, rem = e - L::OFF_H;
        while (rem >= PI - a) { rem -= PI - a; ++a; }
        return 6 + a;
    }
    constexpr MomSplitTable() : part(), slot(), count(), entry() {
        int load[NP] = {}, fill[NP] = {}, where[NATOM] = {};
        bool done[NATOM] = {};
        for (int n = 0; n < NATOM; ++n) {
            int best = -1;
            for (int t = 0; t < NATOM; ++t)
                if (!done[t] && (best < 0 || atom_cost(t) > atom_cost(best))) best = t;
            // least loaded part among those with room: the accumulator count of a part is capped one above the even share (the
            // part with the most accumulators sets the kernel's register count)
            int p = -1;
            for (int q = 0; q < NP; ++q)
                if (fill[q] + atom_size(best) <= CAP && (p < 0 || load[q] < load[p])) p = q;
            if (p < 0) {
                p = 0;
                for (int q = 1; q < NP; ++q)
                    if (load[q] < load[p]) p = q;
            }
            fill[p] += atom_size(best);
            done[best] = true;
            where[best] = p;
            load[p] += atom_cost(best);
        }
        for (int e = 0; e < N; ++e) {
            const int p = where[atom_of(e)];
            part[e] = static_cast<short>(p);
            slot[e] = count[p];
            entry[p][count[p]] = static_cast<short>(e);
            ++count[p];
        }
    }
    constexpr int max_count() const {
        int m = 0;
        for (int p = 0; p < NP; ++p) m = count[p] > m ? count[p] : m;
        return m;
    }
};
template <int PI, int NP>
struct MomSplit {
    static constexpr MomSplitTable<PI, NP> T{};
};

template <int MODEL, typename T>
CBA_HD void mom_rows(const T* bc, const T* intr, const T* sd, T X, T Y, T uo, T vo, double* w) {
    constexpr int PI = IntrSize<MODEL>::value;
    using R = MomRows<PI>;
    T rt[2], Pt[3], dut[3], dvt[3], Juit[PI], Jvit[PI];
    reproj_core<MODEL, T>(bc, intr, sd, X, Y, uo, vo, rt, Pt, dut, dvt, Juit, Jvit);
    w[0] = rt[0]; w[1] = rt[1];
    for (int k = 0; k < 3; ++k) { w[2 + k] = dut[k]; w[5 + k] = dvt[k]; }
    int n = 8;
    for (int j = 0; j < PI; ++j)
        if (R::u_live(j)) w[n++] = Juit[j];
    for (int j = 0; j < PI; ++j)
        if (R::v_live(j)) w[n++] = Jvit[j];
}

// One observation's contribution to the entries of the moment row that part PART keeps (MomSplitTable: entry e lives in
// acc[slot[e]] of part[e]), from its moment rows w and its target point (x, y).
template <int PI, int NPARTS, int PART>
CBA_HD void mom_accumulate(const double* w, double x, double y, double* acc) {
    using L = MomLayout<PI>;
    using R = MomRows<PI>;
    const double ru = w[0], rv = w[1];
    double du[3], dv[3], Jui[PI], Jvi[PI];
    for (int k = 0; k < 3; ++k) { du[k] = w[2 + k]; dv[k] = w[5 + k]; }
    {
        int n = 8;
        for (int j = 0; j < PI; ++j) Jui[j] = R::u_live(j) ? w[n++] : (j == 2 ? 1.0 : 0.0);
        for (int j = 0; j < PI; ++j) Jvi[j] = R::v_live(j) ? w[n++] : (j == 3 ? 1.0 : 0.0);
        Jvi[1] = Jui[4];  // d v / d fy = d u / d skew (not shipped twice)
    }
    const double m[3] = {1.0, x, y};
    const double mm[6] = {1.0, x, y, x * x, x * y, y * y};
    using S = MomSplit<PI, NPARTS>;
#define CBA_ACC(E, VALUE) if (S::T.part[E] == PART) acc[S::T.slot[E]] += (VALUE)
#define CBA_FMA(E, A, B) if (S::T.part[E] == PART) acc[S::T.slot[E]] = __builtin_fma((A), (B), acc[S::T.slot[E]])
    {   // Q = du du^T + dv dv^T and its six moments
        int kl = 0;
        for (int k = 0; k < 3; ++k)
            for (int l = k; l < 3; ++l, ++kl) {
                const double q = __builtin_fma(dv[k], dv[l], du[k] * du[l]);
                CBA_ACC(L::OFF_Q + kl, q);
                for (int a = 1; a < 6; ++a) CBA_FMA(L::OFF_Q + a * 6 + kl, mm[a], q);
            }
    }
    for (int k = 0; k < 3; ++k) {  // q = du r_u + dv r_v
        const double q = __builtin_fma(dv[k], rv, du[k] * ru);
        CBA_ACC(L::OFF_q + k, q);
        CBA_FMA(L::OFF_q + 3 + k, m[1], q);
        CBA_FMA(L::OFF_q + 6 + k, m[2], q);
    }
    for (int k = 0; k < 3; ++k)  // E = du Jui^T + dv Jvi^T
        for (int j = 0; j < PI; ++j) {
            const bool hu = !(j == 1 || j == 3), hv = !(j == 0 || j == 2 || j == 4);
            double ekj = 0.0;
            if (hu) ekj = du[k] * Jui[j];
            if (hv) ekj = __builtin_fma(dv[k], Jvi[j], ekj);
            CBA_ACC(L::OFF_E + k * PI + j, ekj);
            CBA_FMA(L::OFF_E + (3 + k) * PI + j, m[1], ekj);
            CBA_FMA(L::OFF_E + (6 + k) * PI + j, m[2], ekj);
        }
    {   // intrinsics-intrinsics, gradient, |r|^2
        int e = L::OFF_H;
        for (int a = 0; a < PI; ++a)
            for (int b = a; b < PI; ++b, ++e) {
                const bool hu = !(a == 1 || a == 3) && !(b == 1 || b == 3), hv = !(a == 0 || a == 2 || a == 4) && !(b == 0 || b == 2 || b == 4);
                if (hu) CBA_FMA(e, Jui[a], Jui[b]);
                if (hv) CBA_FMA(e, Jvi[a], Jvi[b]);
            }
        for (int a = 0; a < PI; ++a) {
            if (!(a == 1 || a == 3)) CBA_FMA(L::OFF_G + a, Jui[a], ru);
            if (!(a == 0 || a == 2 || a == 4)) CBA_FMA(L::OFF_G + a, Jvi[a], rv);
        }
        CBA_FMA(L::OFF_S, ru, ru);
        CBA_FMA(L::OFF_S, rv, rv);
    }
#undef CBA_ACC
#undef CBA_FMA
}

// both in one go (the one-wavefront-per-tile kernels and the CPU test build)
template <int MODEL, int NPARTS, int PART, typename T>
CBA_HD void mom_point(const T* bc, const T* intr, const T* sd, T X, T Y, T uo, T vo, double* acc) {
    constexpr int PI = IntrSize<MODEL>::value;
    double w[MomRows<PI>::N];
    mom_rows<MODEL, T>(bc, intr, sd, X, Y, uo, vo, w);
    mom_accumulate<PI, NPARTS, PART>(w, static_cast<double>(X), static_cast<double>(Y), acc);
}

// ---- Huber (ceres::HuberLoss + Corrector with rho'' <= 0): weight = rho'(s), rho(s) ------------
CBA_HD void huber(double s, double delta, double* rho, double* w) {
    if (delta > 0.0 && s > delta * delta) {
        const double rt = sqrt(s);
        *rho = 2.0 * delta * rt - delta * delta;
        const double ww = delta / rt;
        *w = ww > 2.2250738585072014e-308 ? ww : 2.2250738585072014e-308;
    } else {
        *rho = s;
        *w = 1.0;
    }
}

// ---- ceres::QuaternionManifold::Plus ----------------------------------------------------------
CBA_HD void quat_plus(const double* q, const double* d, double* out) {
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n == 0.0) { for (int i = 0; i < 4; ++i) out[i] = q[i]; return; }
    const double s = sin(n) / n;
    const double a0 = cos(n), a1 = s * d[0], a2 = s * d[1], a3 = s * d[2];
    out[0] = a0 * q[0] - a1 * q[1] - a2 * q[2] - a3 * q[3];
    out[1] = a0 * q[1] + a1 * q[0] + a2 * q[3] - a3 * q[2];
    out[2] = a0 * q[2] - a1 * q[3] + a2 * q[0] + a3 * q[1];
    out[3] = a0 * q[3] + a1 * q[2] - a2 * q[1] + a3 * q[0];
}

}  // namespace cba
