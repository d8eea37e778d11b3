// oracle/models.hpp — TEST INFRASTRUCTURE ONLY.
//
// Scalar-templated restatement of the reference's projection math and pose
// helpers.  Every function cites the reference lines it follows.  T is double
// or orc::Jet<N>.  No Eigen: 3-vectors are T[3], 3x3 matrices are row-major
// T[9].
#pragma once
#include "jet.hpp"

namespace orc {

enum CameraModel { PINHOLE_BC = 0, SCHEIMPFLUG = 1 };
inline int intr_size(int model) { return model == SCHEIMPFLUG ? 12 : 10; }

// ---- pose helpers: src/estimation/detail/observationutils.h ----------------

// quat_array_to_rotmat, observationutils.h:20-24: Eigen::Quaternion<T>(w,x,y,z)
// .toRotationMatrix() with NO normalisation.  Eigen's formula (third-party,
// restated from Eigen/src/Geometry/Quaternion.h): tx=2x, ty=2y, tz=2z, ...
template <typename T>
inline void quat_to_rotmat(const T* q, T* R) {
    const T w = q[0], x = q[1], y = q[2], z = q[3];
    const T tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const T twx = tx * w, twy = ty * w, twz = tz * w;
    const T txx = tx * x, txy = ty * x, txz = tz * x;
    const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

template <typename T>
inline void mat3_mul(const T* A, const T* B, T* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
template <typename T>
inline void mat3_vec(const T* A, const T* x, T* y) {
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
template <typename T>
inline void mat3_transpose(const T* A, T* At) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) At[3 * i + j] = A[3 * j + i];
}

// invert_transform, observationutils.h:26-32: (R,t)^-1 = (R^T, -R^T t)
template <typename T>
inline void invert_transform(const T* R, const T* t, T* Ri, T* ti) {
    mat3_transpose(R, Ri);
    T tmp[3];
    mat3_vec(Ri, t, tmp);
    for (int i = 0; i < 3; ++i) ti[i] = -tmp[i];
}

// product, observationutils.h:34-41: (R1,t1)*(R2,t2) = (R1 R2, R1 t2 + t1)
template <typename T>
inline void product(const T* R1, const T* t1, const T* R2, const T* t2, T* R, T* t) {
    mat3_mul(R1, R2, R);
    T tmp[3];
    mat3_vec(R1, t2, tmp);
    for (int i = 0; i < 3; ++i) t[i] = tmp[i] + t1[i];
}

// ---- Brown-Conrady: include/calib/models/distortion.h:91-116 ---------------
// coeffs = [k1..kn, p1, p2]; the camera traits always pass n = 3
// (pinhole.h:117-133, k_num_dist_coeffs = 5).
template <typename T>
inline void apply_distortion(const T& x, const T& y, const T* coeffs, int num_coeffs, T* xd, T* yd) {
    const int num_radial = num_coeffs - 2;
    T r2 = x * x + y * y;
    T radial = T(1.0);
    T rpow = r2;
    for (int i = 0; i < num_radial; ++i) {
        radial += coeffs[i] * rpow;
        rpow *= r2;
    }
    const T p1 = coeffs[num_radial];
    const T p2 = coeffs[num_radial + 1];
    *xd = x * radial + T(2.0) * p1 * x * y + p2 * (r2 + T(2.0) * x * x);
    *yd = y * radial + p1 * (r2 + T(2.0) * y * y) + T(2.0) * p2 * x * y;
}

// denormalize, include/calib/models/camera_matrix.h:41-46
template <typename T>
inline void denormalize(const T& fx, const T& fy, const T& cx, const T& cy, const T& skew,
                        const T& x, const T& y, T* u, T* v) {
    *u = fx * x + skew * y + cx;
    *v = fy * y + cy;
}

// PinholeCamera::project(Vec3), include/calib/models/pinhole.h:102-107;
// param vector [fx,fy,cx,cy,skew,k1,k2,k3,p1,p2], pinhole.h:117-133.
template <typename T>
inline void project_pinhole(const T* intr, const T* P, T* uv) {
    const T x = P[0] / P[2];  // hnormalized()
    const T y = P[1] / P[2];
    T xd, yd;
    apply_distortion(x, y, intr + 5, 5, &xd, &yd);
    denormalize(intr[0], intr[1], intr[2], intr[3], intr[4], xd, yd, &uv[0], &uv[1]);
}

// ScheimpflugCamera::project, include/calib/models/scheimpflug.h:139-181;
// params = pinhole 10 + [tau_x, tau_y], scheimpflug.h:234-247.
template <typename T>
inline void project_scheimpflug(const T* intr, const T* P, T* uv) {
    const T tau_x = intr[10], tau_y = intr[11];
    const T ctx = cos(tau_x), stx = sin(tau_x), cty = cos(tau_y), sty = sin(tau_y);
    // rot_sensor rows (scheimpflug.h:150-152)
    const T Rs[9] = {cty,  stx * sty, ctx * sty,
                     T(0.0), ctx,     -stx,
                     -sty, stx * cty, ctx * cty};
    // columns: axis (0), base (1), normal (2)
    const T sden = Rs[2] * P[0] + Rs[5] * P[1] + Rs[8] * P[2];
    const T mx = (Rs[0] * P[0] + Rs[3] * P[1] + Rs[6] * P[2]) / sden;
    const T my = (Rs[1] * P[0] + Rs[4] * P[1] + Rs[7] * P[2]) / sden;
    const T s0 = Rs[8];
    const T mx0 = Rs[6] / s0;
    const T my0 = Rs[7] / s0;
    // camera.project(Vec3(dx, dy, 1)) goes through hnormalized(): divide by 1
    const T dP[3] = {mx - mx0, my - my0, T(1.0)};
    T px[2];
    project_pinhole(intr, dP, px);
    // apply_linear_intrinsics, pinhole.h:148-153: fx, fy, skew only
    T su, sv;
    denormalize(intr[0], intr[1], T(0.0), T(0.0), intr[4], mx0, my0, &su, &sv);
    uv[0] = px[0] + su;
    uv[1] = px[1] + sv;
}

template <typename T>
inline void project(int model, const T* intr, const T* P, T* uv) {
    if (model == SCHEIMPFLUG) project_scheimpflug(intr, P, uv);
    else project_pinhole(intr, P, uv);
}

// ---- rigid chains ----------------------------------------------------------
enum Chain { CHAIN_INTRINSIC = 0, CHAIN_EXTRINSIC = 1, CHAIN_BUNDLE = 2 };

// intrinsicresidual.h:22-23: c_T_t is the view's own pose
template <typename T>
inline void chain_intrinsic(const T* q, const T* t, T* R, T* tr) {
    quat_to_rotmat(q, R);
    for (int i = 0; i < 3; ++i) tr[i] = t[i];
}
// extrinsicsresidual.h:14-20,30-33: c_T_t = c_T_r * r_T_t
template <typename T>
inline void chain_extrinsic(const T* c_q_r, const T* c_t_r, const T* r_q_t, const T* r_t_t, T* R, T* tr) {
    T Rcr[9], Rrt[9];
    quat_to_rotmat(c_q_r, Rcr);
    quat_to_rotmat(r_q_t, Rrt);
    product(Rcr, c_t_r, Rrt, r_t_t, R, tr);
}
// bundleresidual.h:15-27,39-43: c_T_t = (g_T_c)^-1 * (b_T_g)^-1 * b_T_t
template <typename T>
inline void chain_bundle(const T* b_q_t, const T* b_t_t, const T* g_q_c, const T* g_t_c,
                         const double* b_R_g, const double* b_t_g, T* R, T* tr) {
    T Rbt[9], Rgc[9], Rbg[9], tbg[3];
    quat_to_rotmat(b_q_t, Rbt);
    quat_to_rotmat(g_q_c, Rgc);
    for (int i = 0; i < 9; ++i) Rbg[i] = T(b_R_g[i]);
    for (int i = 0; i < 3; ++i) tbg[i] = T(b_t_g[i]);
    T Rcg[9], tcg[3], Rgb[9], tgb[3], Rcb[9], tcb[3];
    invert_transform(Rgc, g_t_c, Rcg, tcg);
    invert_transform(Rbg, tbg, Rgb, tgb);
    product(Rcg, tcg, Rgb, tgb, Rcb, tcb);
    product(Rcb, tcb, Rbt, b_t_t, R, tr);
}

// one observation: intrinsicresidual.h:27-33 (identical loop body in
// extrinsicsresidual.h:37-44 and bundleresidual.h:47-53)
template <typename T>
inline void reproject_point(int model, const T* intr, const T* R, const T* tr, double X, double Y,
                            double u, double v, T* r2) {
    // point = (X, Y, 0); point = R*point + t
    const T P[3] = {R[0] * X + R[1] * Y + tr[0], R[3] * X + R[4] * Y + tr[1],
                    R[6] * X + R[7] * Y + tr[2]};
    T uv[2];
    project(model, intr, P, uv);
    r2[0] = uv[0] - u;
    r2[1] = uv[1] - v;
}

// ---- Eigen matrix -> quaternion -> angle-axis (third-party, restated) ------
// Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Other,3,3>
template <typename T>
inline void rotmat_to_quat(const T* m, T* q /*w,x,y,z*/) {
    T t = m[0] + m[4] + m[8];
    if (t > T(0.0)) {
        t = sqrt(t + T(1.0));
        q[0] = T(0.5) * t;
        t = T(0.5) / t;
        q[1] = (m[7] - m[5]) * t;
        q[2] = (m[2] - m[6]) * t;
        q[3] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3;
        const int k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + T(1.0));
        q[1 + i] = T(0.5) * t;
        t = T(0.5) / t;
        q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
// Eigen/src/Geometry/AngleAxis.h AngleAxis::operator=(QuaternionBase)
template <typename T>
inline void quat_to_angle_axis(const T* q, T* angle, T* axis) {
    T n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    // (stableNorm fallback for n < eps is value-identical for our purposes)
    if (n != 0.0) {
        *angle = T(2.0) * atan2(n, abs(q[0]));
        if (q[0] < T(0.0)) n = -n;
        axis[0] = q[1] / n; axis[1] = q[2] / n; axis[2] = q[3] / n;
    } else {
        *angle = T(0.0);
        axis[0] = T(1.0); axis[1] = T(0.0); axis[2] = T(0.0);
    }
}

// AxXbResidual::operator(), src/estimation/residuals/handeyeresidual.h:25-49
template <typename T>
inline void axxb_residual(const T* q, const T* t, const double* RA, const double* RB,
                          const double* tA, const double* tB, T* r6) {
    T RX[9], A[9], Bt[9], RXt[9];
    quat_to_rotmat(q, RX);
    for (int i = 0; i < 9; ++i) A[i] = T(RA[i]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Bt[3 * i + j] = T(RB[3 * j + i]);
    mat3_transpose(RX, RXt);
    T M1[9], M2[9], RS[9];
    mat3_mul(A, RX, M1);
    mat3_mul(M1, Bt, M2);
    mat3_mul(M2, RXt, RS);
    T qs[4], angle, axis[3];
    rotmat_to_quat(RS, qs);
    quat_to_angle_axis(qs, &angle, axis);
    // tra_e = (rot_a - I) tra_x - (rot_x tra_b - tra_a)
    T AmI[9];
    for (int i = 0; i < 9; ++i) AmI[i] = A[i];
    AmI[0] = AmI[0] - T(1.0); AmI[4] = AmI[4] - T(1.0); AmI[8] = AmI[8] - T(1.0);
    T e1[3], e2[3];
    mat3_vec(AmI, t, e1);
    const T tb[3] = {T(tB[0]), T(tB[1]), T(tB[2])};
    mat3_vec(RX, tb, e2);
    r6[0] = angle * axis[0];
    r6[1] = angle * axis[1];
    r6[2] = angle * axis[2];
    for (int i = 0; i < 3; ++i) r6[3 + i] = e1[i] - (e2[i] - T(tA[i]));
}

// ---- ceres::QuaternionManifold (third-party, restated) ---------------------
// Plus(q, d) = q_d (x) q with q_d = [cos|d|, sin|d|/|d| * d]; identity if |d|=0.
inline void quat_plus(const double* q, const double* d, double* out) {
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n == 0.0) {
        for (int i = 0; i < 4; ++i) out[i] = q[i];
        return;
    }
    const double s = std::sin(n) / n;
    const double qd[4] = {std::cos(n), s * d[0], s * d[1], s * d[2]};
    // Hamilton product qd * q, storage (w,x,y,z)
    out[0] = qd[0] * q[0] - qd[1] * q[1] - qd[2] * q[2] - qd[3] * q[3];
    out[1] = qd[0] * q[1] + qd[1] * q[0] + qd[2] * q[3] - qd[3] * q[2];
    out[2] = qd[0] * q[2] - qd[1] * q[3] + qd[2] * q[0] + qd[3] * q[1];
    out[3] = qd[0] * q[3] + qd[1] * q[2] - qd[2] * q[1] + qd[3] * q[0];
}
// PlusJacobian at q (4x3, row-major): d(q_d (x) q)/dd at d = 0
inline void quat_plus_jacobian(const double* q, double* J43) {
    J43[0] = -q[1]; J43[1] = -q[2]; J43[2] = -q[3];
    J43[3] = q[0];  J43[4] = q[3];  J43[5] = -q[2];
    J43[6] = -q[3]; J43[7] = q[0];  J43[8] = q[1];
    J43[9] = q[2];  J43[10] = -q[1]; J43[11] = q[0];
}

}  // namespace orc
