// schur_math.hpp — per-view / per-block bodies of the Schur-complement step, as
// __host__ __device__ inline code (the HIP kernels in backend_hip.hip are thin index wrappers;
// tests/cpu_backend runs the same bodies on the host to check the LM logic without a GPU).
//
// Per private view v (pose A of the INTRINSIC / EXTRINSIC chains), with w_b = rho'(s_b) the Huber
// weight of residual block b (ceres corrector with rho'' <= 0: J~ = sqrt(w) J, r~ = sqrt(w) r):
//   H_pp = sum_b w_b H_b[A,A]        g_p = sum_b w_b g_b[A]        E_b = w_b H_b[A, shared cols]
//   (H_pp + D_p) = L L^T             y = L^-1 g_p                  Z_b = L^-1 E_b
// Reduced system contributions: S -= sum_v Z_v^T Z_v,  g_red -= sum_v Z_v^T y_v, and the
// back-substitution  delta_p = -L^-T (y + Z_v delta_c).
// D_p is the Levenberg-Marquardt diagonal in Ceres' convention with Jacobi scaling
// (levenberg_marquardt_strategy.cc): D_ii = clamp(s_i^2 H_ii, 1e-6, 1e32) / (radius s_i^2),
// s_i = 1 / (1 + sqrt(H_ii(x0))).
#pragma once
#include "reproj_math.hpp"

namespace cba {

struct SchurDims {
    int PL, NH, NACC, PSH, PC, n_cams, chain;
};

CBA_HD int hidx(int PL, int i, int j) { return i * PL - i * (i - 1) / 2 + (j - i); }
CBA_HD int hidx_sym(int PL, int i, int j) { return i <= j ? hidx(PL, i, j) : hidx(PL, j, i); }

constexpr double LM_MIN_DIAG = 1e-6, LM_MAX_DIAG = 1e32;

CBA_HD double lm_diag(double hii, double scale2, double radius) {
    double ds = hii * scale2;
    ds = ds < LM_MIN_DIAG ? LM_MIN_DIAG : (ds > LM_MAX_DIAG ? LM_MAX_DIAG : ds);
    return ds / radius / scale2;
}

// in-place lower Cholesky of a 6x6 (row-major full storage); false if not positive definite
// (fp64 division is ~40 dependent instructions on the GPU: one reciprocal per pivot, multiplied through)
CBA_HD bool chol6(double* A) {
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        A[j * 6 + j] = d;
        const double r = 1.0 / d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * r;
        }
    }
    return true;
}
// b <- L^-1 b with the reciprocal diagonal rd[i] = 1 / L[i][i] supplied (many right-hand sides per factor)
CBA_HD void fwd6r(const double* L, const double* rd, double* b) {
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * b[k];
        b[i] = s * rd[i];
    }
}
CBA_HD void fwd6(const double* L, double* b) {  // b <- L^-1 b
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * b[k];
        b[i] = s / L[i * 6 + i];
    }
}
CBA_HD void bwd6(const double* L, double* b) {  // b <- L^-T b
    for (int i = 5; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * b[k];
        b[i] = s / L[i * 6 + i];
    }
}

// Private-view elimination.  Outputs: L (36, lower), y (6), D (6), gp (6), scale2 (6, written when
// init_scale), Z of every block of the view ([6][PSH] at blk_Z + b*6*PSH), *gmax = the view's
// contribution to Ceres' gradient max-norm.  Returns false if the damped H_pp is not PD.
// The elimination of one view in two pieces, so that a wavefront can share it (backend_hip.hip k_schur_view_wave: every lane
// runs the factor part redundantly — it is a latency chain, not work — and the lanes split the Z columns); the serial form below
// is the same pieces in sequence, so both give bit-identical results.
//   schur_view_factor: H_pp, g_p, scale, damping, Cholesky, y.  F (36) = the factor, rd (6) = reciprocal pivots, in the caller's
//   registers; L / y / D / gp / scale2 / gmax are written only when `store` (one lane).  Returns false if not positive definite.
CBA_HD bool schur_view_factor(const SchurDims& d, int nb, const int32_t* blks, const double* blk_acc, const double* blk_w,
                              double radius, bool init_scale, bool constrained, const double* xview7, double* scale2, double* L,
                              double* y, double* D, double* gp, double* gmax, bool store, double* F, double* rd) {
    double gl[6], sc2[6];
    for (int i = 0; i < 36; ++i) F[i] = 0.0;
    for (int i = 0; i < 6; ++i) gl[i] = 0.0;
    for (int k = 0; k < nb; ++k) {
        const int b = blks[k];
        const double w = blk_w[b];
        const double* acc = blk_acc + static_cast<long long>(b) * d.NACC;
        for (int i = 0; i < 6; ++i) {
            for (int j = i; j < 6; ++j) F[i * 6 + j] += w * acc[hidx(d.PL, i, j)];
            gl[i] += w * acc[d.NH + i];
        }
    }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < i; ++j) F[i * 6 + j] = F[j * 6 + i];
    for (int i = 0; i < 6; ++i) {
        if (init_scale) { const double s = 1.0 / (1.0 + sqrt(F[i * 6 + i])); sc2[i] = s * s; }
        else sc2[i] = scale2[i];
    }
    // gradient max-norm contribution (trust_region_minimizer.cc: inf-norm of the tangent gradient,
    // or of Plus(x, -g) - x when the problem is bounds-constrained)
    double gm = 0.0;
    if (!constrained) {
        for (int i = 0; i < 6; ++i) gm = fmax(gm, fabs(gl[i]));
    } else {
        const double ng[3] = {-gl[0], -gl[1], -gl[2]};
        double qn[4];
        quat_plus(xview7, ng, qn);
        for (int i = 0; i < 4; ++i) gm = fmax(gm, fabs(qn[i] - xview7[i]));
        for (int i = 3; i < 6; ++i) gm = fmax(gm, fabs(gl[i]));
    }
    double Dl[6];
    for (int i = 0; i < 6; ++i) {
        Dl[i] = lm_diag(F[i * 6 + i], sc2[i], radius);
        F[i * 6 + i] += Dl[i];
    }
    const bool ok = chol6(F);  // F now holds the factor: every substitution reads this local copy, not L
    double yl[6];
    for (int i = 0; i < 6; ++i) { yl[i] = gl[i]; rd[i] = 1.0 / F[i * 6 + i]; }
    if (ok) fwd6r(F, rd, yl);
    if (store) {
        *gmax = gm;
        for (int i = 0; i < 36; ++i) L[i] = F[i];
        for (int i = 0; i < 6; ++i) { gp[i] = gl[i]; D[i] = Dl[i]; y[i] = yl[i]; if (init_scale) scale2[i] = sc2[i]; }
    }
    return ok;
}
// column c of Z_b = L^-1 (w_b H_b[A, shared column c])
CBA_HD void schur_view_zcol(const SchurDims& d, const double* F, const double* rd, double w, const double* acc, int c, double* Z) {
    double e[6];
    for (int i = 0; i < 6; ++i) e[i] = w * acc[hidx(d.PL, i, 6 + c)];
    fwd6r(F, rd, e);
    for (int i = 0; i < 6; ++i) Z[i * d.PSH + c] = e[i];
}

CBA_HD bool schur_view_body(const SchurDims& d, int nb, const int32_t* blks, const double* blk_acc, const double* blk_w,
                            bool fixed, double radius, bool init_scale, bool constrained, const double* xview7,
                            double* scale2, double* L, double* y, double* D, double* gp, double* blk_Z, double* gmax) {
    if (fixed) {
        for (int i = 0; i < 36; ++i) L[i] = (i % 7 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 6; ++i) { y[i] = 0.0; D[i] = 0.0; gp[i] = 0.0; if (init_scale) scale2[i] = 1.0; }
        for (int k = 0; k < nb; ++k) {
            double* Z = blk_Z + static_cast<long long>(blks[k]) * 6 * d.PSH;
            for (int i = 0; i < 6 * d.PSH; ++i) Z[i] = 0.0;
        }
        *gmax = 0.0;
        return true;
    }
    double F[36], rd[6];
    if (!schur_view_factor(d, nb, blks, blk_acc, blk_w, radius, init_scale, constrained, xview7, scale2, L, y, D, gp, gmax, true, F, rd))
        return false;
    for (int k = 0; k < nb; ++k) {
        const int b = blks[k];
        for (int c = 0; c < d.PSH; ++c)
            schur_view_zcol(d, F, rd, blk_w[b], blk_acc + static_cast<long long>(b) * d.NACC, c, blk_Z + static_cast<long long>(b) * 6 * d.PSH);
    }
    return true;
}

// delta_p = -L^-T (y + sum_b Z_b delta_c[cols of b]); trial pose = Plus(x, delta_p).
// out4 = the view's share of [ |x_trial - x|^2, |x|^2 (ambient, 7 numbers), g^T d, d^T H d ] where the last
// two are the terms of Ceres' model_cost_change = -g^T d - 1/2 d^T H d that involve the private block:
//   g_p^T d_p   and   d_p^T H_pp d_p + 2 d_p^T E d_c  =  |rhs|^2 - d_p^T D d_p - 2 rhs^T a,
// with a = Z d_c, rhs = y + a = -L^T d_p, H_pp = L L^T - D, E = L Z.
// (two pieces again: a = Z d_c is a sum over the view's blocks and shared columns that a wavefront can split, the rest is a short
// serial chain — backend_hip.hip k_backsub_wave; the serial form is the two in sequence)
CBA_HD void backsub_view_finish(bool fixed, const double* a, const double* L, const double* y, const double* D, const double* gp,
                                const double* x7, double* delta_p, double* xt7, double* out4) {
    double xn = 0.0;
    for (int i = 0; i < 7; ++i) xn += x7[i] * x7[i];
    out4[1] = fixed ? 0.0 : xn;  // constant blocks are not part of Ceres' reduced state vector
    if (fixed) {
        for (int i = 0; i < 6; ++i) delta_p[i] = 0.0;
        for (int i = 0; i < 7; ++i) xt7[i] = x7[i];
        out4[0] = out4[2] = out4[3] = 0.0;
        return;
    }
    double rhs[6];
    double rr = 0.0, ra = 0.0;
    for (int i = 0; i < 6; ++i) { rhs[i] = y[i] + a[i]; rr += rhs[i] * rhs[i]; ra += rhs[i] * a[i]; }
    bwd6(L, rhs);
    double gd = 0.0, dDd = 0.0;
    for (int i = 0; i < 6; ++i) { delta_p[i] = -rhs[i]; gd += gp[i] * delta_p[i]; dDd += D[i] * delta_p[i] * delta_p[i]; }
    quat_plus(x7, delta_p, xt7);
    for (int i = 0; i < 3; ++i) xt7[4 + i] = x7[4 + i] + delta_p[3 + i];
    double s2 = 0.0;
    for (int i = 0; i < 7; ++i) s2 += (xt7[i] - x7[i]) * (xt7[i] - x7[i]);
    out4[0] = s2;
    out4[2] = gd;
    out4[3] = rr - dDd - 2.0 * ra;
}

CBA_HD void backsub_view_body(const SchurDims& d, int nb, const int32_t* blks, const int32_t* blk_cam, const double* blk_Z,
                              const double* delta_sh, bool fixed, const double* L, const double* y, const double* D,
                              const double* gp, const double* x7, double* delta_p, double* xt7, double* out4) {
    double a[6];
    for (int i = 0; i < 6; ++i) a[i] = 0.0;
    if (!fixed)
        for (int k = 0; k < nb; ++k) {
            const int b = blks[k];
            const double* Z = blk_Z + static_cast<long long>(b) * 6 * d.PSH;
            const double* dc = delta_sh + blk_cam[b] * d.PC;
            for (int i = 0; i < 6; ++i) {
                double s = 0.0;
                for (int c = 0; c < d.PSH; ++c) s += Z[i * d.PSH + c] * dc[c];
                a[i] += s;
            }
        }
    backsub_view_finish(fixed, a, L, y, D, gp, x7, delta_p, xt7, out4);
}

// Line search (line_search.hpp): the private pose of a view at step size a along the step of the last back-substitution,
// xt = Plus(x, a * delta_p); out2 = { |xt - x|^2, |x|^2 } (zero for a constant view, as in backsub_view_body)
CBA_HD void scale_step_view_body(bool fixed, double a, const double* x7, const double* delta_p, double* xt7, double* out2) {
    if (fixed) {
        for (int i = 0; i < 7; ++i) xt7[i] = x7[i];
        out2[0] = out2[1] = 0.0;
        return;
    }
    double d[6], xn = 0.0, s2 = 0.0;
    for (int i = 0; i < 6; ++i) d[i] = a * delta_p[i];
    quat_plus(x7, d, xt7);
    for (int i = 0; i < 3; ++i) xt7[4 + i] = x7[4 + i] + d[3 + i];
    for (int i = 0; i < 7; ++i) { xn += x7[i] * x7[i]; s2 += (xt7[i] - x7[i]) * (xt7[i] - x7[i]); }
    out2[0] = s2;
    out2[1] = xn;
}

// ... and the view's share of the directional derivative there: (sum over the view's blocks of w_b g_b[0..6)) . delta_p
CBA_HD double view_slope_body(const SchurDims& d, int nb, const int32_t* blks, const double* blk_acc, const double* blk_w, bool fixed,
                              const double* delta_p) {
    if (fixed) return 0.0;
    double s = 0.0;
    for (int k = 0; k < nb; ++k) {
        const int b = blks[k];
        const double* g = blk_acc + static_cast<long long>(b) * d.NACC + d.NH;
        double t = 0.0;
        for (int i = 0; i < 6; ++i) t += g[i] * delta_p[i];
        s += blk_w[b] * t;
    }
    return s;
}

// Z of view v at global shared column g, row k (0 if the view has no block on that column's camera)
CBA_HD double z_entry(const SchurDims& d, const int32_t* view_cam_blk, const double* blk_Z, int v, int g, int k, int nsh) {
    if (g >= nsh) return 0.0;
    const int cam = g / d.PC, lc = g - cam * d.PC;
    const int b = view_cam_blk[static_cast<long long>(v) * d.n_cams + cam];
    return b < 0 ? 0.0 : blk_Z[(static_cast<long long>(b) * 6 + k) * d.PSH + lc];
}

}  // namespace cba
