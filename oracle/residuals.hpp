// oracle/residuals.hpp — TEST INFRASTRUCTURE ONLY.
//
// The reference's Ceres cost functors, evaluated with orc::Jet exactly as
// ceres::AutoDiffCostFunction would: one residual block per VIEW with 2N
// residuals (intrinsicresidual.h:37-48, extrinsicsresidual.h:48-59,
// bundleresidual.h:58-68) and one 6-residual block per motion pair
// (handeyeresidual.h:51-53).
#pragma once
#include <limits>
#include <stdexcept>
#include <vector>

#include "lm.hpp"
#include "models.hpp"

namespace orc {

struct ViewData {
    int n = 0;
    const double *X = nullptr, *Y = nullptr, *u = nullptr, *v = nullptr;
};

// Generic driver: evaluates `fn(params as T*, point index) -> r2` per point with
// NJ-wide jets over the concatenated parameter blocks.
template <int CHAIN, int MODEL>
struct ReprojBlock final : ResidualBlock {
    static constexpr int PI = MODEL == SCHEIMPFLUG ? 12 : 10;
    static constexpr int NJ = (CHAIN == CHAIN_INTRINSIC ? 7 : 14) + PI;
    ViewData view;
    double bRg[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // bundle: b_T_g rotation (row-major)
    double btg[3] = {0, 0, 0};                    // bundle: b_T_g translation

    explicit ReprojBlock(const ViewData& v) : view(v) {
        // create(): throws on empty view (intrinsicresidual.h:38-40 etc.)
        if (v.n <= 0) throw std::invalid_argument("No observations provided");
        nres = 2 * v.n;
    }

    template <typename T>
    void residuals(const T* const* p, T* r) const {
        T R[9], t[3];
        const T* intr;
        if constexpr (CHAIN == CHAIN_INTRINSIC) {
            chain_intrinsic(p[0], p[1], R, t);
            intr = p[2];
        } else if constexpr (CHAIN == CHAIN_EXTRINSIC) {
            chain_extrinsic(p[0], p[1], p[2], p[3], R, t);
            intr = p[4];
        } else {
            chain_bundle(p[0], p[1], p[2], p[3], bRg, btg, R, t);
            intr = p[4];
        }
        for (int i = 0; i < view.n; ++i)
            reproject_point(MODEL, intr, R, t, view.X[i], view.Y[i], view.u[i], view.v[i], r + 2 * i);
    }

    void evaluate(const double* const* x, double* r, double** J) const override {
        constexpr int NB = CHAIN == CHAIN_INTRINSIC ? 3 : 5;
        static const int sizes3[3] = {4, 3, PI};
        static const int sizes5[5] = {4, 3, 4, 3, PI};
        const int* sizes = CHAIN == CHAIN_INTRINSIC ? sizes3 : sizes5;
        if (!J) {
            residuals<double>(x, r);
            return;
        }
        using JT = Jet<NJ>;
        std::vector<JT> storage(NJ);
        const JT* ptr[5];
        int off = 0;
        for (int b = 0; b < NB; ++b) {
            ptr[b] = &storage[off];
            for (int k = 0; k < sizes[b]; ++k) storage[off + k] = JT(x[b][k], off + k);
            off += sizes[b];
        }
        std::vector<JT> rj(nres);
        residuals<JT>(ptr, rj.data());
        for (int i = 0; i < nres; ++i) {
            r[i] = rj[i].a;
            off = 0;
            for (int b = 0; b < NB; ++b) {
                if (J[b])
                    for (int k = 0; k < sizes[b]; ++k) J[b][static_cast<size_t>(i) * sizes[b] + k] = rj[i].v[off + k];
                off += sizes[b];
            }
        }
    }
};

struct AxXbBlock final : ResidualBlock {
    double RA[9], RB[9], tA[3], tB[3];
    AxXbBlock(const double* ra, const double* rb, const double* ta, const double* tb) {
        for (int i = 0; i < 9; ++i) { RA[i] = ra[i]; RB[i] = rb[i]; }
        for (int i = 0; i < 3; ++i) { tA[i] = ta[i]; tB[i] = tb[i]; }
        nres = 6;
    }
    void evaluate(const double* const* x, double* r, double** J) const override {
        if (!J) {
            axxb_residual<double>(x[0], x[1], RA, RB, tA, tB, r);
            return;
        }
        using JT = Jet<7>;
        JT q[4], t[3], rj[6];
        for (int k = 0; k < 4; ++k) q[k] = JT(x[0][k], k);
        for (int k = 0; k < 3; ++k) t[k] = JT(x[1][k], 4 + k);
        axxb_residual<JT>(q, t, RA, RB, tA, tB, rj);
        for (int i = 0; i < 6; ++i) {
            r[i] = rj[i].a;
            if (J[0]) for (int k = 0; k < 4; ++k) J[0][i * 4 + k] = rj[i].v[k];
            if (J[1]) for (int k = 0; k < 3; ++k) J[1][i * 3 + k] = rj[i].v[4 + k];
        }
    }
};

// ceres::AngleAxisRotatePoint (third-party ceres/rotation.h, restated): Rodrigues for theta^2 > eps,
// first-order Taylor otherwise.
template <typename T>
inline void angle_axis_rotate_point(const T* aa, const T* pt, T* out) {
    const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > T(2.220446049250313e-16)) {
        const T theta = sqrt(theta2);
        const T c = cos(theta), s = sin(theta), ti = T(1.0) / theta;
        const T w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
        const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
        const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - c);
        for (int i = 0; i < 3; ++i) out[i] = pt[i] * c + wxp[i] * s + w[i] * tmp;
    } else {
        const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
        for (int i = 0; i < 3; ++i) out[i] = pt[i] + wxp[i];
    }
}

// argmin |A x - b| the way the reference computes it (models/distortion.h:290-294): thin SVD of the design matrix and
// x = V diag(1 / sigma_j) U^T b over the singular values above Eigen's default rank threshold (SVDBase::threshold():
// min(rows, cols) * epsilon, relative to the largest) - the MINIMUM-NORM solution when A is rank deficient (near-collinear
// points), where normal equations would divide by a rounding-noise pivot.  The SVD is a one-sided Jacobi (Hestenes) iteration on
// the columns of A - every operation on T, so the derivative parts of Jets ride through the rotations exactly as they ride
// through Eigen::JacobiSVD in the reference (the rotation pattern is decided by the scalar parts).  Eigen is a third-party
// dependency absent from /root/reference: restated, like the rest of this directory.
// A is rows x m row-major (destroyed: it leaves as U Sigma), m <= 8.
template <typename T>
void lstsq_svd(std::vector<T>& A, int rows, int m, const std::vector<T>& b, T* x) {
    std::vector<T> V(static_cast<size_t>(m) * m, T(0.0));
    for (int j = 0; j < m; ++j) V[j * m + j] = T(1.0);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < m - 1; ++p)
            for (int q = p + 1; q < m; ++q) {
                T al(0.0), be(0.0), ga(0.0);
                for (int r = 0; r < rows; ++r) {
                    const T& wp = A[static_cast<size_t>(r) * m + p];
                    const T& wq = A[static_cast<size_t>(r) * m + q];
                    al = al + wp * wp; be = be + wq * wq; ga = ga + wp * wq;
                }
                const double a0 = scalar_of(al), b0 = scalar_of(be), g0 = scalar_of(ga);
                if (a0 == 0.0 || b0 == 0.0) continue;
                const double rel = std::fabs(g0) / std::sqrt(a0 * b0);
                off = std::max(off, rel);
                if (rel <= 1e-16) continue;
                const T zeta = (be - al) / (T(2.0) * ga);
                const T t = (scalar_of(zeta) >= 0.0 ? T(1.0) : T(-1.0)) / (abs(zeta) + sqrt(T(1.0) + zeta * zeta));
                const T c = T(1.0) / sqrt(T(1.0) + t * t), sn = c * t;
                for (int r = 0; r < rows; ++r) {
                    const T wp = A[static_cast<size_t>(r) * m + p], wq = A[static_cast<size_t>(r) * m + q];
                    A[static_cast<size_t>(r) * m + p] = c * wp - sn * wq;
                    A[static_cast<size_t>(r) * m + q] = sn * wp + c * wq;
                }
                for (int r = 0; r < m; ++r) {
                    const T vp = V[r * m + p], vq = V[r * m + q];
                    V[r * m + p] = c * vp - sn * vq;
                    V[r * m + q] = sn * vp + c * vq;
                }
            }
        if (off <= 1e-15) break;
    }
    std::vector<T> s2(m, T(0.0)), wb(m, T(0.0));
    double smax2 = 0.0;
    for (int j = 0; j < m; ++j) {
        for (int r = 0; r < rows; ++r) {
            s2[j] = s2[j] + A[static_cast<size_t>(r) * m + j] * A[static_cast<size_t>(r) * m + j];
            wb[j] = wb[j] + A[static_cast<size_t>(r) * m + j] * b[r];
        }
        smax2 = std::max(smax2, scalar_of(s2[j]));
    }
    const double thr = static_cast<double>(std::min(rows, m)) * 2.220446049250313e-16;
    for (int i = 0; i < m; ++i) x[i] = T(0.0);
    for (int j = 0; j < m; ++j) {
        if (!(std::sqrt(scalar_of(s2[j])) > thr * std::sqrt(smax2))) continue;  // below the rank threshold: left out (minimum norm)
        const T coef = wb[j] / s2[j];
        for (int i = 0; i < m; ++i) x[i] = x[i] + V[i * m + j] * coef;
    }
}

// PlanarPoseVPResidual::operator(), src/estimation/optim/planarpose.cpp:39-57:
//   to_observation (observationutils.h:97-113) -> fit_distortion_full (models/distortion.h:229-295)
//   -> residual = A alpha - b with alpha = argmin |A alpha - b|.
// The reference solves the least squares with a thin JacobiSVD on Jets (distortion.h:290-294): so does the oracle (lstsq_svd
// above), including the minimum-norm behaviour on rank-deficient designs.  (The product uses normal equations and an analytic
// Golub-Pereyra derivative: the same minimiser at full column rank, compared in tests/.)
struct PlanarPoseVPBlock final : ResidualBlock {
    ViewData view;
    double K[5];  // fx fy cx cy skew
    int num_radial;
    PlanarPoseVPBlock(const ViewData& v, const double* kmtx5, int nr) : view(v), num_radial(nr) {
        for (int i = 0; i < 5; ++i) K[i] = kmtx5[i];
        if (v.n < 8) throw std::invalid_argument("fit_distortion_full needs at least 8 observations");  // distortion.h:236-239
        nres = 2 * v.n;
    }
    template <typename T>
    void residuals(const T* pose6, T* r, T* alpha_out) const {
        const int m = num_radial + 2, N = view.n;
        const T fx(K[0]), fy(K[1]), cx(K[2]), cy(K[3]), skew(K[4]);
        std::vector<T> A(static_cast<size_t>(2 * N) * m), b(2 * N);
        for (int i = 0; i < N; ++i) {
            const T pt[3] = {T(view.X[i]), T(view.Y[i]), T(0.0)};
            T pc[3];
            angle_axis_rotate_point(pose6, pt, pc);
            for (int k = 0; k < 3; ++k) pc[k] = pc[k] + pose6[3 + k];
            const T iz = T(1.0) / pc[2];
            const T x = pc[0] * iz, y = pc[1] * iz;
            const T r2 = x * x + y * y;
            T* Au = &A[static_cast<size_t>(2 * i) * m];
            T* Av = &A[static_cast<size_t>(2 * i + 1) * m];
            T rpow = r2;
            for (int j = 0; j < num_radial; ++j) {
                Au[j] = fx * x * rpow + skew * y * rpow;
                Av[j] = fy * y * rpow;
                rpow = rpow * r2;
            }
            Au[num_radial] = fx * (T(2.0) * x * y) + skew * (r2 + T(2.0) * y * y);
            Au[num_radial + 1] = fx * (r2 + T(2.0) * x * x) + skew * (T(2.0) * x * y);
            Av[num_radial] = fy * (r2 + T(2.0) * y * y);
            Av[num_radial + 1] = fy * (T(2.0) * x * y);
            b[2 * i] = T(view.u[i]) - (fx * x + skew * y + cx);
            b[2 * i + 1] = T(view.v[i]) - (fy * y + cy);
        }
        // least squares by SVD (distortion.h:290-294); A is needed again for the residual: the SVD works on a copy
        std::vector<T> al(m);
        {
            std::vector<T> W = A;
            lstsq_svd<T>(W, 2 * N, m, b, al.data());
        }
        for (int row = 0; row < 2 * N; ++row) {
            T sacc = -b[row];
            for (int a = 0; a < m; ++a) sacc = sacc + A[static_cast<size_t>(row) * m + a] * al[a];
            r[row] = sacc;
        }
        if (alpha_out)
            for (int a = 0; a < m; ++a) alpha_out[a] = al[a];
    }
    void evaluate(const double* const* x, double* r, double** J) const override {
        if (!J) { residuals<double>(x[0], r, nullptr); return; }
        using JT = Jet<6>;
        JT p[6];
        for (int k = 0; k < 6; ++k) p[k] = JT(x[0][k], k);
        std::vector<JT> rj(nres);
        residuals<JT>(p, rj.data(), nullptr);
        for (int i = 0; i < nres; ++i) {
            r[i] = rj[i].a;
            if (J[0]) for (int k = 0; k < 6; ++k) J[0][static_cast<size_t>(i) * 6 + k] = rj[i].v[k];
        }
    }
};

// CalibVPResidual::operator(), src/estimation/residuals/intrinsicsemidltresidual.h:34-58: parameter blocks
// [intr(5) | quat_0(4) tran_0(3) | quat_1 tran_1 | ...]; every view's points go through
// planar_observables_to_observables (observationutils.h:78-95: point = c_se3_t * (X, Y, 0), xn = x/z, yn = y/z with the
// rotation from quat_array_to_rotmat, no normalisation), then ONE fit_distortion_full over all observations
// (distortion.h:229-295) and residual = A alpha - b.  Least squares by SVD on Jets (lstsq_svd), as the reference does.
// Jets are NJ = 5 + 7 * VMAX wide: the oracle handles up to VMAX views (enough for the tests).
struct CalibVPBlock final : ResidualBlock {
    static constexpr int VMAX = 6, NJ = 5 + 7 * VMAX;
    std::vector<ViewData> views;
    int num_radial, total = 0;
    CalibVPBlock(const std::vector<ViewData>& v, int nr) : views(v), num_radial(nr) {
        if (static_cast<int>(v.size()) > VMAX) throw std::invalid_argument("oracle CalibVPBlock: too many views for the fixed Jet width");
        for (const auto& w : v) total += w.n;
        nres = 2 * total;
    }
    // p[0] = intr5, p[1 + 2 i] = quat_i, p[2 + 2 i] = tran_i.  Returns false when fit_distortion_full would (N < 8).
    template <typename T>
    bool residuals(const T* const* p, T* r, T* alpha_out) const {
        const int m = num_radial + 2, N = total;
        if (N < 8) return false;  // distortion.h:235-238
        const T fx = p[0][0], fy = p[0][1], cx = p[0][2], cy = p[0][3], skew = p[0][4];
        std::vector<T> A(static_cast<size_t>(2 * N) * m), b(2 * N);
        int row = 0;
        for (size_t vi = 0; vi < views.size(); ++vi) {
            T R[9];
            quat_to_rotmat<T>(p[1 + 2 * vi], R);
            const T* t = p[2 + 2 * vi];
            const ViewData& view = views[vi];
            for (int i = 0; i < view.n; ++i, row += 2) {
                const T X(view.X[i]), Y(view.Y[i]);
                const T pc0 = R[0] * X + R[1] * Y + t[0], pc1 = R[3] * X + R[4] * Y + t[1], pc2 = R[6] * X + R[7] * Y + t[2];
                const T x = pc0 / pc2, y = pc1 / pc2;
                const T r2 = x * x + y * y;
                T* Au = &A[static_cast<size_t>(row) * m];
                T* Av = &A[static_cast<size_t>(row + 1) * m];
                T rpow = r2;
                for (int j = 0; j < num_radial; ++j) {
                    Au[j] = fx * x * rpow + skew * y * rpow;
                    Av[j] = fy * y * rpow;
                    rpow = rpow * r2;
                }
                Au[num_radial] = fx * (T(2.0) * x * y) + skew * (r2 + T(2.0) * y * y);
                Au[num_radial + 1] = fx * (r2 + T(2.0) * x * x) + skew * (T(2.0) * x * y);
                Av[num_radial] = fy * (r2 + T(2.0) * y * y);
                Av[num_radial + 1] = fy * (T(2.0) * x * y);
                b[row] = T(view.u[i]) - (fx * x + skew * y + cx);
                b[row + 1] = T(view.v[i]) - (fy * y + cy);
            }
        }
        std::vector<T> al(m);
        {
            std::vector<T> W = A;
            lstsq_svd<T>(W, 2 * N, m, b, al.data());  // distortion.h:290-294
        }
        for (int rw = 0; rw < 2 * N; ++rw) {
            T sacc = -b[rw];
            for (int a = 0; a < m; ++a) sacc = sacc + A[static_cast<size_t>(rw) * m + a] * al[a];
            r[rw] = sacc;
        }
        if (alpha_out)
            for (int a = 0; a < m; ++a) alpha_out[a] = al[a];
        return true;
    }
    void evaluate(const double* const* x, double* r, double** J) const override {
        const int nb = 1 + 2 * static_cast<int>(views.size());
        if (!J) {
            if (!residuals<double>(x, r, nullptr))
                for (int i = 0; i < nres; ++i) r[i] = std::numeric_limits<double>::quiet_NaN();
            return;
        }
        using JT = Jet<NJ>;
        std::vector<std::vector<JT>> store(nb);
        std::vector<const JT*> ptr(nb);
        int col = 0;
        for (int bk = 0; bk < nb; ++bk) {
            const int sz = bk == 0 ? 5 : (bk % 2 == 1 ? 4 : 3);
            store[bk].resize(sz);
            for (int k = 0; k < sz; ++k) store[bk][k] = JT(x[bk][k], col++);
            ptr[bk] = store[bk].data();
        }
        std::vector<JT> rj(nres);
        const bool ok = residuals<JT>(ptr.data(), rj.data(), nullptr);
        col = 0;
        for (int bk = 0; bk < nb; ++bk) {
            const int sz = bk == 0 ? 5 : (bk % 2 == 1 ? 4 : 3);
            for (int i = 0; i < nres; ++i) {
                if (bk == 0) r[i] = ok ? rj[i].a : std::numeric_limits<double>::quiet_NaN();
                if (J[bk]) for (int k = 0; k < sz; ++k) J[bk][static_cast<size_t>(i) * sz + k] = ok ? rj[i].v[col + k] : 0.0;
            }
            col += sz;
        }
    }
};

// HomographyResidual::operator(), src/estimation/optim/homography.cpp:103-130: 8 parameters (H22 = 1),
// uvw = H [x y 1]^T, residual = hnormalized(uvw) - (u, v); ONE residual block per correspondence (:132-142).
template <typename T>
inline void homography_residual(const T* h, double x, double y, double u, double v, T* r) {
    const T uvw0 = h[0] * T(x) + h[1] * T(y) + h[2];
    const T uvw1 = h[3] * T(x) + h[4] * T(y) + h[5];
    const T uvw2 = h[6] * T(x) + h[7] * T(y) + T(1.0);
    r[0] = uvw0 / uvw2 - T(u);
    r[1] = uvw1 / uvw2 - T(v);
}
struct HomographyBlock final : ResidualBlock {
    double x, y, u, v;
    HomographyBlock(double x_, double y_, double u_, double v_) : x(x_), y(y_), u(u_), v(v_) { nres = 2; }
    void evaluate(const double* const* p, double* r, double** J) const override {
        if (!J) { homography_residual<double>(p[0], x, y, u, v, r); return; }
        using JT = Jet<8>;
        JT h[8], rj[2];
        for (int k = 0; k < 8; ++k) h[k] = JT(p[0][k], k);
        homography_residual<JT>(h, x, y, u, v, rj);
        for (int i = 0; i < 2; ++i) {
            r[i] = rj[i].a;
            if (J[0]) for (int k = 0; k < 8; ++k) J[0][i * 8 + k] = rj[i].v[k];
        }
    }
};

inline std::unique_ptr<ResidualBlock> make_reproj_block(int chain, int model, const ViewData& v,
                                                        const double* bTg12) {
    auto fill = [&](auto* blk) {
        if (bTg12) {
            for (int i = 0; i < 9; ++i) blk->bRg[i] = bTg12[i];
            for (int i = 0; i < 3; ++i) blk->btg[i] = bTg12[9 + i];
        }
        return std::unique_ptr<ResidualBlock>(blk);
    };
    if (chain == CHAIN_INTRINSIC)
        return model == SCHEIMPFLUG ? fill(new ReprojBlock<CHAIN_INTRINSIC, SCHEIMPFLUG>(v))
                                    : fill(new ReprojBlock<CHAIN_INTRINSIC, PINHOLE_BC>(v));
    if (chain == CHAIN_EXTRINSIC)
        return model == SCHEIMPFLUG ? fill(new ReprojBlock<CHAIN_EXTRINSIC, SCHEIMPFLUG>(v))
                                    : fill(new ReprojBlock<CHAIN_EXTRINSIC, PINHOLE_BC>(v));
    return model == SCHEIMPFLUG ? fill(new ReprojBlock<CHAIN_BUNDLE, SCHEIMPFLUG>(v))
                                : fill(new ReprojBlock<CHAIN_BUNDLE, PINHOLE_BC>(v));
}

}  // namespace orc
