// oracle/residuals.hpp — TEST INFRASTRUCTURE ONLY.
//
// The reference's Ceres cost functors, evaluated with orc::Jet exactly as
// ceres::AutoDiffCostFunction would: one residual block per VIEW with 2N
// residuals (intrinsicresidual.h:37-48, extrinsicsresidual.h:48-59,
// bundleresidual.h:58-68) and one 6-residual block per motion pair
// (handeyeresidual.h:51-53).
#pragma once
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
