// tests/cpu_backend/semidlt_cpu.cpp — TEST-ONLY host run of semidlt_math.hpp + semidlt_core.hpp (the semi-DLT variable
// projection the GPU evaluates one wavefront per view) with the single-thread cooperative group.  Never linked into the product.
#include <cstring>
#include <string>
#include <vector>

#include "../../calibration_amd/csrc/semidlt_core.hpp"
#include "../../calibration_amd/csrc/semidlt_math.hpp"

using namespace cba;

namespace {
// Views [v0, v0 + Vl) of V are local (their observations behind off / X / Y / u / v); the poses cover all V views.  With a reduce
// callback the evaluator follows the exchange protocol of the product's sharded evaluator (semidlt.hip HipSemiDlt): the pass-1
// sums cross the ranks before alpha is formed, the per-view tables are gathered as a sum of zero-padded tables.
struct SerialSemiDlt final : SemiDltEval {
    const int64_t* off;
    const double *X, *Y, *u, *v;
    int Vl, v0 = 0;
    cba_allreduce_fn fn = nullptr;
    void* user = nullptr;
    std::vector<int64_t> counts;
    SerialSemiDlt(int n_views, const int64_t* o, const double* x, const double* y, const double* uu, const double* vv, int num_radial)
        : off(o), X(x), Y(y), u(uu), v(vv), Vl(n_views) {
        V = n_views; nr = num_radial; n_obs = o[n_views];
        for (int i = 0; i < V; ++i) counts.push_back(o[i + 1] - o[i]);
    }
    SerialSemiDlt(int n_local, const int64_t* o, const double* x, const double* y, const double* uu, const double* vv, int num_radial,
                  int n_total, int first_view, cba_allreduce_fn f, void* usr)
        : off(o), X(x), Y(y), u(uu), v(vv), Vl(n_local), v0(first_view), fn(f), user(usr) {
        V = n_total; nr = num_radial;
        std::vector<double> c(V, 0.0);
        for (int i = 0; i < Vl; ++i) c[v0 + i] = static_cast<double>(o[i + 1] - o[i]);
        reduce(c.data(), V);
        n_obs = 0;
        for (int i = 0; i < V; ++i) { counts.push_back(static_cast<int64_t>(c[i] + 0.5)); n_obs += counts.back(); }
    }
    void reduce(double* buf, int64_t n) const {
        if (fn && fn(buf, n, user) != 0) throw std::runtime_error("allreduce callback failed");
    }
    SDView view(int i, const double* poses7) const {  // i: local index
        SDView W;
        W.n = static_cast<int>(off[i + 1] - off[i]);
        W.X = X + off[i]; W.Y = Y + off[i]; W.u = u + off[i]; W.v = v + off[i];
        block_consts<CH_INTRINSIC>(poses7 + 7 * static_cast<size_t>(v0 + i), nullptr, nullptr, W.bc);
        return W;
    }
    template <int NR>
    void normal_t(const double* k, const double* p, double* N, double* rhs) const {
        using L = SDLayout<NR>;
        constexpr int m = L::M;
        double acc[L::N1] = {0}, one[L::N1];
        SerialCoop co;
        for (int i = 0; i < Vl; ++i) {
            const SDView W = view(i, p);
            sd_pass1<NR>(W, k, co, one);
            for (int e = 0; e < L::N1; ++e) acc[e] += one[e];
        }
        reduce(acc, L::N1);
        int e = 0;
        for (int a = 0; a < m; ++a)
            for (int c = 0; c <= a; ++c, ++e) { N[a * m + c] = acc[e]; N[c * m + a] = acc[e]; }
        for (int a = 0; a < m; ++a) rhs[a] = acc[e + a];
    }
    template <int NR>
    bool evaluate_t(const double* k, const double* p, double* N, double* rhs, double* al, double* per_view) const {
        constexpr int m = NR + 2;
        normal_t<NR>(k, p, N, rhs);
        double Lc[m * m];
        for (int a = 0; a < m * m; ++a) Lc[a] = N[a];
        if (!vp_chol<m>(Lc)) return false;
        for (int a = 0; a < m; ++a) al[a] = rhs[a];
        vp_chol_solve<m>(Lc, al);
        SerialCoop co;
        if (fn) std::memset(per_view, 0, sizeof(double) * static_cast<size_t>(V) * SDLayout<NR>::N2);
        for (int i = 0; i < Vl; ++i) {
            const SDView W = view(i, p);
            sd_pass2_part<NR, 1, 0>(W, k, al, co, per_view + static_cast<size_t>(v0 + i) * SDLayout<NR>::N2);
        }
        reduce(per_view, static_cast<int64_t>(V) * SDLayout<NR>::N2);
        return true;
    }
    template <int NR>
    void resid_t(const double* k, const double* p, const double* al, double* s) const {
        SerialCoop co;
        for (int i = 0; i < V; ++i) s[i] = 0.0;
        for (int i = 0; i < Vl; ++i) { const SDView W = view(i, p); s[v0 + i] = sd_resid<NR>(W, k, al, co); }
        reduce(s, V);
    }
    void normal(const double* k, const double* p, double* N, double* rhs) override {
        switch (nr) { case 0: normal_t<0>(k, p, N, rhs); break; case 1: normal_t<1>(k, p, N, rhs); break;
                      case 2: normal_t<2>(k, p, N, rhs); break; default: normal_t<3>(k, p, N, rhs); }
    }
    bool evaluate(const double* k, const double* p, double* N, double* rhs, double* al, double* pv) override {
        switch (nr) { case 0: return evaluate_t<0>(k, p, N, rhs, al, pv); case 1: return evaluate_t<1>(k, p, N, rhs, al, pv);
                      case 2: return evaluate_t<2>(k, p, N, rhs, al, pv); default: return evaluate_t<3>(k, p, N, rhs, al, pv); }
    }
    void resid(const double* k, const double* p, const double* al, double* s) override {
        switch (nr) { case 0: resid_t<0>(k, p, al, s); break; case 1: resid_t<1>(k, p, al, s); break;
                      case 2: resid_t<2>(k, p, al, s); break; default: resid_t<3>(k, p, al, s); }
    }
};
thread_local std::string g_sd_err;
}  // namespace

extern "C" {

const char* hm_semidlt_last_error(void) { return g_sd_err.c_str(); }

// dense tangent-space H (n x n), g (n), cost and alpha at (kappa, poses); n = (optimize_skew ? 5 : 4) + 6 V
int hm_semidlt_linearise(int n_views, const int64_t* off, const double* X, const double* Y, const double* u, const double* v,
                         const double* kappa5, const double* poses7, int num_radial, const cba_options* o, double* H, double* g,
                         double* cost, double* alpha) {
    SerialSemiDlt ev(n_views, off, X, Y, u, v, num_radial);
    SemiDltDriver drv(ev, *o);
    drv.kappa.assign(kappa5, kappa5 + 5);
    drv.poses.assign(poses7, poses7 + 7 * static_cast<size_t>(n_views));
    SemiDltSystem S;
    if (!drv.linearise(drv.kappa, drv.poses, S)) return 1;
    std::vector<double> Hd;
    S.dense(Hd);
    std::memcpy(H, Hd.data(), sizeof(double) * Hd.size());
    std::memcpy(g, S.g.data(), sizeof(double) * S.n);
    *cost = S.cost;
    for (int a = 0; a < S.m; ++a) alpha[a] = S.alpha[a];
    return 0;
}

// one damped solve through the arrow + Woodbury path: delta = -(H + diag(dlm))^-1 g  (checked against a dense solve by the tests)
int hm_semidlt_step(int n_views, const int64_t* off, const double* X, const double* Y, const double* u, const double* v,
                    const double* kappa5, const double* poses7, int num_radial, const cba_options* o, const double* dlm, double* delta) {
    SerialSemiDlt ev(n_views, off, X, Y, u, v, num_radial);
    SemiDltDriver drv(ev, *o);
    drv.kappa.assign(kappa5, kappa5 + 5);
    drv.poses.assign(poses7, poses7 + 7 * static_cast<size_t>(n_views));
    SemiDltSystem S;
    if (!drv.linearise(drv.kappa, drv.poses, S)) return 1;
    std::vector<double> d(dlm, dlm + S.n), out;
    if (!S.solve(d, out)) return 2;
    std::memcpy(delta, out.data(), sizeof(double) * S.n);
    return 0;
}

// same outputs as the product's semidlt_solve / the oracle's orc_semidlt_solve
int hm_semidlt_solve(int n_views, const int64_t* off, const double* X, const double* Y, const double* u, const double* v, double* kappa5,
                     double* poses7, int num_radial, const double* lo, const double* hi, const int32_t* fixed_idx, const double* fixed_val,
                     int n_fixed, const cba_options* o, cba_summary* summary, double* distortion, double* view_errors, double* cov) {
    try {
        SerialSemiDlt ev(n_views, off, X, Y, u, v, num_radial);
        SemiDltDriver drv(ev, *o);
        if (lo && hi) { drv.bounds.enabled = true; for (int k = 0; k < 5; ++k) { drv.bounds.lo[k] = lo[k]; drv.bounds.hi[k] = hi[k]; } }
        drv.solve(kappa5, poses7, summary);
        SemiDltResult res;
        double ssr = 0.0;
        drv.finish(fixed_idx, fixed_val, n_fixed, off, res, &ssr);
        for (int a = 0; a < num_radial + 2; ++a) distortion[a] = res.alpha[a];
        for (int i = 0; i < n_views; ++i) view_errors[i] = res.view_errors[i];
        if (cov) {
            const size_t dim = 5 + 7 * static_cast<size_t>(n_views);
            std::memset(cov, 0, sizeof(double) * dim * dim);
            std::vector<double> c;
            if (drv.covariance(ssr, c)) std::memcpy(cov, c.data(), sizeof(double) * dim * dim);
        }
        return 0;
    } catch (const std::exception& e) {
        g_sd_err = e.what();
        return 1;
    }
}
// the same solve with the views sharded over ranks (what cba_optimize_intrinsics_semidlt_sharded does on the GPU): this rank's views
// [first_view, first_view + n_local) of n_total; kappa5 / poses7 / view_errors / cov cover the whole problem
int hm_semidlt_solve_sharded(int n_local, const int64_t* off, const double* X, const double* Y, const double* u, const double* v, int n_total,
                             int first_view, double* kappa5, double* poses7, int num_radial, const double* lo, const double* hi,
                             const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const cba_options* o, cba_summary* summary,
                             double* distortion, double* view_errors, double* cov, cba_allreduce_fn fn, void* user) {
    try {
        SerialSemiDlt ev(n_local, off, X, Y, u, v, num_radial, n_total, first_view, fn, user);
        SemiDltDriver drv(ev, *o);
        if (lo && hi) { drv.bounds.enabled = true; for (int k = 0; k < 5; ++k) { drv.bounds.lo[k] = lo[k]; drv.bounds.hi[k] = hi[k]; } }
        drv.solve(kappa5, poses7, summary);
        SemiDltResult res;
        double ssr = 0.0;
        std::vector<int64_t> off_all(static_cast<size_t>(n_total) + 1, 0);
        for (int i = 0; i < n_total; ++i) off_all[i + 1] = off_all[i] + ev.counts[i];
        drv.finish(fixed_idx, fixed_val, n_fixed, off_all.data(), res, &ssr);
        for (int a = 0; a < num_radial + 2; ++a) distortion[a] = res.alpha[a];
        for (int i = 0; i < n_total; ++i) view_errors[i] = res.view_errors[i];
        if (cov) {
            const size_t dim = 5 + 7 * static_cast<size_t>(n_total);
            std::memset(cov, 0, sizeof(double) * dim * dim);
            std::vector<double> c;
            if (drv.covariance(ssr, c)) std::memcpy(cov, c.data(), sizeof(double) * dim * dim);
        }
        return 0;
    } catch (const std::exception& e) {
        g_sd_err = e.what();
        return 1;
    }
}
}
