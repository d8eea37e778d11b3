// semidlt.hip — optimize_intrinsics_semidlt (src/estimation/optim/intrinsicssemidlt.cpp:155-191) on the GPU.
//
// One evaluation of the variable-projection functor (CalibVPResidual, intrinsicsemidltresidual.h:19-73) is
//   k_sd_pass1   one wavefront per view: N_v = A^T A, (A^T b)_v                     (semidlt_math.hpp)
//   k_sd_alpha   one thread: fixed-order sum over views, m x m Cholesky, alpha
//   k_sd_pass2   one wavefront per view: W^T W, W^T r, A^T W, dA^T r, |r|^2 in three register-sized parts
// on one stream with ONE device-to-host copy at the end; the O(#views) linear algebra of the LM step runs on the
// host (semidlt_core.hpp).  Lanes stride over the view's points (unit-stride loads), sums cross the wave in DPP.
#include "engine.hpp"
#include "semidlt_core.hpp"
#include "semidlt_math.hpp"

namespace cba {

constexpr int SD_WAVES = 4;

__device__ __forceinline__ bool sd_view_setup(int n_views, const int64_t* off, const double* X, const double* Y, const double* u,
                                              const double* v, const double* poses, SDView& V, int* view) {
    const int i = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * SD_WAVES + (threadIdx.x >> 6)));
    if (i >= n_views) return false;
    *view = i;
    V.n = static_cast<int>(off[i + 1] - off[i]);
    V.X = X + off[i]; V.Y = Y + off[i]; V.u = u + off[i]; V.v = v + off[i];
    block_consts<CH_INTRINSIC>(poses + 7 * static_cast<int64_t>(i), nullptr, nullptr, V.bc);
    return true;
}

template <int NR>
__global__ __launch_bounds__(64 * SD_WAVES) void k_sd_pass1(int n_views, const int64_t* __restrict__ off, const double* __restrict__ X,
                                                            const double* __restrict__ Y, const double* __restrict__ u,
                                                            const double* __restrict__ v, const double* __restrict__ kappa,
                                                            const double* __restrict__ poses, double* __restrict__ out1) {
    SDView V;
    int view;
    if (!sd_view_setup(n_views, off, X, Y, u, v, poses, V, &view)) return;
    double K[5];
    for (int k = 0; k < 5; ++k) K[k] = kappa[k];
    WaveCoop co;
    sd_pass1<NR>(V, K, co, out1 + static_cast<int64_t>(view) * SDLayout<NR>::N1);
}

// sums[0 .. m*m) = N (full), [m*m .. m*m+m) = A^T b, then alpha (m), then ok flag
template <int NR>
__global__ void k_sd_alpha(int n_views, const double* __restrict__ out1, double* __restrict__ sums) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using L = SDLayout<NR>;
    constexpr int m = L::M;
    double acc[L::N1];
    for (int e = 0; e < L::N1; ++e) acc[e] = 0.0;
    for (int v = 0; v < n_views; ++v)
        for (int e = 0; e < L::N1; ++e) acc[e] += out1[static_cast<int64_t>(v) * L::N1 + e];
    double Nf[m * m], al[m];
    int e = 0;
    for (int a = 0; a < m; ++a)
        for (int c = 0; c <= a; ++c, ++e) { Nf[a * m + c] = acc[e]; Nf[c * m + a] = acc[e]; }
    for (int a = 0; a < m; ++a) al[a] = acc[e + a];
    for (int a = 0; a < m * m; ++a) sums[a] = Nf[a];
    for (int a = 0; a < m; ++a) sums[m * m + a] = al[a];
    const bool ok = vp_chol<m>(Nf);
    if (ok) vp_chol_solve<m>(Nf, al);
    for (int a = 0; a < m; ++a) sums[m * m + m + a] = ok ? al[a] : 0.0;
    sums[m * m + 2 * m] = ok ? 1.0 : 0.0;
}

template <int NR>
__global__ __launch_bounds__(64 * SD_WAVES) void k_sd_pass2(int n_views, const int64_t* __restrict__ off, const double* __restrict__ X,
                                                            const double* __restrict__ Y, const double* __restrict__ u,
                                                            const double* __restrict__ v, const double* __restrict__ kappa,
                                                            const double* __restrict__ poses, const double* __restrict__ alpha_dev,
                                                            double* __restrict__ out2) {
    SDView V;
    int view;
    if (!sd_view_setup(n_views, off, X, Y, u, v, poses, V, &view)) return;
    constexpr int m = NR + 2;
    double K[5], al[m];
    for (int k = 0; k < 5; ++k) K[k] = kappa[k];
    for (int a = 0; a < m; ++a) al[a] = alpha_dev[a];
    WaveCoop co;
    double* out = out2 + static_cast<int64_t>(view) * SDLayout<NR>::N2;
    sd_pass2_part<NR, 3, 0>(V, K, al, co, out);
    sd_pass2_part<NR, 3, 1>(V, K, al, co, out);
    sd_pass2_part<NR, 3, 2>(V, K, al, co, out);
}

template <int NR>
__global__ __launch_bounds__(64 * SD_WAVES) void k_sd_resid(int n_views, const int64_t* __restrict__ off, const double* __restrict__ X,
                                                            const double* __restrict__ Y, const double* __restrict__ u,
                                                            const double* __restrict__ v, const double* __restrict__ kappa,
                                                            const double* __restrict__ poses, const double* __restrict__ alpha_dev,
                                                            double* __restrict__ s_view) {
    SDView V;
    int view;
    if (!sd_view_setup(n_views, off, X, Y, u, v, poses, V, &view)) return;
    constexpr int m = NR + 2;
    double K[5], al[m];
    for (int k = 0; k < 5; ++k) K[k] = kappa[k];
    for (int a = 0; a < m; ++a) al[a] = alpha_dev[a];
    WaveCoop co;
    const double s = sd_resid<NR>(V, K, al, co);
    if (co.lane() == 0) s_view[view] = s;
}

namespace {
struct HipSemiDlt final : SemiDltEval {
    StreamLease lease;  // before the buffers: released after them
    hipStream_t stream = lease;
    DevBuf<double> X, Y, u, v, kappa, poses, out1, sums, out2, alpha, sview;
    DevBuf<int64_t> off;
    dim3 grid, block;
    HipSemiDlt(int n_views, const int64_t* view_offset, const double* hX, const double* hY, const double* hu, const double* hv, int num_radial) {
        V = n_views; nr = num_radial; n_obs = view_offset[n_views];
        const size_t n = static_cast<size_t>(std::max<int64_t>(n_obs, 1));
        X.alloc(n); Y.alloc(n); u.alloc(n); v.alloc(n);
        X.upload(hX, n_obs, stream); Y.upload(hY, n_obs, stream); u.upload(hu, n_obs, stream); v.upload(hv, n_obs, stream);
        off.alloc(V + 1); off.upload(view_offset, V + 1, stream);
        kappa.alloc(5); poses.alloc(7 * static_cast<size_t>(V));
        out1.alloc(static_cast<size_t>(V) * 20); sums.alloc(64); alpha.alloc(8);
        out2.alloc(static_cast<size_t>(V) * n2()); sview.alloc(V);
        grid = dim3((V + SD_WAVES - 1) / SD_WAVES); block = dim3(64 * SD_WAVES);
        CBA_HIP(hipStreamSynchronize(stream));
    }
    ~HipSemiDlt() override { (void)hipStreamSynchronize(stream); }
    void put(const double* kappa5, const double* poses7) {
        kappa.upload(kappa5, 5, stream);
        poses.upload(poses7, 7 * static_cast<size_t>(V), stream);
    }
#define SD_DISPATCH(KERNEL, G, B, ...)                                                                  \
    switch (nr) {                                                                                      \
        case 0: hipLaunchKernelGGL(KERNEL<0>, G, B, 0, stream, __VA_ARGS__); break;                    \
        case 1: hipLaunchKernelGGL(KERNEL<1>, G, B, 0, stream, __VA_ARGS__); break;                    \
        case 2: hipLaunchKernelGGL(KERNEL<2>, G, B, 0, stream, __VA_ARGS__); break;                    \
        default: hipLaunchKernelGGL(KERNEL<3>, G, B, 0, stream, __VA_ARGS__); break;                   \
    }
    void launch_pass1() {
        SD_DISPATCH(k_sd_pass1, grid, block, V, off.p, X.p, Y.p, u.p, v.p, kappa.p, poses.p, out1.p)
        SD_DISPATCH(k_sd_alpha, dim3(1), dim3(64), V, out1.p, sums.p)
    }
    void normal(const double* kappa5, const double* poses7, double* N, double* rhs) override {
        put(kappa5, poses7);
        launch_pass1();
        CBA_HIP(hipGetLastError());
        double h[64];
        sums.download(h, 64, stream);
        CBA_HIP(hipStreamSynchronize(stream));
        const int mm = m();
        for (int a = 0; a < mm * mm; ++a) N[a] = h[a];
        for (int a = 0; a < mm; ++a) rhs[a] = h[mm * mm + a];
    }
    bool evaluate(const double* kappa5, const double* poses7, double* N, double* rhs, double* al, double* per_view) override {
        put(kappa5, poses7);
        launch_pass1();
        const int mm = m();
        SD_DISPATCH(k_sd_pass2, grid, block, V, off.p, X.p, Y.p, u.p, v.p, kappa.p, poses.p, sums.p + mm * mm + mm, out2.p)
        CBA_HIP(hipGetLastError());
        double h[64];
        sums.download(h, 64, stream);
        out2.download(per_view, static_cast<size_t>(V) * n2(), stream);
        CBA_HIP(hipStreamSynchronize(stream));
        for (int a = 0; a < mm * mm; ++a) N[a] = h[a];
        for (int a = 0; a < mm; ++a) { rhs[a] = h[mm * mm + a]; al[a] = h[mm * mm + mm + a]; }
        return h[mm * mm + 2 * mm] > 0.5;
    }
    void resid(const double* kappa5, const double* poses7, const double* al, double* s_view) override {
        put(kappa5, poses7);
        alpha.upload(al, m(), stream);
        SD_DISPATCH(k_sd_resid, grid, block, V, off.p, X.p, Y.p, u.p, v.p, kappa.p, poses.p, alpha.p, sview.p)
        CBA_HIP(hipGetLastError());
        sview.download(s_view, V, stream);
        CBA_HIP(hipStreamSynchronize(stream));
    }
#undef SD_DISPATCH
};
}  // namespace

void semidlt_solve(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                   double* kappa5, double* poses7, int num_radial, const double* bounds_lo, const double* bounds_hi,
                   const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const cba_options* o, cba_summary* summary,
                   double* distortion, double* view_errors, double* cov, int device) {
    CBA_HIP(hipSetDevice(device));
    HipSemiDlt ev(n_views, view_offset, X, Y, u, v, num_radial);
    SemiDltDriver drv(ev, *o);
    if (bounds_lo && bounds_hi) {
        drv.bounds.enabled = true;
        for (int k = 0; k < 5; ++k) { drv.bounds.lo[k] = bounds_lo[k]; drv.bounds.hi[k] = bounds_hi[k]; }
    }
    drv.solve(kappa5, poses7, summary);
    SemiDltResult res;
    double ssr = 0.0;
    drv.finish(fixed_idx, fixed_val, n_fixed, view_offset, res, &ssr);
    const int m = num_radial + 2;
    if (distortion) for (int a = 0; a < m; ++a) distortion[a] = res.alpha[a];
    if (view_errors) for (int i = 0; i < n_views; ++i) view_errors[i] = res.view_errors[i];
    if (cov) {
        const size_t dim = 5 + 7 * static_cast<size_t>(n_views);
        std::memset(cov, 0, sizeof(double) * dim * dim);
        std::vector<double> c;
        if (drv.covariance(ssr, c)) std::memcpy(cov, c.data(), sizeof(double) * dim * dim);
    }
}

}  // namespace cba
