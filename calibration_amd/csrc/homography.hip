// homography.hip — batched optimize_homography (src/estimation/optim/homography.cpp:144-175) on the GPU.
//
// k_homography: ONE WAVEFRONT PER VIEW runs that view's whole 8-parameter Levenberg-Marquardt refinement in-kernel
// (hom_math.hpp / small_lm.hpp): lanes stride over the correspondences (unit-stride 512-byte loads of X, Y, u, v),
// the 36 + 8 + 3 partial sums cross the wave in DPP, and every lane takes the same 8x8 Cholesky step.  No host
// round trips; views are independent problems, so a batch of thousands fills the chip (one view = one wave, 4 views
// per 256-thread workgroup).  Every wave leaves after at most max_iterations iterations: the grid always drains.
#include <chrono>
#include <cmath>
#include <cstdio>

#include "engine.hpp"
#include "hom_math.hpp"

namespace cba {

constexpr int HOM_WAVES_PER_BLOCK = 4;

__global__ __launch_bounds__(64 * HOM_WAVES_PER_BLOCK) void k_homography(
    int n_views, const int64_t* __restrict__ off, const double* __restrict__ X, const double* __restrict__ Y,
    const double* __restrict__ u, const double* __restrict__ v, double huber_delta, double eps, int max_iterations, int want_cov,
    HomResult* __restrict__ res) {
    const int view = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * HOM_WAVES_PER_BLOCK + (threadIdx.x >> 6)));
    if (view >= n_views) return;  // whole wave leaves together
    HomProblem P;
    const int64_t o0 = off[view];
    P.n = static_cast<int>(off[view + 1] - o0);
    P.X = X + o0; P.Y = Y + o0; P.u = u + o0; P.v = v + o0;
    P.huber_delta = huber_delta;
    HomResult r;
    for (int k = 0; k < 8; ++k) r.h[k] = res[view].h[k];
    WaveCoop co;
    hom_solve_view(P, co, eps, max_iterations, want_cov != 0, r);
    if (co.lane() == 0) res[view] = r;
}

void homography_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                      double* h9, const cba_options* o, cba_summary* summaries, double* cov64, int device) {
    if (n_views <= 0) throw std::invalid_argument("At least 4 correspondences are required.");
    if (!view_offset || !X || !Y || !u || !v || !h9 || !o) throw std::invalid_argument("null argument");
    for (int i = 0; i < n_views; ++i) {
        const int64_t n = view_offset[i + 1] - view_offset[i];
        if (n < 4) throw std::invalid_argument("At least 4 correspondences are required.");  // homography.cpp:146-148
        if (n > 0x7fffffff) throw std::invalid_argument("view too large");
    }
    CBA_HIP(hipSetDevice(device));
    StreamLease lease;
    const hipStream_t stream = lease;
    {
        const int64_t n_obs = view_offset[n_views];
        DevBuf<double> dX, dY, du, dv;
        DevBuf<int64_t> doff;
        DevBuf<HomResult> dres;
        dX.alloc(n_obs); dY.alloc(n_obs); du.alloc(n_obs); dv.alloc(n_obs);
        doff.alloc(n_views + 1); dres.alloc(n_views);
        dX.upload(X, n_obs, stream); dY.upload(Y, n_obs, stream); du.upload(u, n_obs, stream); dv.upload(v, n_obs, stream);
        doff.upload(view_offset, n_views + 1, stream);
        std::vector<HomResult> h(n_views);
        for (int i = 0; i < n_views; ++i)  // HomographyBlocks::create (homography.cpp:79-84): the first 8 entries, as given
            for (int k = 0; k < 8; ++k) h[i].h[k] = h9[static_cast<size_t>(i) * 9 + k];
        dres.upload(h.data(), n_views, stream);
        const auto t0 = std::chrono::steady_clock::now();
        const int blocks = (n_views + HOM_WAVES_PER_BLOCK - 1) / HOM_WAVES_PER_BLOCK;
        hipLaunchKernelGGL(k_homography, dim3(blocks), dim3(64 * HOM_WAVES_PER_BLOCK), 0, stream, n_views, doff.p, dX.p, dY.p, du.p,
                           dv.p, o->huber_delta, o->epsilon, o->max_iterations, (cov64 && o->compute_covariance) ? 1 : 0, dres.p);
        CBA_HIP(hipGetLastError());
        dres.download(h.data(), n_views, stream);
        CBA_HIP(hipStreamSynchronize(stream));
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < n_views; ++i) {
            const HomResult& r = h[i];
            double* H = h9 + static_cast<size_t>(i) * 9;
            for (int k = 0; k < 8; ++k) H[k] = r.h[k];
            H[8] = 1.0;  // params_to_h (homography.cpp:93-98); the division by H22 (:158-160) is then the identity
            if (cov64) for (int k = 0; k < 64; ++k) cov64[static_cast<size_t>(i) * 64 + k] = r.cov_ok ? r.cov[k] : 0.0;
            if (summaries) {
                cba_summary& s = summaries[i];
                s.termination = r.termination; s.success = r.termination == CBA_TERM_CONVERGENCE;
                s.iterations = r.iterations; s.successful_steps = r.successful_steps;
                s.initial_cost = r.initial_cost; s.final_cost = r.final_cost; s.solve_seconds = secs;
                std::snprintf(s.report, sizeof(s.report), "calibba(homography LM, view %d of %d): termination %d iters=%d cost %.6e -> %.6e",
                              i, n_views, r.termination, r.iterations, r.initial_cost, r.final_cost);
            }
        }
    }
}

}  // namespace cba
