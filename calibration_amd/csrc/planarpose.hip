// planarpose.hip — batched optimize_planar_pose (planarpose.cpp:84-127) on the GPU.
//
// k_planar_pose: ONE WAVEFRONT PER VIEW runs that view's entire variable-projection LM solve
// (vp_math.hpp::vp_solve_view over small_lm.hpp's wave group) in-kernel: lanes stride over the view's points with
// unit-stride loads, the three passes of an evaluation each end in one round of DPP wave sums, and every lane takes
// the same 6x6 step.  No host round trips; views are independent problems (6 unknowns each) and a batch of thousands
// of views fills the chip.  Every wave leaves after at most max_iterations LM iterations, so the grid always drains.
// The view's X, Y, u, v stay L2-resident across the 3 passes per evaluation.
#include <chrono>
#include <cmath>
#include <cstdio>

#include "engine.hpp"
#include "vp_math.hpp"

namespace cba {

constexpr int VP_WAVES_PER_BLOCK = 4;

template <int NR>
__global__ __launch_bounds__(64 * VP_WAVES_PER_BLOCK) void k_planar_pose(
    int n_views, const int64_t* __restrict__ off, const double* __restrict__ X, const double* __restrict__ Y,
    const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ K5, int num_radial, double huber_delta,
    double eps, int max_iterations, int want_cov, VPResult* __restrict__ res) {
    const int i = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * VP_WAVES_PER_BLOCK + (threadIdx.x >> 6)));
    if (i >= n_views) return;  // whole wave leaves together
    VPView V;
    V.n = static_cast<int>(off[i + 1] - off[i]);
    V.X = X + off[i]; V.Y = Y + off[i]; V.u = u + off[i]; V.v = v + off[i];
    for (int k = 0; k < 5; ++k) V.K[k] = K5[k];
    V.num_radial = num_radial;
    VPResult r;
    for (int k = 0; k < 6; ++k) r.pose6[k] = res[i].pose6[k];
    WaveCoop co;
    vp_solve_view<NR>(V, co, huber_delta, eps, max_iterations, want_cov != 0, r);
    if (co.lane() == 0) res[i] = r;
}

void planar_pose_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                       const double* kmtx5, int num_radial, double* pose7, const cba_options* o, cba_summary* summaries,
                       double* distortion, double* rms, double* cov, int device) {
    if (n_views <= 0) throw std::invalid_argument("No observations provided");
    if (num_radial < 0 || num_radial + 2 > VP_MAX_M) throw std::invalid_argument("num_radial must be in [0, 3]");
    if (!view_offset || !X || !Y || !u || !v || !kmtx5 || !pose7 || !o) throw std::invalid_argument("null argument");
    CBA_HIP(hipSetDevice(device));
    StreamLease lease;
    const hipStream_t stream = lease;
    {
        const int64_t n_obs = view_offset[n_views];
        DevBuf<double> dX, dY, du, dv, dK;
        DevBuf<int64_t> doff;
        DevBuf<VPResult> dres;
        dX.alloc(n_obs); dY.alloc(n_obs); du.alloc(n_obs); dv.alloc(n_obs); dK.alloc(5);
        doff.alloc(n_views + 1); dres.alloc(n_views);
        dX.upload(X, n_obs, stream); dY.upload(Y, n_obs, stream); du.upload(u, n_obs, stream); dv.upload(v, n_obs, stream);
        dK.upload(kmtx5, 5, stream); doff.upload(view_offset, n_views + 1, stream);
        std::vector<VPResult> h(n_views);
        for (int i = 0; i < n_views; ++i) {
            // ceres::RotationMatrixToAngleAxis of the initial rotation (planarpose.cpp:89-93)
            const double* p = pose7 + 7 * static_cast<size_t>(i);
            const double nq = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
            const double q[4] = {p[0] / nq, p[1] / nq, p[2] / nq, p[3] / nq};
            quat_to_angle_axis_ceres(q, h[i].pose6);
            for (int k = 0; k < 3; ++k) h[i].pose6[3 + k] = p[4 + k];
        }
        dres.upload(h.data(), n_views, stream);
        const auto t0 = std::chrono::steady_clock::now();
        const dim3 grid((n_views + VP_WAVES_PER_BLOCK - 1) / VP_WAVES_PER_BLOCK), block(64 * VP_WAVES_PER_BLOCK);
        const int want_cov = (cov && o->compute_covariance) ? 1 : 0;
#define CBA_VP_LAUNCH(NR)                                                                                                          \
    hipLaunchKernelGGL(k_planar_pose<NR>, grid, block, 0, stream, n_views, doff.p, dX.p, dY.p, du.p, dv.p, dK.p, num_radial,       \
                       o->huber_delta, o->epsilon, o->max_iterations, want_cov, dres.p)
        switch (num_radial) {  // the design-matrix width nr + 2 is a compile-time constant of the kernel: no stack arrays
            case 0: CBA_VP_LAUNCH(0); break;
            case 1: CBA_VP_LAUNCH(1); break;
            case 2: CBA_VP_LAUNCH(2); break;
            default: CBA_VP_LAUNCH(3); break;
        }
#undef CBA_VP_LAUNCH
        CBA_HIP(hipGetLastError());
        dres.download(h.data(), n_views, stream);
        CBA_HIP(hipStreamSynchronize(stream));
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const int m = num_radial + 2;
        for (int i = 0; i < n_views; ++i) {
            const VPResult& r = h[i];
            double* p = pose7 + 7 * static_cast<size_t>(i);
            angle_axis_to_quat_ceres(r.pose6, p);  // axisangle_to_pose, planarpose.cpp:73-82
            for (int k = 0; k < 3; ++k) p[4 + k] = r.pose6[3 + k];
            if (distortion) for (int k = 0; k < m; ++k) distortion[static_cast<size_t>(i) * m + k] = r.alpha[k];
            if (rms) rms[i] = r.rms;
            if (cov) for (int k = 0; k < 36; ++k) cov[static_cast<size_t>(i) * 36 + k] = r.cov_ok ? r.cov[k] : 0.0;
            if (summaries) {
                cba_summary& s = summaries[i];
                s.termination = r.termination; s.success = r.termination == CBA_TERM_CONVERGENCE;
                s.iterations = r.iterations; s.successful_steps = r.successful_steps;
                s.initial_cost = r.initial_cost; s.final_cost = r.final_cost; s.solve_seconds = secs;
                std::snprintf(s.report, sizeof(s.report), "calibba(planar-pose VP LM, view %d of %d): termination %d iters=%d cost %.6e -> %.6e",
                              i, n_views, r.termination, r.iterations, r.initial_cost, r.final_cost);
            }
        }
    }
}

}  // namespace cba
