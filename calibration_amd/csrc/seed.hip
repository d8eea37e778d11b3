// seed.hip — batched estimate_planar_pose (src/estimation/linear/planarpose_linear.cpp:54-76) on the GPU: one wavefront
// per view runs seed_math.hpp::planar_seed_view (three strided passes over the view's points + O(1) per-lane algebra).
// SURVEY.md §8(f) rank 1 — the per-view seeds that optimize_intrinsics / optimize_intrinsics_semidlt start from.
#include "engine.hpp"
#include "seed_math.hpp"

namespace cba {

constexpr int SEED_WAVES = 4;

__global__ __launch_bounds__(64 * SEED_WAVES) void k_planar_seed(int n_views, const int64_t* __restrict__ off, const double* __restrict__ X,
                                                                 const double* __restrict__ Y, const double* __restrict__ u,
                                                                 const double* __restrict__ v, const double* __restrict__ K5,
                                                                 double* __restrict__ pose7) {
    const int i = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * SEED_WAVES + (threadIdx.x >> 6)));
    if (i >= n_views) return;  // whole wave leaves together
    double K[5], p[7];
    for (int k = 0; k < 5; ++k) K[k] = K5[k];
    WaveCoop co;
    planar_seed_view(static_cast<int>(off[i + 1] - off[i]), X + off[i], Y + off[i], u + off[i], v + off[i], K, co, p);
    if (co.lane() == 0)
        for (int k = 0; k < 7; ++k) pose7[7 * static_cast<int64_t>(i) + k] = p[k];
}

// estimate_homography's DLT path (homography.cpp:31-43) for a batch of views: H9 [n_views][9] row-major, ok [n_views]
__global__ __launch_bounds__(64 * SEED_WAVES) void k_dlt_homography(int n_views, const int64_t* __restrict__ off, const double* __restrict__ X,
                                                                    const double* __restrict__ Y, const double* __restrict__ u,
                                                                    const double* __restrict__ v, double* __restrict__ H9,
                                                                    int32_t* __restrict__ ok) {
    const int i = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * SEED_WAVES + (threadIdx.x >> 6)));
    if (i >= n_views) return;
    const double K[5] = {1.0, 1.0, 0.0, 0.0, 0.0};
    double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    WaveCoop co;
    const bool good = dlt_homography_view(static_cast<int>(off[i + 1] - off[i]), X + off[i], Y + off[i], u + off[i], v + off[i], K, co, H);
    if (co.lane() == 0) {
        for (int k = 0; k < 9; ++k) H9[9 * static_cast<int64_t>(i) + k] = H[k];
        ok[i] = good ? 1 : 0;
    }
}

void dlt_homography_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                          double* H9, int32_t* ok, int device) {
    CBA_HIP(hipSetDevice(device));
    StreamLease lease;
    const hipStream_t stream = lease;
    {
        const int64_t n_obs = view_offset[n_views];
        DevBuf<double> dX, dY, du, dv, dH;
        DevBuf<int64_t> doff;
        DevBuf<int32_t> dok;
        const size_t n = static_cast<size_t>(std::max<int64_t>(n_obs, 1));
        dX.alloc(n); dY.alloc(n); du.alloc(n); dv.alloc(n); dH.alloc(9 * static_cast<size_t>(n_views)); dok.alloc(n_views);
        doff.alloc(n_views + 1);
        dX.upload(X, n_obs, stream); dY.upload(Y, n_obs, stream); du.upload(u, n_obs, stream); dv.upload(v, n_obs, stream);
        doff.upload(view_offset, n_views + 1, stream);
        hipLaunchKernelGGL(k_dlt_homography, dim3((n_views + SEED_WAVES - 1) / SEED_WAVES), dim3(64 * SEED_WAVES), 0, stream, n_views,
                           doff.p, dX.p, dY.p, du.p, dv.p, dH.p, dok.p);
        CBA_HIP(hipGetLastError());
        dH.download(H9, 9 * static_cast<size_t>(n_views), stream);
        dok.download(ok, n_views, stream);
        CBA_HIP(hipStreamSynchronize(stream));
    }
}

void planar_seed_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                       const double* kmtx5, double* pose7, int device) {
    CBA_HIP(hipSetDevice(device));
    StreamLease lease;
    const hipStream_t stream = lease;
    {
        const int64_t n_obs = view_offset[n_views];
        DevBuf<double> dX, dY, du, dv, dK, dP;
        DevBuf<int64_t> doff;
        const size_t n = static_cast<size_t>(std::max<int64_t>(n_obs, 1));
        dX.alloc(n); dY.alloc(n); du.alloc(n); dv.alloc(n); dK.alloc(5); dP.alloc(7 * static_cast<size_t>(n_views));
        doff.alloc(n_views + 1);
        dX.upload(X, n_obs, stream); dY.upload(Y, n_obs, stream); du.upload(u, n_obs, stream); dv.upload(v, n_obs, stream);
        dK.upload(kmtx5, 5, stream); doff.upload(view_offset, n_views + 1, stream);
        hipLaunchKernelGGL(k_planar_seed, dim3((n_views + SEED_WAVES - 1) / SEED_WAVES), dim3(64 * SEED_WAVES), 0, stream, n_views,
                           doff.p, dX.p, dY.p, du.p, dv.p, dK.p, dP.p);
        CBA_HIP(hipGetLastError());
        dP.download(pose7, 7 * static_cast<size_t>(n_views), stream);
        CBA_HIP(hipStreamSynchronize(stream));
    }
}

}  // namespace cba
