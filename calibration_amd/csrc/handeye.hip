// handeye.hip — AX = XB refinement on the GPU (optimize_handeye, handeye.cpp:60-78).
//
// k_axxb: one thread per ordered pose pair (i < j).  Pairs are NOT materialised (the reference's
// build_all_pairs stores 192 B per pair, 384 MB at 2000 poses): the thread rebuilds the motion pair
// from the two poses' rotations/translations (24 doubles each, L2-resident), applies the reference's
// pair filter, evaluates residual + analytic tangent Jacobian, applies the per-pair Huber weight and
// accumulates [H | g | cost | count] — wave-shuffle + LDS reduction per workgroup, then a fixed-order
// two-level sum over workgroups (no atomics; bitwise reproducible).  Compute-bound: ~1.2 kFLOP per pair.
#include <rccl/rccl.h>

#include "engine.hpp"
#include "handeye_core.hpp"
#include "seed_math.hpp"
#include "wave_reduce.hpp"

namespace cba {

// MODE 0: the AX = XB residual blocks of optimize_handeye; MODE 1 / 2: the rotation / translation sums of the Tsai-Lenz
// all-pairs seed estimate_handeye_dlt (handeyedlt.cpp:84-137) — same pair enumeration, same filter, same reduction.
template <int MODE>
__global__ __launch_bounds__(256) void k_axxb(int n, const double* __restrict__ poses /*[n][24]: Rb tb Rc tc*/,
                                              const double* __restrict__ X /*RX(9) tX(3)*/, double min_angle,
                                              double axis_eps, double huber_delta, int i_first, double* __restrict__ partial) {
    __shared__ double sh[4][AXXB_NACC];
    const int i = i_first + blockIdx.y;  // this launch covers first poses [i_first, i_first + gridDim.y)
    const int j = blockIdx.x * 256 + threadIdx.x;
    double acc[AXXB_NACC];
#pragma unroll
    for (int e = 0; e < AXXB_NACC; ++e) acc[e] = 0.0;
    if (j > i && j < n) {
        const double* pi = poses + 24 * static_cast<int64_t>(i);
        const double* pj = poses + 24 * static_cast<int64_t>(j);
        double RA[9], RB[9], tA[3], tB[3];
        if (motion_pair(pi, pi + 9, pj, pj + 9, pi + 12, pi + 21, pj + 12, pj + 21, min_angle, axis_eps, RA, RB, tA, tB)) {
            if (MODE == 0) {
                double r[6], J[36];
                axxb_point(X, X + 9, RA, RB, tA, tB, r, J);
                axxb_accumulate(r, J, huber_delta, acc);
            } else {
                tsai_lenz_accumulate(MODE - 1, RA, RB, tA, tB, X, acc);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < AXXB_NACC; ++e) {
        const double v = wave_sum63(acc[e]);
        if (lane == 63) sh[wave][e] = v;
    }
    __syncthreads();
    if (threadIdx.x < AXXB_NACC)
        partial[(static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * AXXB_NACC + threadIdx.x] =
            (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// out[b][e] = sum of rows [b * chunk, (b + 1) * chunk) of a row-major [n_rows][AXXB_NACC] table, in a FIXED order: thread
// (column e, group r) adds rows r, r + 8, ... of the chunk, the 8 group sums are combined in group order through LDS.
// Applied twice (chunks of 64 rows, then the <= few hundred chunk sums) it replaces one wave walking all ~16 000 workgroup
// partials of a 2000-pose problem serially (3.8 ms per evaluation, 17x the pair kernel itself).
__global__ __launch_bounds__(256) void k_axxb_chunk_sum(int64_t n_rows, int64_t chunk, const double* __restrict__ rows,
                                                        double* __restrict__ out) {
    __shared__ double sh[8][32];
    const int e = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * chunk, t1 = t0 + chunk < n_rows ? t0 + chunk : n_rows;
    double s = 0.0;
    if (e < AXXB_NACC)
        for (int64_t t = t0 + r; t < t1; t += 8) s += rows[t * AXXB_NACC + e];
    sh[r][e] = s;
    __syncthreads();
    if (r == 0 && e < AXXB_NACC) {
        double tot = 0.0;
        for (int k = 0; k < 8; ++k) tot += sh[k][e];
        out[static_cast<int64_t>(blockIdx.x) * AXXB_NACC + e] = tot;
    }
}

namespace {
struct HipAxxb final : AxxbEval {
    int n;
    StreamLease lease;  // before the buffers: released after them
    hipStream_t stream = lease;
    DevBuf<double> poses, X, partial, partial2, out;
    int64_t n_rows = 0, n_chunks = 0;
    dim3 grid;
    int i_first = 0;              // first poses [i_first, i_first + grid.y) are this rank's (axxb_rank_range)
    cba_allreduce_fn reduce = nullptr;  // in-place sum of the 29 accumulated values over ranks (host callback)
    void* reduce_user = nullptr;
    void* rccl = nullptr;               // ... or an RCCL communicator: ncclAllReduce of the device-resident sums on this stream
    HipAxxb(int n_poses, const double* bTg, const double* cTt, cba_allreduce_fn fn = nullptr, void* user = nullptr, int n_ranks = 1,
            int rank = 0, void* rccl_comm = nullptr)
        : n(n_poses), reduce(n_ranks > 1 ? fn : nullptr), reduce_user(user), rccl(rccl_comm) {
        if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("bad rank / n_ranks");
        if (n_ranks > 1 && !fn && !rccl_comm) throw std::invalid_argument("multi-rank AX = XB needs an all-reduce callback or an RCCL communicator");
        int i_last = std::max(1, n - 1);
        if (n_ranks > 1) axxb_rank_range(n, n_ranks, rank, &i_first, &i_last);
        std::vector<double> h(static_cast<size_t>(n) * 24);
        for (int k = 0; k < n; ++k) {
            // R from the (unit) quaternion exactly as Eigen::Isometry3d stored it before populate_quat_tran
            double nb = 0, nc = 0, qb[4], qc[4];
            for (int a = 0; a < 4; ++a) { nb += bTg[7 * k + a] * bTg[7 * k + a]; nc += cTt[7 * k + a] * cTt[7 * k + a]; }
            for (int a = 0; a < 4; ++a) { qb[a] = bTg[7 * k + a] / std::sqrt(nb); qc[a] = cTt[7 * k + a] / std::sqrt(nc); }
            quat_to_rotmat(qb, &h[24 * static_cast<size_t>(k)]);
            quat_to_rotmat(qc, &h[24 * static_cast<size_t>(k) + 12]);
            for (int a = 0; a < 3; ++a) { h[24 * static_cast<size_t>(k) + 9 + a] = bTg[7 * k + 4 + a]; h[24 * static_cast<size_t>(k) + 21 + a] = cTt[7 * k + 4 + a]; }
        }
        grid = dim3((n + 255) / 256, std::max(0, i_last - i_first));  // y = 0: this rank has no pairs
        poses.alloc(h.size()); poses.upload(h.data(), h.size(), stream);
        X.alloc(12); out.alloc(AXXB_NACC);
        n_rows = static_cast<int64_t>(grid.x) * grid.y;
        n_chunks = (n_rows + 63) / 64;
        partial.alloc(static_cast<size_t>(std::max<int64_t>(n_rows, 1)) * AXXB_NACC);
        partial2.alloc(static_cast<size_t>(std::max<int64_t>(n_chunks, 1)) * AXXB_NACC);
        CBA_HIP(hipStreamSynchronize(stream));
    }
    // pair sums of this rank's range at X (already uploaded), two-level fixed-order sum, then the sum over ranks
    template <int MODE>
    void pass(double min_angle, double huber_delta, double* acc) {
        if (grid.y > 0) {
            hipLaunchKernelGGL(k_axxb<MODE>, grid, dim3(256), 0, stream, n, poses.p, X.p, min_angle, 1e-3, huber_delta, i_first, partial.p);
            hipLaunchKernelGGL(k_axxb_chunk_sum, dim3(static_cast<unsigned>(n_chunks)), dim3(256), 0, stream, n_rows, int64_t{64}, partial.p, partial2.p);
            hipLaunchKernelGGL(k_axxb_chunk_sum, dim3(1), dim3(256), 0, stream, n_chunks, n_chunks, partial2.p, out.p);
            CBA_HIP(hipGetLastError());
        } else if (rccl) {
            CBA_HIP(hipMemsetAsync(out.p, 0, AXXB_NACC * sizeof(double), stream));  // this rank has no pairs: it contributes zeros
        }
        if (rccl) {  // the sum over ranks on the device, in place, on this stream: no host staging before the one copy back
            const ncclResult_t r = ncclAllReduce(out.p, out.p, AXXB_NACC, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(rccl), stream);
            if (r != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        }
        if (grid.y > 0 || rccl) {
            out.download(acc, AXXB_NACC, stream);
            CBA_HIP(hipStreamSynchronize(stream));
        } else {
            for (int e = 0; e < AXXB_NACC; ++e) acc[e] = 0.0;
        }
        if (reduce && !rccl && reduce(acc, AXXB_NACC, reduce_user) != 0) throw std::runtime_error("allreduce callback failed");
    }
    ~HipAxxb() override { (void)hipStreamSynchronize(stream); }
    void eval(const double* pose7, double huber_delta, double* acc) override {
        double hx[12];
        quat_to_rotmat(pose7, hx);  // un-normalised, as quat_array_to_rotmat (observationutils.h:20-24)
        for (int a = 0; a < 3; ++a) hx[9 + a] = pose7[4 + a];
        X.upload(hx, 12, stream);
        constexpr double kMinAngleDeg = 0.5;  // handeye.cpp:64
        pass<0>(kMinAngleDeg * 3.14159265358979323846 / 180.0, huber_delta, acc);
    }
};
}  // namespace

// estimate_handeye_dlt (handeyedlt.cpp:126-137): two passes over all pose pairs on the device (rotation sums, then
// translation sums at the estimated R_X), two 3x3 ridge solves on the host.  pose7 out.
void handeye_dlt(int n_poses, const double* bTg, const double* cTt, double min_angle_deg, double* pose7, int device, cba_allreduce_fn fn,
                 void* user, int n_ranks, int rank, void* rccl_comm) {
    if (n_poses < 2 || !bTg || !cTt)  // handeyedlt.cpp:56-58
        throw std::runtime_error("Inconsistent hand-eye input sizes");
    CBA_HIP(hipSetDevice(device));
    HipAxxb ev(n_poses, bTg, cTt, fn, user, n_ranks, rank, rccl_comm);
    const double min_angle = min_angle_deg * 3.14159265358979323846 / 180.0;
    auto pass = [&](int mode, const double* RX, double* acc) {
        double hx[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        if (RX) for (int k = 0; k < 9; ++k) hx[k] = RX[k];
        ev.X.upload(hx, 12, ev.stream);
        if (mode == 0) ev.pass<1>(min_angle, 0.0, acc);
        else ev.pass<2>(min_angle, 0.0, acc);
    };
    double acc[AXXB_NACC], w[3], RX[9], t[3];
    pass(0, nullptr, acc);
    if (acc[9] < 0.5)  // handeyedlt.cpp:76-79
        throw std::runtime_error("No valid motion pairs after filtering. Increase motion or relax thresholds.");
    if (!tsai_lenz_solve(acc, 1e-12, w)) throw std::runtime_error("Tsai-Lenz rotation system is singular");
    exp_so3(w, RX);
    pass(1, RX, acc);
    if (!tsai_lenz_solve(acc, 1e-12, t)) throw std::runtime_error("Tsai-Lenz translation system is singular");
    seed_rotmat_to_quat(RX, pose7);
    for (int k = 0; k < 3; ++k) pose7[4 + k] = t[k];
}

void handeye_solve(int n_poses, const double* bTg, const double* cTt, double* pose7, const cba_options* o, cba_summary* s,
                   double* cov, int device, cba_allreduce_fn fn, void* user, int n_ranks, int rank, void* rccl_comm) {
    if (n_poses < 2 || !bTg || !cTt)  // handeyedlt.cpp:56-58
        throw std::runtime_error("Inconsistent hand-eye input sizes");
    CBA_HIP(hipSetDevice(device));
    HipAxxb ev(n_poses, bTg, cTt, fn, user, n_ranks, rank, rccl_comm);
    handeye_lm(ev, pose7, *o, s, (cov && o->compute_covariance) ? cov : nullptr);
}

}  // namespace cba
