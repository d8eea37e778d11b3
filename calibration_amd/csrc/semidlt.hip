// semidlt.hip — optimize_intrinsics_semidlt (src/estimation/optim/intrinsicssemidlt.cpp:155-191) on the GPU.
//
// One evaluation of the variable-projection functor (CalibVPResidual, intrinsicsemidltresidual.h:19-73) is
//   k_sd_pass1   one wavefront per view: N_v = A^T A, (A^T b)_v                     (semidlt_math.hpp)
//   k_sd_alpha   one thread: fixed-order sum over views, m x m Cholesky, alpha
//   k_sd_pass2   one wavefront per view: W^T W, W^T r, A^T W, dA^T r, |r|^2 in three register-sized parts
// on one stream with ONE device-to-host copy at the end; the O(#views) linear algebra of the LM step runs on the
// host (semidlt_core.hpp).  Lanes stride over the view's points (unit-stride loads), sums cross the wave in DPP.
//
// Several GPUs (one process each; cba_optimize_intrinsics_semidlt_sharded / _rccl): the views are sharded, each rank holds the
// observations of its own range [v0, v0 + n_local) and ALL poses.  The O(#observations) passes run on the local views; two sums
// cross the ranks per evaluation, both in device memory when the transport is RCCL: the m(m+1)/2 + m sums of pass 1 (between
// k_sd_sum1 and k_sd_alpha_finish, so that every rank eliminates with the same alpha) and the table of per-view sums, which every
// rank fills for its own rows and zeroes elsewhere (the all-reduce is then a gather).  The O(#views) arrow / Woodbury step of
// semidlt_core.hpp runs redundantly and identically on every rank from the identical table.
#include <rccl/rccl.h>
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

// the two halves of k_sd_alpha with the ranks' exchange between them: raw[0 .. N1) = this rank's sums in view order ...
template <int NR>
__global__ void k_sd_sum1(int n_views, const double* __restrict__ out1, double* __restrict__ raw) {
    using L = SDLayout<NR>;
    const int e = threadIdx.x;
    if (blockIdx.x != 0 || e >= L::N1) return;
    double acc = 0.0;
    for (int v = 0; v < n_views; ++v) acc += out1[static_cast<int64_t>(v) * L::N1 + e];
    raw[e] = acc;
}
// ... and, from the sums over all ranks, what k_sd_alpha leaves in `sums`
template <int NR>
__global__ void k_sd_alpha_finish(const double* __restrict__ raw, double* __restrict__ sums) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using L = SDLayout<NR>;
    constexpr int m = L::M;
    double Nf[m * m], al[m];
    int e = 0;
    for (int a = 0; a < m; ++a)
        for (int c = 0; c <= a; ++c, ++e) { Nf[a * m + c] = raw[e]; Nf[c * m + a] = raw[e]; }
    for (int a = 0; a < m; ++a) al[a] = raw[e + a];
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
    DevBuf<double> X, Y, u, v, kappa, poses, out1, sums, raw, out2, alpha, sview;
    DevBuf<int64_t> off;
    PinnedBuf<double> stage;  // host-callback transport: the exchanged range on its way through the host
    dim3 grid, block;
    int Vl = 0, v0 = 0;  // this rank's views [v0, v0 + Vl) of the V views of the problem
    cba_allreduce_fn fn = nullptr;
    void* user = nullptr;
    void* rccl = nullptr;
    bool multi = false;
    std::vector<int64_t> counts;  // observations of every view of the problem (the per-view RMS needs them)

    // n_views_total < 0: a single-rank problem (all views local, no transport)
    HipSemiDlt(int n_local, const int64_t* view_offset, const double* hX, const double* hY, const double* hu, const double* hv, int num_radial,
               int n_views_total = -1, int first_view = 0, cba_allreduce_fn allreduce = nullptr, void* allreduce_user = nullptr,
               void* rccl_comm = nullptr) {
        Vl = n_local; v0 = n_views_total < 0 ? 0 : first_view; V = n_views_total < 0 ? n_local : n_views_total; nr = num_radial;
        fn = allreduce; user = allreduce_user; rccl = rccl_comm; multi = n_views_total >= 0;
        if (multi && !fn && !rccl) throw std::invalid_argument("a sharded semi-DLT solve needs a transport");
        if (v0 < 0 || v0 + Vl > V) throw std::invalid_argument("view range outside the problem");
        const int64_t n_loc = view_offset[Vl];
        const size_t n = static_cast<size_t>(std::max<int64_t>(n_loc, 1));
        X.alloc(n); Y.alloc(n); u.alloc(n); v.alloc(n);
        X.upload(hX, n_loc, stream); Y.upload(hY, n_loc, stream); u.upload(hu, n_loc, stream); v.upload(hv, n_loc, stream);
        off.alloc(Vl + 1); off.upload(view_offset, Vl + 1, stream);
        kappa.alloc(5); poses.alloc(7 * static_cast<size_t>(V));
        out1.alloc(static_cast<size_t>(std::max(1, Vl)) * 20); sums.alloc(64); raw.alloc(64); alpha.alloc(8);
        out2.alloc(static_cast<size_t>(V) * n2()); sview.alloc(V);
        grid = dim3((std::max(1, Vl) + SD_WAVES - 1) / SD_WAVES); block = dim3(64 * SD_WAVES);
        counts.assign(V, 0);
        for (int i = 0; i < Vl; ++i) counts[v0 + i] = view_offset[i + 1] - view_offset[i];
        if (multi) {  // every rank learns every view's size: the same zero-padded sum as the tables of an evaluation
            std::vector<double> c(V, 0.0);
            for (int i = 0; i < Vl; ++i) c[v0 + i] = static_cast<double>(counts[v0 + i]);
            sview.upload(c.data(), V, stream);
            exchange(sview, V);
            sview.download(c.data(), V, stream);
            CBA_HIP(hipStreamSynchronize(stream));
            for (int i = 0; i < V; ++i) counts[i] = static_cast<int64_t>(c[i] + 0.5);
        }
        n_obs = 0;
        for (int i = 0; i < V; ++i) n_obs += counts[i];
        CBA_HIP(hipStreamSynchronize(stream));
    }
    ~HipSemiDlt() override { (void)hipStreamSynchronize(stream); }
    // sum buf[0 .. count) over the ranks, in device memory (RCCL, in place on the evaluator's stream) or through the host callback
    void exchange(DevBuf<double>& buf, size_t count) {
        if (rccl) {
            const ncclResult_t r = ncclAllReduce(buf.p, buf.p, count, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(rccl), stream);
            if (r != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
            return;
        }
        stage.reserve(count);
        buf.download(stage.p, count, stream);
        CBA_HIP(hipStreamSynchronize(stream));
        if (fn(stage.p, static_cast<int64_t>(count), user) != 0) throw std::runtime_error("allreduce callback failed");
        buf.upload(stage.p, count, stream);
    }
    // rows of the other ranks' views -> 0 (this rank's rows were just written by a kernel), then the gather-as-a-sum
    void gather_rows(DevBuf<double>& table, size_t width) {
        if (!multi) return;
        if (v0 > 0) CBA_HIP(hipMemsetAsync(table.p, 0, sizeof(double) * width * static_cast<size_t>(v0), stream));
        if (v0 + Vl < V)
            CBA_HIP(hipMemsetAsync(table.p + width * static_cast<size_t>(v0 + Vl), 0, sizeof(double) * width * static_cast<size_t>(V - v0 - Vl), stream));
        exchange(table, width * static_cast<size_t>(V));
    }
    void put(const double* kappa5, const double* poses7) {
        kappa.upload(kappa5, 5, stream);
        poses.upload(poses7, 7 * static_cast<size_t>(V), stream);
    }
    const double* my_poses() const { return poses.p + 7 * static_cast<size_t>(v0); }
#define SD_DISPATCH(KERNEL, G, B, ...)                                                                  \
    switch (nr) {                                                                                      \
        case 0: hipLaunchKernelGGL(KERNEL<0>, G, B, 0, stream, __VA_ARGS__); break;                    \
        case 1: hipLaunchKernelGGL(KERNEL<1>, G, B, 0, stream, __VA_ARGS__); break;                    \
        case 2: hipLaunchKernelGGL(KERNEL<2>, G, B, 0, stream, __VA_ARGS__); break;                    \
        default: hipLaunchKernelGGL(KERNEL<3>, G, B, 0, stream, __VA_ARGS__); break;                   \
    }
    void launch_pass1() {
        if (Vl > 0) SD_DISPATCH(k_sd_pass1, grid, block, Vl, off.p, X.p, Y.p, u.p, v.p, kappa.p, my_poses(), out1.p)
        if (!multi) {
            SD_DISPATCH(k_sd_alpha, dim3(1), dim3(64), Vl, out1.p, sums.p)
            return;
        }
        const int mm = m();
        SD_DISPATCH(k_sd_sum1, dim3(1), dim3(64), Vl, out1.p, raw.p)
        exchange(raw, static_cast<size_t>(mm * (mm + 1) / 2 + mm));
        SD_DISPATCH(k_sd_alpha_finish, dim3(1), dim3(64), raw.p, sums.p)
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
        if (Vl > 0)
            SD_DISPATCH(k_sd_pass2, grid, block, Vl, off.p, X.p, Y.p, u.p, v.p, kappa.p, my_poses(), sums.p + mm * mm + mm,
                        out2.p + static_cast<size_t>(v0) * n2())
        CBA_HIP(hipGetLastError());
        gather_rows(out2, static_cast<size_t>(n2()));
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
        if (Vl > 0) SD_DISPATCH(k_sd_resid, grid, block, Vl, off.p, X.p, Y.p, u.p, v.p, kappa.p, my_poses(), alpha.p, sview.p + v0)
        CBA_HIP(hipGetLastError());
        gather_rows(sview, 1);
        sview.download(s_view, V, stream);
        CBA_HIP(hipStreamSynchronize(stream));
    }
#undef SD_DISPATCH
};

void semidlt_run(HipSemiDlt& ev, double* kappa5, double* poses7, const double* bounds_lo, const double* bounds_hi, const int32_t* fixed_idx,
                 const double* fixed_val, int n_fixed, const cba_options* o, cba_summary* summary, double* distortion, double* view_errors,
                 double* cov) {
    SemiDltDriver drv(ev, *o);
    if (bounds_lo && bounds_hi) {
        drv.bounds.enabled = true;
        for (int k = 0; k < 5; ++k) { drv.bounds.lo[k] = bounds_lo[k]; drv.bounds.hi[k] = bounds_hi[k]; }
    }
    drv.solve(kappa5, poses7, summary);
    SemiDltResult res;
    double ssr = 0.0;
    std::vector<int64_t> off_all(static_cast<size_t>(ev.V) + 1, 0);
    for (int i = 0; i < ev.V; ++i) off_all[i + 1] = off_all[i] + ev.counts[i];
    drv.finish(fixed_idx, fixed_val, n_fixed, off_all.data(), res, &ssr);
    const int m = ev.m();
    if (distortion) for (int a = 0; a < m; ++a) distortion[a] = res.alpha[a];
    if (view_errors) for (int i = 0; i < ev.V; ++i) view_errors[i] = res.view_errors[i];
    if (cov) {
        const size_t dim = 5 + 7 * static_cast<size_t>(ev.V);
        std::memset(cov, 0, sizeof(double) * dim * dim);
        std::vector<double> c;
        if (drv.covariance(ssr, c)) std::memcpy(cov, c.data(), sizeof(double) * dim * dim);
    }
}
}  // namespace

void semidlt_solve(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                   double* kappa5, double* poses7, int num_radial, const double* bounds_lo, const double* bounds_hi,
                   const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const cba_options* o, cba_summary* summary,
                   double* distortion, double* view_errors, double* cov, int device) {
    CBA_HIP(hipSetDevice(device));
    HipSemiDlt ev(n_views, view_offset, X, Y, u, v, num_radial);
    semidlt_run(ev, kappa5, poses7, bounds_lo, bounds_hi, fixed_idx, fixed_val, n_fixed, o, summary, distortion, view_errors, cov);
}

// The same solve with the views sharded over ranks: this rank holds the observations of views [first_view, first_view + n_local)
// of n_views_total; kappa5, poses7 [n_views_total][7], view_errors [n_views_total] and cov cover the WHOLE problem and come out
// identical on every rank.  Transport: fn (host callback) or rccl_comm (ncclComm_t).
void semidlt_solve_sharded(int n_local, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                           int n_views_total, int first_view, double* kappa5, double* poses7, int num_radial, const double* bounds_lo,
                           const double* bounds_hi, const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const cba_options* o,
                           cba_summary* summary, double* distortion, double* view_errors, double* cov, int device, cba_allreduce_fn fn,
                           void* user, void* rccl_comm) {
    CBA_HIP(hipSetDevice(device));
    HipSemiDlt ev(n_local, view_offset, X, Y, u, v, num_radial, n_views_total, first_view, fn, user, rccl_comm);
    semidlt_run(ev, kappa5, poses7, bounds_lo, bounds_hi, fixed_idx, fixed_val, n_fixed, o, summary, distortion, view_errors, cov);
}

}  // namespace cba
