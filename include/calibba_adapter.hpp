// calibba_adapter.hpp — header-only C++ adapter that gives libcalibba.so the EXACT signatures of the
// reference's refinement entry points, so `calib::pipeline` and the tests can link against it instead of
// calib_estimation_optim.  It needs Eigen and the reference's own headers (calib/models/*,
// calib/estimation/optim/*), therefore it is compiled in the reference's tree, not in this repository
// (neither Eigen nor Ceres exist in the build image).  It only flattens AoS containers into the SoA
// buffers of include/calibba.h and maps status codes back to the reference's exception types.
//
//   replaces                                               (reference file:line)
//   calib::optimize_intrinsics<CameraT>                    src/estimation/optim/intrinsics.cpp:98-120
//   calib::optimize_extrinsics<CameraT>                    src/estimation/optim/extrinsics.cpp:174-196
//   calib::optimize_bundle<CameraT>                        src/estimation/optim/bundle.cpp:147-170
//   calib::optimize_handeye                                src/estimation/optim/handeye.cpp:60-78
//   calib::estimate_and_optimize_handeye                   src/estimation/optim/handeye.cpp:80-87
//   calib::optimize_intrinsics_semidlt                     src/estimation/optim/intrinsicssemidlt.cpp:155-191
//   calib::optimize_planar_pose                            src/estimation/optim/planarpose.cpp:84-127
//   calib::optimize_homography                             src/estimation/optim/homography.cpp:144-175
#pragma once
#include <Eigen/Geometry>
#include <algorithm>
#include <array>
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "calib/estimation/linear/planarpose.h"
#include "calib/estimation/optim/bundle.h"
#include "calib/estimation/optim/extrinsics.h"
#include "calib/estimation/optim/handeye.h"
#include "calib/estimation/optim/homography.h"
#include "calib/estimation/optim/intrinsics.h"
#include "calib/estimation/optim/planarpose.h"
#include "calib/models/scheimpflug.h"
#include "calibba.h"

namespace calibba_adapter {

inline void check(cba_status st) {
    if (st == CBA_OK) return;
    if (st == CBA_ERR_INVALID_ARGUMENT) throw std::invalid_argument(cba_last_error());
    throw std::runtime_error(cba_last_error());
}

template <class CameraT> struct ModelOf { static constexpr int value = CBA_CAMERA_PINHOLE_BC; };
template <class Inner> struct ModelOf<calib::ScheimpflugCamera<Inner>> { static constexpr int value = CBA_CAMERA_SCHEIMPFLUG; };

inline cba_options make_options(const calib::OptimOptions& core) {
    cba_options o;
    cba_options_default(&o);
    o.optimizer = static_cast<int32_t>(core.optimizer);
    o.huber_delta = core.huber_delta;
    o.epsilon = core.epsilon;
    o.max_iterations = core.max_iterations;
    o.compute_covariance = core.compute_covariance;
    o.verbose = core.verbose;
    return o;
}

inline void fill_core(const cba_summary& s, const cba_options& o, const std::vector<double>& cov, Eigen::Index dim,
                      calib::OptimResult& core) {
    core.success = s.success != 0;
    core.final_cost = s.final_cost;
    core.report = s.report;
    if (o.compute_covariance && dim > 0) {
        // row-major symmetric -> Eigen (symmetric, so the storage order is immaterial); the engine zero-fills
        // the matrix when the Jacobian is rank deficient, where the reference leaves it empty
        Eigen::MatrixXd m = Eigen::Map<const Eigen::MatrixXd>(cov.data(), dim, dim);
        if (m.diagonal().cwiseAbs().maxCoeff() > 0.0) core.covariance = std::move(m);
    }
}

// The reference's PlanarObservation {Eigen::Vector2d object_xy, image_uv} (linear/planarpose.h:22-26) is four contiguous
// doubles, so a PlanarView's storage IS the {X, Y, u, v} record array cba_reproj_create_aos reads in place: the three
// reprojection solvers below never copy an observation on the host.
static_assert(sizeof(calib::PlanarObservation) == 4 * sizeof(double), "PlanarObservation must be {object_xy, image_uv} = 4 doubles");
struct Records {
    std::vector<const double*> obs;
    std::vector<int64_t> off{0};
    std::vector<int32_t> cam, view;
    std::vector<double> bTg;
    void push(const calib::PlanarView& pv, int c, int vw) {
        obs.push_back(reinterpret_cast<const double*>(pv.data()));
        off.push_back(off.back() + static_cast<int64_t>(pv.size()));
        cam.push_back(c); view.push_back(vw);
    }
};

// what the one-shot C entry points do (capi.cpp one_shot), on records: handle, solve, parameters back, covariance, release
inline void solve_records(cba_reproj_problem& d, const Records& r, const cba_options& o, cba_summary* sum, std::vector<double>& cov) {
    d.n_blocks = static_cast<int32_t>(r.obs.size());
    d.blk_offset = r.off.data();
    cba_reproj* h = nullptr;
    check(cba_reproj_create_aos(&d, r.obs.data(), 0, &h));
    struct Release { cba_reproj* h; ~Release() { cba_reproj_destroy(h); } } release{h};
    check(cba_reproj_solve(h, &o, sum));
    check(cba_reproj_get_params(h, d.intr, d.cam_pose, d.view_pose, d.target_pose));
    if (!cov.empty()) {
        const cba_status st = cba_reproj_covariance(h, &o, cov.data());
        if (st == CBA_ERR_RUNTIME) std::fill(cov.begin(), cov.end(), 0.0);  // rank deficient: the reference leaves the matrix empty
        else check(st);
    }
}

inline void pose_in(const Eigen::Isometry3d& T, double* p7) { cba_pose_from_matrix(T.data(), p7); }
inline Eigen::Isometry3d pose_out(const double* p7) {
    Eigen::Isometry3d T;
    cba_pose_to_matrix(p7, T.data());
    return T;
}

template <calib::camera_model CameraT>
auto optimize_intrinsics(const std::vector<calib::PlanarView>& views, const CameraT& init_camera,
                         std::vector<Eigen::Isometry3d> init_c_se3_t, const calib::IntrinsicsOptimOptions& opts = {})
    -> calib::IntrinsicsOptimizationResult<CameraT> {
    using Traits = calib::CameraTraits<CameraT>;
    std::array<double, Traits::param_count> intr{};
    Traits::to_array(init_camera, intr);
    if (views.size() < 4)  // intrinsics.cpp:92-96
        throw std::invalid_argument("Insufficient views for calibration (at least 4 required).");
    Records s;
    for (size_t i = 0; i < views.size(); ++i) s.push(views[i], 0, static_cast<int>(i));
    std::vector<double> poses(7 * init_c_se3_t.size());
    for (size_t i = 0; i < init_c_se3_t.size(); ++i) pose_in(init_c_se3_t[i], &poses[7 * i]);
    cba_options o = make_options(opts.core);
    o.optimize_skew = opts.optimize_skew;
    cba_summary sum{};
    const Eigen::Index dim = static_cast<Eigen::Index>(Traits::param_count + 7 * views.size());
    std::vector<double> cov(o.compute_covariance ? static_cast<size_t>(dim * dim) : 0);
    o.optimize_intrinsics = 1;
    cba_reproj_problem d{};
    d.chain = CBA_CHAIN_INTRINSIC; d.camera_model = ModelOf<CameraT>::value;
    d.n_cams = 1; d.n_views = static_cast<int32_t>(views.size());
    d.intr = intr.data(); d.view_pose = poses.data();
    solve_records(d, s, o, &sum, cov);
    calib::IntrinsicsOptimizationResult<CameraT> res;
    res.camera = Traits::template from_array<double>(intr.data());
    res.c_se3_t.resize(views.size());
    for (size_t i = 0; i < views.size(); ++i) res.c_se3_t[i] = pose_out(&poses[7 * i]);
    fill_core(sum, o, cov, dim, res.core);
    return res;
}

template <calib::camera_model CameraT>
auto optimize_extrinsics(const std::vector<calib::MulticamPlanarView>& views, const std::vector<CameraT>& init_cameras,
                         const std::vector<Eigen::Isometry3d>& init_c_se3_r, const std::vector<Eigen::Isometry3d>& init_r_se3_t,
                         const calib::ExtrinsicOptions& opts = {}) -> calib::ExtrinsicOptimizationResult<CameraT> {
    using Traits = calib::CameraTraits<CameraT>;
    constexpr size_t P = Traits::param_count;
    const size_t C = init_cameras.size(), V = views.size();
    if (init_c_se3_r.size() != C || init_r_se3_t.size() != V)  // extrinsics.cpp:162-172
        throw std::invalid_argument("Incompatible pose vector sizes for joint optimization");
    std::vector<double> intr(C * P), cams(7 * C), tgts(7 * V);
    for (size_t c = 0; c < C; ++c) {
        std::array<double, P> a{};
        Traits::to_array(init_cameras[c], a);
        std::copy(a.begin(), a.end(), intr.begin() + c * P);
        pose_in(init_c_se3_r[c], &cams[7 * c]);
    }
    for (size_t v = 0; v < V; ++v) pose_in(init_r_se3_t[v], &tgts[7 * v]);
    Records s;
    for (size_t v = 0; v < V; ++v)
        for (size_t c = 0; c < C; ++c)
            if (!views[v][c].empty()) s.push(views[v][c], static_cast<int>(c), static_cast<int>(v));  // extrinsics.cpp:94-96
    cba_options o = make_options(opts.core);
    o.optimize_intrinsics = opts.optimize_intrinsics;
    o.optimize_skew = opts.optimize_skew;
    o.optimize_extrinsics = opts.optimize_extrinsics;
    cba_summary sum{};
    const Eigen::Index dim = static_cast<Eigen::Index>(C * (P + 7) + 7 * V);
    std::vector<double> cov(o.compute_covariance ? static_cast<size_t>(dim * dim) : 0);
    cba_reproj_problem d{};
    d.chain = CBA_CHAIN_EXTRINSIC; d.camera_model = ModelOf<CameraT>::value;
    d.n_cams = static_cast<int32_t>(C); d.n_views = static_cast<int32_t>(V);
    d.blk_view = s.view.data(); d.blk_cam = s.cam.data();
    d.intr = intr.data(); d.cam_pose = cams.data(); d.view_pose = tgts.data();
    solve_records(d, s, o, &sum, cov);
    calib::ExtrinsicOptimizationResult<CameraT> res;
    res.cameras.resize(C); res.c_se3_r.resize(C); res.r_se3_t.resize(V);
    for (size_t c = 0; c < C; ++c) {
        res.cameras[c] = Traits::template from_array<double>(&intr[c * P]);
        res.c_se3_r[c] = pose_out(&cams[7 * c]);
    }
    for (size_t v = 0; v < V; ++v) res.r_se3_t[v] = pose_out(&tgts[7 * v]);
    fill_core(sum, o, cov, dim, res.core);
    return res;
}

template <calib::camera_model CameraT>
auto optimize_bundle(const std::vector<calib::BundleObservation>& observations, const std::vector<CameraT>& initial_cameras,
                     const std::vector<Eigen::Isometry3d>& init_g_se3_c, const Eigen::Isometry3d& init_b_se3_t,
                     const calib::BundleOptions& opts = {}) -> calib::BundleResult<CameraT> {
    using Traits = calib::CameraTraits<CameraT>;
    constexpr size_t P = Traits::param_count;
    const size_t C = initial_cameras.size();
    if (C == 0) throw std::invalid_argument("No camera intrinsics provided");    // bundle.cpp:139-141
    if (observations.empty()) throw std::invalid_argument("No observations provided");  // bundle.cpp:142-144
    std::vector<double> intr(C * P), g(7 * C), bt(7);
    for (size_t c = 0; c < C; ++c) {
        std::array<double, P> a{};
        Traits::to_array(initial_cameras[c], a);
        std::copy(a.begin(), a.end(), intr.begin() + c * P);
        pose_in(init_g_se3_c[c], &g[7 * c]);
    }
    pose_in(init_b_se3_t, bt.data());
    Records s;
    for (const auto& ob : observations) {
        s.push(ob.view, static_cast<int>(ob.camera_index), 0);
        const Eigen::Matrix3d R = ob.b_se3_g.linear();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) s.bTg.push_back(R(r, c));  // row-major rotation
        for (int k = 0; k < 3; ++k) s.bTg.push_back(ob.b_se3_g.translation()(k));
    }
    cba_options o = make_options(opts.core);
    o.optimize_intrinsics = opts.optimize_intrinsics;
    o.optimize_skew = opts.optimize_skew;
    o.optimize_extrinsics = opts.optimize_hand_eye;
    o.optimize_target_pose = opts.optimize_target_pose;
    cba_summary sum{};
    const Eigen::Index dim = static_cast<Eigen::Index>(C * (P + 7) + 7);
    std::vector<double> cov(o.compute_covariance ? static_cast<size_t>(dim * dim) : 0);
    cba_reproj_problem d{};
    d.chain = CBA_CHAIN_BUNDLE; d.camera_model = ModelOf<CameraT>::value;
    d.n_cams = static_cast<int32_t>(C); d.n_views = 0;
    d.blk_cam = s.cam.data(); d.blk_b_T_g = s.bTg.data();
    d.intr = intr.data(); d.cam_pose = g.data(); d.target_pose = bt.data();
    solve_records(d, s, o, &sum, cov);
    calib::BundleResult<CameraT> res;
    res.cameras.resize(C); res.g_se3_c.resize(C);
    for (size_t c = 0; c < C; ++c) {
        res.cameras[c] = Traits::template from_array<double>(&intr[c * P]);
        res.g_se3_c[c] = pose_out(&g[7 * c]);
    }
    res.b_se3_t = pose_out(bt.data());
    fill_core(sum, o, cov, dim, res.core);
    return res;
}

inline auto optimize_handeye(const std::vector<Eigen::Isometry3d>& base_se3_gripper,
                             const std::vector<Eigen::Isometry3d>& camera_se3_target,
                             const Eigen::Isometry3d& init_gripper_se3_ref, const calib::OptimOptions& options = {})
    -> calib::HandeyeResult {
    if (base_se3_gripper.size() < 2 || base_se3_gripper.size() != camera_se3_target.size())  // handeyedlt.cpp:56-58
        throw std::runtime_error("Inconsistent hand-eye input sizes");
    const size_t n = base_se3_gripper.size();
    std::vector<double> bg(7 * n), ct(7 * n), x(7), cov(49);
    for (size_t i = 0; i < n; ++i) { pose_in(base_se3_gripper[i], &bg[7 * i]); pose_in(camera_se3_target[i], &ct[7 * i]); }
    pose_in(init_gripper_se3_ref, x.data());
    cba_options o = make_options(options);
    cba_summary sum{};
    check(cba_optimize_handeye(static_cast<int32_t>(n), bg.data(), ct.data(), x.data(), &o, &sum, cov.data()));
    calib::HandeyeResult res;
    res.g_se3_c = pose_out(x.data());
    fill_core(sum, o, cov, 7, res.core);
    return res;
}

// estimate_and_optimize_handeye (handeye.cpp:80-87): the Tsai-Lenz all-pairs seed and the refinement, both on the device
inline auto estimate_and_optimize_handeye(const std::vector<Eigen::Isometry3d>& base_se3_gripper,
                                          const std::vector<Eigen::Isometry3d>& camera_se3_target, double min_angle_deg = 1.0,
                                          const calib::OptimOptions& options = {}) -> calib::HandeyeResult {
    if (base_se3_gripper.size() < 2 || base_se3_gripper.size() != camera_se3_target.size())  // handeyedlt.cpp:56-58
        throw std::runtime_error("Inconsistent hand-eye input sizes");
    const size_t n = base_se3_gripper.size();
    std::vector<double> bg(7 * n), ct(7 * n), x(7), cov(49);
    for (size_t i = 0; i < n; ++i) { pose_in(base_se3_gripper[i], &bg[7 * i]); pose_in(camera_se3_target[i], &ct[7 * i]); }
    cba_options o = make_options(options);
    cba_summary sum{};
    check(cba_estimate_and_optimize_handeye(static_cast<int32_t>(n), bg.data(), ct.data(), min_angle_deg, x.data(), &o, &sum, cov.data()));
    calib::HandeyeResult res;
    res.g_se3_c = pose_out(x.data());
    fill_core(sum, o, cov, 7, res.core);
    return res;
}

inline void flatten_view(const calib::PlanarView& view, std::vector<double>& X, std::vector<double>& Y, std::vector<double>& u,
                         std::vector<double>& v) {
    X.reserve(view.size()); Y.reserve(view.size()); u.reserve(view.size()); v.reserve(view.size());
    for (const auto& ob : view) {
        X.push_back(ob.object_xy.x()); Y.push_back(ob.object_xy.y());
        u.push_back(ob.image_uv.x()); v.push_back(ob.image_uv.y());
    }
}

inline auto optimize_intrinsics_semidlt(const std::vector<calib::PlanarView>& views, const calib::CameraMatrix& initial_guess,
                                        const calib::IntrinsicsOptimOptions& opts = {})
    -> calib::IntrinsicsOptimizationResult<calib::PinholeCamera<calib::BrownConradyd>> {
    calib::IntrinsicsOptimizationResult<calib::PinholeCamera<calib::BrownConradyd>> res;
    if (views.size() < 4) {  // intrinsicssemidlt.cpp:163-166
        std::cerr << "Insufficient views for calibration (at least 4 required)." << '\n';
        return res;
    }
    std::vector<int64_t> off{0};
    std::vector<double> X, Y, u, v, poses(7 * views.size());
    for (size_t i = 0; i < views.size(); ++i) {
        flatten_view(views[i], X, Y, u, v);
        off.push_back(static_cast<int64_t>(X.size()));
        pose_in(calib::estimate_planar_pose(views[i], initial_guess), &poses[7 * i]);  // IntrinsicBlocks::create, :37-40 (host seed)
    }
    double K[5] = {initial_guess.fx, initial_guess.fy, initial_guess.cx, initial_guess.cy, initial_guess.skew};
    double lo[5], hi[5];
    if (opts.bounds.has_value()) {
        const auto& b = *opts.bounds;
        const double l[5] = {b.fx_min, b.fy_min, b.cx_min, b.cy_min, b.skew_min}, h[5] = {b.fx_max, b.fy_max, b.cx_max, b.cy_max, b.skew_max};
        for (int k = 0; k < 5; ++k) { lo[k] = l[k]; hi[k] = h[k]; }
    }
    std::vector<int32_t> fidx(opts.fixed_distortion_indices.begin(), opts.fixed_distortion_indices.end());
    std::vector<double> fval(fidx.size(), 0.0);
    for (size_t i = 0; i < fidx.size() && i < opts.fixed_distortion_values.size(); ++i) fval[i] = opts.fixed_distortion_values[i];
    cba_options o = make_options(opts.core);
    o.optimize_skew = opts.optimize_skew;
    o.compute_covariance = 1;  // the reference computes it unconditionally (:184-188)
    cba_summary sum{};
    const Eigen::Index dim = static_cast<Eigen::Index>(5 + 7 * views.size());
    std::vector<double> dist(static_cast<size_t>(opts.num_radial) + 2), verr(views.size()), cov(static_cast<size_t>(dim * dim));
    check(cba_optimize_intrinsics_semidlt(static_cast<int32_t>(views.size()), off.data(), X.data(), Y.data(), u.data(), v.data(), K,
                                          poses.data(), opts.num_radial, opts.bounds ? lo : nullptr, opts.bounds ? hi : nullptr,
                                          fidx.empty() ? nullptr : fidx.data(), fval.empty() ? nullptr : fval.data(),
                                          static_cast<int32_t>(fidx.size()), &o, &sum, dist.data(), verr.data(), cov.data()));
    res.camera.kmtx.fx = K[0]; res.camera.kmtx.fy = K[1]; res.camera.kmtx.cx = K[2]; res.camera.kmtx.cy = K[3]; res.camera.kmtx.skew = K[4];
    res.camera.distortion.coeffs = Eigen::Map<const Eigen::VectorXd>(dist.data(), static_cast<Eigen::Index>(dist.size()));
    res.c_se3_t.resize(views.size());
    for (size_t i = 0; i < views.size(); ++i) res.c_se3_t[i] = pose_out(&poses[7 * i]);
    res.view_errors = verr;
    fill_core(sum, o, cov, dim, res.core);
    return res;
}

inline auto optimize_planar_pose(const calib::PlanarView& view, const calib::CameraMatrix& intrinsics,
                                 const Eigen::Isometry3d& init_pose, const calib::PlanarPoseOptions& opts = {})
    -> calib::PlanarPoseResult {
    std::vector<double> X, Y, u, v;
    flatten_view(view, X, Y, u, v);
    const double K[5] = {intrinsics.fx, intrinsics.fy, intrinsics.cx, intrinsics.cy, intrinsics.skew};
    double p7[7], rms = 0.0;
    pose_in(init_pose, p7);
    std::vector<double> dist(static_cast<size_t>(opts.num_radial) + 2), cov(36);
    cba_options o = make_options(opts.core);
    cba_summary sum{};
    check(cba_optimize_planar_pose(static_cast<int32_t>(view.size()), X.data(), Y.data(), u.data(), v.data(), K, opts.num_radial, p7,
                                   &o, &sum, dist.data(), &rms, cov.data()));
    calib::PlanarPoseResult res;
    res.pose = pose_out(p7);
    res.distortion = Eigen::Map<const Eigen::VectorXd>(dist.data(), static_cast<Eigen::Index>(dist.size()));
    res.reprojection_error = rms;
    fill_core(sum, o, cov, 6, res.core);
    return res;
}

inline auto optimize_homography(const calib::PlanarView& data, const Eigen::Matrix3d& init_h, const calib::OptimOptions& options = {})
    -> calib::OptimizeHomographyResult {
    if (data.size() < 4) throw std::invalid_argument("At least 4 correspondences are required.");  // homography.cpp:146-148
    std::vector<double> X, Y, u, v;
    flatten_view(data, X, Y, u, v);
    double h9[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) h9[3 * r + c] = init_h(r, c);  // row-major, as HomographyBlocks::create reads it
    std::vector<double> cov(64);
    cba_options o = make_options(options);
    cba_summary sum{};
    check(cba_optimize_homography(static_cast<int32_t>(data.size()), X.data(), Y.data(), u.data(), v.data(), h9, &o, &sum, cov.data()));
    calib::OptimizeHomographyResult res;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) res.homography(r, c) = h9[3 * r + c];
    fill_core(sum, o, cov, 8, res.core);
    return res;
}

}  // namespace calibba_adapter
