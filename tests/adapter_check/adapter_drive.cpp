// Test program for include/calibba_adapter.hpp: instantiates EVERY adapter function for both camera types the reference
// instantiates (PinholeCamera<BrownConradyd>, ScheimpflugCamera<PinholeCamera<BrownConradyd>>; intrinsics.cpp:122-132,
// extrinsics.cpp:198-207, bundle.cpp:172-179) against the test-only stand-in declarations under stand_ins/, and — when run
// on a GPU box — drives each of them on a synthetic noise-free scene and checks ground-truth recovery and the exception
// mapping of SURVEY.md §8(b).  CPU tier: `g++ -std=c++20 -fsyntax-only` of this file (tests/test_adapter_compiles.py);
// GPU tier: built by tests/adapter_check/Makefile against libcalibba.so and executed.
// Observations are synthesised with the oracle's projection (oracle/models.hpp) — test infrastructure on both sides.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "calibba_adapter.hpp"
#include "models.hpp"  // oracle/

namespace ad = calibba_adapter;
using Eigen::Isometry3d;
using Pinhole = calib::PinholeCamera<calib::BrownConradyd>;
using Scheimpflug = calib::ScheimpflugCamera<Pinhole>;

// explicit instantiation of every templated entry point for both camera types: the type check proper
template auto ad::optimize_intrinsics<Pinhole>(const std::vector<calib::PlanarView>&, const Pinhole&, std::vector<Isometry3d>,
                                               const calib::IntrinsicsOptimOptions&) -> calib::IntrinsicsOptimizationResult<Pinhole>;
template auto ad::optimize_intrinsics<Scheimpflug>(const std::vector<calib::PlanarView>&, const Scheimpflug&, std::vector<Isometry3d>,
                                                   const calib::IntrinsicsOptimOptions&) -> calib::IntrinsicsOptimizationResult<Scheimpflug>;
template auto ad::optimize_extrinsics<Pinhole>(const std::vector<calib::MulticamPlanarView>&, const std::vector<Pinhole>&,
                                               const std::vector<Isometry3d>&, const std::vector<Isometry3d>&,
                                               const calib::ExtrinsicOptions&) -> calib::ExtrinsicOptimizationResult<Pinhole>;
template auto ad::optimize_extrinsics<Scheimpflug>(const std::vector<calib::MulticamPlanarView>&, const std::vector<Scheimpflug>&,
                                                   const std::vector<Isometry3d>&, const std::vector<Isometry3d>&,
                                                   const calib::ExtrinsicOptions&) -> calib::ExtrinsicOptimizationResult<Scheimpflug>;
template auto ad::optimize_bundle<Pinhole>(const std::vector<calib::BundleObservation>&, const std::vector<Pinhole>&,
                                           const std::vector<Isometry3d>&, const Isometry3d&, const calib::BundleOptions&)
    -> calib::BundleResult<Pinhole>;
template auto ad::optimize_bundle<Scheimpflug>(const std::vector<calib::BundleObservation>&, const std::vector<Scheimpflug>&,
                                               const std::vector<Isometry3d>&, const Isometry3d&, const calib::BundleOptions&)
    -> calib::BundleResult<Scheimpflug>;

// ---- tiny SE(3) helpers on the stand-in Isometry3d ---------------------------------------------------------------
static Isometry3d make_pose(double ax, double ay, double az, double angle, double tx, double ty, double tz) {
    const double n = std::sqrt(ax * ax + ay * ay + az * az);
    const double k[3] = {ax / n, ay / n, az / n}, c = std::cos(angle), s = std::sin(angle);
    Isometry3d T = Isometry3d::Identity();
    auto& m = T.matrix();
    const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m(i, j) = (i == j ? c : 0.0) + s * K[3 * i + j] + (1 - c) * k[i] * k[j];
    m(0, 3) = tx; m(1, 3) = ty; m(2, 3) = tz;
    return T;
}
static Isometry3d mul(const Isometry3d& A, const Isometry3d& B) {
    Isometry3d C = Isometry3d::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = j == 3 ? A.matrix()(i, 3) : 0.0;
            for (int k = 0; k < 3; ++k) s += A.matrix()(i, k) * B.matrix()(k, j);
            C.matrix()(i, j) = s;
        }
    return C;
}
static Isometry3d inv(const Isometry3d& A) {
    Isometry3d C = Isometry3d::Identity();
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int k = 0; k < 3; ++k) { C.matrix()(i, k) = A.matrix()(k, i); s -= A.matrix()(k, i) * A.matrix()(k, 3); }
        C.matrix()(i, 3) = s;
    }
    return C;
}
static double pose_gap(const Isometry3d& A, const Isometry3d& B) {
    double d = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) d = std::max(d, std::abs(A.matrix()(i, j) - B.matrix()(i, j)));
    return d;
}

static int failures = 0;
#define EXPECT(cond, ...)                                                   \
    do {                                                                    \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } \
    } while (0)

static const double kGT[12] = {1000, 1005, 640, 360, 0, -0.12, 0.02, 0.0005, -0.0007, 0.001, 0.02, -0.015};

static calib::PlanarView synth_view(int model, const double* intr, const Isometry3d& c_T_t) {
    calib::PlanarView view;
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 11; ++c) {
            const double X = (c - 5) * 0.03, Y = (r - 3.5) * 0.03;
            const auto& m = c_T_t.matrix();
            const double P[3] = {m(0, 0) * X + m(0, 1) * Y + m(0, 3), m(1, 0) * X + m(1, 1) * Y + m(1, 3), m(2, 0) * X + m(2, 1) * Y + m(2, 3)};
            double uv[2];
            orc::project(model, intr, P, uv);
            view.push_back({Eigen::Vector2d(X, Y), Eigen::Vector2d(uv[0], uv[1])});
        }
    return view;
}

template <class CameraT> CameraT make_camera(const double* p) { return calib::CameraTraits<CameraT>::template from_array<double>(p); }
template <class CameraT> double camera_gap(const CameraT& cam, const double* gt) {
    std::array<double, calib::CameraTraits<CameraT>::param_count> a{};
    calib::CameraTraits<CameraT>::to_array(cam, a);
    double d = 0;
    for (size_t i = 0; i < a.size(); ++i) d = std::max(d, std::abs(a[i] - gt[i]) / std::max(1.0, std::abs(gt[i])));
    return d;
}

static std::vector<Isometry3d> view_poses(int n, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<Isometry3d> out;
    for (int i = 0; i < n; ++i)
        out.push_back(make_pose(U(rng), U(rng), 0.3 * U(rng), 0.15 + 0.35 * std::abs(U(rng)), 0.05 * U(rng), 0.05 * U(rng), 0.9 + 0.2 * U(rng)));
    return out;
}

template <class CameraT>
static void drive_intrinsics(const char* name) {
    const int model = ad::ModelOf<CameraT>::value;
    const auto gt_poses = view_poses(12, 7);
    std::vector<calib::PlanarView> views;
    for (const auto& T : gt_poses) views.push_back(synth_view(model, kGT, T));
    double init[12] = {970, 1035, 645, 356, 0, 0, 0, 0, 0, 0, 0.0, 0.0};
    std::vector<Isometry3d> init_poses;
    for (const auto& T : gt_poses) init_poses.push_back(mul(make_pose(1, 2, 3, 0.01, 0.002, -0.001, 0.004), T));
    calib::IntrinsicsOptimOptions opts;
    opts.core.epsilon = 1e-12;
    auto res = ad::optimize_intrinsics(views, make_camera<CameraT>(init), init_poses, opts);
    EXPECT(res.core.success, "%s intrinsics did not converge: %s", name, res.core.report.c_str());
    EXPECT(camera_gap(res.camera, kGT) < 1e-6, "%s intrinsics gap %.3e", name, camera_gap(res.camera, kGT));
    double pg = 0;
    for (size_t i = 0; i < gt_poses.size(); ++i) pg = std::max(pg, pose_gap(res.c_se3_t[i], gt_poses[i]));
    EXPECT(pg < 1e-6, "%s view pose gap %.3e", name, pg);
    const Eigen::Index dim = static_cast<Eigen::Index>(calib::CameraTraits<CameraT>::param_count + 7 * views.size());
    EXPECT(res.core.covariance.rows() == dim && res.core.covariance.cols() == dim, "%s covariance is %ld x %ld", name,
           static_cast<long>(res.core.covariance.rows()), static_cast<long>(res.core.covariance.cols()));
    EXPECT(res.view_errors.empty(), "view_errors must stay empty (intrinsics.cpp:98-120)");
    bool threw = false;
    try {
        views.resize(3); init_poses.resize(3);
        (void)ad::optimize_intrinsics(views, make_camera<CameraT>(init), init_poses, opts);
    } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "<4 views must throw std::invalid_argument (intrinsics.cpp:92-96)");
    std::printf("  optimize_intrinsics<%s> ok: %s\n", name, res.core.report.c_str());
}

template <class CameraT>
static void drive_extrinsics(const char* name) {
    const int model = ad::ModelOf<CameraT>::value;
    const std::vector<Isometry3d> c_T_r = {Isometry3d::Identity(), make_pose(0, 1, 0, -0.2, -0.25, 0.01, 0.03)};
    const auto r_T_t = view_poses(10, 5);
    std::vector<calib::MulticamPlanarView> views;
    for (const auto& T : r_T_t) {
        calib::MulticamPlanarView mv;
        for (const auto& C : c_T_r) mv.push_back(synth_view(model, kGT, mul(C, T)));
        views.push_back(mv);
    }
    views[3][1].clear();  // a camera that did not see the target in one view (extrinsics.cpp:94-96)
    double init[12] = {990, 1015, 642, 358, 0, -0.1, 0, 0, 0, 0, 0.015, -0.01};
    std::vector<CameraT> cams(2, make_camera<CameraT>(init));
    std::vector<Isometry3d> ic = {c_T_r[0], mul(make_pose(1, 0, 1, 0.01, 0.003, 0.001, -0.002), c_T_r[1])}, it;
    for (const auto& T : r_T_t) it.push_back(mul(make_pose(0, 1, 1, 0.008, -0.002, 0.001, 0.003), T));
    it[0] = r_T_t[0];  // the first target pose is held constant when intrinsics are optimised (extrinsics.cpp:123-126)
    calib::ExtrinsicOptions opts;
    opts.core.epsilon = 1e-12;
    auto res = ad::optimize_extrinsics(views, cams, ic, it, opts);
    EXPECT(res.core.success, "%s extrinsics did not converge: %s", name, res.core.report.c_str());
    for (int c = 0; c < 2; ++c) {
        EXPECT(camera_gap(res.cameras[c], kGT) < 1e-6, "%s extrinsics camera %d gap %.3e", name, c, camera_gap(res.cameras[c], kGT));
        EXPECT(pose_gap(res.c_se3_r[c], c_T_r[c]) < 1e-6, "%s c_se3_r[%d] gap %.3e", name, c, pose_gap(res.c_se3_r[c], c_T_r[c]));
    }
    bool threw = false;
    try { it.pop_back(); (void)ad::optimize_extrinsics(views, cams, ic, it, opts); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "pose-vector size mismatch must throw std::invalid_argument (extrinsics.cpp:169-171)");
    std::printf("  optimize_extrinsics<%s> ok: %s\n", name, res.core.report.c_str());
}

template <class CameraT>
static void drive_bundle(const char* name) {
    const int model = ad::ModelOf<CameraT>::value;
    const Isometry3d g_T_c = make_pose(1, -1, 2, 0.1, 0.03, -0.02, 0.05), b_T_t = make_pose(0, 0, 1, 0.05, 0.0, 0.0, 1.0);
    std::vector<calib::BundleObservation> obs;
    std::mt19937 rng(137);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int i = 0; i < 10; ++i) {
        calib::BundleObservation ob;
        ob.b_se3_g = make_pose(U(rng), U(rng), U(rng), 0.1 + 0.25 * std::abs(U(rng)), 0.05 * U(rng), 0.05 * U(rng), 0.05 * U(rng));
        ob.camera_index = 0;
        ob.view = synth_view(model, kGT, mul(mul(inv(g_T_c), inv(ob.b_se3_g)), b_T_t));  // bundleresidual.h:15-27
        obs.push_back(ob);
    }
    std::vector<CameraT> cams = {make_camera<CameraT>(kGT)};
    calib::BundleOptions opts;
    opts.core.epsilon = 1e-12;
    auto res = ad::optimize_bundle(obs, cams, {mul(make_pose(1, 1, 0, 0.02, 0.004, -0.003, 0.002), g_T_c)},
                                   mul(make_pose(0, 1, 0, 0.01, 0.002, 0.002, -0.003), b_T_t), opts);
    EXPECT(res.core.success, "%s bundle did not converge: %s", name, res.core.report.c_str());
    EXPECT(pose_gap(res.g_se3_c[0], g_T_c) < 1e-6, "%s g_se3_c gap %.3e", name, pose_gap(res.g_se3_c[0], g_T_c));
    EXPECT(pose_gap(res.b_se3_t, b_T_t) < 1e-6, "%s b_se3_t gap %.3e", name, pose_gap(res.b_se3_t, b_T_t));
    bool threw = false;
    try { (void)ad::optimize_bundle({}, cams, {g_T_c}, b_T_t, opts); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "no observations must throw std::invalid_argument (bundle.cpp:142-144)");
    std::printf("  optimize_bundle<%s> ok: %s\n", name, res.core.report.c_str());
}

static void drive_handeye() {
    const Isometry3d X = make_pose(1, 2, -1, 0.3, 0.04, -0.03, 0.08), b_T_t = make_pose(0, 1, 0, 0.2, 0.3, 0.1, 1.2);
    std::vector<Isometry3d> bg, ct;
    std::mt19937 rng(2024);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int i = 0; i < 14; ++i) {
        bg.push_back(make_pose(U(rng), U(rng), U(rng), 0.2 + 0.5 * std::abs(U(rng)), 0.2 * U(rng), 0.2 * U(rng), 0.2 * U(rng)));
        ct.push_back(mul(mul(inv(X), inv(bg.back())), b_T_t));
    }
    calib::OptimOptions o;
    o.epsilon = 1e-12;
    auto r1 = ad::optimize_handeye(bg, ct, mul(make_pose(1, 0, 0, 0.03, 0.005, 0.002, -0.004), X), o);
    EXPECT(r1.core.success && pose_gap(r1.g_se3_c, X) < 1e-8, "optimize_handeye gap %.3e (%s)", pose_gap(r1.g_se3_c, X), r1.core.report.c_str());
    auto r2 = ad::estimate_and_optimize_handeye(bg, ct, 1.0, o);
    EXPECT(r2.core.success && pose_gap(r2.g_se3_c, X) < 1e-8, "estimate_and_optimize_handeye gap %.3e", pose_gap(r2.g_se3_c, X));
    EXPECT(r2.core.covariance.rows() == 7, "hand-eye covariance must be 7 x 7");
    bool threw = false;
    try { ct.pop_back(); (void)ad::optimize_handeye(bg, ct, X, o); } catch (const std::runtime_error&) { threw = true; }
    EXPECT(threw, "inconsistent hand-eye sizes must throw std::runtime_error (handeyedlt.cpp:56-58)");
    std::printf("  optimize_handeye / estimate_and_optimize_handeye ok: %s\n", r2.core.report.c_str());
}

// the host seed the adapter's optimize_intrinsics_semidlt calls, defined here on the library's batched seed
auto calib::estimate_planar_pose(PlanarView view, const CameraMatrix& k) -> Eigen::Isometry3d {
    std::vector<double> X, Y, u, v;
    ad::flatten_view(view, X, Y, u, v);
    const int64_t off[2] = {0, static_cast<int64_t>(view.size())};
    const double K[5] = {k.fx, k.fy, k.cx, k.cy, k.skew};
    double p7[7];
    ad::check(cba_estimate_planar_pose_batch(1, off, X.data(), Y.data(), u.data(), v.data(), K, p7));
    return ad::pose_out(p7);
}

static void drive_small_solvers() {
    const double intr[10] = {1000, 1005, 640, 360, 0, -0.12, 0.02, 0, -0.0007, 0.001};  // num_radial = 2
    const calib::CameraMatrix K{1000, 1005, 640, 360, 0};
    const auto poses = view_poses(8, 17);
    std::vector<calib::PlanarView> views;
    for (const auto& T : poses) views.push_back(synth_view(orc::PINHOLE_BC, intr, T));
    calib::PlanarPoseOptions po;
    po.core.epsilon = 1e-12;
    auto pr = ad::optimize_planar_pose(views[0], K, mul(make_pose(1, 1, 1, 0.02, 0.003, -0.002, 0.004), poses[0]), po);
    EXPECT(pr.core.success && pose_gap(pr.pose, poses[0]) < 1e-6, "planar pose gap %.3e (%s)", pose_gap(pr.pose, poses[0]), pr.core.report.c_str());
    EXPECT(pr.distortion.size() == 4 && std::abs(pr.distortion[0] + 0.12) < 1e-6, "planar-pose distortion k1 = %.9f", pr.distortion.size() ? pr.distortion[0] : 0.0);
    EXPECT(pr.reprojection_error < 1e-6, "planar-pose rms %.3e", pr.reprojection_error);

    // homography of an undistorted view: H ~ K [r1 r2 t]
    const double pin[10] = {1000, 1005, 640, 360, 0, 0, 0, 0, 0, 0};
    const auto hv = synth_view(orc::PINHOLE_BC, pin, poses[1]);
    Eigen::Matrix3d H0;
    const auto& m = poses[1].matrix();
    for (int c = 0; c < 3; ++c) {
        const int src = c == 2 ? 3 : c;
        H0(0, c) = pin[0] * m(0, src) + pin[2] * m(2, src);
        H0(1, c) = pin[1] * m(1, src) + pin[3] * m(2, src);
        H0(2, c) = m(2, src);
    }
    Eigen::Matrix3d Hinit = H0;
    for (int i = 0; i < 9; ++i) Hinit(i) = H0(i) / H0(2, 2) * (1.0 + 1e-3 * ((i * 7) % 5 - 2));
    auto hr = ad::optimize_homography(hv, Hinit);
    double hg = 0;
    for (int i = 0; i < 9; ++i) hg = std::max(hg, std::abs(hr.homography(i) / hr.homography(2, 2) - H0(i) / H0(2, 2)) / std::max(1.0, std::abs(H0(i) / H0(2, 2))));
    EXPECT(hr.core.success && hg < 1e-8, "homography gap %.3e (%s)", hg, hr.core.report.c_str());
    bool threw = false;
    try { calib::PlanarView three(hv.begin(), hv.begin() + 3); (void)ad::optimize_homography(three, Hinit); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw, "<4 correspondences must throw std::invalid_argument (homography.cpp:146-148)");

    calib::IntrinsicsOptimOptions so;
    so.core.epsilon = 1e-12;
    auto sr = ad::optimize_intrinsics_semidlt(views, calib::CameraMatrix{980, 1020, 635, 365, 0}, so);
    EXPECT(sr.core.success, "semi-DLT did not converge: %s", sr.core.report.c_str());
    EXPECT(std::abs(sr.camera.kmtx.fx - 1000) < 1e-4 && std::abs(sr.camera.kmtx.cy - 360) < 1e-4, "semi-DLT K: fx %.6f cy %.6f", sr.camera.kmtx.fx, sr.camera.kmtx.cy);
    EXPECT(sr.camera.distortion.coeffs.size() == 4 && std::abs(sr.camera.distortion.coeffs[0] + 0.12) < 1e-6, "semi-DLT k1");
    EXPECT(sr.view_errors.size() == views.size(), "semi-DLT view_errors");
    std::printf("  optimize_planar_pose / optimize_homography / optimize_intrinsics_semidlt ok\n");
}

int main() {
    if (cba_device_count() <= 0) {
        std::printf("adapter_drive: no HIP device; libcalibba has no CPU fallback\n");
        return 77;
    }
    drive_intrinsics<Pinhole>("PinholeCamera<BrownConradyd>");
    drive_intrinsics<Scheimpflug>("ScheimpflugCamera<PinholeCamera<BrownConradyd>>");
    drive_extrinsics<Pinhole>("PinholeCamera<BrownConradyd>");
    drive_extrinsics<Scheimpflug>("ScheimpflugCamera<PinholeCamera<BrownConradyd>>");
    drive_bundle<Pinhole>("PinholeCamera<BrownConradyd>");
    drive_bundle<Scheimpflug>("ScheimpflugCamera<PinholeCamera<BrownConradyd>>");
    drive_handeye();
    drive_small_solvers();
    std::printf(failures ? "adapter_drive: %d FAILURE(S)\n" : "adapter_drive: all ok\n", failures);
    return failures ? 1 : 0;
}
