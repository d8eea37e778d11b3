// TEST-ONLY STAND-IN for the calib:: types include/calibba_adapter.hpp touches (names, members and defaults as in the
// reference's public headers: include/calib/estimation/optim/{optimize,intrinsics,extrinsics,bundle,handeye,planarpose,
// homography}.h, include/calib/estimation/linear/{planarpose,extrinsics}.h, include/calib/models/{camera_matrix,pinhole,
// distortion,scheimpflug,cameramodel}.h).  Declarations only — no projection code, no serialisation, no Ceres: it lets the
// adapter be type-checked and driven end to end in an image without Eigen / nlohmann / Boost.PFR.  It pins nothing.
#pragma once
#include <array>
#include <cstddef>
#include <optional>
#include <string>
#include <vector>

#include "eigen_min.h"

namespace calib {

template <typename Scalar>
struct CameraMatrixT final {
    Scalar fx = Scalar(0), fy = Scalar(0), cx = Scalar(0), cy = Scalar(0), skew = Scalar(0);
};
using CameraMatrix = CameraMatrixT<double>;

struct CalibrationBounds final {
    double fx_min = 0.0, fx_max = 2000.0, fy_min = 0.0, fy_max = 2000.0;
    double cx_min = 0.0, cx_max = 1280.0, cy_min = 0.0, cy_max = 720.0;
    double skew_min = -0.01, skew_max = 0.01;
};

template <typename Cam>
concept camera_model = requires { typename Cam::Scalar; };

template <typename CamT> struct CameraTraits;

template <typename T>
struct BrownConrady final {
    using Scalar = T;
    Eigen::Matrix<T, Eigen::Dynamic, 1> coeffs;
};
using BrownConradyd = BrownConrady<double>;

template <typename DistortionT>
class PinholeCamera final {
  public:
    using Scalar = typename DistortionT::Scalar;
    CameraMatrixT<Scalar> kmtx;
    DistortionT distortion;
    PinholeCamera() = default;
    PinholeCamera(const CameraMatrixT<Scalar>& matrix, const DistortionT& d) : kmtx(matrix), distortion(d) {}
};

template <typename DistortionT>
struct CameraTraits<PinholeCamera<DistortionT>> {
    static constexpr size_t param_count = 10;
    template <typename T>
    static auto from_array(const T* intr) -> PinholeCamera<BrownConrady<T>> {
        PinholeCamera<BrownConrady<T>> cam;
        cam.kmtx = CameraMatrixT<T>{intr[0], intr[1], intr[2], intr[3], intr[4]};
        cam.distortion.coeffs = Eigen::Matrix<T, Eigen::Dynamic, 1>(5);
        for (int i = 0; i < 5; ++i) cam.distortion.coeffs[i] = intr[5 + i];
        return cam;
    }
    static void to_array(const PinholeCamera<DistortionT>& cam, std::array<double, param_count>& arr) {
        arr = {cam.kmtx.fx, cam.kmtx.fy, cam.kmtx.cx, cam.kmtx.cy, cam.kmtx.skew, 0, 0, 0, 0, 0};
        for (int i = 0; i < 5 && i < cam.distortion.coeffs.size(); ++i) arr[5 + i] = cam.distortion.coeffs[i];
    }
};

template <camera_model CameraT>
struct ScheimpflugCamera final {
    using Scalar = typename CameraT::Scalar;
    CameraT camera;
    Scalar tau_x{0}, tau_y{0};
    ScheimpflugCamera() = default;
    ScheimpflugCamera(const CameraT& cam, Scalar tx, Scalar ty) : camera(cam), tau_x(tx), tau_y(ty) {}
};

template <camera_model CameraT>
struct CameraTraits<ScheimpflugCamera<CameraT>> {
    static constexpr size_t param_count = CameraTraits<CameraT>::param_count + 2;
    template <typename T>
    static auto from_array(const T* intr) -> ScheimpflugCamera<decltype(CameraTraits<CameraT>::from_array(intr))> {
        auto cam = CameraTraits<CameraT>::from_array(intr);
        return ScheimpflugCamera<decltype(cam)>(cam, intr[param_count - 2], intr[param_count - 1]);
    }
    static void to_array(const ScheimpflugCamera<CameraT>& cam, std::array<double, param_count>& arr) {
        std::array<double, CameraTraits<CameraT>::param_count> inner{};
        CameraTraits<CameraT>::to_array(cam.camera, inner);
        for (size_t i = 0; i < inner.size(); ++i) arr[i] = inner[i];
        arr[param_count - 2] = cam.tau_x;
        arr[param_count - 1] = cam.tau_y;
    }
};

struct PlanarObservation {
    Eigen::Vector2d object_xy;
    Eigen::Vector2d image_uv;
};
using PlanarView = std::vector<PlanarObservation>;
using MulticamPlanarView = std::vector<PlanarView>;

// the host seed optimize_intrinsics_semidlt calls (src/estimation/linear/planarpose_linear.cpp:54-76); the test
// program defines it on top of cba_estimate_planar_pose_batch
auto estimate_planar_pose(PlanarView view, const CameraMatrix& intrinsics) -> Eigen::Isometry3d;

enum class OptimizerType { DEFAULT, SPARSE_SCHUR, DENSE_SCHUR, DENSE_QR };

struct OptimOptions final {
    OptimizerType optimizer = OptimizerType::DEFAULT;
    double huber_delta = 1.0;
    double epsilon = 1e-9;
    int max_iterations = 1000;
    bool compute_covariance = true;
    bool verbose = false;
};

struct OptimResult final {
    bool success = false;
    Eigen::MatrixXd covariance;
    std::string report = "Empty";
    double final_cost = 0.0;
};

struct IntrinsicsOptimOptions final {
    OptimOptions core;
    int num_radial = 2;
    bool optimize_skew = false;
    std::optional<CalibrationBounds> bounds = std::nullopt;
    std::vector<int> fixed_distortion_indices;
    std::vector<double> fixed_distortion_values;
};

template <camera_model CameraT>
struct IntrinsicsOptimizationResult final {
    OptimResult core;
    CameraT camera;
    std::vector<Eigen::Isometry3d> c_se3_t;
    std::vector<double> view_errors;
};

template <camera_model CameraT>
struct ExtrinsicOptimizationResult final {
    OptimResult core;
    std::vector<CameraT> cameras;
    std::vector<Eigen::Isometry3d> c_se3_r;
    std::vector<Eigen::Isometry3d> r_se3_t;
};

struct ExtrinsicOptions final {
    OptimOptions core;
    bool optimize_intrinsics = true;
    bool optimize_skew = false;
    bool optimize_extrinsics = true;
};

struct BundleObservation final {
    PlanarView view;
    Eigen::Isometry3d b_se3_g;
    size_t camera_index = 0;
};

struct BundleOptions final {
    OptimOptions core;
    bool optimize_intrinsics = false;
    bool optimize_skew = false;
    bool optimize_target_pose = true;
    bool optimize_hand_eye = true;
};

template <camera_model CameraT>
struct BundleResult final {
    OptimResult core;
    std::vector<CameraT> cameras;
    std::vector<Eigen::Isometry3d> g_se3_c;
    Eigen::Isometry3d b_se3_t;
};

struct HandeyeResult final {
    OptimResult core;
    Eigen::Isometry3d g_se3_c;
};

struct PlanarPoseOptions final {
    OptimOptions core;
    int num_radial = 2;
};

struct PlanarPoseResult final {
    OptimResult core;
    Eigen::Isometry3d pose;
    Eigen::VectorXd distortion;
    double reprojection_error = 0.0;
};

struct OptimizeHomographyResult final {
    OptimResult core;
    Eigen::Matrix3d homography;
};

}  // namespace calib
