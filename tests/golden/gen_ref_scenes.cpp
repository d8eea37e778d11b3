// tests/golden/gen_ref_scenes.cpp — generator of tests/golden/kat_scenes_stdrng.json (TEST INFRASTRUCTURE).
//
// The random scenes of the reference's hot-path tests, drawn from the SAME random stream the reference binary draws
// them from: std::mt19937 + a fresh std::uniform_real_distribution<double> per draw (tests/unit/utils.h:163-181),
// libstdc++, compiled with g++ like the reference's CI.  tests/golden/gen_golden.py redraws these scenes from numpy's
// MT19937 (same engine, different real-number transform), so its poses differ from the reference binary's; these do not.
//
//   scene                 reference test                                              seed
//   intrinsics_noskew     tests/unit/intrinsics_optimize_test.cpp:8-61                 RNG(7)
//   intrinsics_skew       tests/unit/intrinsics_optimize_test.cpp:63-113               RNG(5)
//   bundle_noskew / skew  tests/unit/bundle_test.cpp:9-81, 83-154                      RNG(7)
//   bundle_distortion     tests/unit/bundle_test.cpp:156-210                           RNG(137)  (NOT substituted)
//   axxb_refine           tests/unit/handeye_test.cpp:101-152                          RNG(2024)
//
// What is and is not bit-exact: the engine and the uniform transform are libstdc++'s own, so every draw equals the
// reference's.  `Vector3d dt(rng.uni(..), rng.uni(..), rng.uni(..))` (utils.h:214-215) has unspecified argument evaluation
// order; the same expression shape is used here (a three-argument constructor call) and g++ makes the same choice for both.
// Eigen's arithmetic (AngleAxis::toRotationMatrix, Isometry products and inverse) is restated below in its documented
// operation order (third-party, not in /root/reference); contraction differences could move a pixel by an ulp.
// The projection is the oracle's restatement of pinhole.h / distortion.h (oracle/models.hpp).
//
// Build + run:  g++ -O0 -std=c++20 -ffp-contract=off -I../../oracle gen_ref_scenes.cpp -o /tmp/gen_ref_scenes && /tmp/gen_ref_scenes > kat_scenes_stdrng.json
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <numbers>
#include <random>
#include <string>
#include <vector>

#include "models.hpp"  // oracle/: project()

struct V3 {
    double x, y, z;
    V3(const double& a, const double& b, const double& c) : x(a), y(b), z(c) {}
};
struct Iso {  // R row-major, t
    double R[9], t[3];
};
static Iso identity() { return Iso{{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
static Iso mul(const Iso& A, const Iso& B) {  // Eigen Isometry product: (A.R B.R, A.R B.t + A.t)
    Iso C;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A.R[3 * i] * B.R[j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
        C.t[i] = A.R[3 * i] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2] + A.t[i];
    }
    return C;
}
static Iso inv(const Iso& A) {  // Transform<Isometry>::inverse(): (R^T, -R^T t)
    Iso C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A.R[3 * j + i];
    for (int i = 0; i < 3; ++i) C.t[i] = -(C.R[3 * i] * A.t[0] + C.R[3 * i + 1] * A.t[1] + C.R[3 * i + 2] * A.t[2]);
    return C;
}
// axis_angle_to_R (utils.h:53-56): identity below 1e-16, else Eigen::AngleAxisd(angle, axis.normalized()).toRotationMatrix()
static void axis_angle_to_R(const V3& axis, double angle, double* R) {
    if (angle < 1e-16) { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; for (int i = 0; i < 9; ++i) R[i] = I[i]; return; }
    const double n = std::sqrt(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
    const double a[3] = {axis.x / n, axis.y / n, axis.z / n};
    // Eigen/src/Geometry/AngleAxis.h toRotationMatrix (third-party, restated)
    const double s = std::sin(angle), c = std::cos(angle);
    const double sa[3] = {s * a[0], s * a[1], s * a[2]}, ca[3] = {(1 - c) * a[0], (1 - c) * a[1], (1 - c) * a[2]};
    double tmp;
    tmp = ca[0] * a[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * a[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * a[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * a[0] + c; R[4] = ca[1] * a[1] + c; R[8] = ca[2] * a[2] + c;
}
static Iso make_pose(const V3& t, const V3& axis, double angle) {  // utils.h:58-64
    Iso T = identity();
    axis_angle_to_R(axis, angle, T.R);
    T.t[0] = t.x; T.t[1] = t.y; T.t[2] = t.z;
    return T;
}
static double deg2rad(double d) { return d * std::numbers::pi / 180.0; }

struct RNG {  // utils.h:163-181: a fresh distribution object per draw
    std::mt19937 gen;
    explicit RNG(uint32_t seed) : gen(seed) {}
    double uni(double a, double b) {
        std::uniform_real_distribution<double> d(a, b);
        return d(gen);
    }
    V3 rand_unit_axis() {
        double z = uni(-1.0, 1.0);
        double t = uni(0.0, 2.0 * std::numbers::pi);
        double r = std::sqrt(1.0 - z * z);
        return V3(r * std::cos(t), r * std::sin(t), z);
    }
};

// SimulatedHandEye (utils.h:183-251)
struct Sim {
    Iso g_T_c, b_T_t;
    std::vector<double> cam;  // 10
    std::vector<Iso> b_T_g, c_T_t;
    std::vector<std::pair<double, double>> grid;
    std::vector<std::vector<double>> views;  // rows of X, Y, u, v
    void make_sequence(size_t n, RNG& rng) {
        Iso T = identity();
        for (size_t k = 0; k < n; ++k) {
            b_T_g.push_back(T);
            c_T_t.push_back(mul(mul(inv(g_T_c), inv(T)), b_T_t));
            if (k + 1 < n) {
                const double ang = deg2rad(rng.uni(5.0, 25.0));
                const V3 ax = rng.rand_unit_axis();
                const V3 dt(rng.uni(-0.10, 0.10), rng.uni(-0.10, 0.10), rng.uni(-0.10, 0.10));
                T = mul(T, make_pose(dt, ax, ang));
            }
        }
    }
    void make_target_grid(int rows, int cols, double spacing) {
        const double x0 = -0.5 * (cols - 1) * spacing, y0 = -0.5 * (rows - 1) * spacing;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) grid.emplace_back(x0 + c * spacing, y0 + r * spacing);
    }
    void render_pixels() {
        for (const Iso& T : c_T_t) {
            std::vector<double> v;
            for (const auto& p : grid) {
                const double Pc[3] = {T.R[0] * p.first + T.R[1] * p.second + T.R[2] * 0.0 + T.t[0],
                                      T.R[3] * p.first + T.R[4] * p.second + T.R[5] * 0.0 + T.t[1],
                                      T.R[6] * p.first + T.R[7] * p.second + T.R[8] * 0.0 + T.t[2]};
                if (Pc[2] <= 1e-6) continue;
                double uv[2];
                orc::project(orc::PINHOLE_BC, cam.data(), Pc, uv);
                v.insert(v.end(), {p.first, p.second, uv[0], uv[1]});
            }
            views.push_back(v);
        }
    }
};

// ---- JSON ----------------------------------------------------------------------------------------------------------
static std::string num(double v) { char b[40]; std::snprintf(b, sizeof b, "%.17g", v); return b; }
static std::string arr(const std::vector<double>& v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + num(v[i]);
    return s + "]";
}
static std::string mat4(const Iso& T) {
    std::string s = "[";
    for (int i = 0; i < 3; ++i) s += "[" + num(T.R[3 * i]) + "," + num(T.R[3 * i + 1]) + "," + num(T.R[3 * i + 2]) + "," + num(T.t[i]) + "],";
    return s + "[0,0,0,1]]";
}
static std::string view_json(const std::vector<double>& v) {
    std::string s = "[";
    for (size_t i = 0; i + 3 < v.size(); i += 4) s += (i ? "," : "") + arr({v[i], v[i + 1], v[i + 2], v[i + 3]});
    return s + "]";
}
static std::vector<double> cam10(double fx, double fy, double cx, double cy, double skew, std::vector<double> dist = {0, 0, 0, 0, 0}) {
    std::vector<double> c = {fx, fy, cx, cy, skew};
    c.insert(c.end(), dist.begin(), dist.end());
    return c;
}

static std::string scene_intrinsics(uint32_t seed, double skew, double ffx, double ffy, double dcx, double dcy, bool opt_skew, double tol_skew,
                                    const char* ref) {
    RNG rng(seed);
    Sim sim{identity(), make_pose(V3(0.0, 0.0, 2.0), V3(0, 0, 1), 0.0), cam10(1000, 1005, 640, 360, skew)};
    sim.make_sequence(15, rng);
    sim.make_target_grid(8, 11, 0.02);
    sim.render_pixels();
    std::vector<double> c0 = sim.cam;
    c0[0] *= ffx; c0[1] *= ffy; c0[2] += dcx; c0[3] -= dcy; c0[4] = 0.0;
    std::string s = "{\"kind\":\"intrinsics\",\"views\":[";
    for (size_t i = 0; i < sim.views.size(); ++i) s += (i ? "," : "") + view_json(sim.views[i]);
    s += "],\"cam_gt\":" + arr(sim.cam) + ",\"cam_init\":" + arr(c0) + ",\"optimize_skew\":" + (opt_skew ? "true" : "false") +
         ",\"tol_K\":1e-06,\"tol_skew\":" + num(tol_skew) + ",\"max_final_cost\":1e-06,\"seed\":" + std::to_string(seed) + ",\"ref\":\"" + ref + "\"}";
    return s;
}

static std::string obs_json(const Sim& sim) {
    std::string s = "[";
    for (size_t i = 0; i < sim.views.size(); ++i)
        s += std::string(i ? "," : "") + "{\"view\":" + view_json(sim.views[i]) + ",\"b_T_g\":" + mat4(sim.b_T_g[i]) + ",\"cam\":0}";
    return s + "]";
}

static Iso perturbed(const Iso& X, const V3& dt, const V3& axis, double angle) {  // X0.translation() += dt; X0.linear() = R(axis, angle) * X0.linear()
    Iso X0 = X, D = identity();
    X0.t[0] += dt.x; X0.t[1] += dt.y; X0.t[2] += dt.z;
    axis_angle_to_R(axis, angle, D.R);
    Iso out = X0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out.R[3 * i + j] = D.R[3 * i] * X0.R[j] + D.R[3 * i + 1] * X0.R[3 + j] + D.R[3 * i + 2] * X0.R[6 + j];
    return out;
}

static std::string scene_bundle(double skew, bool opt_skew, double tol_skew) {  // bundle_test.cpp:9-81 / 83-154
    RNG rng(7);
    Sim sim{make_pose(V3(0.03, 0.00, 0.12), V3(0, 1, 0), deg2rad(8.0)), make_pose(V3(0.5, -0.1, 0.8), V3(1, 0, 0), deg2rad(14.0)),
            cam10(1000, 1005, 640, 360, skew)};
    sim.make_sequence(25, rng);
    sim.make_target_grid(8, 11, 0.02);
    sim.render_pixels();
    // cam0.skew = cam_gt.skew in the no-skew test (bundle_test.cpp:38); 0 in the skew test (bundle_test.cpp:112-113)
    const std::vector<double> c0 = cam10(1000 * 0.97, 1005 * 1.03, 640 + 5.0, 360 - 4.0, opt_skew ? 0.0 : skew);
    const Iso g0 = perturbed(sim.g_T_c, V3(-0.01, 0.006, -0.004), V3(0.3, 0.7, -0.2), deg2rad(2.0));
    return "{\"kind\":\"bundle\",\"obs\":" + obs_json(sim) + ",\"cams_gt\":[" + arr(sim.cam) + "],\"cams_init\":[" + arr(c0) + "],\"g_T_c_gt\":[" +
           mat4(sim.g_T_c) + "],\"g_T_c_init\":[" + mat4(g0) + "],\"b_T_t_gt\":" + mat4(sim.b_T_t) + ",\"b_T_t_init\":" + mat4(sim.b_T_t) +
           ",\"opts\":{\"optimize_intrinsics\":true,\"optimize_skew\":" + (opt_skew ? "true" : "false") +
           ",\"huber_delta\":-1.0},\"tol_rot_deg\":1e-06,\"tol_trans\":1e-06,\"tol_K\":1e-06,\"tol_skew\":" + num(tol_skew) +
           ",\"seed\":7,\"ref\":\"tests/unit/bundle_test.cpp:9-154\"}";
}

static std::string scene_bundle_distortion() {  // bundle_test.cpp:156-210, seed 137 as in the reference
    RNG rng(137);
    Sim sim{make_pose(V3(0.03, 0.00, 0.12), V3(0, 1, 0), deg2rad(8.0)), make_pose(V3(0.5, -0.1, 80), V3(1, 0, 0), deg2rad(14.0)),
            cam10(900, 905, 640, 360, 0.0, {-0.12, 0.02, 0.0005, -0.0007, 0.001})};
    sim.make_sequence(22, rng);
    sim.make_target_grid(7, 10, 0.022);
    sim.render_pixels();
    std::vector<double> c0 = sim.cam;
    for (int i = 5; i < 10; ++i) c0[i] = 0.0;
    const Iso X0 = perturbed(sim.g_T_c, V3(0.01, 0.006, -0.003), V3(0.1, 0.8, 0.1), deg2rad(2.0));
    double worst = 0;  // largest normalised radius over all rendered points (diagnostic: how far outside the polynomial's sane range)
    size_t culled = 0;
    for (size_t k = 0; k < sim.c_T_t.size(); ++k) {
        const Iso& T = sim.c_T_t[k];
        culled += sim.grid.size() - sim.views[k].size() / 4;
        for (const auto& p : sim.grid) {
            const double z = T.R[6] * p.first + T.R[7] * p.second + T.t[2];
            if (z <= 1e-6) continue;
            const double x = (T.R[0] * p.first + T.R[1] * p.second + T.t[0]) / z, y = (T.R[3] * p.first + T.R[4] * p.second + T.t[1]) / z;
            worst = std::fmax(worst, std::sqrt(x * x + y * y));
        }
    }
    return "{\"kind\":\"bundle\",\"obs\":" + obs_json(sim) + ",\"cams_gt\":[" + arr(sim.cam) + "],\"cams_init\":[" + arr(c0) + "],\"g_T_c_gt\":[" +
           mat4(sim.g_T_c) + "],\"g_T_c_init\":[" + mat4(X0) + "],\"b_T_t_gt\":" + mat4(sim.b_T_t) + ",\"b_T_t_init\":" + mat4(sim.b_T_t) +
           ",\"opts\":{\"optimize_intrinsics\":true,\"optimize_skew\":false,\"huber_delta\":1.0},\"tol_rot_deg\":0.1,\"tol_trans\":0.02,"
           "\"tol_dist\":1e-05,\"seed_used\":137,\"max_normalised_radius\":" + num(worst) + ",\"culled_points\":" + std::to_string(culled) +
           ",\"ref\":\"tests/unit/bundle_test.cpp:156-210\"}";
}

static std::string scene_axxb() {  // handeye_test.cpp:101-152
    RNG rng(2024);
    const Iso X = make_pose(V3(0.02, -0.01, 0.09), rng.rand_unit_axis(), deg2rad(10.0));
    const Iso bTt = make_pose(V3(0.25, 0.05, 0.55), rng.rand_unit_axis(), deg2rad(18.0));
    Sim sim{X, bTt, cam10(950, 960, 640, 360, 0.0)};
    sim.make_sequence(18, rng);
    Iso X0 = X;
    {
        const V3 ax = rng.rand_unit_axis();
        Iso D = identity();
        axis_angle_to_R(ax, deg2rad(2.0), D.R);
        Iso R0 = X0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R0.R[3 * i + j] = D.R[3 * i] * X0.R[j] + D.R[3 * i + 1] * X0.R[3 + j] + D.R[3 * i + 2] * X0.R[6 + j];
        X0 = R0;
        X0.t[0] += 0.01; X0.t[1] += -0.005; X0.t[2] += 0.004;
    }
    std::string s = "{\"kind\":\"handeye\",\"b_T_g\":[";
    for (size_t i = 0; i < sim.b_T_g.size(); ++i) s += (i ? "," : "") + mat4(sim.b_T_g[i]);
    s += "],\"c_T_t\":[";
    for (size_t i = 0; i < sim.c_T_t.size(); ++i) s += (i ? "," : "") + mat4(sim.c_T_t[i]);
    s += "],\"X_gt\":" + mat4(X) + ",\"X_init\":" + mat4(X0) +
         ",\"opts\":{\"max_iterations\":60,\"huber_delta\":1.0},\"tol_rot_deg\":0.05,\"tol_trans\":0.002,\"seed\":2024,"
         "\"ref\":\"tests/unit/handeye_test.cpp:101-152\"}";
    return s;
}

int main() {
    std::printf("{\"intrinsics_noskew\":%s,\n", scene_intrinsics(7, 0.0, 0.97, 1.03, 5.0, 4.0, false, 1e-9, "tests/unit/intrinsics_optimize_test.cpp:8-61").c_str());
    std::printf("\"intrinsics_skew\":%s,\n", scene_intrinsics(5, 0.001, 0.95, 1.05, 10.0, 6.0, true, 1e-8, "tests/unit/intrinsics_optimize_test.cpp:63-113").c_str());
    std::printf("\"bundle_noskew\":%s,\n", scene_bundle(0.0, false, 1e-9).c_str());
    std::printf("\"bundle_skew\":%s,\n", scene_bundle(0.001, true, 1e-6).c_str());
    std::printf("\"bundle_distortion\":%s,\n", scene_bundle_distortion().c_str());
    std::printf("\"axxb_refine\":%s}\n", scene_axxb().c_str());
    return 0;
}
