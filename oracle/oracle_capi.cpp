// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE ONLY.  Never linked into or loaded by the
// product (calibration_amd/, libcalibba.so).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg load liboracle.so.
//
// C entry points over the CPU restatement of the reference's Ceres path.  The problem
// description structs are the product's public ones (include/calibba.h) so a test can hand
// the identical buffers to both sides.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../include/calibba.h"
#include "lm.hpp"
#include "models.hpp"
#include "residuals.hpp"

using namespace orc;

static thread_local std::string g_err;

namespace {

struct Built {
    Problem prob;
    std::vector<int> intr_id, camq_id, camt_id, viewq_id, viewt_id;
    int tq_id = -1, tt_id = -1;
    std::vector<int> cov_order;
};

// Builds the ceres::Problem equivalent of
//   intrinsics.cpp:63-90 (build_problem), extrinsics.cpp:86-160, bundle.cpp:83-133.
void build(const cba_reproj_problem& d, const cba_options& o, Built& B) {
    const int P = intr_size(d.camera_model);
    Problem& p = B.prob;
    if (d.chain == CBA_CHAIN_BUNDLE) {
        if (d.n_cams <= 0) throw std::invalid_argument("No camera intrinsics provided");
        if (d.n_blocks <= 0) throw std::invalid_argument("No observations provided");
    }
    for (int c = 0; c < d.n_cams; ++c) B.intr_id.push_back(p.add_param(d.intr + static_cast<size_t>(c) * P, P));
    if (d.chain != CBA_CHAIN_INTRINSIC)
        for (int c = 0; c < d.n_cams; ++c) {
            B.camq_id.push_back(p.add_param(d.cam_pose + 7 * static_cast<size_t>(c), 4, BLK_QUAT));
            B.camt_id.push_back(p.add_param(d.cam_pose + 7 * static_cast<size_t>(c) + 4, 3));
        }
    if (d.chain != CBA_CHAIN_BUNDLE)
        for (int v = 0; v < d.n_views; ++v) {
            B.viewq_id.push_back(p.add_param(d.view_pose + 7 * static_cast<size_t>(v), 4, BLK_QUAT));
            B.viewt_id.push_back(p.add_param(d.view_pose + 7 * static_cast<size_t>(v) + 4, 3));
        }
    if (d.chain == CBA_CHAIN_BUNDLE) {
        B.tq_id = p.add_param(d.target_pose, 4, BLK_QUAT);
        B.tt_id = p.add_param(d.target_pose + 4, 3);
    }
    for (int b = 0; b < d.n_blocks; ++b) {
        ViewData vd;
        vd.n = static_cast<int>(d.blk_offset[b + 1] - d.blk_offset[b]);
        vd.X = d.X + d.blk_offset[b]; vd.Y = d.Y + d.blk_offset[b];
        vd.u = d.u + d.blk_offset[b]; vd.v = d.v + d.blk_offset[b];
        const int c = d.blk_cam ? d.blk_cam[b] : 0;
        if (c < 0 || c >= d.n_cams) throw std::invalid_argument("camera index out of range");
        auto rb = make_reproj_block(d.chain, d.camera_model, vd,
                                    d.chain == CBA_CHAIN_BUNDLE ? d.blk_b_T_g + 12 * static_cast<size_t>(b) : nullptr);
        if (d.chain == CBA_CHAIN_INTRINSIC) {
            const int v = d.blk_view ? d.blk_view[b] : b;
            if (v < 0 || v >= d.n_views) throw std::invalid_argument("view index out of range");
            rb->pb = {B.viewq_id[v], B.viewt_id[v], B.intr_id[c]};
        } else if (d.chain == CBA_CHAIN_EXTRINSIC) {
            const int v = d.blk_view[b];
            if (v < 0 || v >= d.n_views) throw std::invalid_argument("view index out of range");
            rb->pb = {B.camq_id[c], B.camt_id[c], B.viewq_id[v], B.viewt_id[v], B.intr_id[c]};
        } else {
            rb->pb = {B.tq_id, B.tt_id, B.camq_id[c], B.camt_id[c], B.intr_id[c]};
        }
        p.residuals.push_back(std::move(rb));
    }
    // constraints
    const int idx_fx = 0, idx_fy = 1, idx_skew = 4;  // pinhole.h:118-121
    auto intr_constraints = [&](int id) {
        p.set_lower(id, idx_fx, 0.0);
        p.set_lower(id, idx_fy, 0.0);
        if (!o.optimize_skew) p.params[id].subset_const = idx_skew;
    };
    if (d.chain == CBA_CHAIN_INTRINSIC) {  // intrinsics.cpp:77-87
        for (int id : B.intr_id) intr_constraints(id);
        B.cov_order = B.intr_id;           // intrinsics.cpp:34-50
        for (int id : B.viewq_id) B.cov_order.push_back(id);
        for (int id : B.viewt_id) B.cov_order.push_back(id);
    } else if (d.chain == CBA_CHAIN_EXTRINSIC) {  // extrinsics.cpp:110-150
        if (!o.optimize_intrinsics) {
            for (int id : B.intr_id) p.params[id].constant = true;
        } else if (d.n_views > 0 && d.first_view_global == 0) {
            p.params[B.viewq_id[0]].constant = true;
            p.params[B.viewt_id[0]].constant = true;
        }
        if (!o.optimize_extrinsics) {
            for (int id : B.camq_id) p.params[id].constant = true;
            for (int id : B.camt_id) p.params[id].constant = true;
        } else if (d.n_cams > 0) {
            p.params[B.camq_id[0]].constant = true;
            p.params[B.camt_id[0]].constant = true;
        }
        for (int id : B.intr_id) intr_constraints(id);
        B.cov_order = B.intr_id;  // extrinsics.cpp:50-67
        for (int id : B.camq_id) B.cov_order.push_back(id);
        for (int id : B.camt_id) B.cov_order.push_back(id);
        for (int id : B.viewq_id) B.cov_order.push_back(id);
        for (int id : B.viewt_id) B.cov_order.push_back(id);
    } else {  // bundle.cpp:98-131
        if (!o.optimize_target_pose) { p.params[B.tq_id].constant = true; p.params[B.tt_id].constant = true; }
        if (!o.optimize_extrinsics) {
            for (int id : B.camq_id) p.params[id].constant = true;
            for (int id : B.camt_id) p.params[id].constant = true;
        }
        if (!o.optimize_intrinsics) {
            for (int id : B.intr_id) p.params[id].constant = true;
        } else {
            for (int id : B.intr_id) intr_constraints(id);
        }
        B.cov_order = B.intr_id;  // bundle.cpp:48-68
        for (int id : B.camq_id) B.cov_order.push_back(id);
        for (int id : B.camt_id) B.cov_order.push_back(id);
        B.cov_order.push_back(B.tq_id);
        B.cov_order.push_back(B.tt_id);
    }
}

LMOptions to_lm(const cba_options& o, int threads) {
    LMOptions l;
    l.huber_delta = o.huber_delta; l.epsilon = o.epsilon; l.max_iterations = o.max_iterations;
    l.verbose = o.verbose; l.num_threads = threads;
    if (const char* env = std::getenv("ORC_LINE_SEARCH")) l.line_search = std::atoi(env);
    return l;
}

void fill_summary(const LMSummary& s, double secs, cba_summary* out) {
    out->success = s.termination == TERM_CONVERGENCE;
    out->termination = s.termination;
    out->iterations = s.iterations;
    out->successful_steps = s.successful_steps;
    out->initial_cost = s.initial_cost;
    out->final_cost = s.final_cost;
    out->solve_seconds = secs;
    std::snprintf(out->report, sizeof(out->report), "oracle(dense LM): %s iters=%d cost %.6e -> %.6e", s.message,
                  s.iterations, s.initial_cost, s.final_cost);
}

template <typename F>
int guarded(F&& f) {
    try {
        f();
        return CBA_OK;
    } catch (const std::invalid_argument& e) {
        g_err = e.what();
        return CBA_ERR_INVALID_ARGUMENT;
    } catch (const std::runtime_error& e) {
        g_err = e.what();
        return CBA_ERR_RUNTIME;
    } catch (const std::exception& e) {
        g_err = e.what();
        return CBA_ERR_INTERNAL;
    }
}

}  // namespace

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

// closed-form KAT hook: uv = project(model, intr, P)
void orc_project(int model, const double* intr, const double* P3, double* uv2) { project<double>(model, intr, P3, uv2); }

void orc_quat_to_rotmat(const double* q, double* R9) { quat_to_rotmat<double>(q, R9); }
void orc_rotmat_to_quat(const double* R9, double* q) { rotmat_to_quat<double>(R9, q); }
void orc_quat_plus(const double* q, const double* d3, double* out) { quat_plus(q, d3, out); }

// Residuals + Jacobians of every block at the given parameters, in Ceres layout:
//   r[2*n_obs]; J row-major [2*n_obs][P] in the tangent space and local column order of
//   cba_local_columns() ( [poseA d,t | poseB d,t | intr] ), poseA = view pose (INTRINSIC,
//   EXTRINSIC) or target pose (BUNDLE); poseB = camera pose.
// Jamb (optional) gets the raw ambient autodiff Jacobian [2*n_obs][7 or 14 + P] with columns
//   [qA(4) tA(3) | qB(4) tB(3) | intr].
int orc_reproj_eval(const cba_reproj_problem* d, double* r, double* J, double* Jamb) {
    return guarded([&] {
        const int P = intr_size(d->camera_model);
        const bool two = d->chain != CBA_CHAIN_INTRINSIC;
        const int PL = (two ? 12 : 6) + P;
        const int PA = (two ? 14 : 7) + P;
        for (int b = 0; b < d->n_blocks; ++b) {
            ViewData vd;
            vd.n = static_cast<int>(d->blk_offset[b + 1] - d->blk_offset[b]);
            vd.X = d->X + d->blk_offset[b]; vd.Y = d->Y + d->blk_offset[b];
            vd.u = d->u + d->blk_offset[b]; vd.v = d->v + d->blk_offset[b];
            const int c = d->blk_cam ? d->blk_cam[b] : 0;
            auto rb = make_reproj_block(d->chain, d->camera_model, vd,
                                        d->chain == CBA_CHAIN_BUNDLE ? d->blk_b_T_g + 12 * static_cast<size_t>(b) : nullptr);
            const double* intr = d->intr + static_cast<size_t>(c) * P;
            const double *pA, *pB = nullptr;
            const double* x[5];
            const int nres = 2 * vd.n;
            std::vector<double> jq0(static_cast<size_t>(nres) * 4), jt0(static_cast<size_t>(nres) * 3), jq1(static_cast<size_t>(nres) * 4),
                jt1(static_cast<size_t>(nres) * 3), ji(static_cast<size_t>(nres) * P);
            double* Jp[5];
            // map ceres parameter order -> (A, B) roles
            if (d->chain == CBA_CHAIN_INTRINSIC) {
                pA = d->view_pose + 7 * static_cast<size_t>(d->blk_view ? d->blk_view[b] : b);
                x[0] = pA; x[1] = pA + 4; x[2] = intr;
                Jp[0] = jq0.data(); Jp[1] = jt0.data(); Jp[2] = ji.data();
            } else if (d->chain == CBA_CHAIN_EXTRINSIC) {
                pB = d->cam_pose + 7 * static_cast<size_t>(c);
                pA = d->view_pose + 7 * static_cast<size_t>(d->blk_view[b]);
                x[0] = pB; x[1] = pB + 4; x[2] = pA; x[3] = pA + 4; x[4] = intr;
                Jp[0] = jq1.data(); Jp[1] = jt1.data(); Jp[2] = jq0.data(); Jp[3] = jt0.data(); Jp[4] = ji.data();
            } else {
                pA = d->target_pose;
                pB = d->cam_pose + 7 * static_cast<size_t>(c);
                x[0] = pA; x[1] = pA + 4; x[2] = pB; x[3] = pB + 4; x[4] = intr;
                Jp[0] = jq0.data(); Jp[1] = jt0.data(); Jp[2] = jq1.data(); Jp[3] = jt1.data(); Jp[4] = ji.data();
            }
            double* rb_r = r + 2 * d->blk_offset[b];
            rb->evaluate(x, rb_r, (J || Jamb) ? Jp : nullptr);
            if (!J && !Jamb) continue;
            double PJA[12], PJB[12];
            quat_plus_jacobian(pA, PJA);
            if (two) quat_plus_jacobian(pB, PJB);
            for (int i = 0; i < nres; ++i) {
                const size_t row = static_cast<size_t>(2 * d->blk_offset[b] + i);
                if (J) {
                    double* o = J + row * PL;
                    for (int k = 0; k < 3; ++k) {
                        double s = 0;
                        for (int m = 0; m < 4; ++m) s += jq0[static_cast<size_t>(i) * 4 + m] * PJA[m * 3 + k];
                        o[k] = s;
                        o[3 + k] = jt0[static_cast<size_t>(i) * 3 + k];
                    }
                    int off = 6;
                    if (two) {
                        for (int k = 0; k < 3; ++k) {
                            double s = 0;
                            for (int m = 0; m < 4; ++m) s += jq1[static_cast<size_t>(i) * 4 + m] * PJB[m * 3 + k];
                            o[6 + k] = s;
                            o[9 + k] = jt1[static_cast<size_t>(i) * 3 + k];
                        }
                        off = 12;
                    }
                    for (int k = 0; k < P; ++k) o[off + k] = ji[static_cast<size_t>(i) * P + k];
                }
                if (Jamb) {
                    double* o = Jamb + row * PA;
                    for (int m = 0; m < 4; ++m) o[m] = jq0[static_cast<size_t>(i) * 4 + m];
                    for (int m = 0; m < 3; ++m) o[4 + m] = jt0[static_cast<size_t>(i) * 3 + m];
                    int off = 7;
                    if (two) {
                        for (int m = 0; m < 4; ++m) o[7 + m] = jq1[static_cast<size_t>(i) * 4 + m];
                        for (int m = 0; m < 3; ++m) o[11 + m] = jt1[static_cast<size_t>(i) * 3 + m];
                        off = 14;
                    }
                    for (int k = 0; k < P; ++k) o[off + k] = ji[static_cast<size_t>(i) * P + k];
                }
            }
        }
    });
}

// cost = 1/2 sum_b rho(|r_b|^2)
int orc_reproj_cost(const cba_reproj_problem* d, double huber_delta, double* cost) {
    return guarded([&] {
        cba_options o; std::memset(&o, 0, sizeof(o));
        o.huber_delta = huber_delta; o.optimize_intrinsics = 1; o.optimize_extrinsics = 1; o.optimize_target_pose = 1;
        Built B;
        build(*d, o, B);
        B.prob.finalize_layout();
        std::vector<double> x;
        B.prob.gather(x);
        LMOptions l = to_lm(o, 1);
        B.prob.evaluate(x, l, false, cost, nullptr, nullptr);
    });
}

// LM solve of the whole problem, parameters updated in place (like the reference's blocks).
int orc_reproj_solve(const cba_reproj_problem* d, const cba_options* o, int threads, cba_summary* out) {
    return guarded([&] {
        Built B;
        build(*d, *o, B);
        LMSummary s;
        const auto t0 = std::chrono::steady_clock::now();
        B.prob.solve(to_lm(*o, threads), &s);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fill_summary(s, secs, out);
    });
}

int64_t orc_reproj_covariance_dim(const cba_reproj_problem* d) {
    const int P = intr_size(d->camera_model);
    int64_t n = static_cast<int64_t>(d->n_cams) * P;
    if (d->chain != CBA_CHAIN_INTRINSIC) n += 7LL * d->n_cams;
    if (d->chain != CBA_CHAIN_BUNDLE) n += 7LL * d->n_views;
    if (d->chain == CBA_CHAIN_BUNDLE) n += 7;
    return n;
}

// covariance at the current parameters, reference layout (ceresutils.h:69-126)
int orc_reproj_covariance(const cba_reproj_problem* d, const cba_options* o, double* cov) {
    return guarded([&] {
        Built B;
        build(*d, *o, B);
        std::vector<double> c;
        int dim = 0;
        if (!B.prob.covariance(to_lm(*o, 1), B.cov_order, &c, &dim)) throw std::runtime_error("covariance: rank deficient Jacobian");
        std::memcpy(cov, c.data(), sizeof(double) * c.size());
    });
}

// CPU baseline: wall seconds for `repeats` full residual+Jacobian evaluations of all blocks
// with `threads` workers over blocks (the reference parallelises the same way:
// copts.num_threads = hardware_concurrency, ceresutils.h:30).  Only block range [b0,b1).
double orc_reproj_bench_eval(const cba_reproj_problem* d, int b0, int b1, int threads, int repeats) {
    cba_options o; std::memset(&o, 0, sizeof(o));
    o.huber_delta = 1.0; o.optimize_intrinsics = 1; o.optimize_extrinsics = 1; o.optimize_target_pose = 1;
    Built B;
    try { build(*d, o, B); } catch (...) { return -1.0; }
    B.prob.finalize_layout();
    std::vector<double> x;
    B.prob.gather(x);
    const int nt = std::max(1, threads);
    std::vector<double> sink(nt, 0.0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < repeats; ++rep) {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) {
            th.emplace_back([&, t] {
                // Jacobian evaluation only (autodiff), no normal-equation assembly
                std::vector<double> r, Jbuf;
                for (int b = b0 + t; b < b1; b += nt) {
                    const ResidualBlock& rb = *B.prob.residuals[b];
                    r.assign(rb.nres, 0.0);
                    size_t jt = 0;
                    for (int id : rb.pb) jt += static_cast<size_t>(rb.nres) * B.prob.params[id].size;
                    Jbuf.assign(jt, 0.0);
                    const double* xp[5]; double* Jp[5];
                    size_t off = 0;
                    for (size_t k = 0; k < rb.pb.size(); ++k) {
                        const ParamBlock& p = B.prob.params[rb.pb[k]];
                        xp[k] = p.aoff >= 0 ? &x[p.aoff] : p.x;
                        Jp[k] = &Jbuf[off];
                        off += static_cast<size_t>(rb.nres) * p.size;
                    }
                    rb.evaluate(xp, r.data(), Jp);
                    sink[t] += r[0] + Jbuf[0];
                }
            });
        }
        for (auto& t : th) t.join();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (sink[0] == 1.2345e-300) std::printf(" ");
    return secs;
}

// ---- planar pose (variable projection) ---------------------------------------------------------
// residual (2N) and Jacobian (2N x 6, row-major) of PlanarPoseVPResidual at pose6 = [angle-axis, t]
int orc_planar_vp_eval(int n, const double* X, const double* Y, const double* u, const double* v, const double* kmtx5,
                       int num_radial, const double* pose6, double* r, double* J, double* alpha) {
    return guarded([&] {
        ViewData vd; vd.n = n; vd.X = X; vd.Y = Y; vd.u = u; vd.v = v;
        PlanarPoseVPBlock blk(vd, kmtx5, num_radial);
        const double* x[1] = {pose6};
        double* Jp[1] = {J};
        blk.evaluate(x, r, J ? Jp : nullptr);
        if (alpha) { std::vector<double> rr(2 * n); blk.residuals<double>(pose6, rr.data(), alpha); }
    });
}

// optimize_planar_pose core (planarpose.cpp:84-127) with pose6 in/out:
//   rms = sqrt(ssr / 2N), cov66 = (J~^T J~)^-1 * ssr / max(1, 2N - 6)  (ceresutils.h:117-123), distortion = alpha
int orc_planar_pose_solve(int n, const double* X, const double* Y, const double* u, const double* v, const double* kmtx5,
                          int num_radial, double* pose6, const cba_options* o, cba_summary* out, double* distortion,
                          double* rms, double* cov66) {
    return guarded([&] {
        ViewData vd; vd.n = n; vd.X = X; vd.Y = Y; vd.u = u; vd.v = v;
        Problem p;
        const int id = p.add_param(pose6, 6);
        auto rb = std::make_unique<PlanarPoseVPBlock>(vd, kmtx5, num_radial);
        const PlanarPoseVPBlock* raw = rb.get();
        rb->pb = {id};
        p.residuals.push_back(std::move(rb));
        LMSummary s;
        const auto t0 = std::chrono::steady_clock::now();
        p.solve(to_lm(*o, 1), &s);
        fill_summary(s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
        std::vector<double> r(2 * n), al(num_radial + 2);
        raw->residuals<double>(pose6, r.data(), al.data());
        double ssr = 0;
        for (double e : r) ssr += e * e;
        if (rms) *rms = std::sqrt(ssr / (2 * n));
        if (distortion) for (int k = 0; k < num_radial + 2; ++k) distortion[k] = al[k];
        if (cov66) {
            std::vector<double> c; int dim = 0;
            if (p.covariance(to_lm(*o, 1), {id}, &c, &dim)) {
                const double vf = ssr / std::max(1, 2 * n - 6);
                for (int k = 0; k < 36; ++k) cov66[k] = c[k] * vf;
            } else std::memset(cov66, 0, sizeof(double) * 36);
        }
    });
}

// ---- homography -------------------------------------------------------------------------------
// residual (2) + Jacobian (2x8 row-major) of one correspondence
void orc_homography_eval(const double* h8, double x, double y, double u, double v, double* r2, double* J28) {
    HomographyBlock blk(x, y, u, v);
    const double* xp[1] = {h8};
    double* Jp[1] = {J28};
    blk.evaluate(xp, r2, J28 ? Jp : nullptr);
}

// optimize_homography core (homography.cpp:144-175): h9 row-major in/out (first 8 entries are the parameters, H22 := 1);
// cov64 = (J~^T J~)^-1 * ssr / max(1, 2N - 8) with ssr = sum of squared LOSS-CORRECTED residuals (:163-170 call
// Problem::Evaluate with default EvaluateOptions, whose apply_loss_function is true — Ceres, third-party).
int orc_homography_solve(int n, const double* X, const double* Y, const double* u, const double* v, double* h9,
                         const cba_options* o, cba_summary* out, double* cov64) {
    return guarded([&] {
        if (n < 4) throw std::invalid_argument("At least 4 correspondences are required.");
        Problem p;
        const int id = p.add_param(h9, 8);
        for (int i = 0; i < n; ++i) {
            auto rb = std::make_unique<HomographyBlock>(X[i], Y[i], u[i], v[i]);
            rb->pb = {id};
            p.residuals.push_back(std::move(rb));
        }
        LMSummary s;
        const auto t0 = std::chrono::steady_clock::now();
        p.solve(to_lm(*o, 1), &s);
        fill_summary(s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
        h9[8] = 1.0;
        if (cov64) {
            double ssr = 0;
            for (int i = 0; i < n; ++i) {
                double r[2];
                homography_residual<double>(h9, X[i], Y[i], u[i], v[i], r);
                const double sq = r[0] * r[0] + r[1] * r[1];
                double w = 1.0;
                if (o->huber_delta > 0 && sq > o->huber_delta * o->huber_delta) w = o->huber_delta / std::sqrt(sq);
                ssr += w * sq;
            }
            std::vector<double> c; int dim = 0;
            if (p.covariance(to_lm(*o, 1), {id}, &c, &dim)) {
                const double vf = ssr / std::max(1, 2 * n - 8);
                for (int k = 0; k < 64; ++k) cov64[k] = c[k] * vf;
            } else std::memset(cov64, 0, sizeof(double) * 64);
        }
    });
}

// ---- AX = XB ---------------------------------------------------------------------------------
// residuals (6) + ambient Jacobians (6x4, 6x3) + tangent Jacobian (6x6) for one pair
void orc_axxb_eval(const double* q, const double* t, const double* RA, const double* RB, const double* tA,
                   const double* tB, double* r6, double* Jq64, double* Jt63, double* Jtan66) {
    AxXbBlock blk(RA, RB, tA, tB);
    const double* x[2] = {q, t};
    double jq[24], jt[18];
    double* Jp[2] = {jq, jt};
    blk.evaluate(x, r6, Jp);
    if (Jq64) std::memcpy(Jq64, jq, sizeof(jq));
    if (Jt63) std::memcpy(Jt63, jt, sizeof(jt));
    if (Jtan66) {
        double PJ[12];
        quat_plus_jacobian(q, PJ);
        for (int i = 0; i < 6; ++i) {
            for (int k = 0; k < 3; ++k) {
                double s = 0;
                for (int m = 0; m < 4; ++m) s += jq[i * 4 + m] * PJ[m * 3 + k];
                Jtan66[i * 6 + k] = s;
                Jtan66[i * 6 + 3 + k] = jt[i * 3 + k];
            }
        }
    }
}

// optimize_handeye core (handeye.cpp:45-58,69-76) on pre-built motion pairs:
// pairs[i] = [RA(9) RB(9) tA(3) tB(3)] row-major.  pose7 in/out; cov77 optional.
int orc_axxb_solve(int n_pairs, const double* pairs, double* pose7, const cba_options* o, cba_summary* out,
                   double* cov77) {
    return guarded([&] {
        if (n_pairs <= 0) throw std::runtime_error("No valid motion pairs after filtering. Increase motion or relax thresholds.");
        Problem p;
        const int qid = p.add_param(pose7, 4, BLK_QUAT);
        const int tid = p.add_param(pose7 + 4, 3);
        for (int i = 0; i < n_pairs; ++i) {
            const double* m = pairs + 24 * static_cast<size_t>(i);
            auto rb = std::make_unique<AxXbBlock>(m, m + 9, m + 18, m + 21);
            rb->pb = {qid, tid};
            p.residuals.push_back(std::move(rb));
        }
        LMSummary s;
        const auto t0 = std::chrono::steady_clock::now();
        p.solve(to_lm(*o, 1), &s);
        fill_summary(s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
        if (cov77) {
            std::vector<double> c; int dim = 0;
            if (p.covariance(to_lm(*o, 1), {qid, tid}, &c, &dim)) std::memcpy(cov77, c.data(), sizeof(double) * 49);
            else std::memset(cov77, 0, sizeof(double) * 49);
        }
    });
}

}  // extern "C"

// ---- semi-DLT intrinsics (variable projection over all views) ------------------------------------
namespace {
struct SemiDltProblem {
    Problem p;
    std::vector<ViewData> views;
    int kid = -1;
    std::vector<int> qid, tid;
    const CalibVPBlock* raw = nullptr;
    SemiDltProblem(int n_views, const int64_t* off, const double* X, const double* Y, const double* u, const double* v, double* kappa5,
                   double* poses7, int num_radial, const double* lo, const double* hi, const cba_options& o) {
        for (int i = 0; i < n_views; ++i) {
            ViewData vd; vd.n = static_cast<int>(off[i + 1] - off[i]);
            vd.X = X + off[i]; vd.Y = Y + off[i]; vd.u = u + off[i]; vd.v = v + off[i];
            views.push_back(vd);
        }
        // build_problem, intrinsicssemidlt.cpp:93-143
        kid = p.add_param(kappa5, 5);
        auto rb = std::make_unique<CalibVPBlock>(views, num_radial);
        raw = rb.get();
        rb->pb.push_back(kid);
        for (int i = 0; i < n_views; ++i) {
            qid.push_back(p.add_param(poses7 + 7 * static_cast<size_t>(i), 4, BLK_QUAT));
            tid.push_back(p.add_param(poses7 + 7 * static_cast<size_t>(i) + 4, 3));
            rb->pb.push_back(qid.back());
            rb->pb.push_back(tid.back());
        }
        p.residuals.push_back(std::move(rb));
        if (lo && hi)
            for (int k = 0; k < 5; ++k) { p.set_lower(kid, k, lo[k]); p.set_upper(kid, k, hi[k]); }
        if (!o.optimize_skew) p.params[kid].subset_const = 4;
    }
};
}  // namespace

extern "C" {
// residual (2N), ambient Jacobian blocks concatenated per row [intr5 | q0(4) t0(3) | q1 t1 ...] (2N x (5 + 7V)), alpha
int orc_semidlt_eval(int n_views, const int64_t* off, const double* X, const double* Y, const double* u, const double* v,
                     const double* kappa5, const double* poses7, int num_radial, double* r, double* Jamb, double* alpha) {
    return guarded([&] {
        std::vector<double> k(kappa5, kappa5 + 5), ps(poses7, poses7 + 7 * static_cast<size_t>(n_views));
        cba_options o; std::memset(&o, 0, sizeof(o)); o.optimize_skew = 1;
        SemiDltProblem sp(n_views, off, X, Y, u, v, k.data(), ps.data(), num_radial, nullptr, nullptr, o);
        const int nres = sp.raw->nres, nb = 1 + 2 * n_views, width = 5 + 7 * n_views;
        std::vector<const double*> xp(nb);
        std::vector<std::vector<double>> Js(nb);
        std::vector<double*> Jp(nb);
        xp[0] = k.data();
        for (int i = 0; i < n_views; ++i) { xp[1 + 2 * i] = &ps[7 * static_cast<size_t>(i)]; xp[2 + 2 * i] = &ps[7 * static_cast<size_t>(i) + 4]; }
        for (int bk = 0; bk < nb; ++bk) { const int sz = bk == 0 ? 5 : (bk % 2 == 1 ? 4 : 3); Js[bk].assign(static_cast<size_t>(nres) * sz, 0.0); Jp[bk] = Js[bk].data(); }
        sp.raw->evaluate(xp.data(), r, Jamb ? Jp.data() : nullptr);
        if (Jamb) {
            int col = 0;
            for (int bk = 0; bk < nb; ++bk) {
                const int sz = bk == 0 ? 5 : (bk % 2 == 1 ? 4 : 3);
                for (int i = 0; i < nres; ++i)
                    for (int c = 0; c < sz; ++c) Jamb[static_cast<size_t>(i) * width + col + c] = Js[bk][static_cast<size_t>(i) * sz + c];
                col += sz;
            }
        }
        if (alpha) { std::vector<double> rr(nres); sp.raw->residuals<double>(xp.data(), rr.data(), alpha); }
    });
}

// optimize_intrinsics_semidlt core (intrinsicssemidlt.cpp:155-191) with kappa5 and poses in/out:
// distortion = solve_full with the fixed entries (:74-90), view_errors (:137-153), cov = ceres::Covariance over
// [intr, quats, trans] * ssr / max(1, 2N - (5 + 7V)) (:184-188), zeros when rank deficient.
int orc_semidlt_solve(int n_views, const int64_t* off, const double* X, const double* Y, const double* u, const double* v, double* kappa5,
                      double* poses7, int num_radial, const double* lo, const double* hi, const int32_t* fixed_idx,
                      const double* fixed_val, int n_fixed, const cba_options* o, cba_summary* out, double* distortion,
                      double* view_errors, double* cov) {
    return guarded([&] {
        SemiDltProblem sp(n_views, off, X, Y, u, v, kappa5, poses7, num_radial, lo, hi, *o);
        LMSummary s;
        const auto t0 = std::chrono::steady_clock::now();
        sp.p.solve(to_lm(*o, 1), &s);
        fill_summary(s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), out);
        // solve_full: design matrix at the final point, coefficients listed in fixed_idx held at fixed_val
        const int m = num_radial + 2, N = sp.raw->total;
        std::vector<double> A(static_cast<size_t>(2 * N) * m), b(2 * N);
        {
            int row = 0;
            for (int vi = 0; vi < n_views; ++vi) {
                double R[9];
                quat_to_rotmat<double>(poses7 + 7 * static_cast<size_t>(vi), R);
                const double* t = poses7 + 7 * static_cast<size_t>(vi) + 4;
                const ViewData& w = sp.views[vi];
                for (int i = 0; i < w.n; ++i, row += 2) {
                    const double pc0 = R[0] * w.X[i] + R[1] * w.Y[i] + t[0], pc1 = R[3] * w.X[i] + R[4] * w.Y[i] + t[1],
                                 pc2 = R[6] * w.X[i] + R[7] * w.Y[i] + t[2];
                    const double x = pc0 / pc2, y = pc1 / pc2, r2 = x * x + y * y;
                    const double fx = kappa5[0], fy = kappa5[1], cx = kappa5[2], cy = kappa5[3], skew = kappa5[4];
                    double* Au = &A[static_cast<size_t>(row) * m];
                    double* Av = &A[static_cast<size_t>(row + 1) * m];
                    double rpow = r2;
                    for (int j = 0; j < num_radial; ++j) { Au[j] = fx * x * rpow + skew * y * rpow; Av[j] = fy * y * rpow; rpow *= r2; }
                    Au[num_radial] = fx * (2 * x * y) + skew * (r2 + 2 * y * y);
                    Au[num_radial + 1] = fx * (r2 + 2 * x * x) + skew * (2 * x * y);
                    Av[num_radial] = fy * (r2 + 2 * y * y);
                    Av[num_radial + 1] = fy * (2 * x * y);
                    b[row] = w.u[i] - (fx * x + skew * y + cx);
                    b[row + 1] = w.v[i] - (fy * y + cy);
                }
            }
        }
        if (N < 8) throw std::runtime_error("Failed to compute distortion parameters");
        std::vector<double> alpha(m, 0.0);
        std::vector<char> fixed(m, 0);
        for (int i = 0; i < n_fixed; ++i) {
            if (fixed_idx[i] < 0 || fixed_idx[i] >= m) throw std::invalid_argument("Fixed distortion index out of range");
            if (!fixed[fixed_idx[i]]) { fixed[fixed_idx[i]] = 1; alpha[fixed_idx[i]] = fixed_val ? fixed_val[i] : 0.0; }
        }
        std::vector<int> fr;
        for (int a = 0; a < m; ++a) if (!fixed[a]) fr.push_back(a);
        const int nf = static_cast<int>(fr.size());
        if (nf > 0) {  // distortion.h:333-356: the free columns, the right-hand side minus the fixed columns' share, thin SVD solve
            std::vector<double> Af(static_cast<size_t>(2 * N) * nf), badj(2 * N), af(nf);
            for (int rw = 0; rw < 2 * N; ++rw) {
                badj[rw] = b[rw];
                for (int a = 0; a < m; ++a) if (fixed[a]) badj[rw] -= A[static_cast<size_t>(rw) * m + a] * alpha[a];
                for (int i = 0; i < nf; ++i) Af[static_cast<size_t>(rw) * nf + i] = A[static_cast<size_t>(rw) * m + fr[i]];
            }
            lstsq_svd<double>(Af, 2 * N, nf, badj, af.data());
            for (int i = 0; i < nf; ++i) {
                if (!std::isfinite(af[i])) throw std::runtime_error("Failed to compute distortion parameters");
                alpha[fr[i]] = af[i];
            }
        }
        double ssr = 0;
        {
            int row = 0;
            for (int vi = 0; vi < n_views; ++vi) {
                double sv = 0;
                for (int i = 0; i < 2 * sp.views[vi].n; ++i, ++row) {
                    double rr = -b[row];
                    for (int a = 0; a < m; ++a) rr += A[static_cast<size_t>(row) * m + a] * alpha[a];
                    sv += rr * rr;
                }
                ssr += sv;
                if (view_errors) view_errors[vi] = std::sqrt(sv / (2.0 * sp.views[vi].n));
            }
        }
        if (distortion) for (int a = 0; a < m; ++a) distortion[a] = alpha[a];
        if (cov) {
            const size_t dim = 5 + 7 * static_cast<size_t>(n_views);
            std::memset(cov, 0, sizeof(double) * dim * dim);
            std::vector<int> order{sp.kid};
            for (int id : sp.qid) order.push_back(id);
            for (int id : sp.tid) order.push_back(id);
            std::vector<double> c; int d = 0;
            if (sp.p.covariance(to_lm(*o, 1), order, &c, &d)) {
                const double vf = ssr / std::max(1, 2 * N - static_cast<int>(dim));
                for (size_t k = 0; k < dim * dim; ++k) cov[k] = c[k] * vf;
            }
        }
    });
}
}  // extern "C"

