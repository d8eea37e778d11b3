/* calibba.h — C ABI of libcalibba.so, the MI355X (gfx950) bundle-adjustment engine that
 * drops in behind VitalyVorobyev/calibration's `calib::estimation_optim` refinement API.
 *
 * Plain C: pointers, sizes, POD structs.  No C++/Eigen/torch types cross this boundary.
 * All floating-point data is IEEE fp64; all matrices row-major unless stated.
 * Every entry point returns a cba_status; cba_last_error() gives the message a C++ adapter
 * re-throws (std::invalid_argument for CBA_ERR_INVALID_ARGUMENT, std::runtime_error for
 * CBA_ERR_RUNTIME — the exception types of the reference, SURVEY.md §8b "Errors").
 *
 * Reference interface each group replaces (paths relative to the reference repo):
 *   cba_options                 include/calib/estimation/optim/optimize.h:24-33 (OptimOptions)
 *                               + intrinsics.h:13-20, extrinsics.h:22-27, bundle.h:30-36
 *   cba_summary                 include/calib/estimation/optim/optimize.h:35-40 (OptimResult)
 *   cba_optimize_intrinsics     include/calib/estimation/optim/intrinsics.h:35-39
 *                               (src/estimation/optim/intrinsics.cpp:98-120)
 *   cba_optimize_extrinsics     include/calib/estimation/optim/extrinsics.h:29-34
 *                               (src/estimation/optim/extrinsics.cpp:174-196)
 *   cba_optimize_bundle         include/calib/estimation/optim/bundle.h:58-63
 *                               (src/estimation/optim/bundle.cpp:147-170)
 *   cba_optimize_handeye        include/calib/estimation/optim/handeye.h:40-43
 *                               (src/estimation/optim/handeye.cpp:60-78)
 *   cba_optimize_planar_pose    include/calib/estimation/optim/planarpose.h:24-26
 *                               (src/estimation/optim/planarpose.cpp:84-127)
 *   cba_optimize_intrinsics_semidlt  include/calib/estimation/optim/intrinsics.h (optimize_intrinsics_semidlt)
 *                               (src/estimation/optim/intrinsicssemidlt.cpp:155-191)
 *   cba_optimize_homography     include/calib/estimation/optim/homography.h:17-18
 *                               (src/estimation/optim/homography.cpp:144-175)
 *   cba_reproj_* (handle API)   the ceres::Problem the reference builds and solves inside those
 *                               functions (intrinsics.cpp:63-90, extrinsics.cpp:86-160,
 *                               bundle.cpp:83-133, detail/ceresutils.h:27-43,69-126); exposed so
 *                               observations can stay resident in HBM across solves and so the
 *                               residual+Jacobian evaluation (ceres::CostFunction::Evaluate of
 *                               residuals/intrinsicresidual.h:20-35, extrinsicsresidual.h:28-46,
 *                               bundleresidual.h:36-56) can be called and timed on its own.
 *
 * Pose convention: a rigid transform is 7 doubles [qw, qx, qy, qz, tx, ty, tz] — exactly the
 * parameter blocks the reference hands to Ceres (observationutils.h:43-48 populate_quat_tran;
 * quaternion storage w,x,y,z, observationutils.h:20-24).  Results come back un-normalised, as in
 * the reference's blocks; the caller applies restore_pose (observationutils.h:50-62).
 * cba_pose_from_matrix / cba_pose_to_matrix restate those two helpers for hosts without Eigen.
 *
 * Camera parameter vectors follow CameraTraits (include/calib/models/pinhole.h:117-133,
 * scheimpflug.h:234-261):
 *   CBA_CAMERA_PINHOLE_BC   10: [fx, fy, cx, cy, skew, k1, k2, k3, p1, p2]
 *   CBA_CAMERA_SCHEIMPFLUG  12: the above + [tau_x, tau_y]
 *
 * Threading: entry points are re-entrant; one handle must not be used from two threads at once.
 */
#ifndef CALIBBA_H
#define CALIBBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBA_VERSION_STRING "0.1.0"

typedef enum cba_status {
    CBA_OK = 0,
    CBA_ERR_INVALID_ARGUMENT = 1, /* reference throws std::invalid_argument */
    CBA_ERR_RUNTIME = 2,          /* reference throws std::runtime_error */
    CBA_ERR_NO_DEVICE = 3,        /* no usable gfx950 device: the engine has NO CPU fallback */
    CBA_ERR_HIP = 4,              /* HIP / RCCL runtime error */
    CBA_ERR_INTERNAL = 5
} cba_status;

typedef enum cba_chain {
    CBA_CHAIN_INTRINSIC = 0, /* c_T_t = view pose                       (intrinsicresidual.h) */
    CBA_CHAIN_EXTRINSIC = 1, /* c_T_t = c_T_r * r_T_t                   (extrinsicsresidual.h) */
    CBA_CHAIN_BUNDLE = 2     /* c_T_t = g_T_c^-1 * b_T_g^-1 * b_T_t     (bundleresidual.h) */
} cba_chain;

typedef enum cba_camera_model { CBA_CAMERA_PINHOLE_BC = 0, CBA_CAMERA_SCHEIMPFLUG = 1 } cba_camera_model;

typedef enum cba_termination {
    CBA_TERM_CONVERGENCE = 0,    /* ceres::CONVERGENCE  -> success = true (ceresutils.h:42) */
    CBA_TERM_NO_CONVERGENCE = 1, /* max_iterations hit  -> success = false */
    CBA_TERM_FAILURE = 2
} cba_termination;

/* OptimOptions + the per-stage switches.  cba_options_default() gives the reference defaults. */
typedef struct cba_options {
    int32_t optimizer;            /* OptimizerType 0..3; accepted, ignored (one Schur solver) */
    int32_t max_iterations;       /* 1000 */
    double huber_delta;           /* 1.0; <= 0 disables the loss (intrinsics.cpp:70-71) */
    double epsilon;               /* 1e-9: function, gradient and parameter tolerance */
    int32_t compute_covariance;   /* 1 */
    int32_t verbose;              /* 0 */
    int32_t optimize_intrinsics;  /* ExtrinsicOptions (default 1) / BundleOptions (default 0);
                                     ignored (always on) for the intrinsic chain */
    int32_t optimize_skew;        /* 0 */
    int32_t optimize_extrinsics;  /* ExtrinsicOptions::optimize_extrinsics (1) /
                                     BundleOptions::optimize_hand_eye (1) */
    int32_t optimize_target_pose; /* BundleOptions::optimize_target_pose (1) */
} cba_options;

/* OptimResult.  `covariance` is fetched separately (cba_reproj_covariance). */
typedef struct cba_summary {
    int32_t success;     /* termination == CONVERGENCE */
    int32_t termination; /* cba_termination */
    int32_t iterations;
    int32_t successful_steps;
    double initial_cost;
    double final_cost; /* 1/2 sum_blocks rho(|r_block|^2), as ceres::Solver::Summary */
    double solve_seconds;
    char report[192]; /* brief report string (engine's own wording) */
} cba_summary;

/* One reprojection bundle problem.  A "residual block" is one planar view seen by one camera,
 * exactly one ceres residual block of the reference (2*N residuals, one loss per block).
 *
 *   chain INTRINSIC: n_cams = 1; block b has private pose view_pose[blk_view[b]] (c_T_t)
 *   chain EXTRINSIC: cam_pose[c] = c_T_r, view_pose[v] = r_T_t; block b = (view blk_view[b], cam blk_cam[b])
 *   chain BUNDLE:    cam_pose[c] = g_T_c, target_pose = b_T_t, blk_b_T_g[b] = robot pose (data);
 *                    n_views = 0
 * Observations are SoA over all blocks, CSR-indexed by blk_offset.
 */
typedef struct cba_reproj_problem {
    int32_t chain;        /* cba_chain */
    int32_t camera_model; /* cba_camera_model */
    int32_t n_blocks;
    int32_t n_cams;
    int32_t n_views;
    int32_t reserved0;
    int64_t first_view_global; /* index of local view 0 in the whole (multi-GPU) problem; the
                                  gauge rule "first target pose constant" (extrinsics.cpp:123-126)
                                  applies to global view 0 */
    const int64_t* blk_offset; /* [n_blocks + 1] */
    const int32_t* blk_cam;    /* [n_blocks] */
    const int32_t* blk_view;   /* [n_blocks]; ignored for BUNDLE */
    const double* blk_b_T_g;   /* BUNDLE: [n_blocks][12] = rotation row-major (9) + translation (3) */
    const double* X;           /* [n_obs] target-plane x  (PlanarObservation::object_xy) */
    const double* Y;           /* [n_obs] target-plane y */
    const double* u;           /* [n_obs] pixel u         (PlanarObservation::image_uv) */
    const double* v;           /* [n_obs] pixel v */
    double* intr;              /* [n_cams][10|12]  in/out */
    double* cam_pose;          /* [n_cams][7]      in/out (NULL for INTRINSIC) */
    double* view_pose;         /* [n_views][7]     in/out (NULL for BUNDLE) */
    double* target_pose;       /* [7]              in/out (BUNDLE only) */
} cba_reproj_problem;

typedef struct cba_reproj cba_reproj; /* opaque: owns device buffers + one HIP stream */

/* ---- library ------------------------------------------------------------------------------ */
const char* cba_version(void);
const char* cba_last_error(void); /* thread-local, valid until the next call on this thread */
int32_t cba_device_count(void);   /* number of visible HIP devices (0 if none) */
/* Handles and the one-shot calls return their device / page-locked blocks (up to 16 MiB each, 256 MiB per kind and device) and
 * their stream to a process-wide cache instead of the runtime: a pipeline that calls optimize_* stage after stage pays for the
 * allocations once (releasing them was 1.9 ms of a 5 ms call at the reference's test sizes).  This frees what the cache holds. */
void cba_trim_cache(void);
/* The device used by every entry point that takes neither a handle nor a device argument (the one-shot cba_optimize_* calls, the
 * batched per-view solvers and seeds).  Default 0.  With one process per GPU call cba_set_device(LOCAL_RANK) once: the batched
 * solvers have no exchange step, so several GPUs simply take slices of the views. */
cba_status cba_set_device(int32_t device);
int32_t cba_get_device(void);
void cba_options_default(cba_options* o);
int32_t cba_intrinsics_size(int32_t camera_model); /* 10 or 12 */
int32_t cba_local_columns(int32_t chain, int32_t camera_model); /* tangent columns per observation:
                                     INTRINSIC 6+P, otherwise 12+P  [poseA d(3) t(3) | poseB d(3) t(3) | intr] */

/* populate_quat_tran / restore_pose (observationutils.h:43-62) for 4x4 column-major matrices
 * (the memory layout of Eigen::Isometry3d::data()). */
void cba_pose_from_matrix(const double* m44_colmajor, double* pose7);
void cba_pose_to_matrix(const double* pose7, double* m44_colmajor);

/* ---- handle API --------------------------------------------------------------------------- */
/* Validates like the reference (empty block -> INVALID_ARGUMENT "No observations provided",
 * bad indices -> INVALID_ARGUMENT), copies observations into padded SoA device arrays and the
 * parameters into device blocks.  Host buffers may be freed after this returns. */
cba_status cba_reproj_create(const cba_reproj_problem* desc, int32_t device, cba_reproj** out);
/* The same from array-of-structures input, read in place: blk_obs[b] points to block b's observations as interleaved
 * {object_x, object_y, image_u, image_v} records (32 bytes each) — exactly the memory of the reference's
 * std::vector<PlanarObservation> (include/calib/estimation/linear/planarpose.h:22-26: two Eigen::Vector2d), so a binding
 * passes view.data() and never builds X / Y / u / v arrays (desc->X, Y, u, v are ignored and may be NULL; desc->blk_offset
 * still gives the record counts).  SURVEY.md §8(f) rank 4: at 1.6e8 observations the caller-side AoS -> SoA copy alone is
 * 5 GB read + 5 GB written. */
cba_status cba_reproj_create_aos(const cba_reproj_problem* desc, const double* const* blk_obs, int32_t device, cba_reproj** out);
void cba_reproj_destroy(cba_reproj* h);
cba_status cba_reproj_set_params(cba_reproj* h, const double* intr, const double* cam_pose,
                                 const double* view_pose, const double* target_pose);
cba_status cba_reproj_get_params(cba_reproj* h, double* intr, double* cam_pose, double* view_pose,
                                 double* target_pose);
int64_t cba_reproj_num_observations(const cba_reproj* h);

/* Residual + tangent-space Jacobian of every observation at the current parameters ("Mode A").
 * Output stays in HBM as SoA (r: [2][n_pad], J: [2*P][n_pad]); cba_reproj_eval_fetch copies it
 * out in Ceres layout: r[2*n_obs] interleaved (u0,v0,u1,v1,..) per block, J row-major
 * [2*n_obs][P] with the local column order of cba_local_columns().  Quaternion columns are in
 * the tangent space of ceres::QuaternionManifold (ambient 2Nx4 Jacobian times PlusJacobian). */
cba_status cba_reproj_eval(cba_reproj* h);
cba_status cba_reproj_eval_fetch(cba_reproj* h, double* r, double* J);
/* The same for the residual blocks [b0, b1) only: r [2 * n], J [2 * n][P] with n = observations of those blocks, indexed from
 * the first observation of block b0.  For problems whose full Mode A output (59 GB at BASELINE config 3) must not cross PCIe. */
cba_status cba_reproj_eval_fetch_blocks(cba_reproj* h, int32_t b0, int32_t b1, double* r, double* J);
/* Runs `iters` back-to-back evaluations on the handle's stream bracketed by HIP events;
 * returns the average milliseconds per evaluation of the dominant kernel region. */
cba_status cba_reproj_eval_timed(cba_reproj* h, int32_t warmup, int32_t iters, double* ms_per_eval);

/* The same for one Mode B pass (all launches that produce the per-block normal equations of a linearisation). */
cba_status cba_reproj_normal_eq_timed(cba_reproj* h, int32_t warmup, int32_t iters, double* ms_per_pass);

/* Arithmetic type of the per-observation kernels (BASELINE config 5's fp32-vs-fp64 study): 0 = fp64
 * (default), 1 = fp32: observations and per-block constants are rounded to fp32 once, the residual and
 * Jacobian rows are evaluated in fp32, and EVERY accumulator (J^T J, J^T r, |r|^2), the Schur step and the
 * LM stay fp64.  Affects cba_reproj_eval / _eval_timed / _cost / _block_normal_eq / _solve / _covariance.
 * With fp32 the Mode A output is float: fetch it with cba_reproj_eval_fetch_f32 (same layout as
 * cba_reproj_eval_fetch). */
cba_status cba_reproj_set_scalar(cba_reproj* h, int32_t scalar);
cba_status cba_reproj_eval_fetch_f32(cba_reproj* h, float* r, float* J);

/* Cost 1/2 sum rho(|r_b|^2) at the current parameters (residual-only pass). */
cba_status cba_reproj_cost(cba_reproj* h, double huber_delta, double* cost);

/* Per-block normal-equation blocks at the current parameters ("Mode B", unweighted):
 * out[b] = [ upper triangle of J_b^T J_b row-major (P(P+1)/2) | J_b^T r_b (P) | |r_b|^2 (1) ]. */
cba_status cba_reproj_block_normal_eq(cba_reproj* h, double* out);
int64_t cba_reproj_block_normal_eq_size(const cba_reproj* h); /* doubles per block */

/* Levenberg-Marquardt solve (what solve_problem + ceres::Solve do, ceresutils.h:27-43). */
cba_status cba_reproj_solve(cba_reproj* h, const cba_options* opts, cba_summary* summary);

/* How cba_reproj_solve runs the iteration.  0: stage by stage (every stage a kernel launch on the handle's stream; the only form
 * for multi-rank handles, verbose solves and the fp32 study).  The reduced system, the step decision, the radius update and the
 * next trial point of the shared blocks are the work of ONE single-workgroup controller kernel right behind the packed exchange
 * (csrc/lm_ctl.hip): the host queues launch sequences and reads a control record, it takes no part in the arithmetic.
 * 2: "resident" - the whole solve in ONE launch of a single-workgroup kernel, for problems too small to fill the chip (the
 * sizes the reference's own tests and pipelines run); falls back to 0 when the kernel cannot take the problem (reduced system
 * wider than 80, > 16 cameras, a transport set).  1 (default): resident below the measured crossover with the staged form
 * (intrinsic chain: n_views + 0.0105 n_obs <= 28, e.g. 10 views x 88 points; 20 x 88 is already faster staged), staged otherwise.
 * 3 (diagnostic, for A/B measurements): as 0, but the reduced solve and the step decision run on the host from a copy of the
 * reduced pack, as they did before the controller existed.
 * All forms follow the same rules, take the same decisions and agree to rounding. */
cba_status cba_reproj_set_lm_mode(cba_reproj* h, int32_t mode);

/* What the last host-driven cba_reproj_solve on this handle exchanged between ranks (one packed
 * sum-all-reduce per linear solve).  A trial point is linearised ahead of the accept decision, so an accepted step whose
 * gain ratio is >= 0.937 (Ceres then grows the radius by its maximum factor 3, which is the radius the elimination was
 * made with) costs exactly ONE collective; stats8 = {all-reduce calls, all-reduced doubles, speculative steps,
 * of those accepted with the predicted radius, accepted with another radius (+1 re-elimination and collective),
 * rejected steps, trust-region steps that went through the projected Armijo line search of bounds-constrained problems,
 * line-search evaluations (one collective each)}.  All zero after a resident-kernel solve.  CBA_LM_SPECULATE=0 selects the
 * two-exchange sequence, CBA_LM_LINE_SEARCH=0 switches the line search off. */
cba_status cba_reproj_solve_stats(const cba_reproj* h, int64_t stats8[8]);

/* Covariance in the reference's layout (ceresutils.h:69-126): dense symmetric, AMBIENT block
 * sizes, block order = get_param_blocks() of the stage (intrinsics.cpp:34-50,
 * extrinsics.cpp:50-67, bundle.cpp:48-68).  Returns CBA_ERR_RUNTIME if rank deficient (the
 * reference then leaves the matrix empty). */
int64_t cba_reproj_covariance_dim(const cba_reproj* h);
cba_status cba_reproj_covariance(cba_reproj* h, const cba_options* opts, double* cov /*[dim*dim]*/);

/* The same matrix restricted to the SHARED blocks — [intr[c]..., camera quats, camera trans] (for BUNDLE: every block) — i.e.
 * the marginal covariance of everything but the per-view poses, from the Schur-reduced system: O(#views) work and a
 * (shared dim)^2 result where the all-block-pairs matrix of ceresutils.h:80-84 is O(#views^2) (393 MB at 1000 views, 6.3 GB
 * at 4000).  Equal to the corresponding rows/columns of cba_reproj_covariance. */
int64_t cba_reproj_covariance_shared_dim(const cba_reproj* h);
cba_status cba_reproj_covariance_shared(cba_reproj* h, const cba_options* opts, double* cov /*[dim*dim]*/);
/* ... and the per-view blocks on demand: the marginal covariance of the poses of the listed views
 * (indices local to this handle), cov7x7 [n_sel][7][7] in ambient coordinates [quaternion (4), translation (3)] - the diagonal
 * blocks cba_reproj_covariance holds for those views, from the same Schur pieces, O(#views + n_sel) work instead of
 * O(#views^2); zeros for a view the gauge holds constant.  Same rank test as cba_reproj_covariance_shared.
 * INTRINSIC / EXTRINSIC chains (the bundle chain has no per-view poses).  ceresutils.h:69-126. */
cba_status cba_reproj_covariance_views(cba_reproj* h, const cba_options* opts, int32_t n_sel, const int32_t* view_idx,
                                       double* cov7x7);

/* ---- multi-GPU: views sharded across ranks, one sum-all-reduce per LM linear solve ---------- */
/* Host-buffer callback (any transport: gloo, MPI, ...): in-place sum of buf[count] over ranks. */
typedef int32_t (*cba_allreduce_fn)(double* buf, int64_t count, void* user);
cba_status cba_reproj_set_allreduce(cba_reproj* h, cba_allreduce_fn fn, void* user, int32_t n_ranks, int32_t rank);
/* RCCL-native: device-buffer ncclAllReduce on the handle's stream (xGMI within a node). */
#define CBA_RCCL_UNIQUE_ID_BYTES 128
cba_status cba_rccl_unique_id(uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES]);
cba_status cba_reproj_init_rccl(cba_reproj* h, const uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES],
                                int32_t n_ranks, int32_t rank);

/* ---- one-shot entry points mirroring the reference's free functions ------------------------- */
/* optimize_intrinsics: >= 4 views else INVALID_ARGUMENT (intrinsics.cpp:92-96).
 * cov may be NULL; otherwise [(P + 7*n_views)^2], order [intr, quats..., trans...]. */
cba_status cba_optimize_intrinsics(int32_t camera_model, int32_t n_views, const int64_t* view_offset,
                                   const double* X, const double* Y, const double* u, const double* v,
                                   double* intr, double* c_T_t /*[n_views][7]*/, const cba_options* opts,
                                   cba_summary* summary, double* cov);
/* optimize_extrinsics: views[v][c] = block; empty (view,cam) pairs are simply absent.
 * cov order: [intr[c]..., cam quats..., cam trans..., view quats..., view trans...]. */
cba_status cba_optimize_extrinsics(int32_t camera_model, int32_t n_cams, int32_t n_views, int32_t n_blocks,
                                   const int64_t* blk_offset, const int32_t* blk_view, const int32_t* blk_cam,
                                   const double* X, const double* Y, const double* u, const double* v,
                                   double* intr, double* c_T_r, double* r_T_t, const cba_options* opts,
                                   cba_summary* summary, double* cov);
/* optimize_bundle: n_cams == 0 / n_blocks == 0 -> INVALID_ARGUMENT (bundle.cpp:139-144).
 * cov order: [intr[c]..., g quats..., g trans..., b quat, b tran]. */
cba_status cba_optimize_bundle(int32_t camera_model, int32_t n_cams, int32_t n_blocks,
                               const int64_t* blk_offset, const int32_t* blk_cam, const double* blk_b_T_g,
                               const double* X, const double* Y, const double* u, const double* v,
                               double* intr, double* g_T_c, double* b_T_t, const cba_options* opts,
                               cba_summary* summary, double* cov);
/* optimize_handeye: AX = XB refinement over all motion pairs (handeye.cpp:60-78; pairs per
 * src/estimation/linear/handeyedlt.cpp:51-81 with min angle 0.5 deg).  Poses are 7-vectors.
 * RUNTIME error for < 2 poses / size mismatch / no valid pairs.  cov: [7*7] or NULL. */
cba_status cba_optimize_handeye(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                double* g_T_c /*[7] in/out*/, const cba_options* opts, cba_summary* summary,
                                double* cov);

/* estimate_handeye_dlt (include/calib/estimation/linear/handeye.h, src/estimation/linear/handeyedlt.cpp:126-137): the all-pairs
 * Tsai-Lenz seed — rotation from sum skew(alpha+beta) x = beta - alpha, translation from sum (R_A - I) t = R_X t_B - t_A, both
 * ridge 1e-12 — over the pairs that pass the filter at min_angle_deg (:25-49).  O(n^2) pairs are enumerated on the device.
 * g_T_c [7] out.  RUNTIME error for < 2 poses / no valid pairs.
 * estimate_and_optimize_handeye (include/calib/estimation/optim/handeye.h:64-67, handeye.cpp:80-87): that seed, then
 * cba_optimize_handeye. */
cba_status cba_estimate_handeye_dlt(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                    double min_angle_deg, double* g_T_c /*[7] out*/);
cba_status cba_estimate_and_optimize_handeye(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                             double min_angle_deg /*reference default 1.0*/, double* g_T_c /*[7] out*/,
                                             const cba_options* opts, cba_summary* summary, double* cov);

/* The same two steps on several GPUs (AX = XB row): every rank passes ALL n poses; rank r evaluates the pairs
 * (i, j > i) whose first pose i lies in its range (ranges balanced by pair count) and the 29 accumulated values
 * [H | g | cost | #pairs] of every evaluation are summed over ranks through `fn` (the callback of cba_reproj_set_allreduce), so
 * every rank runs the same LM on the same sums.  estimate != 0: start from the all-pairs Tsai-Lenz seed (g_T_c out), else refine
 * g_T_c in place.  `device`: this rank's GPU. */
cba_status cba_estimate_and_optimize_handeye_sharded(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                                     double min_angle_deg, int32_t estimate, double* g_T_c,
                                                     const cba_options* opts, cba_summary* summary, double* cov,
                                                     cba_allreduce_fn fn, void* user, int32_t n_ranks, int32_t rank, int32_t device);

/* ... and with RCCL over xGMI as the transport (BASELINE configs[3]: "8 MI355X"): `id` is the 128-byte unique id of
 * cba_rccl_unique_id(), created by one rank and handed to all of them by the caller; every rank calls this entry point (it is a
 * collective: the communicator is created inside, over the ranks' devices, and torn down at the end).  The 29 sums of every
 * evaluation are reduced IN PLACE in device memory on the evaluation's stream (ncclAllReduce), then copied back once.  A rank
 * that fails aborts the communicator so that its peers fail too instead of waiting. */
cba_status cba_estimate_and_optimize_handeye_rccl(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                                  double min_angle_deg, int32_t estimate, double* g_T_c,
                                                  const cba_options* opts, cba_summary* summary, double* cov,
                                                  const uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank,
                                                  int32_t device);

/* optimize_planar_pose (include/calib/estimation/optim/planarpose.h:24-26, src/estimation/optim/planarpose.cpp:84-127):
 * pose refinement of ONE planar view for fixed K = [fx, fy, cx, cy, skew] by variable projection over the
 * Brown-Conrady coefficients (num_radial radial + 2 tangential, PlanarPoseOptions::num_radial default 2).
 * pose7 in/out; distortion [num_radial + 2] = fitted coefficients; reprojection_error = sqrt(ssr / 2N);
 * cov36 = 6x6 covariance of [angle-axis, t] scaled by ssr / max(1, 2N - 6) (zeros if rank deficient), may be NULL.
 * Fewer than 8 observations: the reference's functor fails to evaluate and Ceres reports FAILURE
 * (success = false), not an exception; same here.
 * The _batch form solves n_views independent views in one launch (one GPU thread per view); arrays are
 * per view: pose7 [n_views][7], summaries [n_views], distortion [n_views][num_radial + 2], etc. */
cba_status cba_optimize_planar_pose(int32_t n, const double* X, const double* Y, const double* u, const double* v,
                                    const double* kmtx5, int32_t num_radial, double* pose7, const cba_options* opts,
                                    cba_summary* summary, double* distortion, double* reprojection_error, double* cov36);
cba_status cba_optimize_planar_pose_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                          const double* u, const double* v, const double* kmtx5, int32_t num_radial,
                                          double* pose7, const cba_options* opts, cba_summary* summaries, double* distortion,
                                          double* reprojection_error, double* cov36);

/* optimize_homography (include/calib/estimation/optim/homography.h:17-18, src/estimation/optim/homography.cpp:144-175):
 * refinement of the 8 free entries of a plane-to-image homography (H22 = 1), one 2-residual block PER
 * CORRESPONDENCE, each with its own Huber loss (homography.cpp:132-142).  h9 in/out, row-major 3x3: the first 8
 * entries are taken as given (HomographyBlocks::create :79-84), H22 comes back as 1.  Fewer than 4 correspondences:
 * INVALID_ARGUMENT (:146-148).  cov64 = 8x8 covariance scaled by ssr / max(1, 2N - 8), ssr from the loss-corrected
 * residuals (:163-173 evaluate the ceres::Problem with its default EvaluateOptions); zeros if rank deficient; may be
 * NULL.  The _batch form refines n_views independent views in one launch (one wavefront per view). */
cba_status cba_optimize_homography(int32_t n, const double* X, const double* Y, const double* u, const double* v,
                                   double* h9, const cba_options* opts, cba_summary* summary, double* cov64);
cba_status cba_optimize_homography_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                         const double* u, const double* v, double* h9 /*[n_views][9]*/,
                                         const cba_options* opts, cba_summary* summaries, double* cov64 /*[n_views][64]*/);

/* optimize_intrinsics_semidlt (include/calib/estimation/optim/intrinsics.h, src/estimation/optim/intrinsicssemidlt.cpp:155-191):
 * refinement of K = [fx, fy, cx, cy, skew] and one pose per view with the Brown-Conrady coefficients of ALL views
 * eliminated by linear least squares inside the cost (CalibVPResidual, residuals/intrinsicsemidltresidual.h:19-73): one
 * residual block, one Huber loss, QuaternionManifold per view, skew held by a SubsetManifold unless opts->optimize_skew,
 * optional box bounds on K (CalibrationBounds; both pointers NULL = none).
 * kmtx5 in/out; c_T_t [n_views][7] in/out — the reference seeds these inside the call with calib::estimate_planar_pose
 * (host code of calib::estimation_linear, intrinsicssemidlt.cpp:37-40); the adapter calls it and passes the result.
 * After the solve the coefficients are re-fitted with the listed entries held fixed (solve_full :74-90, distortion.h:296-363):
 * distortion [num_radial + 2] = [k1.., p1, p2]; view_errors [n_views] = per-view RMS (:137-153);
 * cov [(5 + 7 n_views)^2], block order [K, all quaternions, all translations], scaled by ssr / max(1, 2N - (5 + 7 n_views))
 * (:184-188), zeros if rank deficient; may be NULL.  Fewer than 4 views: CBA_OK with summary->success = 0 and nothing written
 * (the reference prints a message and returns a default result, :163-166). */
cba_status cba_optimize_intrinsics_semidlt(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                           const double* u, const double* v, double* kmtx5, double* c_T_t, int32_t num_radial,
                                           const double* bounds_lo5, const double* bounds_hi5, const int32_t* fixed_distortion_indices,
                                           const double* fixed_distortion_values, int32_t n_fixed, const cba_options* opts,
                                           cba_summary* summary, double* distortion, double* view_errors, double* cov);

/* The same refinement with the VIEWS sharded over ranks (one process per GPU; BASELINE configs[3] names 8): this rank passes the
 * observations of views [first_view, first_view + n_views_local) of n_views_total (view_offset [n_views_local + 1] into its own
 * X, Y, u, v) and the seeds of the whole problem (kmtx5, c_T_t [n_views_total][7], identical on every rank).  The
 * O(#observations) passes run on the local views; per evaluation the ranks exchange the m(m+1)/2 + m sums that determine the
 * eliminated distortion coefficients and the table of per-view sums (78 + 22 m doubles per view, every rank filling its own rows),
 * after which the O(#views) step of the solver runs identically on every rank.  Outputs cover the whole problem and are identical
 * on every rank: kmtx5, c_T_t, distortion, view_errors [n_views_total], cov [(5 + 7 n_views_total)^2].  Agrees with the
 * single-GPU call to rounding (the sums are added in another order).  Transport: the host callback of cba_reproj_set_allreduce
 * (_sharded), or RCCL on device memory with the id from cba_rccl_unique_id on rank 0 (_rccl: a communicator is created for the
 * call; a rank that fails aborts it so that its peers' collectives fail instead of hanging).
 * src/estimation/optim/intrinsicssemidlt.cpp:155-191. */
cba_status cba_optimize_intrinsics_semidlt_sharded(int32_t n_views_local, const int64_t* view_offset, const double* X, const double* Y,
                                                   const double* u, const double* v, int32_t n_views_total, int32_t first_view,
                                                   double* kmtx5, double* c_T_t, int32_t num_radial, const double* bounds_lo5,
                                                   const double* bounds_hi5, const int32_t* fixed_distortion_indices,
                                                   const double* fixed_distortion_values, int32_t n_fixed, const cba_options* opts,
                                                   cba_summary* summary, double* distortion, double* view_errors, double* cov,
                                                   cba_allreduce_fn fn, void* user, int32_t n_ranks, int32_t rank, int32_t device);
cba_status cba_optimize_intrinsics_semidlt_rccl(int32_t n_views_local, const int64_t* view_offset, const double* X, const double* Y,
                                                const double* u, const double* v, int32_t n_views_total, int32_t first_view,
                                                double* kmtx5, double* c_T_t, int32_t num_radial, const double* bounds_lo5,
                                                const double* bounds_hi5, const int32_t* fixed_distortion_indices,
                                                const double* fixed_distortion_values, int32_t n_fixed, const cba_options* opts,
                                                cba_summary* summary, double* distortion, double* view_errors, double* cov,
                                                const uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank, int32_t device);

/* estimate_homography, DLT path (include/calib/estimation/linear/homography.h, src/estimation/optim/homography.cpp:31-43 ->
 * HomographyEstimator::fit, src/estimation/linear/homographyestimator.cpp:123-146): Hartley-normalised DLT of every view in one
 * launch.  h9 [n_views][9] row-major = T_dst^-1 Hn T_src with Hn(2,2) = 1, returned WITHOUT a final rescale exactly as the reference
 * does (homographyestimator.cpp:70, 79-87); success [n_views] = 0 where fit fails (< 4 correspondences, non-finite H), h9 is then
 * the identity.  The natural seed of cba_optimize_homography_batch. */
cba_status cba_estimate_homography_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                         const double* u, const double* v, double* h9, int32_t* success);

/* estimate_planar_pose (include/calib/estimation/linear/planarpose.h:38-110, src/estimation/linear/planarpose_linear.cpp:54-76)
 * for a batch of views in one launch: pixels normalised by K = [fx, fy, cx, cy, skew], Hartley-normalised DLT homography
 * (src/estimation/linear/homographyestimator.cpp:17-87), pose_from_homography_normalized (planarpose_linear.cpp:17-52).
 * pose7 [n_views][7] out; a view with fewer than 4 points gets the identity, as in the reference (:55-57).
 * This is the seed optimize_intrinsics' callers and optimize_intrinsics_semidlt (intrinsicssemidlt.cpp:37-40) start from. */
cba_status cba_estimate_planar_pose_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                          const double* u, const double* v, const double* kmtx5, double* pose7);

#ifdef __cplusplus
}
#endif
#endif /* CALIBBA_H */
