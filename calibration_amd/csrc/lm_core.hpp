// lm_core.hpp — host Levenberg-Marquardt driver over the Schur-reduced camera system.
//
// Restates what the reference delegates to ceres::Solve (src/estimation/detail/ceresutils.h:27-43):
// Ceres 2.x TrustRegionMinimizer + LevenbergMarquardtStrategy with the options the reference sets
// (function/gradient/parameter tolerance = epsilon, max_num_iterations) and Ceres' defaults
// (initial radius 1e4, max 1e16, min 1e-32, min_relative_decrease 1e-3, LM diagonal clamp
// [1e-6, 1e32], jacobi_scaling, monotonic steps, 5 consecutive invalid steps -> FAILURE), the
// per-residual-block Huber corrector, QuaternionManifold / SubsetManifold / constant blocks and the
// fx, fy >= 0 projection.  Ceres is a third-party dependency absent from /root/reference: see
// DESIGN.md "Solver semantics" for the restated rules and the one known deviation (no Armijo
// line search on bounds-constrained problems).
//
// Exchange pattern: every linear solve exchanges ONE packed buffer (PackLayout).  A trial point is
// linearised speculatively (Backend::sys_step) so that the step statistics and the next system travel together: an accepted
// step whose gain ratio is >= 0.937 (radius x 3, the usual case once the iteration converges) costs exactly one all-reduce.
//
// All O(#observations) and O(#views) arithmetic happens in the Backend (HIP kernels); this file
// only handles the reduced system (<= a few hundred unknowns), the accept/reject control flow,
// the shared parameter update and the one sum-all-reduce per linear solve.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>
#include <vector>

#include "dense.hpp"
#include "exp_env.hpp"
#include "line_search.hpp"
#include "lm_ctl.hpp"
#include "reproj_math.hpp"
#include "schur_math.hpp"
#include "structure.hpp"

namespace cba {

struct TrialStats {
    double gd = 0, dHd = 0;       // this rank's views' share of g^T d and d^T H d (private + cross terms)
    double step2 = 0, xnorm2 = 0; // private (per-view) share of |x+ - x|^2 and |x|^2
    double cost = 0;              // 1/2 sum_b rho(s_b) at the trial point, this rank's blocks
};

using AllReduce = std::function<void(double*, int64_t)>;

// Layout of the ONE packed buffer a linear solve exchanges between ranks: everything is a sum over ranks;
// the private-gradient max travels as one slot per rank (sum of a one-hot vector), so a single sum-all-reduce serves all.
//   [ step statistics (6) | per-camera weighted sums (n_cams * NACC) | cost | #failed views | S_schur (upper triangle, packed
//     row-major: n (n + 1) / 2) | g_schur (n) | gmax slots ]
struct PackLayout {
    int64_t stats = 0, cam = 6, cost = 0, nfail = 0, S = 0, g = 0, gmax = 0, size = 0;
    int n = 0, n_ranks = 1;
    // GD, DHD: the views' share of g^T d and d^T H d; SLOPE (line_eval): the views' share of the directional derivative
    enum { GD = 0, DHD = 1, STEP2 = 2, XNORM2 = 3, TRIAL_COST = 4, SLOPE = 5 };
    PackLayout() = default;
    PackLayout(const Structure& s, int ranks) : n(s.nsh), n_ranks(ranks) {
        cost = cam + static_cast<int64_t>(s.n_cams) * s.NACC;
        nfail = cost + 1;
        S = nfail + 1;
        g = S + static_cast<int64_t>(n) * (n + 1) / 2;
        gmax = g + n;
        size = gmax + n_ranks;
    }
    // the symmetric n x n matrix <-> its packed upper triangle in the pack
    void pack_S(const std::vector<double>& full, double* dst) const {
        for (int i = 0; i < n; ++i)
            for (int j = i; j < n; ++j) *dst++ = full[static_cast<size_t>(i) * n + j];
    }
    void unpack_S(const double* src, std::vector<double>& full) const {
        full.resize(static_cast<size_t>(n) * n);
        for (int i = 0; i < n; ++i)
            for (int j = i; j < n; ++j) {
                const double v = *src++;
                full[static_cast<size_t>(i) * n + j] = v;
                full[static_cast<size_t>(j) * n + i] = v;
            }
    }
};

struct Backend {
    virtual ~Backend() = default;
    virtual void set_view_fixed(const std::vector<int32_t>& fixed) = 0;
    // shared parameter copy `which` (0 current, 1 trial): intr [n_cams][PI], cam [n_cams][7], target [7]
    virtual void upload_shared(int which, const double* intr, const double* cam, const double* target) = 0;
    // at copy 0: block constants, Mode B normal equations, Huber weights, weighted per-camera sums
    // cam_acc [n_cams][NACC]; cost2 = {1/2 sum rho(s_b), sum s_b}
    virtual void normal_eq(double huber, std::vector<double>& cam_acc, double cost2[2]) = 0;
    // per-view elimination with the given radius; S_schur [nsh*nsh], g_schur [nsh] (this rank's views)
    virtual void schur(double radius, bool init_scale, bool constrained, std::vector<double>& S, std::vector<double>& g,
                       double* gmax_priv, int* nfail) = 0;
    // both of the above for a new linearisation; a device backend overrides it to queue everything and wait once
    virtual void normal_eq_schur(double huber, std::vector<double>& cam_acc, double cost2[2], double radius, bool init_scale,
                                 bool constrained, std::vector<double>& S, std::vector<double>& g, double* gmax_priv, int* nfail) {
        normal_eq(huber, cam_acc, cost2);
        schur(radius, init_scale, constrained, S, g, gmax_priv, nfail);
    }
    // back-substitute delta_sh, write private trial poses (copy 1), model terms, trial cost
    virtual void trial(const double* delta_sh, double huber, TrialStats* st) = 0;
    virtual void accept() = 0;  // private copy 1 -> copy 0
    virtual void download_private(double* view_pose) = 0;
    virtual void download_blocks(std::vector<double>& acc, std::vector<double>& w) = 0;

    // ---- packed systems: `pack` (PackLayout) leaves every call holding the sums OVER ALL RANKS after exactly ONE
    // sum-all-reduce.  A backend with a device-side collective packs and reduces on the device (backend_hip.hip over RCCL);
    // these defaults pack on the host and call `ar`.
    // a new linearisation at copy 0 + elimination with `radius`: fills [cam .. gmax]
    virtual void sys_new(double huber, double radius, bool init_scale, bool constrained, const PackLayout& L, const AllReduce& ar,
                         int rank, double* pack) {
        std::vector<double> cam_acc, S, g;
        double cost2[2] = {0, 0}, gm = 0;
        int nf = 0;
        normal_eq_schur(huber, cam_acc, cost2, radius, init_scale, constrained, S, g, &gm, &nf);
        std::fill(pack, pack + L.size, 0.0);
        std::copy(cam_acc.begin(), cam_acc.end(), pack + L.cam);
        pack[L.cost] = cost2[0];
        pack[L.nfail] = nf;
        L.pack_S(S, pack + L.S);
        std::copy(g.begin(), g.end(), pack + L.g);
        pack[L.gmax + rank] = gm;
        ar(pack + L.cam, L.size - L.cam);
    }
    // re-elimination of the current linearisation with another radius: fills [nfail .. g]
    virtual void sys_resolve(double radius, bool constrained, const PackLayout& L, const AllReduce& ar, int rank, double* pack) {
        std::vector<double> S, g;
        double gm = 0;
        int nf = 0;
        schur(radius, false, constrained, S, g, &gm, &nf);
        pack[L.nfail] = nf;
        L.pack_S(S, pack + L.S);
        std::copy(g.begin(), g.end(), pack + L.g);
        ar(pack + L.nfail, L.gmax - L.nfail);
        (void)rank;
    }
    // The SPECULATIVE step: back-substitute delta_sh (trial poses, the views' model-cost terms), then linearise AT THE TRIAL
    // POINT (copy 1, uploaded before) into the backend's second set of block sums, and eliminate with `radius_next` — the radius the
    // step will have if it is accepted with a gain ratio >= 0.937 (Ceres grows the radius by its maximum factor 3 then).
    // Fills the whole pack: the step statistics AND the next system travel in ONE all-reduce.  accept_step() makes the trial
    // linearisation the current one; after a rejected step the current block sums are untouched (sys_resolve works on them).
    // Returns false when the backend has no speculative path (the driver then uses trial() / sys_new()).
    virtual bool sys_step(const double* delta_sh, double huber, double radius_next, bool constrained, const PackLayout& L,
                          const AllReduce& ar, int rank, double* pack) {
        (void)delta_sh; (void)huber; (void)radius_next; (void)constrained; (void)L; (void)ar; (void)rank; (void)pack;
        return false;
    }
    virtual void accept_step() {}
    // One sample of the line search (line_search.hpp) at x (+) a * step, where `step` is the step of the last trial() / sys_step()
    // (its private part is still held by the backend; the shared part, already scaled, has been uploaded to copy 1): private trial
    // poses = Plus(x, a * step_p), the cost there and, if want_slope, the linearisation there (into the second set of block sums)
    // for the directional derivative along the step.  Fills TRIAL_COST, STEP2, XNORM2, SLOPE (private share) and, with want_slope,
    // the camera sums; one all-reduce.
    virtual void line_eval(double a, double huber, bool want_slope, const PackLayout& L, const AllReduce& ar, int rank, double* pack) = 0;
    // collectives the backend issued itself (device-side packing + RCCL): the driver adds them to its ExchangeStats
    int64_t device_allreduce_calls = 0, device_allreduce_doubles = 0;

    // ---- the controller form of the iteration (lm_ctl.hpp): the reduced solve, the step decision and the shared parameter
    // blocks live with the backend (in device memory on the GPU); the driver only reads the control record the controller
    // publishes after every exchange and queues the launch sequence it asks for.  Every ctl_* call below queues work and returns;
    // ctl_wait() returns the record of the LAST queued controller invocation.
    struct CtlSetup {
        double eps = 0;
        int max_iterations = 0;
        bool constrained = false, line_search = true, speculate = true, intr_var = true, target_var = false;
        const std::vector<char>*active = nullptr, *cam_var = nullptr;
        const double *intr = nullptr, *cam = nullptr, *target = nullptr;  // the start point (projected onto the bounds)
    };
    virtual bool ctl_begin(const CtlSetup& cs, const PackLayout& L) { (void)cs; (void)L; return false; }  // false: no controller
    // linearise at the current point, eliminate with the controller's radius, pack, ONE exchange, controller (CTL_NEW)
    virtual void ctl_new(double huber, bool first, const PackLayout& L, const AllReduce& ar, int rank) { (void)huber; (void)first; (void)L; (void)ar; (void)rank; }
    // eliminate the current linearisation again with the controller's radius, pack, ONE exchange, controller (CTL_RESOLVED)
    virtual void ctl_resolve(const PackLayout& L, const AllReduce& ar, int rank) { (void)L; (void)ar; (void)rank; }
    // the pending trial step (its shared part is the controller's): back-substitution, then the cost at the trial point or
    // (speculative) its linearisation and elimination with the predicted radius; pack, ONE exchange, controller (CTL_STEP)
    virtual void ctl_step(double huber, bool speculative, const PackLayout& L, const AllReduce& ar, int rank) { (void)huber; (void)speculative; (void)L; (void)ar; (void)rank; }
    // the last trial was accepted: private poses (and, after a speculative step, block sums and weights) trial -> current
    virtual void ctl_accept(bool blocks) { (void)blocks; }
    // Queue the head of the next speculative step (back-substitution, block constants, Mode B at the trial point) right behind the
    // controller invocation queued last, BEFORE its decision is known; the launches do nothing unless the controller's CS_GO flag
    // says that this is the step it asks for.  false: not supported.  If the record then confirms CS_GO the driver calls
    // ctl_step_tail() (the rest of the step), otherwise ctl_prelaunch_cancel().
    virtual bool ctl_prelaunch() { return false; }
    virtual void ctl_prelaunch_cancel() {}
    virtual void ctl_step_tail(double huber, const PackLayout& L, const AllReduce& ar, int rank) { (void)huber; (void)L; (void)ar; (void)rank; }
    virtual const double* ctl_wait() { return nullptr; }
    // current shared blocks and the pending shared step, for the host side of a line search / the end of the solve
    virtual void ctl_fetch(double* intr, double* cam, double* target, double* delta) { (void)intr; (void)cam; (void)target; (void)delta; }
    // the host ran a line search on the pending step: scal = the control scalars with CS_LS_* filled in; controller (CTL_LS_DONE)
    virtual void ctl_line_search_done(const double* scal) { (void)scal; }
};

// what a solve exchanged (cba_reproj_solve_stats)
struct ExchangeStats {
    int64_t allreduce_calls = 0, allreduce_doubles = 0;
    int32_t speculative_steps = 0;   // trial points linearised ahead of the accept decision
    int32_t speculation_hits = 0;    // ... accepted with the predicted radius: ONE collective for the whole LM step
    int32_t speculation_misses = 0;  // ... accepted with another radius: one re-elimination + collective more
    int32_t rejected_steps = 0;
    int32_t line_searches = 0;         // trust-region steps that failed the Armijo test at step size 1 (bounds-constrained problems)
    int32_t line_search_evaluations = 0;
};

class LMDriver {
  public:
    LMDriver(const Structure& s, Backend& be, std::vector<double>& intr, std::vector<double>& cam,
             std::vector<double>& view, std::vector<double>& target, AllReduce ar, int n_ranks, int rank)
        : s_(s), be_(be), intr_(intr), cam_(cam), view_(view), target_(target), n_ranks_(n_ranks), rank_(rank), L_(s, n_ranks) {
        ar_ = [this, raw = std::move(ar)](double* buf, int64_t n) {  // every exchange of the solve goes through here or is
            ++xs_.allreduce_calls;                                    // reported by the backend through note_exchange()
            xs_.allreduce_doubles += n;
            raw(buf, n);
        };
        pack_.assign(static_cast<size_t>(L_.size), 0.0);
        if (const char* env = std::getenv("CBA_LM_SPECULATE")) speculate_ = std::atoi(env) != 0;
        if (const char* env = std::getenv("CBA_LM_LINE_SEARCH")) line_search_ = std::atoi(env) != 0;
        if (const char* env = cba_exp_env("CBA_LM_CTL")) use_ctl_ = std::atoi(env) != 0;
        if (const char* env = cba_exp_env("CBA_LM_PIPELINE")) pipeline_ = std::atoi(env) != 0;
        if (const char* env = cba_exp_env("CBA_LM_PRELAUNCH")) prelaunch_ = std::atoi(env) != 0;
    }
    LMDriver(const LMDriver&) = delete;  // ar_ captures `this`
    LMDriver& operator=(const LMDriver&) = delete;
    void set_speculate(bool on) { speculate_ = on; }
    void set_line_search(bool on) { line_search_ = on; }
    void set_controller(bool on) { use_ctl_ = on; }  // false: the host-side form of the iteration (comparison / diagnosis)
    const ExchangeStats& exchange_stats() const { return xs_; }

    // ---- masks: which blocks Ceres would hold constant ----------------------------------------
    void setup(const cba_options& o) {
        const int n = s_.nsh;
        active_.assign(n, 1);
        intr_var_ = s_.chain == CBA_CHAIN_INTRINSIC ? true : o.optimize_intrinsics != 0;
        cam_var_.assign(s_.n_cams, 0);
        target_var_ = false;
        std::vector<int32_t> fixed(s_.n_views, 0);
        auto mask_intr = [&](int base) {
            for (int k = 0; k < s_.PI; ++k) active_[base + k] = intr_var_ ? 1 : 0;
            if (!o.optimize_skew) active_[base + 4] = 0;  // SubsetManifold({idx_skew}), pinhole.h:121
        };
        if (s_.chain == CBA_CHAIN_INTRINSIC) {
            mask_intr(0);
        } else if (s_.chain == CBA_CHAIN_EXTRINSIC) {  // extrinsics.cpp:110-150
            for (int c = 0; c < s_.n_cams; ++c) {
                cam_var_[c] = (o.optimize_extrinsics && c != 0) ? 1 : 0;
                for (int k = 0; k < 6; ++k) active_[c * s_.PC + k] = cam_var_[c];
                mask_intr(c * s_.PC + 6);
            }
            if (o.optimize_intrinsics && s_.n_views > 0 && s_.first_view_global == 0) fixed[0] = 1;
        } else {  // bundle.cpp:98-131
            target_var_ = o.optimize_target_pose != 0;
            for (int k = 0; k < 6; ++k) active_[k] = target_var_ ? 1 : 0;
            for (int c = 0; c < s_.n_cams; ++c) {
                cam_var_[c] = o.optimize_extrinsics ? 1 : 0;
                for (int k = 0; k < 6; ++k) active_[6 + c * s_.PC + k] = cam_var_[c];
                mask_intr(6 + c * s_.PC + 6);
            }
        }
        // bounds fx, fy >= 0 exist on every variable intrinsics block (intrinsics.cpp:81-82,
        // extrinsics.cpp:142-144, bundle.cpp:118-123) => Ceres treats the problem as constrained
        constrained_ = intr_var_;
        view_fixed_ = fixed;
        be_.set_view_fixed(fixed);
    }

    // the constant / gauge masks of `o`, for a solver that runs the iteration itself (resident_lm.hip)
    struct Masks {
        std::vector<char> active, cam_var;
        bool intr_var, target_var, constrained;
    };
    Masks masks(const cba_options& o) {
        setup(o);
        return Masks{active_, cam_var_, intr_var_, target_var_, constrained_};
    }

    void solve(const cba_options& o, cba_summary* out) {
        if (use_ctl_ && solve_ctl(o, out)) return;
        solve_host(o, out);
    }

    // The iteration with the controller (lm_ctl.hpp) next to the data: this loop takes no decision and does no arithmetic on
    // the reduced system — it reads the control record and queues what the record asks for.  One wait per trial step; a
    // re-elimination (rejected step, radius miss) and the trial step behind it are queued together.
    bool solve_ctl(const cba_options& o, cba_summary* out) {
        const auto t0 = std::chrono::steady_clock::now();
        setup(o);
        const double huber = o.huber_delta;
        project_shared();
        xs_ = ExchangeStats();
        Backend::CtlSetup cs;
        cs.eps = o.epsilon; cs.max_iterations = o.max_iterations;
        cs.constrained = constrained_; cs.line_search = line_search_; cs.speculate = speculate_;
        cs.intr_var = intr_var_; cs.target_var = target_var_;
        cs.active = &active_; cs.cam_var = &cam_var_;
        cs.intr = intr_.data(); cs.cam = cam_.data(); cs.target = target_.data();
        if (!be_.ctl_begin(cs, L_)) return false;
        static const bool timing = [] { const char* e = std::getenv("CBA_LM_TIMING"); return e && e[0] == '1'; }();
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        double t_wait = 0, t_queue = 0;
        int n_waits = 0, n_head_queued = 0;  // n_head_queued: waits that returned with the next step's head already on the stream
        double rec[CS_COUNT];
        auto wait = [&] {
            const auto a = now();
            const double* r = be_.ctl_wait();
            std::memcpy(rec, r, sizeof(rec));
            t_wait += us(a, now());
            ++n_waits;
        };
        be_.ctl_new(huber, true, L_, ar_, rank_);
        wait();
        double last_printed = 0;
        bool pre = false;  // the head of a speculative step is queued behind the controller invocation being waited for
        while (rec[CS_TERM] < 0.0) {
            const auto q0 = now();
            const int expect = static_cast<int>(rec[CS_EXPECT]);
            if (o.verbose && rec[CS_P_ITER] > last_printed) {
                last_printed = rec[CS_P_ITER];
                std::printf("[cba] it %3d cost %.12e cand %.12e rel %.3e radius %.3e |g| %.3e%s\n", static_cast<int>(rec[CS_P_ITER]),
                            rec[CS_P_COST], rec[CS_CAND_COST], rec[CS_REL], rec[CS_P_RADIUS], rec[CS_P_GMAX],
                            rec[CS_SPECULATED] != 0.0 ? " (speculative)" : "");
            }
            bool queued_spec = false;  // the last thing queued in this pass is a speculative step
            if (pre && rec[CS_GO] != 0.0) {  // as predicted: the step's head is already running; queue the rest
                ++n_head_queued;
                be_.ctl_step_tail(huber, L_, ar_, rank_);
                queued_spec = true;
            } else {
                if (pre) be_.ctl_prelaunch_cancel();
                if (rec[CS_ACCEPT] != 0.0) be_.ctl_accept(rec[CS_ACCEPT] == 2.0);
                if (expect == CTL_STEP) {
                    be_.ctl_step(huber, rec[CS_STEP_SPEC] != 0.0, L_, ar_, rank_);
                    queued_spec = rec[CS_STEP_SPEC] != 0.0;
                } else if (expect == CTL_RESOLVED) {
                    be_.ctl_resolve(L_, ar_, rank_);
                    // what follows the re-elimination does not depend on its result (unless the reduced system turns out unsolvable,
                    // in which case the controller ignores the step): queue it behind, one wait for both
                    if (pipeline_ && rec[CS_WILL_END] == 0.0) {
                        be_.ctl_step(huber, rec[CS_STEP_SPEC] != 0.0, L_, ar_, rank_);
                        queued_spec = rec[CS_STEP_SPEC] != 0.0;
                    }
                } else if (expect == CTL_NEW) {
                    be_.ctl_new(huber, false, L_, ar_, rank_);
                } else if (expect == CTL_LINE_SEARCH) {
                    ctl_line_search(huber, rec);
                } else {
                    throw std::runtime_error("LM controller: unexpected request");
                }
            }
            pre = prelaunch_ && queued_spec && be_.ctl_prelaunch();
            t_queue += us(q0, now());
            wait();
        }
        if (pre) be_.ctl_prelaunch_cancel();
        // the accepted point: shared blocks from the controller, private poses from the backend
        {
            std::vector<double> d(std::max(1, s_.nsh));
            be_.ctl_fetch(intr_.data(), cam_.data(), target_.data(), d.data());
        }
        if (!view_.empty()) be_.download_private(view_.data());
        xs_.allreduce_calls += be_.device_allreduce_calls;
        xs_.allreduce_doubles += be_.device_allreduce_doubles;
        be_.device_allreduce_calls = be_.device_allreduce_doubles = 0;
        xs_.speculative_steps = static_cast<int32_t>(rec[CS_N_SPEC]);
        xs_.speculation_hits = static_cast<int32_t>(rec[CS_N_HITS]);
        xs_.speculation_misses = static_cast<int32_t>(rec[CS_N_MISSES]);
        xs_.rejected_steps = static_cast<int32_t>(rec[CS_N_REJECTED]);
        cost_ = rec[CS_COST];
        const int term = static_cast<int>(rec[CS_TERM]), iter = static_cast<int>(rec[CS_ITER]);
        out->termination = term;
        out->success = term == CBA_TERM_CONVERGENCE;
        out->iterations = iter;
        out->successful_steps = static_cast<int>(rec[CS_SUCCESSFUL]);
        out->initial_cost = rec[CS_INITIAL_COST];
        out->final_cost = cost_;
        out->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (timing)
            std::fprintf(stderr, "[cba timing] solve %.1f us, %d iterations, %d waits: waiting %.1f us, host between a wait and the end of its "
                         "launches %.1f us (%.1f per wait; the host computes nothing there: it reads the record and queues launches), %d of the "
                         "waits returned with the next step's head already queued behind the controller (chip idle for the host: 0), "
                         "controller invocations out of turn %d\n", out->solve_seconds * 1e6, iter, n_waits, t_wait, t_queue,
                         t_queue / std::max(1, n_waits), n_head_queued, static_cast<int>(rec[CS_N_WASTED]));
        if (timing && rec[CS_PROF + CP_PUBLISH] > 0.0) {
            static const char* const names[CTL_NPROF] = {"entry", "adopt", "gradient norm", "assemble", "factorise", "back-substitute", "plus",
                                                         "model terms", "decide", "publish", "", ""};
            std::fprintf(stderr, "[cba timing] controller kernel phases, us over %d invocations:", static_cast<int>(rec[CS_SEQ]));
            for (int k = 0; k <= CP_PUBLISH; ++k) std::fprintf(stderr, " %s %.1f |", names[k], rec[CS_PROF + k] * 0.01);
            std::fprintf(stderr, "\n");
        }
        std::snprintf(out->report, sizeof(out->report), "calibba(schur LM, %d rank%s): %s iters=%d cost %.6e -> %.6e", n_ranks_,
                      n_ranks_ > 1 ? "s" : "", ctl_message(static_cast<int>(rec[CS_MSG])), iter, out->initial_cost, cost_);
        return true;
    }

    // Ceres' projected line search on the pending step of a bounds-constrained problem, which failed the Armijo test at step
    // size 1 (line_search.hpp): the samples are queued from here (a handful per search, rare); the decision on the step found goes
    // back to the controller.
    void ctl_line_search(double huber, const double* rec) {
        const int n = s_.nsh;
        ++xs_.line_searches;
        std::vector<double> full(n), scaled(n), tintr, tcam, ttarget;
        be_.ctl_fetch(intr_.data(), cam_.data(), target_.data(), full.data());
        const double cost0 = rec[CS_COST], slope0 = rec[CS_SLOPE0], dmax = rec[CS_DMAX];
        double step2_sh = 0, xnorm2_sh = 0;
        auto sample = [&](double a, bool with_slope) {
            for (int i = 0; i < n; ++i) scaled[i] = a * full[i];
            shared_plus(scaled, tintr, tcam, ttarget, &step2_sh, &xnorm2_sh);
            be_.upload_shared(1, tintr.data(), tcam.data(), ttarget.data());
            be_.line_eval(a, huber, with_slope, L_, ar_, rank_, pack_.data());
            LineSample ls;
            ls.step = a;
            ls.value = pack_[L_.stats + PackLayout::TRIAL_COST];
            ls.has_value = std::isfinite(ls.value);
            if (with_slope && ls.has_value) {
                const std::vector<double> cam_acc(pack_.begin() + L_.cam, pack_.begin() + L_.cost);
                std::vector<double> Ht(static_cast<size_t>(n) * n), gt(n);
                assemble_shared(cam_acc, Ht, gt);
                double sl = pack_[L_.stats + PackLayout::SLOPE];
                for (int i = 0; i < n; ++i) sl += gt[i] * full[i];  // (full is zero on inactive columns)
                ls.slope = sl;
                ls.has_slope = std::isfinite(sl);
            }
            return ls;
        };
        int evals = 0;
        LineSample first;  // the trial point the controller has just judged
        first.step = 1.0; first.value = rec[CS_CAND_COST]; first.has_value = std::isfinite(first.value);
        double a = armijo_line_search(cost0, slope0, dmax, sample, &evals, &first);
        xs_.line_search_evaluations += evals;
        if (!(a > 0.0)) {  // search failed: the full step, re-established on the device (one more sample)
            a = 1.0;
            (void)sample(1.0, false);
            ++xs_.line_search_evaluations;
        }
        double scal[CS_COUNT];
        std::memcpy(scal, rec, sizeof(scal));
        scal[CS_LS_A] = a;
        scal[CS_LS_COST] = pack_[L_.stats + PackLayout::TRIAL_COST];
        scal[CS_LS_STEP2] = pack_[L_.stats + PackLayout::STEP2];
        scal[CS_LS_XNORM2] = pack_[L_.stats + PackLayout::XNORM2];
        scal[CS_LS_STEP2_SH] = step2_sh;
        scal[CS_LS_XNORM2_SH] = xnorm2_sh;
        scal[CS_EXPECT] = CTL_LS_DONE;
        be_.ctl_line_search_done(scal);
    }

    void solve_host(const cba_options& o, cba_summary* out) {
        const auto t0 = std::chrono::steady_clock::now();
        setup(o);
        const int n = s_.nsh;
        const double eps = o.epsilon, huber = o.huber_delta;
        const double min_radius = 1e-32, max_radius = 1e16, min_rel_decrease = 1e-3;
        double radius = 1e4, decrease_factor = 2.0;
        int iter = 0, invalid = 0, successful = 0;

        project_shared();  // Ceres projects the start point onto the bounds
        xs_ = ExchangeStats();
        be_.upload_shared(0, intr_.data(), cam_.data(), target_.data());
        new_system(radius, true, huber);
        const double initial_cost = cost_;
        int term = CBA_TERM_FAILURE;
        const char* msg = "";
        std::vector<double> delta(n, 0.0), tintr, tcam, ttarget;
        // After a rejected step the next trial is evaluated the cheap way (cost only): rejections come in runs, and a
        // speculative linearisation that is thrown away costs a Mode B pass where Mode R would have done.
        bool plain_next = false;
        // ... and the trial point that will END the solve (|cost change| <= eps cost) is not worth linearising either: once the
        // iteration converges the relative cost change of accepted steps falls geometrically (x 1/10 .. 1/100 per step), so it is
        // extrapolated from the last two and the step predicted to fall below eps is evaluated the cheap way.  A wrong guess costs
        // one extra exchange (plain trial, then the new system); a missed one costs a Mode B pass instead of a Mode R pass.
        double rel_last = 0.0, rel_prev = 0.0;

        auto done = [&](int t, const char* m) { term = t; msg = m; };
        // CBA_LM_TIMING=1: where the host side of the iteration spends its time (printed once per solve)
        static const bool timing = [] { const char* e = std::getenv("CBA_LM_TIMING"); return e && e[0] == '1'; }();
        double t_reduced = 0, t_step = 0, t_adopt = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        if (gmax_ <= eps) {
            done(CBA_TERM_CONVERGENCE, "Gradient tolerance reached.");
        } else {
            while (true) {
                if (iter >= o.max_iterations) { done(CBA_TERM_NO_CONVERGENCE, "Maximum number of iterations reached."); break; }
                if (gmax_ <= eps) { done(CBA_TERM_CONVERGENCE, "Gradient tolerance reached."); break; }
                if (radius <= min_radius) { done(CBA_TERM_CONVERGENCE, "Minimum trust region radius reached."); break; }
                ++iter;
                const auto tr0 = now();
                bool valid = solve_reduced(radius, delta);
                t_reduced += us(tr0, now());
                TrialStats st;
                double step2_sh = 0, xnorm2_sh = 0, model_change = 0;
                bool speculated = false;
                // the radius a gain ratio >= 0.937 leads to, in the arithmetic of the update below (radius / (1/3) differs from
                // 3 * radius by one ulp for a quarter of all doubles: the comparison after the step is exact)
                const double radius_spec = std::min(max_radius, radius / (1.0 / 3.0));
                if (valid) {
                    shared_plus(delta, tintr, tcam, ttarget, &step2_sh, &xnorm2_sh);
                    be_.upload_shared(1, tintr.data(), tcam.data(), ttarget.data());
                    const bool expect_convergence = rel_prev > 0.0 && rel_last > 0.0 && rel_last * std::min(1.0, rel_last / rel_prev) <= 4.0 * eps;
                    const auto ts0 = now();
                    if (speculate_ && !plain_next && !expect_convergence)
                        speculated = be_.sys_step(delta.data(), huber, radius_spec, constrained_, L_, ar_, rank_, pack_.data());
                    t_step += us(ts0, now());
                    if (speculated) {  // statistics and the next system arrived in the same exchange
                        ++xs_.speculative_steps;
                        st.gd = pack_[L_.stats + PackLayout::GD]; st.dHd = pack_[L_.stats + PackLayout::DHD];
                        st.step2 = pack_[L_.stats + PackLayout::STEP2]; st.xnorm2 = pack_[L_.stats + PackLayout::XNORM2];
                        st.cost = pack_[L_.stats + PackLayout::TRIAL_COST];
                    } else {
                        be_.trial(delta.data(), huber, &st);
                        double buf[5] = {st.gd, st.dHd, st.step2, st.xnorm2, st.cost};
                        ar_(buf, 5);
                        st.gd = buf[0]; st.dHd = buf[1]; st.step2 = buf[2]; st.xnorm2 = buf[3]; st.cost = buf[4];
                    }
                    // model_cost_change = -(J d)^T (r + J d / 2) = -g^T d - 1/2 d^T H d (trust_region_minimizer.cc);
                    // the views contributed their share, the shared-shared part is added here
                    double gd_sh = 0, dHd_sh = 0;
                    for (int i = 0; i < n; ++i) {
                        if (delta[i] == 0.0) continue;
                        gd_sh += gc_[i] * delta[i];
                        double s = 0;
                        for (int j = 0; j < n; ++j) s += Hcc_[static_cast<size_t>(i) * n + j] * delta[j];
                        dHd_sh += delta[i] * s;
                    }
                    model_change = -(st.gd + gd_sh) - 0.5 * (st.dHd + dHd_sh);
                    if (!(model_change > 0.0) || !std::isfinite(model_change)) valid = false;
                    // Ceres' projected line search on bounds-constrained problems (line_search.hpp).  The trial point just evaluated
                    // IS its first sample (step size 1); only a step that fails the Armijo test there is searched — and then scaled.
                    const double slope0 = st.gd + gd_sh;
                    if (valid && constrained_ && line_search_ && !(std::isfinite(st.cost) && st.cost <= cost_ + 1e-4 * slope0)) {
                        ++xs_.line_searches;
                        speculated = false;  // whatever was linearised at step size 1 is not the point the step will end at
                        const std::vector<double> full = delta;
                        double dmax = 0.0;  // Ceres: max-norm of the whole direction; the views' part is bounded by the shared part's
                        for (double d : full) dmax = std::max(dmax, std::fabs(d));  // scale here (it only gates a 1e-9 cut-off)
                        std::vector<double> scaled(n);
                        auto sample = [&](double a, bool with_slope) {
                            for (int i = 0; i < n; ++i) scaled[i] = a * full[i];
                            shared_plus(scaled, tintr, tcam, ttarget, &step2_sh, &xnorm2_sh);
                            be_.upload_shared(1, tintr.data(), tcam.data(), ttarget.data());
                            be_.line_eval(a, huber, with_slope, L_, ar_, rank_, pack_.data());
                            LineSample ls;
                            ls.step = a;
                            ls.value = pack_[L_.stats + PackLayout::TRIAL_COST];
                            ls.has_value = std::isfinite(ls.value);
                            if (with_slope && ls.has_value) {
                                const std::vector<double> cam_acc(pack_.begin() + L_.cam, pack_.begin() + L_.cost);
                                std::vector<double> Ht(static_cast<size_t>(n) * n), gt(n);
                                assemble_shared(cam_acc, Ht, gt);
                                double sl = pack_[L_.stats + PackLayout::SLOPE];
                                for (int i = 0; i < n; ++i) sl += gt[i] * full[i];  // (full is zero on inactive columns)
                                ls.slope = sl;
                                ls.has_slope = std::isfinite(sl);
                            }
                            return ls;
                        };
                        int evals = 0;
                        LineSample first;  // the trial point just evaluated
                        first.step = 1.0; first.value = st.cost; first.has_value = std::isfinite(first.value);
                        double a = armijo_line_search(cost_, slope0, dmax, sample, &evals, &first);
                        xs_.line_search_evaluations += evals;
                        if (!(a > 0.0)) {  // search failed: the full step, re-established on the device (one more sample)
                            a = 1.0;
                            (void)sample(1.0, false);
                            ++xs_.line_search_evaluations;
                        }
                        for (int i = 0; i < n; ++i) delta[i] = a * full[i];
                        st.cost = pack_[L_.stats + PackLayout::TRIAL_COST];
                        st.step2 = pack_[L_.stats + PackLayout::STEP2];
                        st.xnorm2 = pack_[L_.stats + PackLayout::XNORM2];
                    }
                }
                if (!valid) {
                    if (++invalid >= 5) { done(CBA_TERM_FAILURE, "Number of consecutive invalid steps more than max."); break; }
                    radius *= 0.5;
                    resolve(radius);
                    plain_next = true;
                    continue;
                }
                invalid = 0;
                double cand_cost = st.cost;
                if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
                const double step_norm = std::sqrt(st.step2 + step2_sh);
                const double x_norm = std::sqrt(st.xnorm2 + xnorm2_sh);
                if (step_norm <= eps * (x_norm + eps)) { done(CBA_TERM_CONVERGENCE, "Parameter tolerance reached."); break; }
                const double cost_change = cost_ - cand_cost;
                if (std::fabs(cost_change) <= eps * cost_) { done(CBA_TERM_CONVERGENCE, "Function tolerance reached."); break; }
                const double rel = cost_change / model_change;
                if (o.verbose)
                    std::printf("[cba] it %3d cost %.12e cand %.12e rel %.3e radius %.3e |g| %.3e%s\n", iter, cost_, cand_cost,
                                rel, radius, gmax_, speculated ? " (speculative)" : "");
                if (rel > min_rel_decrease) {
                    intr_ = tintr; cam_ = tcam; target_ = ttarget;
                    ++successful;
                    rel_prev = rel_last;
                    rel_last = std::fabs(cost_change) / cost_;
                    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
                    radius = std::min(max_radius, radius);
                    decrease_factor = 2.0;
                    plain_next = false;
                    if (speculated) {
                        const auto ta0 = now();
                        be_.accept_step();  // the trial linearisation (block sums, weights, poses) becomes the current one
                        adopt_system(false);
                        t_adopt += us(ta0, now());
                        if (radius != radius_spec) {  // gain ratio below 0.937: the elimination was made with another radius
                            ++xs_.speculation_misses;
                            resolve(radius);
                        } else {
                            ++xs_.speculation_hits;
                        }
                    } else {
                        be_.accept();
                        be_.upload_shared(0, intr_.data(), cam_.data(), target_.data());
                        new_system(radius, false, huber);
                    }
                } else {
                    ++xs_.rejected_steps;
                    radius = radius / decrease_factor;
                    decrease_factor *= 2.0;
                    resolve(radius);
                    plain_next = true;
                }
            }
        }
        if (!view_.empty()) be_.download_private(view_.data());
        xs_.allreduce_calls += be_.device_allreduce_calls;
        xs_.allreduce_doubles += be_.device_allreduce_doubles;
        be_.device_allreduce_calls = be_.device_allreduce_doubles = 0;
        out->termination = term;
        out->success = term == CBA_TERM_CONVERGENCE;
        out->iterations = iter;
        out->successful_steps = successful;
        out->initial_cost = initial_cost;
        out->final_cost = cost_;
        out->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (timing)
            std::fprintf(stderr, "[cba timing] solve %.1f us, %d iterations: reduced solve %.1f, speculative step (queue + wait + exchange) %.1f, "
                         "accept + adopt %.1f us\n", out->solve_seconds * 1e6, iter, t_reduced, t_step, t_adopt);
        std::snprintf(out->report, sizeof(out->report), "calibba(schur LM, %d rank%s): %s iters=%d cost %.6e -> %.6e", n_ranks_,
                      n_ranks_ > 1 ? "s" : "", msg, iter, initial_cost, cost_);
    }

    // ---- covariance in the reference's layout (ceresutils.h:69-126) ----------------------------
    int64_t covariance_dim() const {
        int64_t n = static_cast<int64_t>(s_.n_cams) * s_.PI;
        if (s_.chain != CBA_CHAIN_INTRINSIC) n += 7LL * s_.n_cams;
        if (s_.chain != CBA_CHAIN_BUNDLE) n += 7LL * s_.n_views;
        else n += 7;
        return n;
    }

    // Shared blocks only — [intr[c]..., camera quats, camera trans] (everything but the per-view poses): the marginal
    // covariance of the parameters a calibration is run for, O(#views) work and O(shared^2) output where the reference's
    // all-block-pairs matrix is O(#views^2) (6.3 GB at C3).  SURVEY.md §8(f) rank 2.
    int64_t shared_covariance_dim() const {
        if (s_.chain == CBA_CHAIN_BUNDLE) return covariance_dim();
        int64_t n = static_cast<int64_t>(s_.n_cams) * s_.PI;
        if (s_.chain != CBA_CHAIN_INTRINSIC) n += 7LL * s_.n_cams;
        return n;
    }

    // sel_views / view_cov (with shared_only): the marginal covariance of the poses of the listed (local) views on demand,
    // view_cov [n_sel][7 x 7] in ambient coordinates [quaternion (4), translation (3)] - the diagonal blocks the reference-layout
    // matrix holds for those views, from the same Schur pieces: T_vv = H_pp^-1 + W_v S_cc W_v^T, W_v = H_pp^-1 E_v, lifted by
    // the quaternion's PlusJacobian.  A constant view gets zeros.  cov may be nullptr then (only the view blocks are wanted).
    void covariance(const cba_options& o, double* cov, bool shared_only = false, const int32_t* sel_views = nullptr, int n_sel = 0,
                    double* view_cov = nullptr) {
        setup(o);
        const int n = s_.nsh, PL = s_.PL, NH = s_.NH, NACC = s_.NACC;
        if (!shared_only && covariance_dim() > 20000)
            throw std::runtime_error("covariance: dense ambient matrix too large (use cba_reproj_covariance_shared or compute_covariance=false)");
        be_.upload_shared(0, intr_.data(), cam_.data(), target_.data());
        std::vector<double> cam_acc, acc, w;
        double cost2[2];
        be_.normal_eq(o.huber_delta, cam_acc, cost2);
        be_.download_blocks(acc, w);
        if (!view_.empty()) be_.download_private(view_.data());
        // shared Hessian over ALL ranks
        std::vector<double> Hcc(static_cast<size_t>(n) * n, 0.0), gc(n, 0.0);
        {
            std::vector<double> buf = cam_acc;
            ar_(buf.data(), static_cast<int64_t>(buf.size()));
            assemble_shared(buf, Hcc, gc);
        }
        // Rank test.  The reference-layout matrix (cba_reproj_covariance, ceresutils.h:69-126) mimics what ceres::Covariance does with
        // its default SPARSE_QR: a column is dependent when its R diagonal is below 20 (m + n) eps max_j |J_j| (SuiteSparseQR's
        // default tolerance, third-party, restated).  That tolerance grows with the ROW count and says "rank deficient" for any
        // problem of 1e8 observations.  The shared-block covariance (cba_reproj_covariance_shared) is this library's own API with no
        // reference counterpart to be faithful to: it tests what actually matters for the inverse it returns, the reciprocal
        // condition number of the Jacobi-scaled reduced system against ceres::Covariance::Options::min_reciprocal_condition_number
        // = 1e-14 (and positive definiteness of every per-view block).
        double cmax = 0;
        for (int i = 0; i < n; ++i) cmax = std::max(cmax, std::sqrt(Hcc[static_cast<size_t>(i) * n + i]));
        for (int v = 0; v < s_.n_views; ++v)
            for (int64_t k = s_.link_off[v]; k < s_.link_off[v + 1]; ++k)
                for (int i = 0; i < 6; ++i)
                    cmax = std::max(cmax, std::sqrt(w[s_.link_blk[k]] * acc[static_cast<size_t>(s_.link_blk[k]) * NACC + hidx(PL, i, i)]));
        const double rank_tol = 20.0 * static_cast<double>(2 * s_.n_obs + n + 6 * s_.n_views) * 2.220446049250313e-16 * cmax;
        // compact active shared indices; a non-constant column nobody observes is an all-zero Jacobian column:
        // ceres::Covariance::Compute fails on it (rank deficient) and the reference leaves the matrix empty
        std::vector<int> act;
        for (int i = 0; i < n; ++i)
            if (active_[i]) {
                if (Hcc[static_cast<size_t>(i) * n + i] == 0.0) throw std::runtime_error("covariance: rank deficient Jacobian (unobserved parameter)");
                act.push_back(i);
            }
        const int na = static_cast<int>(act.size());
        std::vector<int> free_views;
        for (int v = 0; v < s_.n_views; ++v)
            if (!view_fixed_[v]) free_views.push_back(v);
        const int nv = static_cast<int>(free_views.size());
        // per view: Hpp^-1 (6x6) and W_v = Hpp^-1 E_v (6 x na)
        std::vector<double> Hinv(static_cast<size_t>(nv) * 36), W(static_cast<size_t>(nv) * 6 * std::max(na, 1), 0.0);
        std::vector<double> S0(static_cast<size_t>(na) * na, 0.0);
        for (int a = 0; a < na; ++a)
            for (int b = 0; b < na; ++b) S0[static_cast<size_t>(a) * na + b] = Hcc[static_cast<size_t>(act[a]) * n + act[b]];
        std::vector<int> pos(n, -1);
        for (int a = 0; a < na; ++a) pos[act[a]] = a;
        std::vector<double> Sloc(static_cast<size_t>(na) * na, 0.0);
        for (int fv = 0; fv < nv; ++fv) {
            const int v = free_views[fv];
            std::vector<double> H(36, 0.0), E(static_cast<size_t>(6) * std::max(na, 1), 0.0);
            for (int64_t k = s_.link_off[v]; k < s_.link_off[v + 1]; ++k) {
                const int b = s_.link_blk[k];
                const double* A = &acc[static_cast<size_t>(b) * NACC];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) H[i * 6 + j] += w[b] * A[hidx_sym(PL, i, j)];
                for (int lc = 6; lc < PL; ++lc) {
                    const int g = s_.shared_col(s_.blk_cam[b], lc);
                    if (pos[g] < 0) continue;
                    for (int i = 0; i < 6; ++i) E[static_cast<size_t>(i) * na + pos[g]] += w[b] * A[hidx(PL, i, lc)];
                }
            }
            std::vector<double> L = H;
            if (!chol_inplace(L, 6)) throw std::runtime_error("covariance: rank deficient Jacobian (view block)");
            if (!shared_only)
                for (int i = 0; i < 6; ++i)
                    if (L[i * 6 + i] <= rank_tol) throw std::runtime_error("covariance: rank deficient Jacobian (view block)");
            std::vector<double> Hi;
            chol_inverse(L, 6, Hi);
            std::memcpy(&Hinv[static_cast<size_t>(fv) * 36], Hi.data(), sizeof(double) * 36);
            double* Wv = &W[static_cast<size_t>(fv) * 6 * std::max(na, 1)];
            for (int i = 0; i < 6; ++i)
                for (int a = 0; a < na; ++a) {
                    double sum = 0;
                    for (int k = 0; k < 6; ++k) sum += Hi[i * 6 + k] * E[static_cast<size_t>(k) * na + a];
                    Wv[static_cast<size_t>(i) * na + a] = sum;
                }
            for (int a = 0; a < na; ++a)
                for (int b = 0; b < na; ++b) {
                    double sum = 0;
                    for (int k = 0; k < 6; ++k) sum += E[static_cast<size_t>(k) * na + a] * Wv[static_cast<size_t>(k) * na + b];
                    Sloc[static_cast<size_t>(a) * na + b] += sum;
                }
        }
        if (na > 0) ar_(Sloc.data(), static_cast<int64_t>(Sloc.size()));
        for (size_t i = 0; i < S0.size(); ++i) S0[i] -= Sloc[i];
        std::vector<double> Lc = S0, Scc;
        if (na > 0) {
            if (shared_only) {  // reciprocal condition number of the Jacobi-scaled reduced system (see the rank-test note above)
                std::vector<double> Ss(S0.size());
                for (int a = 0; a < na; ++a)
                    for (int b = 0; b < na; ++b) {
                        const double da = S0[static_cast<size_t>(a) * na + a], db = S0[static_cast<size_t>(b) * na + b];
                        if (!(da > 0.0) || !(db > 0.0)) throw std::runtime_error("covariance: rank deficient Jacobian (reduced system)");
                        Ss[static_cast<size_t>(a) * na + b] = S0[static_cast<size_t>(a) * na + b] / std::sqrt(da * db);
                    }
                double lmin = 0, lmax = 0;
                sym_eig_minmax(Ss, na, &lmin, &lmax);
                if (!(lmin > 0.0) || lmin / lmax < 1e-14)
                    throw std::runtime_error("covariance: rank deficient Jacobian (reduced system: reciprocal condition number below 1e-14)");
            }
            if (!chol_inplace(Lc, na)) throw std::runtime_error("covariance: rank deficient Jacobian (reduced system)");
            if (!shared_only) {
                double dmin = 1e300;
                for (int i = 0; i < na; ++i) dmin = std::min(dmin, Lc[static_cast<size_t>(i) * na + i]);
                if (dmin <= rank_tol) throw std::runtime_error("covariance: rank deficient Jacobian (reduced system)");
            }
            chol_inverse(Lc, na, Scc);
        }
        if (view_cov) {
            std::vector<int> fidx(s_.n_views, -1);
            for (int fv = 0; fv < nv; ++fv) fidx[free_views[fv]] = fv;
            std::vector<double> WS(static_cast<size_t>(6) * std::max(na, 1));
            for (int k = 0; k < n_sel; ++k) {
                const int v = sel_views[k];
                if (v < 0 || v >= s_.n_views) throw std::invalid_argument("covariance: view index out of range");
                double* out = view_cov + static_cast<size_t>(k) * 49;
                for (int i = 0; i < 49; ++i) out[i] = 0.0;
                const int fv = fidx[v];
                if (fv < 0) continue;
                const double* Wv = &W[static_cast<size_t>(fv) * 6 * std::max(na, 1)];
                double T6[36];
                for (int i = 0; i < 6; ++i)
                    for (int b = 0; b < na; ++b) {
                        double sum = 0;
                        for (int a = 0; a < na; ++a) sum += Wv[static_cast<size_t>(i) * na + a] * Scc[static_cast<size_t>(a) * na + b];
                        WS[static_cast<size_t>(i) * na + b] = sum;
                    }
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        double sum = Hinv[static_cast<size_t>(fv) * 36 + i * 6 + j];
                        for (int a = 0; a < na; ++a) sum += WS[static_cast<size_t>(i) * na + a] * Wv[static_cast<size_t>(j) * na + a];
                        T6[i * 6 + j] = sum;
                    }
                const double* q = &view_[7 * static_cast<size_t>(v)];
                double P[7][6] = {{-q[1], -q[2], -q[3], 0, 0, 0}, {q[0], q[3], -q[2], 0, 0, 0}, {-q[3], q[0], q[1], 0, 0, 0},
                                  {q[2], -q[1], q[0], 0, 0, 0}, {0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1}};
                for (int r = 0; r < 7; ++r)
                    for (int c = 0; c < 7; ++c) {
                        double sum = 0;
                        for (int i = 0; i < 6; ++i)
                            for (int j = 0; j < 6; ++j) sum += P[r][i] * T6[i * 6 + j] * P[c][j];
                        out[r * 7 + c] = sum;
                    }
            }
            if (!cov) return;
        }
        // tangent covariance: [shared active (na) | free views (6 each)]
        const int nt = shared_only ? na : na + 6 * nv;
        std::vector<double> T(static_cast<size_t>(nt) * nt, 0.0);
        for (int a = 0; a < na; ++a)
            for (int b = 0; b < na; ++b) T[static_cast<size_t>(a) * nt + b] = Scc[static_cast<size_t>(a) * na + b];
        std::vector<double> WS(shared_only ? 0 : static_cast<size_t>(nv) * 6 * std::max(na, 1), 0.0);  // W_v Scc
        for (int fv = 0; fv < (shared_only ? 0 : nv); ++fv)
            for (int i = 0; i < 6; ++i)
                for (int b = 0; b < na; ++b) {
                    double sum = 0;
                    const double* Wv = &W[static_cast<size_t>(fv) * 6 * na];
                    for (int a = 0; a < na; ++a) sum += Wv[static_cast<size_t>(i) * na + a] * Scc[static_cast<size_t>(a) * na + b];
                    WS[(static_cast<size_t>(fv) * 6 + i) * na + b] = sum;
                    T[static_cast<size_t>(na + 6 * fv + i) * nt + b] = -sum;
                    T[static_cast<size_t>(b) * nt + na + 6 * fv + i] = -sum;
                }
        for (int fv = 0; fv < (shared_only ? 0 : nv); ++fv)
            for (int fw = 0; fw < nv; ++fw)
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        double sum = fv == fw ? Hinv[static_cast<size_t>(fv) * 36 + i * 6 + j] : 0.0;
                        const double* Ww = &W[static_cast<size_t>(fw) * 6 * std::max(na, 1)];
                        for (int a = 0; a < na; ++a) sum += WS[(static_cast<size_t>(fv) * 6 + i) * na + a] * Ww[static_cast<size_t>(j) * na + a];
                        T[static_cast<size_t>(na + 6 * fv + i) * nt + na + 6 * fw + j] = sum;
                    }
        // ambient rows: each is a short linear combination of tangent coordinates
        struct Row { int t[3]; double c[3]; int n; };
        std::vector<Row> rows;
        auto push_zero = [&](int count) { for (int i = 0; i < count; ++i) rows.push_back(Row{{0, 0, 0}, {0, 0, 0}, 0}); };
        auto push_euclid = [&](int tbase_global, int count) {  // shared euclidean columns
            for (int k = 0; k < count; ++k) {
                const int p = pos[tbase_global + k];
                if (p < 0) push_zero(1);
                else rows.push_back(Row{{p, 0, 0}, {1.0, 0, 0}, 1});
            }
        };
        auto push_quat = [&](const double* q, const int* t3) {  // PlusJacobian rows (4x3)
            const double PJ[12] = {-q[1], -q[2], -q[3], q[0], q[3], -q[2], -q[3], q[0], q[1], q[2], -q[1], q[0]};
            for (int r = 0; r < 4; ++r) {
                Row row{{0, 0, 0}, {0, 0, 0}, 0};
                for (int k = 0; k < 3; ++k)
                    if (t3[k] >= 0) { row.t[row.n] = t3[k]; row.c[row.n] = PJ[r * 3 + k]; ++row.n; }
                rows.push_back(row);
            }
        };
        auto sh3 = [&](int gbase, int* t3) { for (int k = 0; k < 3; ++k) t3[k] = pos[gbase + k]; };
        int t3[3];
        // block order of get_param_blocks(): intrinsics.cpp:34-50, extrinsics.cpp:50-67, bundle.cpp:48-68
        for (int c = 0; c < s_.n_cams; ++c) push_euclid(intr_base(c), s_.PI);
        if (s_.chain != CBA_CHAIN_INTRINSIC) {
            for (int c = 0; c < s_.n_cams; ++c) { sh3(campose_base(c), t3); push_quat(&cam_[7 * static_cast<size_t>(c)], t3); }
            for (int c = 0; c < s_.n_cams; ++c) push_euclid(campose_base(c) + 3, 3);
        }
        if (s_.chain != CBA_CHAIN_BUNDLE && shared_only) {
            // per-view blocks omitted
        } else if (s_.chain != CBA_CHAIN_BUNDLE) {
            std::vector<int> fidx(s_.n_views, -1);
            for (int fv = 0; fv < nv; ++fv) fidx[free_views[fv]] = fv;
            for (int v = 0; v < s_.n_views; ++v) {
                if (fidx[v] < 0) { push_zero(4); continue; }
                for (int k = 0; k < 3; ++k) t3[k] = na + 6 * fidx[v] + k;
                push_quat(&view_[7 * static_cast<size_t>(v)], t3);
            }
            for (int v = 0; v < s_.n_views; ++v) {
                if (fidx[v] < 0) { push_zero(3); continue; }
                for (int k = 0; k < 3; ++k) rows.push_back(Row{{na + 6 * fidx[v] + 3 + k, 0, 0}, {1.0, 0, 0}, 1});
            }
        } else {
            sh3(0, t3); push_quat(target_.data(), t3);
            push_euclid(3, 3);
        }
        const int64_t dim = static_cast<int64_t>(rows.size());
        for (int64_t i = 0; i < dim; ++i)
            for (int64_t j = 0; j < dim; ++j) {
                double sum = 0;
                for (int a = 0; a < rows[i].n; ++a)
                    for (int b = 0; b < rows[j].n; ++b)
                        sum += rows[i].c[a] * rows[j].c[b] * T[static_cast<size_t>(rows[i].t[a]) * nt + rows[j].t[b]];
                cov[i * dim + j] = sum;
            }
        (void)NH;
    }

  private:
    int intr_base(int c) const { return s_.chain == CBA_CHAIN_INTRINSIC ? 0 : s_.sh_base + c * s_.PC + 6; }
    int campose_base(int c) const { return s_.sh_base + c * s_.PC; }

    void project_shared() {
        if (!intr_var_) return;
        for (int c = 0; c < s_.n_cams; ++c) {
            double* p = &intr_[static_cast<size_t>(c) * s_.PI];
            p[0] = std::max(p[0], 0.0);
            p[1] = std::max(p[1], 0.0);
        }
    }

    // H_cc / g_c from the (all-reduced) weighted per-camera local sums
    void assemble_shared(const std::vector<double>& cam_acc, std::vector<double>& Hcc, std::vector<double>& gc) const {
        const int n = s_.nsh, PL = s_.PL;
        std::fill(Hcc.begin(), Hcc.end(), 0.0);
        std::fill(gc.begin(), gc.end(), 0.0);
        for (int c = 0; c < s_.n_cams; ++c) {
            const double* A = &cam_acc[static_cast<size_t>(c) * s_.NACC];
            for (int i = 0; i < PL; ++i) {
                const int gi = s_.shared_col(c, i);
                if (gi < 0) continue;
                gc[gi] += A[s_.NH + i];
                for (int j = 0; j < PL; ++j) {
                    const int gj = s_.shared_col(c, j);
                    if (gj < 0) continue;
                    Hcc[static_cast<size_t>(gi) * n + gj] += A[hidx_sym(PL, i, j)];
                }
            }
        }
    }

    // Evaluate J at the current point and eliminate with `radius`: one packed all-reduce.
    void new_system(double radius, bool init_scale, double huber) {
        be_.sys_new(huber, radius, init_scale, constrained_, L_, ar_, rank_, pack_.data());
        adopt_system(init_scale);
    }

    // pack_ holds an all-reduced linearisation (from sys_new or an accepted sys_step): make it the current system
    void adopt_system(bool init_scale) {
        const int n = s_.nsh;
        const std::vector<double> cam_acc(pack_.begin() + L_.cam, pack_.begin() + L_.cost);
        cost_ = pack_[L_.cost];
        nfail_ = static_cast<int>(pack_[L_.nfail] + 0.5);
        L_.unpack_S(pack_.data() + L_.S, Ssch_);
        gsch_.assign(pack_.begin() + L_.g, pack_.begin() + L_.gmax);
        double gm = 0;
        for (int r = 0; r < n_ranks_; ++r) gm = std::max(gm, pack_[L_.gmax + r]);  // max over ranks via per-rank slots
        Hcc_.assign(static_cast<size_t>(n) * n, 0.0);
        gc_.assign(n, 0.0);
        assemble_shared(cam_acc, Hcc_, gc_);
        // columns nobody observes (H_ii == 0) behave like constant blocks
        eff_.assign(n, 0);
        for (int i = 0; i < n; ++i) eff_[i] = active_[i] && Hcc_[static_cast<size_t>(i) * n + i] != 0.0;
        if (init_scale) {
            scale2_.assign(n, 1.0);
            for (int i = 0; i < n; ++i) {
                const double sc = 1.0 / (1.0 + std::sqrt(Hcc_[static_cast<size_t>(i) * n + i]));
                scale2_[i] = sc * sc;
            }
        }
        gmax_ = std::max(gm, shared_gmax());
    }

    void resolve(double radius) {
        be_.sys_resolve(radius, constrained_, L_, ar_, rank_, pack_.data());
        nfail_ = static_cast<int>(pack_[L_.nfail] + 0.5);
        L_.unpack_S(pack_.data() + L_.S, Ssch_);
        gsch_.assign(pack_.begin() + L_.g, pack_.begin() + L_.gmax);
    }

    double shared_gmax() const {
        const int n = s_.nsh;
        double m = 0;
        if (!constrained_) {
            for (int i = 0; i < n; ++i)
                if (eff_[i]) m = std::max(m, std::fabs(gc_[i]));
            return m;
        }
        // |Plus(x, -g) - x|_inf over the shared blocks
        std::vector<double> ng(n, 0.0), ti, tc, tt;
        for (int i = 0; i < n; ++i)
            if (eff_[i]) ng[i] = -gc_[i];
        double s2, x2;
        shared_plus(ng, ti, tc, tt, &s2, &x2);
        for (size_t i = 0; i < ti.size(); ++i) m = std::max(m, std::fabs(ti[i] - intr_[i]));
        for (size_t i = 0; i < tc.size(); ++i) m = std::max(m, std::fabs(tc[i] - cam_[i]));
        for (size_t i = 0; i < tt.size(); ++i) m = std::max(m, std::fabs(tt[i] - target_[i]));
        return m;
    }

    // (H_cc + D_c - S_schur) delta_c = -(g_c - g_schur) on the effective columns
    bool solve_reduced(double radius, std::vector<double>& delta) {
        const int n = s_.nsh;
        if (nfail_ > 0) return false;
        std::vector<int> idx;
        for (int i = 0; i < n; ++i)
            if (eff_[i]) idx.push_back(i);
        const int m = static_cast<int>(idx.size());
        std::fill(delta.begin(), delta.end(), 0.0);
        if (m == 0) return true;
        std::vector<double> A(static_cast<size_t>(m) * m), b(m);
        for (int a = 0; a < m; ++a) {
            const int i = idx[a];
            for (int c = 0; c < m; ++c) {
                const int j = idx[c];
                A[static_cast<size_t>(a) * m + c] = Hcc_[static_cast<size_t>(i) * n + j] - Ssch_[static_cast<size_t>(i) * n + j];
            }
            A[static_cast<size_t>(a) * m + a] += lm_diag_host(Hcc_[static_cast<size_t>(i) * n + i], scale2_[i], radius);
            b[a] = -(gc_[i] - gsch_[i]);
        }
        if (!chol_inplace(A, m)) return false;
        chol_solve(A, m, b.data());
        for (int a = 0; a < m; ++a) {
            if (!std::isfinite(b[a])) return false;
            delta[idx[a]] = b[a];
        }
        return true;
    }

    static double lm_diag_host(double hii, double scale2, double radius) {
        double ds = hii * scale2;
        ds = std::min(std::max(ds, 1e-6), 1e32);
        return ds / radius / scale2;
    }

    // Plus on the shared blocks (+ bounds projection); also the shared share of |x+ - x|^2, |x|^2
    void shared_plus(const std::vector<double>& delta, std::vector<double>& ti, std::vector<double>& tc,
                     std::vector<double>& tt, double* step2, double* xnorm2) const {
        ti = intr_; tc = cam_; tt = target_;
        double s2 = 0, x2 = 0;
        for (int c = 0; c < s_.n_cams; ++c) {
            const int ib = intr_base(c);
            double* p = &ti[static_cast<size_t>(c) * s_.PI];
            if (intr_var_) {
                for (int k = 0; k < s_.PI; ++k) p[k] += delta[ib + k];
                p[0] = std::max(p[0], 0.0);
                p[1] = std::max(p[1], 0.0);
                for (int k = 0; k < s_.PI; ++k) {
                    const double o = intr_[static_cast<size_t>(c) * s_.PI + k];
                    s2 += (p[k] - o) * (p[k] - o);
                    x2 += o * o;
                }
            }
            if (s_.chain != CBA_CHAIN_INTRINSIC && cam_var_[c]) {
                const int pb = campose_base(c);
                const double* q = &cam_[7 * static_cast<size_t>(c)];
                double* o = &tc[7 * static_cast<size_t>(c)];
                quat_plus(q, &delta[pb], o);
                for (int k = 0; k < 3; ++k) o[4 + k] = q[4 + k] + delta[pb + 3 + k];
                for (int k = 0; k < 7; ++k) { s2 += (o[k] - q[k]) * (o[k] - q[k]); x2 += q[k] * q[k]; }
            }
        }
        if (s_.chain == CBA_CHAIN_BUNDLE && target_var_) {
            quat_plus(target_.data(), &delta[0], tt.data());
            for (int k = 0; k < 3; ++k) tt[4 + k] = target_[4 + k] + delta[3 + k];
            for (int k = 0; k < 7; ++k) { s2 += (tt[k] - target_[k]) * (tt[k] - target_[k]); x2 += target_[k] * target_[k]; }
        }
        *step2 = s2;
        *xnorm2 = x2;
    }

    const Structure& s_;
    Backend& be_;
    std::vector<double>&intr_, &cam_, &view_, &target_;
    AllReduce ar_;
    int n_ranks_, rank_;
    PackLayout L_;
    std::vector<double> pack_;
    bool speculate_ = true;
    bool line_search_ = true;
    bool use_ctl_ = true;    // the controller form of the iteration when the backend has one (CBA_LM_CTL=0: host-side form)
    bool pipeline_ = true;   // queue a re-elimination and the step behind it together (CBA_LM_PIPELINE=0: one wait each)
    bool prelaunch_ = true;  // queue the head of the next speculative step before the controller's decision is known (CBA_LM_PRELAUNCH=0)
    ExchangeStats xs_;
    std::vector<char> active_, eff_;
    std::vector<char> cam_var_;
    std::vector<int32_t> view_fixed_;
    bool intr_var_ = true, target_var_ = false, constrained_ = true;
    std::vector<double> Hcc_, gc_, Ssch_, gsch_, scale2_;
    double cost_ = 0, gmax_ = 0;
    int nfail_ = 0;
};

}  // namespace cba
