// lm_ctl.hpp — the CONTROLLER of the Levenberg-Marquardt iteration as __host__ __device__ code: everything the solve does on the
// reduced (shared) system between two exchanges — adopt an all-reduced linearisation (H_cc, g_c, effective columns, Jacobi scale,
// gradient max-norm), factorise and solve (H_cc + D - S) d = -(g_c - g_s), Plus on the shared blocks, the model-cost terms, the
// gain ratio, accept / reject, the radius update, the convergence tests — for ONE cooperating team of threads.
//
// On the GPU the team is one workgroup (lm_ctl.hip: k_lm_ctl, launched on the engine's stream right behind the packed exchange),
// so an LM step never leaves the device: `k_pack -> ncclAllReduce -> k_lm_ctl`, and the host merely reads the control record
// the kernel publishes to learn which launch sequence comes next.  tests/cpu_backend instantiates the same code with a team
// of one host thread (SerialTeam), which is how the CPU tier checks the decisions and the exchange protocol without a GPU.
//
// Restates what the reference delegates to ceres::Solve (src/estimation/detail/ceresutils.h:27-43): Ceres 2.x
// TrustRegionMinimizer + LevenbergMarquardtStrategy with the reference's options (tolerances = epsilon, max_num_iterations) and
// Ceres' defaults; rules and constants as in lm_core.hpp (which keeps the host-side form of the same iteration for comparison).
// Ceres is a third-party dependency absent from /root/reference: parity with its own iteration numerics is unpinned.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/calibba.h"
#include "reproj_math.hpp"
#include "schur_math.hpp"

#if defined(__HIPCC__)
#define CBA_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define CBA_NOINLINE inline
#endif

namespace cba {

// ---- control scalars (doubles; CtlView::scal, published verbatim as the control record) -----------------------------------
enum CtlSlot : int {
    CS_SEQ = 0,        // number of controller invocations so far (the host waits for its own count)
    CS_TERM,           // -1 while the solve runs, else CBA_TERM_*
    CS_MSG,            // CtlMsg
    CS_EXPECT,         // the controller mode that must run next (CtlMode; 0 = none: the solve has ended)
    CS_STEP_SPEC,      // the next trial step is evaluated speculatively (linearised at the trial point) / the cheap way (cost only)
    CS_ACCEPT,         // 0 nothing; 1 the last trial was accepted after a plain evaluation (private poses trial -> current);
                       // 2 accepted after a speculative one (poses + block sums + weights)
    CS_GO,             // 1: this invocation accepted a speculative step with the predicted radius and asks for another speculative
                       // step - the launches the driver queued ahead of this decision (gated on this flag) are the right ones
    CS_WILL_END,       // with CS_EXPECT = RESOLVED: the loop-top tests after the re-elimination will end the solve (no step follows)
    CS_RADIUS, CS_DECREASE, CS_RADIUS_SPEC, CS_COST, CS_GMAX, CS_GMAX_PRIV, CS_INITIAL_COST, CS_REL_LAST, CS_REL_PREV,
    CS_ITER, CS_SUCCESSFUL, CS_INVALID, CS_VALID, CS_PLAIN_NEXT, CS_NFAIL, CS_M,
    CS_STEP2_SH, CS_XNORM2_SH, CS_GD_SH, CS_DHD_SH, CS_DMAX,     // the shared blocks' share of the pending step's statistics
    CS_CAND_COST, CS_REL, CS_MODEL_CHANGE, CS_SLOPE0, CS_SPECULATED,
    CS_P_COST, CS_P_RADIUS, CS_P_GMAX, CS_P_ITER,                // the state a decision was taken in (verbose output)
    CS_N_SPEC, CS_N_HITS, CS_N_MISSES, CS_N_REJECTED, CS_N_WASTED,
    CS_LS_A, CS_LS_COST, CS_LS_STEP2, CS_LS_XNORM2, CS_LS_STEP2_SH, CS_LS_XNORM2_SH,  // host -> controller: result of a line search
    CS_PROF,           // [CTL_NPROF] time per phase of the controller kernel, 10 ns ticks summed over the solve (CtlProf)
    CS_COUNT = 80
};
constexpr int CTL_NPROF = 12;
enum CtlProf : int { CP_ENTRY = 0, CP_ADOPT, CP_GMAX, CP_ASSEMBLE, CP_FACTOR, CP_BACKSOLVE, CP_PLUS, CP_MODEL, CP_DECIDE, CP_PUBLISH };
static_assert(CS_PROF + CTL_NPROF <= CS_COUNT, "control record size");

enum CtlMode : int { CTL_NONE = 0, CTL_NEW = 1, CTL_RESOLVED = 2, CTL_STEP = 3, CTL_LS_DONE = 4, CTL_LINE_SEARCH = 5 };
enum CtlMsg : int { CM_GRADIENT = 0, CM_MAX_ITER, CM_MIN_RADIUS, CM_INVALID_STEPS, CM_PARAMETER, CM_FUNCTION, CM_NONE };
inline const char* ctl_message(int m) {
    static const char* const k[] = {"Gradient tolerance reached.", "Maximum number of iterations reached.",
                                    "Minimum trust region radius reached.", "Number of consecutive invalid steps more than max.",
                                    "Parameter tolerance reached.", "Function tolerance reached.", ""};
    return k[m < 0 || m > CM_NONE ? CM_NONE : m];
}

constexpr int CTL_NB = 8;  // panel width of the blocked factorisation
CBA_HD int ctl_padded(int n) { return (n + CTL_NB - 1) / CTL_NB * CTL_NB; }  // rows of the padded reduced matrix
CBA_HD int ctl_lda(int n) { return ctl_padded(n) | 1; }                        // its row stride (odd: conflict-free column walks in LDS)
constexpr int CTL_LDS_MAX_N = 128;  // the GPU controller keeps the reduced matrix in LDS up to this size (129 x 129 doubles = 133 KB)

// Everything the controller touches, as raw pointers (device memory on the GPU, host vectors in the CPU test build).
struct CtlView {
    // structure (structure.hpp)
    int n, n_cams, PI, PL, NH, NACC, PC, sh_base, chain, n_ranks;
    // pack layout (lm_core.hpp PackLayout)
    int64_t off_stats, off_cam, off_cost, off_nfail, off_S, off_g, off_gmax;
    // options
    double eps;
    int max_iterations, constrained, line_search, speculate, intr_var, target_var;
    // shared parameter packs [intr | cam poses | target pose | (trial only) shared step]
    int64_t pk_cam, pk_target, pk_delta;
    double *x_cur, *x_trial, *x_tmp;
    // state
    double* scal;          // [CS_COUNT]
    const double* pack;    // the all-reduced pack of the exchange this invocation follows
    double* camc;          // [n_cams][NACC] the CURRENT linearisation's per-camera sums: H_cc (block-sparse) is read from them in place
    double *gc, *scale2, *hdiag;  // g_c, Jacobi scale^2, diag(H_cc)  [n]
    const int *colcam, *collc;    // shared column -> (camera or -1 for the bundle chain's target pose, local tangent column)
    int8_t* eff;
    const int8_t *active, *cam_var;
    int* idx;              // effective columns, compact
    // work
    double* A;             // (M + 1) x lda, M = m rounded up to CTL_NB, lda >= M: reduced matrix -> the panels of its lower factor;
                           // row M carries the right-hand side
    int lda;
    double *rdiag, *xs, *Ld;  // reciprocal pivots, solution in compact order [n rounded up to CTL_NB], factors of the diagonal blocks [n / NB][NB * NB]
    int* okflag;           // the factorisation's pivot test, from the thread that took it to the team
    double* lmp;           // [radius of the next elimination, init_scale]: read by the per-view elimination kernels
    double* rec;           // where the control record is published (page-locked host memory on the GPU)
};

struct SerialTeam {
    CBA_HD int tid() const { return 0; }
    CBA_HD int size() const { return 1; }
    CBA_HD void sync() const {}
    CBA_HD double sum(double v) const { return v; }
    CBA_HD double max(double v) const { return v; }
    // number of threads before this one (in thread order) whose flag is set, and the team's total
    CBA_HD int count_before(bool flag, int* total) const { *total = flag ? 1 : 0; return 0; }
    CBA_HD void tick(const CtlView&, int) const {}
    CBA_HD void mark(int) const {}
    CBA_HD void publish(const CtlView& V) const {
        for (int k = 1; k < CS_COUNT; ++k) V.rec[k] = V.scal[k];
        V.rec[CS_SEQ] = V.scal[CS_SEQ];
    }
};

CBA_HD bool ctl_finite(double x) { return fabs(x) <= 1.7976931348623157e308; }

// global shared column i -> (camera, local column); camera -1 = the bundle chain's target pose (structure.hpp shared_col, inverted)
CBA_HD void ctl_decode(const CtlView& V, int i, int* cam, int* lc) {
    if (V.chain == CH_BUNDLE) {
        if (i < 6) { *cam = -1; *lc = i; return; }
        *cam = (i - 6) / V.PC;
        *lc = 6 + (i - 6) - *cam * V.PC;
        return;
    }
    *cam = i / V.PC;
    *lc = 6 + i - *cam * V.PC;
}
template <class VW>
CBA_HD int ctl_intr_base(const VW& V, int c) { return V.chain == CH_INTRINSIC ? 0 : V.sh_base + c * V.PC + 6; }
template <class VW>
CBA_HD int ctl_campose_base(const VW& V, int c) { return V.sh_base + c * V.PC; }
// packed upper triangle of the n x n Schur term as it travels in the pack
CBA_HD int64_t ctl_sidx(int n, int i, int j) { return i <= j ? static_cast<int64_t>(i) * n - static_cast<int64_t>(i) * (i - 1) / 2 + (j - i)
                                                              : static_cast<int64_t>(j) * n - static_cast<int64_t>(j) * (j - 1) / 2 + (i - j); }

// Plus on the shared blocks with the fx, fy >= 0 projection (LMDriver::shared_plus): xo = Plus(x, delta); the team's totals of
// |xo - x|^2 and |x|^2 over the variable blocks.  xo is complete for every thread on return.
// (used for the gradient norm and for the trial point: NOT inlined, so that the quaternion update's sin / cos exist once in the
// code.  Everything it needs travels BY VALUE: a reference to the view or to the team would force the caller to keep them in
// memory - on the GPU that is scratch, and every pointer read back from scratch has lost its address space: the whole controller
// then runs on flat loads and stores instead of LDS / global ones.)
struct CtlPlusView {
    int n_cams, PI, PC, sh_base, chain, intr_var, target_var;
    int64_t pk_cam, pk_target;
    const int8_t* cam_var;
};
struct CtlNorms {
    double step2, xnorm2;
};
CBA_HD CtlPlusView ctl_plus_view(const CtlView& V) {
    return CtlPlusView{V.n_cams, V.PI, V.PC, V.sh_base, V.chain, V.intr_var, V.target_var, V.pk_cam, V.pk_target, V.cam_var};
}
template <class TM>
CBA_NOINLINE CtlNorms ctl_plus(TM tm, CtlPlusView V, const double* x, const double* delta, double* xo) {
    double s2 = 0.0, x2 = 0.0;
    const int PI = V.PI;
    for (int i = tm.tid(); i < V.n_cams * PI; i += tm.size()) {
        const int c = i / PI, k = i - c * PI;
        const double p0 = x[i];
        double p = p0;
        if (V.intr_var) {
            p += delta[ctl_intr_base(V, c) + k];
            if (k < 2) p = fmax(p, 0.0);
            s2 += (p - p0) * (p - p0);
            x2 += p0 * p0;
        }
        xo[i] = p;
    }
    if (V.chain != CH_INTRINSIC)
        for (int c = tm.tid(); c < V.n_cams; c += tm.size()) {
            const double* q = x + V.pk_cam + 7 * c;
            double* o = xo + V.pk_cam + 7 * c;
            for (int k = 0; k < 7; ++k) o[k] = q[k];
            if (V.cam_var[c]) {
                const int pb = ctl_campose_base(V, c);
                quat_plus(q, delta + pb, o);
                for (int k = 0; k < 3; ++k) o[4 + k] = q[4 + k] + delta[pb + 3 + k];
                for (int k = 0; k < 7; ++k) { s2 += (o[k] - q[k]) * (o[k] - q[k]); x2 += q[k] * q[k]; }
            }
        }
    if (V.chain == CH_BUNDLE && tm.tid() == tm.size() - 1) {
        const double* q = x + V.pk_target;
        double* o = xo + V.pk_target;
        for (int k = 0; k < 7; ++k) o[k] = q[k];
        if (V.target_var) {
            quat_plus(q, delta, o);
            for (int k = 0; k < 3; ++k) o[4 + k] = q[4 + k] + delta[3 + k];
            for (int k = 0; k < 7; ++k) { s2 += (o[k] - q[k]) * (o[k] - q[k]); x2 += q[k] * q[k]; }
        }
    }
    CtlNorms r;
    r.step2 = tm.sum(s2);
    r.xnorm2 = tm.sum(x2);
    return r;
}

// the shared blocks' share of Ceres' gradient max-norm (LMDriver::shared_gmax); clobbers xs and x_tmp
template <class TM>
CBA_HD double ctl_shared_gmax(TM& tm, const CtlView& V) {
    double m = 0.0;
    if (!V.constrained) {
        for (int i = tm.tid(); i < V.n; i += tm.size())
            if (V.eff[i]) m = fmax(m, fabs(V.gc[i]));
        return tm.max(m);
    }
    if (V.chain == CH_INTRINSIC) {
        // the shared blocks are intrinsics only (Euclidean, fx, fy >= 0): |Plus(x, -g) - x| entry by entry, in ctl_plus' own arithmetic,
        // without the trip through x_tmp in global memory
        if (V.intr_var)
            for (int i = tm.tid(); i < V.n_cams * V.PI; i += tm.size()) {
                const int c = i / V.PI, k = i - c * V.PI, col = ctl_intr_base(V, c) + k;
                const double p0 = V.x_cur[i];
                double p = p0 + (V.eff[col] ? -V.gc[col] : 0.0);
                if (k < 2) p = fmax(p, 0.0);
                m = fmax(m, fabs(p - p0));
            }
        return tm.max(m);
    }
    for (int i = tm.tid(); i < V.n; i += tm.size()) V.xs[i] = V.eff[i] ? -V.gc[i] : 0.0;
    tm.sync();
    (void)ctl_plus(tm, ctl_plus_view(V), V.x_cur, V.xs, V.x_tmp);
    for (int i = tm.tid(); i < V.n_cams * V.PI; i += tm.size()) m = fmax(m, fabs(V.x_tmp[i] - V.x_cur[i]));
    if (V.chain != CH_INTRINSIC)
        for (int i = tm.tid(); i < 7 * V.n_cams; i += tm.size()) m = fmax(m, fabs(V.x_tmp[V.pk_cam + i] - V.x_cur[V.pk_cam + i]));
    if (V.chain == CH_BUNDLE)
        for (int i = tm.tid(); i < 7; i += tm.size()) m = fmax(m, fabs(V.x_tmp[V.pk_target + i] - V.x_cur[V.pk_target + i]));
    return tm.max(m);
}

// The pack holds an all-reduced linearisation (a new system, or an accepted speculative step): make it the current one
// (LMDriver::adopt_system): H_cc, g_c from the per-camera sums, effective columns, Jacobi scale (first system), cost, gradient norm.
// local tangent column lc of a block of camera c -> global shared column (-1: a private pose column); structure.hpp shared_col
CBA_HD int ctl_shared_col(const CtlView& V, int c, int lc) {
    if (V.chain == CH_BUNDLE) return lc < 6 ? lc : V.sh_base + c * V.PC + (lc - 6);
    return lc < 6 ? -1 : c * V.PC + (lc - 6);
}

// H_cc[i][j] from per-camera sums: zero unless both columns belong to one camera (or to the target pose, which every camera's
// blocks see: summed in camera order)
CBA_HD double ctl_hcc(const CtlView& V, const double* cam, int i, int j) {
    const int ci = V.colcam[i], cj = V.colcam[j];
    const int h = hidx_sym(V.PL, V.collc[i], V.collc[j]);
    if (ci < 0 && cj < 0) {
        double v = 0.0;
        for (int c = 0; c < V.n_cams; ++c) v += cam[static_cast<int64_t>(c) * V.NACC + h];
        return v;
    }
    if (ci < 0 || cj < 0 || ci == cj) return cam[static_cast<int64_t>(ci < 0 ? cj : ci) * V.NACC + h];
    return 0.0;
}
CBA_HD double ctl_gc(const CtlView& V, const double* cam, int i) {
    const int ci = V.colcam[i], li = V.collc[i];
    if (ci < 0) {
        double v = 0.0;
        for (int c = 0; c < V.n_cams; ++c) v += cam[static_cast<int64_t>(c) * V.NACC + V.NH + li];
        return v;
    }
    return cam[static_cast<int64_t>(ci) * V.NACC + V.NH + li];
}

template <class TM>
CBA_HD void ctl_adopt(TM& tm, const CtlView& V, bool init_scale) {
    const int n = V.n;
    const double* cam_acc = V.pack + V.off_cam;
    // the per-camera sums become the controller's own copy: a speculative step that is rejected later leaves the pack holding the
    // trial point's system, and the current H_cc must survive it (H_cc is never formed: n_cams * NACC numbers instead of n^2)
    for (int e = tm.tid(); e < V.n_cams * V.NACC; e += tm.size()) V.camc[e] = cam_acc[e];
    for (int i = tm.tid(); i < n; i += tm.size()) {
        const double hii = ctl_hcc(V, cam_acc, i, i);
        V.hdiag[i] = hii;
        V.gc[i] = ctl_gc(V, cam_acc, i);
        V.eff[i] = V.active[i] && hii != 0.0;  // columns nobody observes behave like constant blocks
        if (init_scale) {
            const double sc = 1.0 / (1.0 + sqrt(hii));
            V.scale2[i] = sc * sc;
        }
    }
    tm.sync();
    int m = 0;  // the effective columns in order: a prefix count over the team per chunk of columns (not one thread walking all n)
    for (int base = 0; base < n; base += tm.size()) {
        const int i = base + tm.tid();
        const bool on = i < n && V.eff[i];
        int tot;
        const int before = tm.count_before(on, &tot);
        if (on) V.idx[m + before] = i;
        m += tot;
    }
    if (tm.tid() == 0) {
        V.scal[CS_M] = m;
        V.scal[CS_COST] = V.pack[V.off_cost];
        V.scal[CS_NFAIL] = floor(V.pack[V.off_nfail] + 0.5);
        double gm = 0.0;
        for (int r = 0; r < V.n_ranks; ++r) gm = fmax(gm, V.pack[V.off_gmax + r]);  // max over ranks through one-hot slots
        V.scal[CS_GMAX_PRIV] = gm;
    }
    tm.sync();
    tm.tick(V, CP_ADOPT);
    const double gsh = ctl_shared_gmax(tm, V);
    if (tm.tid() == 0) V.scal[CS_GMAX] = fmax(V.scal[CS_GMAX_PRIV], gsh);
    tm.sync();
    tm.tick(V, CP_GMAX);
}

// ---- the reduced solve ---------------------------------------------------------------------------------------------------
// sqrt(d) and 1 / sqrt(d) of a pivot.  On the GPU: v_rsq_f64 and two coupled Newton steps (a dozen dependent instructions; the
// library sqrt followed by a division is ~40, and the pivots are THE dependent chain of the factorisation).
CBA_HD void ctl_sqrt_rsqrt(double d, double* sq, double* rsq) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double e = __builtin_fma(-g, g, d);
    g = __builtin_fma(e, h, g);
    *sq = g;
    *rsq = 2.0 * h;
#else
    const double s = sqrt(d);
    *sq = s;
    *rsq = 1.0 / s;
#endif
}

// Blocked right-looking Cholesky of the M x M matrix in A (lower triangle, row-major, stride lda; M a multiple of CTL_NB: the
// caller pads with identity rows) with panels of CTL_NB columns.  Row M of A carries the right-hand side along as one more panel
// row, so that it leaves as L^-1 b (the forward substitution costs no extra pass).  The factor of the p-th diagonal block goes to
// Ld[p] (CTL_NB x CTL_NB, lower; its upper entries are scratch), the panel entries below it stay in A; rdiag[k] = 1 / L[k][k].
// Per panel: every thread that owns a row i >= k0 reads the diagonal block, factorises it in registers (each for itself: a
// latency chain, not work, and nobody waits for a broadcast) and forward-substitutes its own row against it - a row of the
// diagonal block comes out as its row of L, by the same arithmetic; barrier; the team updates the trailing matrix in 4 x 4 tiles
// whose rows and columns are INTERLEAVED (tile (ti, tj) = rows ti + a nt, columns tj + b nt: neighbouring threads touch
// neighbouring rows, conflict-free in LDS with an odd stride; a tile with ti > tj holds 16 distinct unordered pairs, a diagonal
// one writes its mirrored half into the unused upper triangle), other threads the right-hand-side row; barrier.
// Written without guards (padding instead) and with 32-bit offsets: for ONE workgroup the cost is the instruction count along
// the chain panel -> barrier -> tile -> barrier, measured at 4-5 cycles per instruction (tools/probe/ctl_probe.hip).
// *okflag / the return value: every pivot positive and finite.
template <class TM>
CBA_HD bool ctl_cholesky(TM& tm, double* A, int lda, int M, double* Ld, double* rdiag, int* okflag) {
    constexpr int NB = CTL_NB;
    for (int k0 = 0, p = 0; k0 < M; k0 += NB, ++p) {
        tm.mark(0);
        const int b0 = k0 + NB;
        if (k0 + tm.tid() <= M) {  // this thread owns a panel row (thread 0 always does)
            double L[NB][NB], inv[NB];
            const double* Dg = A + k0 * lda + k0;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) L[i][j] = Dg[i * lda + j];
            bool ok = true;
            tm.mark(1);
            // right-looking inside the block: once column j is final, every later entry takes its update at once (independent
            // multiply-adds; the left-looking form is one dependent chain per entry) - the same subtractions in the same order
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const double d = L[j][j];
                if (!(d > 0.0) || !ctl_finite(d)) ok = false;
                double sq, r;
                ctl_sqrt_rsqrt(d, &sq, &r);
                inv[j] = r;
                L[j][j] = sq;
#pragma unroll
                for (int i = j + 1; i < NB; ++i) L[i][j] *= r;
#pragma unroll
                for (int i = j + 1; i < NB; ++i)
#pragma unroll
                    for (int c = j + 1; c <= i; ++c) L[i][c] -= L[i][j] * L[c][j];
            }
            if (tm.tid() == 0) {
                *okflag = ok ? 1 : 0;
#pragma unroll
                for (int c = 0; c < NB; ++c) rdiag[k0 + c] = inv[c];
            }
            tm.mark(2);
            for (int i = k0 + tm.tid(); i <= M; i += tm.size()) {
                double* row = A + i * lda + k0;
                double x[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) x[c] = row[c];
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    x[c] *= inv[c];
#pragma unroll
                    for (int k = c + 1; k < NB; ++k) x[k] -= x[c] * L[k][c];
                }
                double* dst = i < b0 ? Ld + (p * NB + (i - k0)) * NB : row;  // a row of the diagonal block is its row of L
#pragma unroll
                for (int c = 0; c < NB; ++c) dst[c] = x[c];
            }
        }
        tm.mark(3);
        tm.sync();
        tm.mark(4);
        if (!*okflag) return false;
        const int nr = M - b0;  // trailing rows b0 .. M - 1 (a multiple of NB), then the right-hand-side row M
        if (nr > 0) {
            const int nt = nr >> 2;
            const int ntiles = nt * (nt + 1) / 2;
            for (int t = tm.tid(); t < ntiles + nr; t += tm.size()) {
                if (t < ntiles) {
                    int ti = static_cast<int>((sqrtf(8.0f * static_cast<float>(t) + 1.0f) - 1.0f) * 0.5f);
                    ti += ((ti + 1) * (ti + 2) / 2 <= t) ? 1 : 0;
                    ti -= (ti * (ti + 1) / 2 > t) ? 1 : 0;
                    const int tj = t - ti * (ti + 1) / 2;
                    const int ri = (b0 + ti) * lda, rj = (b0 + tj) * lda, st = nt * lda;  // row offsets of a = 0 and the row step
                    double pi[4][NB], pj[4][NB], av[4][4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < NB; ++c) {
                            pi[a][c] = A[ri + a * st + k0 + c];
                            pj[a][c] = A[rj + a * st + k0 + c];
                        }
                    // element (a, b): rows i = b0 + ti + a nt, j = b0 + tj + b nt; it lives at A[max][min] - for a diagonal tile the
                    // mirrored half (a < b) goes to A[i][j] in the unused upper triangle instead
                    const bool diag = ti == tj;
                    int off[4][4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int lo = ri + a * st + (b0 + tj + b * nt);  // A[i][j]
                            const int up = rj + b * st + (b0 + ti + a * nt);  // A[j][i]
                            off[a][b] = (a >= b || diag) ? lo : up;             // (ti > tj, a >= b: i > j; a < b: i < j unless... see below)
                        }
                    // for ti > tj and a < b the pair is ordered by its rows: i = ti + a nt < j = tj + b nt  <=>  a < b (ti, tj < nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) av[a][b] = A[off[a][b]];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            double sacc = av[a][b];
#pragma unroll
                            for (int c = 0; c < NB; ++c) sacc -= pi[a][c] * pj[b][c];
                            A[off[a][b]] = sacc;
                        }
                } else {  // the right-hand-side row: y[j] -= sum_c y[k0 + c] A[j][k0 + c]
                    const int j = b0 + (t - ntiles);
                    const double* yk = A + M * lda + k0;
                    const double* rj = A + j * lda + k0;
                    double sacc = A[M * lda + j];
#pragma unroll
                    for (int c = 0; c < NB; ++c) sacc -= yk[c] * rj[c];
                    A[M * lda + j] = sacc;
                }
            }
        }
        tm.mark(5);
        tm.sync();
        tm.mark(6);
    }
    return true;
}

// x = L^-T y by panels from the last one: y = row M of A (from ctl_cholesky), x -> xs[0 .. M).  Per panel the threads that own
// an earlier row (or one of the panel's) solve the small triangular system, each for itself, then subtract the panel's
// contribution from their row.
template <class TM>
CBA_HD void ctl_backsolve(TM& tm, double* A, int lda, int M, const double* Ld, const double* rdiag, double* xs) {
    constexpr int NB = CTL_NB;
    double* y = A + M * lda;
    for (int k0 = M - NB, p = M / NB - 1; k0 >= 0; k0 -= NB, --p) {
        const bool mine = tm.tid() < k0 + NB;
        double x[NB];
        if (mine) {
            // the panel's triangle, right-hand side and reciprocal pivots first (independent loads), then the dependent chain
            double Lt[NB][NB], yv[NB], rd[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                yv[c] = y[k0 + c];
                rd[c] = rdiag[k0 + c];
#pragma unroll
                for (int k = c + 1; k < NB; ++k) Lt[k][c] = Ld[(p * NB + k) * NB + c];
            }
#pragma unroll
            for (int c = NB - 1; c >= 0; --c) {
                double sacc = yv[c];
#pragma unroll
                for (int k = c + 1; k < NB; ++k) sacc -= Lt[k][c] * x[k];
                x[c] = sacc * rd[c];
            }
        }
        tm.sync();  // every thread has read y[k0 .. k0 + NB) before anyone overwrites it
        if (mine)
            for (int r = tm.tid(); r < k0 + NB; r += tm.size()) {
                if (r >= k0) {
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        if (r - k0 == c) xs[r] = x[c];
                    continue;
                }
                double sacc = y[r], lv[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) lv[c] = A[(k0 + c) * lda + r];
#pragma unroll
                for (int c = 0; c < NB; ++c) sacc -= lv[c] * x[c];
                y[r] = sacc;
            }
        tm.sync();
    }
}

// (H_cc + D - S_schur) delta = -(g_c - g_schur) on the effective columns (LMDriver::solve_reduced); delta -> x_trial[pk_delta ..),
// zero on the other columns.  Returns false (uniformly) when the system cannot be solved.
template <class TM>
CBA_HD bool ctl_solve(TM& tm, const CtlView& V, double radius) {
    const int n = V.n, m = static_cast<int>(V.scal[CS_M]);
    double* delta = V.x_trial + V.pk_delta;
    for (int i = tm.tid(); i < n; i += tm.size()) delta[i] = 0.0;
    if (V.scal[CS_NFAIL] > 0.0) { tm.sync(); return false; }
    if (m == 0) { tm.sync(); return true; }
    const double* S = V.pack + V.off_S;
    const double* gs = V.pack + V.off_g;
    const int M = (m + CTL_NB - 1) / CTL_NB * CTL_NB;  // padded with identity rows: every panel is full, no guards in the factorisation
    // the lower triangle element by element (e -> (r, c) by the triangular root), then the right-hand-side row: independent loads,
    // unrolled so that many are in flight at once (the pack and the camera sums sit in L2: ~1 us per dependent round trip)
    {
        const int tot = M * (M + 1) / 2;
#pragma unroll 8
        for (int e0 = tm.tid(); e0 < tot; e0 += tm.size()) {
            int r = static_cast<int>((sqrtf(8.0f * static_cast<float>(e0) + 1.0f) - 1.0f) * 0.5f);
            r += ((r + 1) * (r + 2) / 2 <= e0) ? 1 : 0;
            r -= (r * (r + 1) / 2 > e0) ? 1 : 0;
            const int c = e0 - r * (r + 1) / 2;
            double val = r == c ? 1.0 : 0.0;  // the identity padding
            if (r < m) {
                const int i = V.idx[r], j = V.idx[c];
                val = ctl_hcc(V, V.camc, i, j) - S[ctl_sidx(n, i, j)];
                if (r == c) val += lm_diag(V.hdiag[i], V.scale2[i], radius);
            }
            V.A[r * V.lda + c] = val;
        }
        double* Arow = V.A + M * V.lda;
        for (int c = tm.tid(); c < M; c += tm.size()) {
            const int i = V.idx[c < m ? c : 0];
            const double b = -(V.gc[i] - gs[i]);
            Arow[c] = c < m ? b : 0.0;
        }
    }
    tm.sync();
    tm.tick(V, CP_ASSEMBLE);
    if (!ctl_cholesky(tm, V.A, V.lda, M, V.Ld, V.rdiag, V.okflag)) return false;
    tm.tick(V, CP_FACTOR);
    ctl_backsolve(tm, V.A, V.lda, M, V.Ld, V.rdiag, V.xs);
    tm.tick(V, CP_BACKSOLVE);
    double bad = 0.0;
    for (int r = tm.tid(); r < m; r += tm.size())
        if (!ctl_finite(V.xs[r])) bad = 1.0;
    if (tm.max(bad) > 0.0) return false;
    for (int r = tm.tid(); r < m; r += tm.size()) delta[V.idx[r]] = V.xs[r];
    tm.sync();
    return true;
}

// ---- the iteration --------------------------------------------------------------------------------------------------------
constexpr double CTL_MIN_RADIUS = 1e-32, CTL_MAX_RADIUS = 1e16, CTL_MIN_REL_DECREASE = 1e-3;

template <class TM>
CBA_HD void ctl_end(TM& tm, const CtlView& V, int term, int msg) {
    if (tm.tid() == 0) {
        V.scal[CS_TERM] = term;
        V.scal[CS_MSG] = msg;
        V.scal[CS_EXPECT] = CTL_NONE;
    }
}

// an invalid step: the linear solve failed or the model did not decrease (halve the radius, eliminate again)
template <class TM>
CBA_HD void ctl_invalid_step(TM& tm, const CtlView& V) {
    if (tm.tid() == 0) {
        const double inv = V.scal[CS_INVALID] + 1.0;
        V.scal[CS_INVALID] = inv;
        if (inv >= 5.0) {
            V.scal[CS_TERM] = CBA_TERM_FAILURE;
            V.scal[CS_MSG] = CM_INVALID_STEPS;
            V.scal[CS_EXPECT] = CTL_NONE;
        } else {
            V.scal[CS_RADIUS] *= 0.5;
            V.scal[CS_PLAIN_NEXT] = 1.0;
            V.scal[CS_EXPECT] = CTL_RESOLVED;
            V.lmp[0] = V.scal[CS_RADIUS];
            V.lmp[1] = 0.0;
        }
    }
}

CBA_HD bool ctl_expect_convergence(const CtlView& V) {
    const double rl = V.scal[CS_REL_LAST], rp = V.scal[CS_REL_PREV];
    return rp > 0.0 && rl > 0.0 && rl * fmin(1.0, rl / rp) <= 4.0 * V.eps;
}
CBA_HD bool ctl_next_step_speculative(const CtlView& V) {
    return V.speculate && V.scal[CS_PLAIN_NEXT] == 0.0 && !ctl_expect_convergence(V);
}
// the loop-top tests of the iteration, on the current state; 0 = go on
CBA_HD int ctl_top_tests(const CtlView& V, int* msg) {
    if (V.scal[CS_ITER] >= V.max_iterations) { *msg = CM_MAX_ITER; return CBA_TERM_NO_CONVERGENCE; }
    if (V.scal[CS_GMAX] <= V.eps) { *msg = CM_GRADIENT; return CBA_TERM_CONVERGENCE; }
    if (V.scal[CS_RADIUS] <= CTL_MIN_RADIUS) { *msg = CM_MIN_RADIUS; return CBA_TERM_CONVERGENCE; }
    return 0;
}

// Top of an iteration on the current system: the termination tests, the reduced solve, the trial point of the shared blocks and
// their share of the step's statistics.  Leaves CS_EXPECT = CTL_STEP (with the kind of evaluation the step gets), or a
// re-elimination after an unsolvable system, or the end of the solve.
template <class TM>
CBA_HD void ctl_iterate(TM& tm, const CtlView& V) {
    int msg = CM_NONE;
    const int t = ctl_top_tests(V, &msg);  // uniform: scalars are read after a barrier
    if (t) { ctl_end(tm, V, t, msg); return; }
    tm.sync();
    if (tm.tid() == 0) V.scal[CS_ITER] += 1.0;
    const double radius = V.scal[CS_RADIUS];
    const bool valid = ctl_solve(tm, V, radius);
    if (!valid) {
        tm.sync();
        ctl_invalid_step(tm, V);
        return;
    }
    const int n = V.n;
    const double* delta = V.x_trial + V.pk_delta;
    double s2, x2;
    const CtlNorms nrm = ctl_plus(tm, ctl_plus_view(V), V.x_cur, delta, V.x_trial);
    s2 = nrm.step2; x2 = nrm.xnorm2;
    tm.tick(V, CP_PLUS);
    // the shared-shared part of Ceres' model cost change -g^T d - 1/2 d^T H d; the views contribute theirs with the step's exchange
    // (H_cc is symmetric: thread (group r, column j) adds the rows r, r + G, ... of column j, so neighbours read neighbours)
    double gd = 0.0, dHd = 0.0, dmax = 0.0;
    for (int i = tm.tid(); i < n; i += tm.size()) {
        const double di = delta[i];
        dmax = fmax(dmax, fabs(di));
        gd += V.gc[i] * di;
    }
    {   // d^T H_cc d = sum over the cameras' blocks of d_c^T H_c d_c (d in the block's local columns; private pose columns excluded)
        const int PL = V.PL, PL2 = PL * PL;
#pragma unroll 4
        for (int e = tm.tid(); e < V.n_cams * PL2; e += tm.size()) {
            const int c = e / PL2, rem = e - c * PL2, li = rem / PL, lj = rem - li * PL;
            const int gi = ctl_shared_col(V, c, li), gj = ctl_shared_col(V, c, lj);
            if (gi < 0 || gj < 0) continue;
            dHd += delta[gi] * V.camc[static_cast<int64_t>(c) * V.NACC + hidx_sym(PL, li, lj)] * delta[gj];
        }
    }
    gd = tm.sum(gd);
    dHd = tm.sum(dHd);
    dmax = tm.max(dmax);
    tm.tick(V, CP_MODEL);
    if (tm.tid() == 0) {
        V.scal[CS_VALID] = 1.0;
        V.scal[CS_STEP2_SH] = s2; V.scal[CS_XNORM2_SH] = x2; V.scal[CS_GD_SH] = gd; V.scal[CS_DHD_SH] = dHd; V.scal[CS_DMAX] = dmax;
        // the radius a gain ratio >= 0.937 leads to, in the arithmetic of the update (radius / (1/3) differs from 3 * radius by one
        // ulp for a quarter of all doubles: the comparison after the step is exact)
        const double rspec = fmin(CTL_MAX_RADIUS, radius / (1.0 / 3.0));
        V.scal[CS_RADIUS_SPEC] = rspec;
        const bool spec = ctl_next_step_speculative(V);
        V.scal[CS_STEP_SPEC] = spec ? 1.0 : 0.0;
        V.scal[CS_EXPECT] = CTL_STEP;
        V.lmp[0] = spec ? rspec : radius;  // the elimination at the trial point is made with the predicted radius
        V.lmp[1] = 0.0;
    }
}

// The decision on a trial step whose statistics (the views' share) arrived with the last exchange; cand / step2 / xnorm2 are the
// totals over the whole state vector.  LMDriver::solve_host from "model_cost_change" to the radius update.  Returns true when
// the step was accepted after a SPECULATIVE evaluation: the pack then holds the next system, which the caller adopts (the big
// pieces - adopt, iterate - appear once in ctl_run: inlined at every use the kernel was 145 KB of code for a 64 KB
// instruction cache, and ran 10x slower than its dependent chains allow).
template <class TM>
CBA_HD bool ctl_decide(TM& tm, const CtlView& V, bool speculated, double model_change, double cand, double step2, double xnorm2) {
    const double eps = V.eps;
    if (tm.tid() == 0) {
        V.scal[CS_INVALID] = 0.0;
        V.scal[CS_P_COST] = V.scal[CS_COST]; V.scal[CS_P_RADIUS] = V.scal[CS_RADIUS]; V.scal[CS_P_GMAX] = V.scal[CS_GMAX];
        V.scal[CS_P_ITER] = V.scal[CS_ITER];
    }
    if (!ctl_finite(cand)) cand = 1.7976931348623157e308;
    const double cost = V.scal[CS_COST];
    const double step_norm = sqrt(step2), x_norm = sqrt(xnorm2);
    const double cost_change = cost - cand;
    const double rel = cost_change / model_change;
    tm.sync();
    if (tm.tid() == 0) { V.scal[CS_CAND_COST] = cand; V.scal[CS_REL] = rel; V.scal[CS_MODEL_CHANGE] = model_change; V.scal[CS_SPECULATED] = speculated ? 1.0 : 0.0; }
    if (step_norm <= eps * (x_norm + eps)) { ctl_end(tm, V, CBA_TERM_CONVERGENCE, CM_PARAMETER); return false; }
    if (fabs(cost_change) <= eps * cost) { ctl_end(tm, V, CBA_TERM_CONVERGENCE, CM_FUNCTION); return false; }
    if (rel > CTL_MIN_REL_DECREASE) {
        // accept: the shared blocks here; the host queues the exchange of the private poses (and block sums) it is told about
        for (int64_t i = tm.tid(); i < V.pk_delta; i += tm.size()) V.x_cur[i] = V.x_trial[i];
        const double radius_old = V.scal[CS_RADIUS];
        const double t3 = 2.0 * rel - 1.0;
        const double radius = fmin(CTL_MAX_RADIUS, radius_old / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3));
        tm.sync();
        if (tm.tid() == 0) {
            V.scal[CS_SUCCESSFUL] += 1.0;
            V.scal[CS_REL_PREV] = V.scal[CS_REL_LAST];
            V.scal[CS_REL_LAST] = fabs(cost_change) / cost;
            V.scal[CS_RADIUS] = radius;
            V.scal[CS_DECREASE] = 2.0;
            V.scal[CS_PLAIN_NEXT] = 0.0;
            V.scal[CS_ACCEPT] = speculated ? 2.0 : 1.0;
            if (!speculated) {
                V.scal[CS_EXPECT] = CTL_NEW;
                V.lmp[0] = radius;
                V.lmp[1] = 0.0;
            }
        }
        tm.sync();
        return speculated;
    }
    if (tm.tid() == 0) {
        V.scal[CS_N_REJECTED] += 1.0;
        const double radius = V.scal[CS_RADIUS] / V.scal[CS_DECREASE];
        V.scal[CS_RADIUS] = radius;
        V.scal[CS_DECREASE] *= 2.0;
        V.scal[CS_PLAIN_NEXT] = 1.0;
        V.scal[CS_EXPECT] = CTL_RESOLVED;
        V.lmp[0] = radius;
        V.lmp[1] = 0.0;
    }
    return false;
}

// One invocation of the controller, right behind an exchange:
//   CTL_NEW       the pack holds a new linearisation at the current point (flag != 0: the start point)
//   CTL_RESOLVED  the pack holds a re-elimination of the current linearisation [nfail | S | g]
//   CTL_STEP      the pack holds the statistics of the pending trial step (flag != 0: and the system linearised there)
//   CTL_LS_DONE   the host ran Ceres' line search on the pending step and left its result in CS_LS_*
// An invocation whose mode is not the one the controller expects (the host queued it ahead of a decision that turned out
// otherwise) changes nothing.  Publishes the control record at the end.
template <class TM>
CBA_HD void ctl_run(TM& tm, const CtlView& V, int mode, int flag) {
    const bool in_turn = V.scal[CS_TERM] < 0.0 && static_cast<int>(V.scal[CS_EXPECT]) == mode &&
                         (mode != CTL_STEP || (V.scal[CS_STEP_SPEC] != 0.0) == (flag != 0));
    tm.sync();
    if (tm.tid() == 0) {
        V.scal[CS_SEQ] += 1.0;
        V.scal[CS_GO] = 0.0;
        if (in_turn) { V.scal[CS_ACCEPT] = 0.0; V.scal[CS_WILL_END] = 0.0; }
        else V.scal[CS_N_WASTED] += 1.0;
    }
    tm.sync();
    tm.tick(V, CP_ENTRY);
    // what this invocation does, in the order: [decide] -> [adopt the pack's system] -> [start the next iteration]
    enum { AFTER_START = 1, AFTER_NEW = 2, AFTER_STEP = 3 };
    int adopt = 0;
    bool iterate = false;
    if (in_turn) {
        if (mode == CTL_NEW) {
            adopt = flag != 0 ? AFTER_START : AFTER_NEW;
        } else if (mode == CTL_RESOLVED) {
            if (tm.tid() == 0) V.scal[CS_NFAIL] = floor(V.pack[V.off_nfail] + 0.5);
            iterate = true;
        } else {  // CTL_STEP, CTL_LS_DONE
            const double* st = V.pack + V.off_stats;
            const bool speculated = mode == CTL_STEP && flag != 0;
            const double gd = st[0] + V.scal[CS_GD_SH], dHd = st[1] + V.scal[CS_DHD_SH];  // PackLayout::GD, DHD
            const double model_change = mode == CTL_STEP ? -gd - 0.5 * dHd : V.scal[CS_MODEL_CHANGE];
            const bool valid = model_change > 0.0 && ctl_finite(model_change);
            double cand = st[4], step2 = st[2] + V.scal[CS_STEP2_SH], xnorm2 = st[3] + V.scal[CS_XNORM2_SH];
            if (mode == CTL_LS_DONE) {
                cand = V.scal[CS_LS_COST];
                step2 = V.scal[CS_LS_STEP2] + V.scal[CS_LS_STEP2_SH];
                xnorm2 = V.scal[CS_LS_XNORM2] + V.scal[CS_LS_XNORM2_SH];
            }
            tm.sync();
            if (tm.tid() == 0 && speculated) V.scal[CS_N_SPEC] += 1.0;
            if (!valid) {
                ctl_invalid_step(tm, V);
            } else if (mode == CTL_STEP && V.constrained && V.line_search && !(ctl_finite(cand) && cand <= V.scal[CS_COST] + 1e-4 * gd)) {
                // Ceres' projected line search on a bounds-constrained problem (line_search.hpp): the trial point just evaluated is
                // its first sample and fails the Armijo test.  The search (a handful of samples, rare) is run by the host.
                if (tm.tid() == 0) {
                    V.scal[CS_MODEL_CHANGE] = model_change;
                    V.scal[CS_SLOPE0] = gd;
                    V.scal[CS_CAND_COST] = cand;  // (the host's search starts from this sample)
                    V.scal[CS_EXPECT] = CTL_LINE_SEARCH;
                }
            } else if (ctl_decide(tm, V, speculated, model_change, cand, step2, xnorm2)) {
                adopt = AFTER_STEP;
            }
        }
        tm.sync();
        tm.tick(V, CP_DECIDE);
    }
    if (adopt) {
        ctl_adopt(tm, V, adopt == AFTER_START);
        if (adopt == AFTER_START) {
            if (tm.tid() == 0) V.scal[CS_INITIAL_COST] = V.scal[CS_COST];
            if (V.scal[CS_GMAX] <= V.eps) ctl_end(tm, V, CBA_TERM_CONVERGENCE, CM_GRADIENT);
            else iterate = true;
        } else if (adopt == AFTER_NEW) {
            iterate = true;
        } else if (V.scal[CS_RADIUS] != V.scal[CS_RADIUS_SPEC]) {  // gain ratio below 0.937: the elimination was made with another radius
            if (tm.tid() == 0) {
                V.scal[CS_N_MISSES] += 1.0;
                V.scal[CS_EXPECT] = CTL_RESOLVED;
                V.lmp[0] = V.scal[CS_RADIUS];
                V.lmp[1] = 0.0;
            }
        } else {
            if (tm.tid() == 0) V.scal[CS_N_HITS] += 1.0;
            iterate = true;
        }
        tm.sync();
    }
    if (iterate) ctl_iterate(tm, V);
    tm.sync();
    if (tm.tid() == 0 && static_cast<int>(V.scal[CS_EXPECT]) == CTL_RESOLVED && V.scal[CS_TERM] < 0.0) {
        // what follows the re-elimination is known now (its result does not enter the loop-top tests): the host may queue it
        // behind the re-elimination without waiting for this record's successor
        int msg;
        V.scal[CS_WILL_END] = ctl_top_tests(V, &msg) ? 1.0 : 0.0;
        V.scal[CS_STEP_SPEC] = ctl_next_step_speculative(V) ? 1.0 : 0.0;
    }
    if (tm.tid() == 0)
        V.scal[CS_GO] = (in_turn && V.scal[CS_TERM] < 0.0 && static_cast<int>(V.scal[CS_EXPECT]) == CTL_STEP && V.scal[CS_STEP_SPEC] != 0.0 &&
                         V.scal[CS_ACCEPT] == 2.0) ? 1.0 : 0.0;
    tm.sync();
    tm.publish(V);
}

// the control scalars at the start of a solve
inline void ctl_reset(double* scal) {
    for (int k = 0; k < CS_COUNT; ++k) scal[k] = 0.0;
    scal[CS_TERM] = -1.0;
    scal[CS_MSG] = CM_NONE;
    scal[CS_EXPECT] = CTL_NEW;
    scal[CS_RADIUS] = 1e4;
    scal[CS_DECREASE] = 2.0;
}

}  // namespace cba
