// resident_lm.hip — the whole Levenberg-Marquardt solve of a SMALL reprojection problem in ONE kernel launch.
//
// The problems the reference's own tests and pipelines solve (15-25 views of an 8 x 11 target: ~2e3 observations) keep
// a 256-CU chip idle: with the host-driven iteration of lm_core.hpp / backend_hip.hip every LM step is ~20 dependent
// kernel launches and two stream synchronisations, ~175 us however little work they carry.
// Here one workgroup of 8 wavefronts stays resident for the whole solve: every stage of the iteration is a phase of the
// kernel separated by workgroup barriers, the accept / reject control flow of the Ceres trust-region loop runs on the
// device, and the host sees one launch, one copy and one synchronisation per solve.
//
// Same arithmetic as the host-driven path, stage by stage: the per-tile Mode B / Mode R bodies (mode_b.hpp), the
// per-view Schur bodies (schur_math.hpp), the reduced system and the Ceres rules restated in lm_core.hpp
// (src/estimation/detail/ceresutils.h:27-43 sets the options; the rules themselves are Ceres 2.x's, DESIGN.md
// "Solver semantics").  Reductions use a fixed order, so a solve is bitwise reproducible; against the host-driven path
// results differ by summation order only (tests/test_gpu_parity.py runs the LM parity cases through both).
//
// Single rank, fp64 only; problems above the size limits, multi-rank handles, verbose solves and the fp32 study take
// the host-driven path (resident_lm_eligible).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "engine.hpp"
#include "lm_core.hpp"
#include "lm_state.hpp"
#include "mode_b.hpp"
#include "schur_math.hpp"

namespace cba {

constexpr int RES_THREADS = 512, RES_WAVES = RES_THREADS / 64;
constexpr int RES_NSH_MAX = 80;  // reduced system lives in LDS: 80 x 81 doubles
constexpr int RES_MAX_CAMS = 16;
constexpr int RES_PROF = 16;    // phase timers (100 MHz wall clock ticks, thread 0), printed when CBA_LM_RESIDENT_PROFILE=1

struct ResidentArgs {
    int n_blocks, n_cams, n_views, nsh, n_tiles, sh_base;
    SchurDims dims;
    const Tile* tiles;
    const int64_t *blk_tile_off, *link_off, *cam_off;
    const int32_t *blk_cam, *blk_view, *link_blk, *view_cam_blk, *cam_blk, *view_fixed;
    const double *X, *Y, *u, *v, *aux;
    double *bc, *sd;
    double *intr[2], *cam[2], *view[2], *target[2];
    double *partial, *blk_acc, *blk_s, *blk_w, *cam_acc;
    double *view_L, *view_y, *view_D, *view_gp, *view_scale2, *blk_Z, *view_delta;
    double *Hcc, *Ssch;
    const int8_t *active, *cam_var;
    int intr_var, target_var, constrained, line_search;
    double huber, eps;
    int max_iterations;
    double* out;  // [8]: termination, iterations, successful steps, initial cost, final cost, message id; [8..8+RES_PROF): phase ticks
};

enum ResidentMsg { MSG_GRADIENT = 0, MSG_MAX_ITER, MSG_MIN_RADIUS, MSG_INVALID_STEPS, MSG_PARAMETER, MSG_FUNCTION, MSG_LINE_SEARCH };
static const char* const kResidentMsg[] = {"Gradient tolerance reached.", "Maximum number of iterations reached.",
                                           "Minimum trust region radius reached.",
                                           "Number of consecutive invalid steps more than max.", "Parameter tolerance reached.",
                                           "Function tolerance reached."};

struct ResidentShared {
    double A[RES_NSH_MAX][RES_NSH_MAX + 1];  // reduced matrix -> its lower Cholesky factor
    double rhs[RES_NSH_MAX], delta[RES_NSH_MAX], gc[RES_NSH_MAX], gsch[RES_NSH_MAX], scale2[RES_NSH_MAX];
    int idx[RES_NSH_MAX];
    int8_t eff[RES_NSH_MAX];
    double red[RES_WAVES][6];
    double grp[RES_THREADS];
    // control block (written by thread 0 between barriers, read by everyone after)
    double radius, decrease_factor, cost, gmax, gmax_priv;
    int m, nfail, valid;
    unsigned long long prof[RES_PROF];
    // the shared parameter blocks, current [0] and trial [1] (Plus and the projected gradient norm work here; the
    // per-observation phases read the global copies, refreshed by publish_shared)
    double p_intr[2][RES_MAX_CAMS * 12], p_cam[2][RES_MAX_CAMS * 7], p_target[2][7];
};

// ---- workgroup reductions (fixed order) ------------------------------------------------------------------------------
// sums of up to 6 values per thread (red[][6]); every thread returns with the totals in v[]
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], ResidentShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double t = wave_sum63(v[k]);
        if (lane == 63) sh.red[wave][k] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double t = 0.0;
        for (int w = 0; w < RES_WAVES; ++w) t += sh.red[w][k];
        v[k] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ double block_max(double v, ResidentShared& sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6][0] = v;
    __syncthreads();
    double m = sh.red[0][0];
    for (int w = 1; w < RES_WAVES; ++w) m = fmax(m, sh.red[w][0]);
    __syncthreads();
    return m;
}

// out(idx, sum_{k < n_terms(idx)} term(idx, k)) for idx < total.  When `total` is small the terms of one sum are dealt
// round-robin to G = 512 / total threads and the G partial sums are added in group order through LDS.
template <class NT, class TF, class OF>
__device__ __forceinline__ void grouped_sums(int total, NT n_terms, TF term, OF out, ResidentShared& sh) {
    const int tid = threadIdx.x;
    if (2 * total <= RES_THREADS) {
        int G = RES_THREADS / total;
        if (G > 16) G = 16;
        const int r = tid / total, idx = tid - r * total;
        if (r < G) {
            double s = 0.0;
            const int n = n_terms(idx);
            for (int k = r; k < n; k += G) s += term(idx, k);
            sh.grp[r * total + idx] = s;
        }
        __syncthreads();
        if (tid < total) {
            double tot = 0.0;
            for (int g = 0; g < G; ++g) tot += sh.grp[g * total + tid];
            out(tid, tot);
        }
    } else {
        for (int idx = tid; idx < total; idx += RES_THREADS) {
            double s = 0.0;
            const int n = n_terms(idx);
            for (int k = 0; k < n; ++k) s += term(idx, k);
            out(idx, s);
        }
    }
    __syncthreads();
}

// ---- layout of the reduced (shared) tangent space: structure.hpp shared_col, inverted -----------------------------------
// global shared column i -> (camera, local column); camera -1 = the bundle chain's target pose (every block has it)
__device__ __forceinline__ void shared_decode(const ResidentArgs& a, int i, int* cam, int* lc) {
    if (a.dims.chain == CBA_CHAIN_BUNDLE) {
        if (i < 6) { *cam = -1; *lc = i; return; }
        *cam = (i - 6) / a.dims.PC;
        *lc = 6 + (i - 6) - *cam * a.dims.PC;
        return;
    }
    *cam = i / a.dims.PC;
    *lc = 6 + i - *cam * a.dims.PC;
}
__device__ __forceinline__ int intr_base(const ResidentArgs& a, int c) {
    return a.dims.chain == CBA_CHAIN_INTRINSIC ? 0 : a.sh_base + c * a.dims.PC + 6;
}
__device__ __forceinline__ int campose_base(const ResidentArgs& a, int c) { return a.sh_base + c * a.dims.PC; }

// LMDriver::shared_plus by the whole workgroup: LDS copy 0 + delta -> LDS copy 1 (Plus on every shared block, fx, fy >= 0
// projection).  *step2 / *xnorm2 receive THIS THREAD's share of |x+ - x|^2 and |x|^2 (the caller block-sums them).
// The caller provides the barrier before copy 1 is read.
__device__ __forceinline__ void shared_plus(const ResidentArgs& a, ResidentShared& sh, const double* delta, double* step2, double* xnorm2) {
    const int PI = a.dims.PL - (a.dims.chain == CBA_CHAIN_INTRINSIC ? 6 : 12);
    const int tid = threadIdx.x;
    double s2 = 0.0, x2 = 0.0;
    for (int i = tid; i < a.n_cams * PI; i += RES_THREADS) {
        const int c = i / PI, k = i - c * PI;
        const double p0 = sh.p_intr[0][i];
        double p = p0;
        if (a.intr_var) {
            p += delta[intr_base(a, c) + k];
            if (k < 2) p = fmax(p, 0.0);
            s2 += (p - p0) * (p - p0);
            x2 += p0 * p0;
        }
        sh.p_intr[1][i] = p;
    }
    if (a.dims.chain != CBA_CHAIN_INTRINSIC)
        for (int c = tid; c < a.n_cams; c += RES_THREADS) {
            const double* q = sh.p_cam[0] + 7 * c;
            double* o = sh.p_cam[1] + 7 * c;
            for (int k = 0; k < 7; ++k) o[k] = q[k];
            if (a.cam_var[c]) {
                const int pb = campose_base(a, c);
                quat_plus(q, delta + pb, o);
                for (int k = 0; k < 3; ++k) o[4 + k] = q[4 + k] + delta[pb + 3 + k];
                for (int k = 0; k < 7; ++k) { s2 += (o[k] - q[k]) * (o[k] - q[k]); x2 += q[k] * q[k]; }
            }
        }
    if (a.dims.chain == CBA_CHAIN_BUNDLE && tid == RES_THREADS - 1) {
        const double* q = sh.p_target[0];
        double* o = sh.p_target[1];
        for (int k = 0; k < 7; ++k) o[k] = q[k];
        if (a.target_var) {
            quat_plus(q, delta, o);
            for (int k = 0; k < 3; ++k) o[4 + k] = q[4 + k] + delta[3 + k];
            for (int k = 0; k < 7; ++k) { s2 += (o[k] - q[k]) * (o[k] - q[k]); x2 += q[k] * q[k]; }
        }
    }
    *step2 = s2;
    *xnorm2 = x2;
}

// LMDriver::shared_gmax: the shared blocks' share of Ceres' gradient max-norm (clobbers sh.delta and LDS copy 1)
__device__ __forceinline__ double shared_gmax(const ResidentArgs& a, ResidentShared& sh) {
    const int tid = threadIdx.x;
    double m = 0.0;
    if (!a.constrained) {
        for (int i = tid; i < a.nsh; i += RES_THREADS)
            if (sh.eff[i]) m = fmax(m, fabs(sh.gc[i]));
        return block_max(m, sh);
    }
    const int PI = a.dims.PL - (a.dims.chain == CBA_CHAIN_INTRINSIC ? 6 : 12);
    for (int i = tid; i < a.nsh; i += RES_THREADS) sh.delta[i] = sh.eff[i] ? -sh.gc[i] : 0.0;
    __syncthreads();
    double s2, x2;
    shared_plus(a, sh, sh.delta, &s2, &x2);
    __syncthreads();
    for (int i = tid; i < a.n_cams * PI; i += RES_THREADS) m = fmax(m, fabs(sh.p_intr[1][i] - sh.p_intr[0][i]));
    if (a.dims.chain != CBA_CHAIN_INTRINSIC)
        for (int i = tid; i < 7 * a.n_cams; i += RES_THREADS) m = fmax(m, fabs(sh.p_cam[1][i] - sh.p_cam[0][i]));
    if (a.dims.chain == CBA_CHAIN_BUNDLE)
        for (int i = tid; i < 7; i += RES_THREADS) m = fmax(m, fabs(sh.p_target[1][i] - sh.p_target[0][i]));
    return block_max(m, sh);
}

// LDS copy `from` of the shared blocks -> global copy `to` (and LDS copy `to`), by the whole workgroup
__device__ __forceinline__ void publish_shared(const ResidentArgs& a, ResidentShared& sh, int from, int to, bool barrier = true) {
    const int PI = a.dims.PL - (a.dims.chain == CBA_CHAIN_INTRINSIC ? 6 : 12);
    for (int i = threadIdx.x; i < a.n_cams * PI; i += RES_THREADS) { const double x = sh.p_intr[from][i]; sh.p_intr[to][i] = x; a.intr[to][i] = x; }
    if (a.dims.chain != CBA_CHAIN_INTRINSIC)
        for (int i = threadIdx.x; i < a.n_cams * 7; i += RES_THREADS) { const double x = sh.p_cam[from][i]; sh.p_cam[to][i] = x; a.cam[to][i] = x; }
    if (a.dims.chain == CBA_CHAIN_BUNDLE)
        for (int i = threadIdx.x; i < 7; i += RES_THREADS) { const double x = sh.p_target[from][i]; sh.p_target[to][i] = x; a.target[to][i] = x; }
    if (barrier) __syncthreads();
}

// ---- phases ------------------------------------------------------------------------------------------------------
template <int CHAIN, int MODEL>
// (the shared blocks come from the LDS copies, so this needs no barrier after publish_shared; the private poses are global)
__device__ __forceinline__ void phase_consts(const ResidentArgs& a, const ResidentShared& sh, int which) {
    for (int b = threadIdx.x; b < a.n_blocks; b += RES_THREADS) {
        const double *pA, *pB = nullptr, *ax = nullptr;
        if (CHAIN == CH_INTRINSIC) {
            pA = a.view[which] + 7 * static_cast<int64_t>(a.blk_view[b]);
        } else if (CHAIN == CH_EXTRINSIC) {
            pA = a.view[which] + 7 * static_cast<int64_t>(a.blk_view[b]);
            pB = sh.p_cam[which] + 7 * a.blk_cam[b];
        } else {
            pA = sh.p_target[which];
            pB = sh.p_cam[which] + 7 * a.blk_cam[b];
            ax = a.aux + 12 * static_cast<int64_t>(b);
        }
        double o[BC_SIZE];
        block_consts<CHAIN>(pA, pB, ax, o);
        for (int i = 0; i < BC_SIZE; ++i) a.bc[static_cast<int64_t>(b) * BC_SIZE + i] = o[i];
    }
    if (MODEL == CAM_SCHEIMPFLUG)
        for (int c = threadIdx.x; c < a.n_cams; c += RES_THREADS) {
            double o[SD_SIZE];
            for (int i = 0; i < SD_SIZE; ++i) o[i] = 0.0;
            scheimpflug_consts(sh.p_intr[which] + 12 * c, o);
            for (int i = 0; i < SD_SIZE; ++i) a.sd[static_cast<int64_t>(c) * SD_SIZE + i] = o[i];
        }
    __syncthreads();
}

template <int CHAIN, int MODEL, int NPARTS, int PART>
struct ModeBParts {
    static __device__ __forceinline__ void run(const Tile t, int lane, const double* bcp, const double* ip, const double* sp,
                                               const ResidentArgs& a, double* row) {
        normal_eq_tile<CHAIN, MODEL, NPARTS, PART, double>(t, lane, bcp, ip, sp, a.X, a.Y, a.u, a.v, row);
        if constexpr (PART + 1 < NPARTS) ModeBParts<CHAIN, MODEL, NPARTS, PART + 1>::run(t, lane, bcp, ip, sp, a, row);
    }
};

// blk_acc[b] = [H | g | s] of every block at parameter copy 0 (bc, sd current)
template <int CHAIN, int MODEL>
__device__ __forceinline__ void phase_mode_b(const ResidentArgs& a) {
    constexpr int PI = IntrSize<MODEL>::value;
    constexpr int NACC = LocalCols<CHAIN, MODEL>::value * (LocalCols<CHAIN, MODEL>::value + 1) / 2 + LocalCols<CHAIN, MODEL>::value + 1;
    // a wavefront has 256 registers here (8 waves on 4 SIMDs): one more part than the chip-wide kernels use
    constexpr int NPARTS = CHAIN == CH_INTRINSIC ? 3 : 6;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool direct = a.n_tiles == a.n_blocks;  // one tile per block: the tile row IS the block row
    for (int w = wave; w < a.n_tiles; w += RES_WAVES) {
        const Tile t = a.tiles[w];
        const int cam = a.blk_cam[t.blk];
        double* row = (direct ? a.blk_acc : a.partial) + static_cast<int64_t>(w) * NACC;
        ModeBParts<CHAIN, MODEL, NPARTS, 0>::run(t, lane, a.bc + static_cast<int64_t>(t.blk) * BC_SIZE, a.intr[0] + static_cast<int64_t>(cam) * PI,
                                                 a.sd + static_cast<int64_t>(cam) * SD_SIZE, a, row);
    }
    __syncthreads();
    if (!direct) {
        for (int idx = threadIdx.x; idx < a.n_blocks * NACC; idx += RES_THREADS) {
            const int b = idx / NACC, e = idx - b * NACC;
            double s = 0.0;
            for (int64_t t = a.blk_tile_off[b]; t < a.blk_tile_off[b + 1]; ++t) s += a.partial[t * NACC + e];
            a.blk_acc[idx] = s;
        }
        __syncthreads();
    }
}

// 1/2 sum_b rho(s_b) at parameter copy `which` (bc, sd built from it); leaves blk_acc / blk_w alone
template <int MODEL>
__device__ __forceinline__ double phase_resid_cost(const ResidentArgs& a, int which, ResidentShared& sh) {
    constexpr int PI = IntrSize<MODEL>::value;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int w = wave; w < a.n_tiles; w += RES_WAVES) {
        const Tile t = a.tiles[w];
        const int cam = a.blk_cam[t.blk];
        const double s = resid_tile<MODEL, double>(t, lane, a.bc + static_cast<int64_t>(t.blk) * BC_SIZE,
                                                   a.intr[which] + static_cast<int64_t>(cam) * PI, a.sd + static_cast<int64_t>(cam) * SD_SIZE,
                                                   a.X, a.Y, a.u, a.v);
        if (lane == 63) a.partial[w] = s;
    }
    __syncthreads();
    double c[1] = {0.0};
    for (int b = threadIdx.x; b < a.n_blocks; b += RES_THREADS) {
        double s = 0.0;
        for (int64_t t = a.blk_tile_off[b]; t < a.blk_tile_off[b + 1]; ++t) s += a.partial[t];
        a.blk_s[b] = s;
        double rho, w;
        huber(s, a.huber, &rho, &w);
        c[0] += 0.5 * rho;
    }
    block_sum(c, sh);
    return c[0];
}

// Huber weights, cost and the weighted per-camera sums of the block normal equations
__device__ __forceinline__ double phase_weights(const ResidentArgs& a, ResidentShared& sh) {
    const int NACC = a.dims.NACC, s_idx = a.dims.NH + a.dims.PL;
    double c[1] = {0.0};
    for (int b = threadIdx.x; b < a.n_blocks; b += RES_THREADS) {
        double rho, w;
        huber(a.blk_acc[static_cast<int64_t>(b) * NACC + s_idx], a.huber, &rho, &w);
        a.blk_w[b] = w;
        c[0] += 0.5 * rho;
    }
    block_sum(c, sh);  // (barriers: blk_w is visible below)
    grouped_sums(
        a.n_cams * NACC, [&](int idx) { const int cam = idx / NACC; return static_cast<int>(a.cam_off[cam + 1] - a.cam_off[cam]); },
        [&](int idx, int k) {
            const int cam = idx / NACC, e = idx - cam * NACC;
            const int b = a.cam_blk[a.cam_off[cam] + k];
            return a.blk_w[b] * a.blk_acc[static_cast<int64_t>(b) * NACC + e];
        },
        [&](int idx, double s) { a.cam_acc[idx] = s; }, sh);
    return c[0];
}

// H_cc, g_c from the per-camera sums (LMDriver::assemble_shared); effective columns; Jacobi scale on the first call
__device__ __forceinline__ void phase_assemble(const ResidentArgs& a, bool init_scale, ResidentShared& sh) {
    const int n = a.nsh, PL = a.dims.PL, NACC = a.dims.NACC, NH = a.dims.NH;
    for (int idx = threadIdx.x; idx < n * n + n; idx += RES_THREADS) {
        if (idx < n * n) {
            const int i = idx / n, j = idx - i * n;
            int ci, li, cj, lj;
            shared_decode(a, i, &ci, &li);
            shared_decode(a, j, &cj, &lj);
            double h = 0.0;
            if (ci < 0 && cj < 0) {
                for (int c = 0; c < a.n_cams; ++c) h += a.cam_acc[static_cast<int64_t>(c) * NACC + hidx_sym(PL, li, lj)];
            } else if (ci < 0 || cj < 0 || ci == cj) {
                h = a.cam_acc[static_cast<int64_t>(ci < 0 ? cj : ci) * NACC + hidx_sym(PL, li, lj)];
            }
            a.Hcc[idx] = h;
        } else {
            const int i = idx - n * n;
            int ci, li;
            shared_decode(a, i, &ci, &li);
            double g = 0.0;
            if (ci < 0) {
                for (int c = 0; c < a.n_cams; ++c) g += a.cam_acc[static_cast<int64_t>(c) * NACC + NH + li];
            } else {
                g = a.cam_acc[static_cast<int64_t>(ci) * NACC + NH + li];
            }
            sh.gc[i] = g;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += RES_THREADS) {
        const double hii = a.Hcc[static_cast<int64_t>(i) * n + i];
        sh.eff[i] = a.active[i] && hii != 0.0;  // columns nobody observes behave like constant blocks
        if (init_scale) {
            const double sc = 1.0 / (1.0 + sqrt(hii));
            sh.scale2[i] = sc * sc;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // compact list of the effective columns (read after the next phase's barriers)
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (sh.eff[i]) sh.idx[m++] = i;
        sh.m = m;
    }
}

// per-view elimination with sh.radius, then S_schur, g_schur
template <class Tick>
__device__ __forceinline__ void phase_schur(const ResidentArgs& a, bool init_scale, ResidentShared& sh, Tick tick) {
    const int n = a.nsh;
    double gm = 0.0, nf[1] = {0.0};
    const double radius = sh.radius;
    for (int v = threadIdx.x; v < a.n_views; v += RES_THREADS) {
        const int nb = static_cast<int>(a.link_off[v + 1] - a.link_off[v]);
        double g = 0.0;
        const bool ok = schur_view_body(a.dims, nb, a.link_blk + a.link_off[v], a.blk_acc, a.blk_w, a.view_fixed[v] != 0, radius, init_scale,
                                        a.constrained != 0, a.view[0] + 7 * static_cast<int64_t>(v), a.view_scale2 + 6 * static_cast<int64_t>(v),
                                        a.view_L + 36 * static_cast<int64_t>(v), a.view_y + 6 * static_cast<int64_t>(v),
                                        a.view_D + 6 * static_cast<int64_t>(v), a.view_gp + 6 * static_cast<int64_t>(v), a.blk_Z, &g);
        if (ok) gm = fmax(gm, g);
        else nf[0] += 1.0;
    }
    tick(11);
    gm = block_max(gm, sh);
    block_sum(nf, sh);
    tick(12);
    if (threadIdx.x == 0) { sh.gmax_priv = gm; sh.nfail = static_cast<int>(nf[0] + 0.5); }
    // S[i][j] = sum_v sum_k Z_v[k][i] Z_v[k][j], g[i] = sum_v sum_k Z_v[k][i] y_v[k]; the column -> (camera, local column)
    // decode is hoisted out of the view loop (z_entry would redo its integer division per term)
    const int PC = a.dims.PC, PSH = a.dims.PSH, n_cams = a.n_cams;
    grouped_sums(
        n * n, [&](int) { return a.n_views; },
        [&](int idx, int v) {
            const int i = idx / n, j = idx - i * n;
            const int ci = i / PC, cj = j / PC;
            const int bi = a.view_cam_blk[static_cast<int64_t>(v) * n_cams + ci], bj = a.view_cam_blk[static_cast<int64_t>(v) * n_cams + cj];
            if (bi < 0 || bj < 0) return 0.0;
            const double* zi = a.blk_Z + static_cast<int64_t>(bi) * 6 * PSH + (i - ci * PC);
            const double* zj = a.blk_Z + static_cast<int64_t>(bj) * 6 * PSH + (j - cj * PC);
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += zi[k * PSH] * zj[k * PSH];
            return s;
        },
        [&](int idx, double s) { a.Ssch[idx] = s; }, sh);
    tick(13);
    grouped_sums(
        n, [&](int) { return a.n_views; },
        [&](int i, int v) {
            const int ci = i / PC;
            const int bi = a.view_cam_blk[static_cast<int64_t>(v) * n_cams + ci];
            if (bi < 0) return 0.0;
            const double* zi = a.blk_Z + static_cast<int64_t>(bi) * 6 * PSH + (i - ci * PC);
            const double* yv = a.view_y + 6 * static_cast<int64_t>(v);
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += zi[k * PSH] * yv[k];
            return s;
        },
        [&](int i, double s) { sh.gsch[i] = s; }, sh);
}

// (H_cc + D_c - S_schur) delta_c = -(g_c - g_schur) on the effective columns (LMDriver::solve_reduced): the matrix is
// assembled by the workgroup, factorised and solved by wavefront 0 with lane r owning rows r and r + 64 (the
// dot-product Cholesky and the substitution order of dense.hpp; the diagonal is applied as a reciprocal).
__device__ __forceinline__ void phase_solve_reduced(const ResidentArgs& a, ResidentShared& sh) {
    const int n = a.nsh;
    if (threadIdx.x == 0) sh.valid = sh.nfail > 0 ? 0 : 1;
    for (int i = threadIdx.x; i < n; i += RES_THREADS) sh.delta[i] = 0.0;
    __syncthreads();
    const int m = sh.m;
    if (!sh.valid || m == 0) return;  // uniform
    const double radius = sh.radius;
    for (int e = threadIdx.x; e < m * m + m; e += RES_THREADS) {
        if (e < m * m) {
            const int r = e / m, c = e - r * m;
            const int i = sh.idx[r], j = sh.idx[c];
            double val = a.Hcc[static_cast<int64_t>(i) * n + j] - a.Ssch[static_cast<int64_t>(i) * n + j];
            if (r == c) val += lm_diag(a.Hcc[static_cast<int64_t>(i) * n + i], sh.scale2[i], radius);
            sh.A[r][c] = val;
        } else {
            const int r = e - m * m;
            const int i = sh.idx[r];
            sh.rhs[r] = -(sh.gc[i] - sh.gsch[i]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int r0 = lane, r1 = lane + 64;
        // value of row k's register (rows < 64 live in x0 of lane k, rows >= 64 in x1 of lane k - 64); k is wave-uniform
        auto row_value = [](double x0, double x1, int k) {
            const double x = k < 64 ? x0 : x1;
            const int l = k & 63;
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
        };
        bool ok = true;
        for (int j = 0; j < m; ++j) {
            // column j: s_r = A[r][j] - sum_{k<j} L[r][k] L[j][k] for r >= j
            double s0 = 0.0, s1 = 0.0;
            if (r0 >= j && r0 < m) { s0 = sh.A[r0][j]; for (int k = 0; k < j; ++k) s0 -= sh.A[r0][k] * sh.A[j][k]; }
            if (r1 >= j && r1 < m) { s1 = sh.A[r1][j]; for (int k = 0; k < j; ++k) s1 -= sh.A[r1][k] * sh.A[j][k]; }
            const double piv = row_value(s0, s1, j);
            if (!(piv > 0.0) || !(fabs(piv) <= 1.7976931348623157e308)) { ok = false; break; }
            const double d = sqrt(piv);
            if (r0 == j) sh.A[r0][j] = d; else if (r0 > j && r0 < m) sh.A[r0][j] = s0 / d;
            if (r1 == j) sh.A[r1][j] = d; else if (r1 > j && r1 < m) sh.A[r1][j] = s1 / d;
            __builtin_amdgcn_wave_barrier();
        }
        if (ok) {
            // forward, column by column: once b_k is final every later row subtracts L[r][k] b_k (per row the same
            // subtraction order as dense.hpp chol_solve); backward likewise from the last column
            double b0 = r0 < m ? sh.rhs[r0] : 0.0, b1 = r1 < m ? sh.rhs[r1] : 0.0;
            const double i0 = r0 < m ? 1.0 / sh.A[r0][r0] : 0.0, i1 = r1 < m ? 1.0 / sh.A[r1][r1] : 0.0;  // one division per row
            for (int k = 0; k < m; ++k) {
                if (r0 == k) b0 *= i0;
                if (r1 == k) b1 *= i1;
                const double bk = row_value(b0, b1, k);
                if (r0 > k && r0 < m) b0 -= sh.A[r0][k] * bk;
                if (r1 > k && r1 < m) b1 -= sh.A[r1][k] * bk;
            }
            for (int k = m - 1; k >= 0; --k) {
                if (r0 == k) b0 *= i0;
                if (r1 == k) b1 *= i1;
                const double bk = row_value(b0, b1, k);
                if (r0 < k) b0 -= sh.A[k][r0] * bk;
                if (r1 < k) b1 -= sh.A[k][r1] * bk;
            }
            if ((r0 < m && !(fabs(b0) <= 1.7976931348623157e308)) || (r1 < m && !(fabs(b1) <= 1.7976931348623157e308))) ok = false;
            ok = __all(ok);
            if (ok) {
                if (r0 < m) sh.delta[sh.idx[r0]] = b0;
                if (r1 < m) sh.delta[sh.idx[r1]] = b1;
            }
        }
        if (lane == 0) sh.valid = ok ? 1 : 0;
    }
    __syncthreads();
    if (!sh.valid) {
        for (int i = threadIdx.x; i < n; i += RES_THREADS) sh.delta[i] = 0.0;
        __syncthreads();
    }
}

// ---- the solve ---------------------------------------------------------------------------------------------------------
template <int CHAIN, int MODEL>
__global__ __launch_bounds__(RES_THREADS) void k_resident_lm(const ResidentArgs a) {
    __shared__ ResidentShared sh;
    const int tid = threadIdx.x;
    const int n = a.nsh;
    const double eps = a.eps;
    constexpr double min_radius = 1e-32, max_radius = 1e16, min_rel_decrease = 1e-3;

    // new linearisation at copy 0 with sh.radius: cost, gmax, Hcc, gc, Ssch, gsch, per-view factors
    unsigned long long last = wall_clock64();
    auto tick = [&](int slot) {  // thread 0: time since the previous tick goes to `slot`
        if (tid == 0) {
            const unsigned long long t = wall_clock64();
            sh.prof[slot] += t - last;
            last = t;
        }
    };
    {
        const int PI = IntrSize<MODEL>::value;
        for (int i = tid; i < a.n_cams * PI; i += RES_THREADS) {
            double x = a.intr[0][i];
            if (a.intr_var && i % PI < 2) x = fmax(x, 0.0);  // Ceres projects the start point onto the bounds (fx, fy >= 0)
            sh.p_intr[0][i] = x;
        }
        if (CHAIN != CH_INTRINSIC)
            for (int i = tid; i < a.n_cams * 7; i += RES_THREADS) sh.p_cam[0][i] = a.cam[0][i];
        if (CHAIN == CH_BUNDLE)
            for (int i = tid; i < 7; i += RES_THREADS) sh.p_target[0][i] = a.target[0][i];
        if (tid == 0) {
            for (int k = 0; k < RES_PROF; ++k) sh.prof[k] = 0;
            sh.radius = 1e4;
            sh.decrease_factor = 2.0;
        }
        __syncthreads();
        publish_shared(a, sh, 0, 0);
    }
    double initial_cost = 0.0;
    int iter = 0, invalid = 0, successful = 0, term = CBA_TERM_FAILURE, msg = MSG_INVALID_STEPS;
    // One loop, every phase once in the code (the kernel is ~15 000 instructions; inlining the linearisation at each of its call
    // sites doubled that): a pass starts with a new linearisation at copy 0 (`linearise`: first pass and after an accepted step)
    // or with the per-view elimination alone at the new radius (`re_eliminate`: after a rejected or invalid step).
    bool linearise = true, re_eliminate = false, first = true;
    while (true) {  // every condition below is workgroup-uniform (read from LDS after a barrier)
        if (linearise) {
            phase_consts<CHAIN, MODEL>(a, sh, 0);
            tick(0);
            phase_mode_b<CHAIN, MODEL>(a);
            tick(1);
            const double cost = phase_weights(a, sh);
            tick(2);
            phase_assemble(a, first, sh);
            tick(3);
            if (tid == 0) sh.cost = cost;
        }
        if (linearise || re_eliminate) {
            phase_schur(a, first && linearise, sh, tick);
            tick(4);
        }
        if (linearise) {
            const double gm_shared = shared_gmax(a, sh);
            if (tid == 0) sh.gmax = fmax(sh.gmax_priv, gm_shared);
            __syncthreads();
            tick(5);
        }
        linearise = re_eliminate = false;
        if (first) {  // (the driver tests the gradient at the start point before it looks at the iteration budget)
            first = false;
            initial_cost = sh.cost;
            if (sh.gmax <= eps) { term = CBA_TERM_CONVERGENCE; msg = MSG_GRADIENT; break; }
        }
        if (iter >= a.max_iterations) { term = CBA_TERM_NO_CONVERGENCE; msg = MSG_MAX_ITER; break; }
        if (sh.gmax <= eps) { term = CBA_TERM_CONVERGENCE; msg = MSG_GRADIENT; break; }
        if (sh.radius <= min_radius) { term = CBA_TERM_CONVERGENCE; msg = MSG_MIN_RADIUS; break; }
        ++iter;
        tick(10);
        phase_solve_reduced(a, sh);
        tick(6);
        bool valid = sh.valid != 0;
        if (valid) {
            // trial point: step2, xnorm2, g^T d, d^T H d (shared blocks' and views' shares together), then the shared-shared part
            // of the model change g_c^T d_c, d_c^T H_cc d_c (computed by the workgroup's LAST threads: the first ones have views)
            double st[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            shared_plus(a, sh, sh.delta, &st[0], &st[1]);
            for (int v = tid; v < a.n_views; v += RES_THREADS) {
                const int nb = static_cast<int>(a.link_off[v + 1] - a.link_off[v]);
                double o4[4];
                backsub_view_body(a.dims, nb, a.link_blk + a.link_off[v], a.blk_cam, a.blk_Z, sh.delta, a.view_fixed[v] != 0,
                                  a.view_L + 36 * static_cast<int64_t>(v), a.view_y + 6 * static_cast<int64_t>(v),
                                  a.view_D + 6 * static_cast<int64_t>(v), a.view_gp + 6 * static_cast<int64_t>(v),
                                  a.view[0] + 7 * static_cast<int64_t>(v), a.view_delta + 6 * static_cast<int64_t>(v),
                                  a.view[1] + 7 * static_cast<int64_t>(v), o4);
                for (int k = 0; k < 4; ++k) st[k] += o4[k];
            }
            for (int i = RES_THREADS - 1 - tid; i < n; i += RES_THREADS) {
                const double di = sh.delta[i];
                if (di != 0.0) {
                    st[4] += sh.gc[i] * di;
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += a.Hcc[static_cast<int64_t>(i) * n + j] * sh.delta[j];
                    st[5] += di * s;
                }
            }
            block_sum(st, sh);
            tick(7);
            const double model_change = -(st[2] + st[4]) - 0.5 * (st[3] + st[5]);
            if (!(model_change > 0.0) || !(fabs(model_change) <= 1.7976931348623157e308)) valid = false;
            if (valid) {
                publish_shared(a, sh, 1, 1, /*barrier=*/false);  // (from == to in LDS: only the global copies change)
                phase_consts<CHAIN, MODEL>(a, sh, 1);
                double cand = phase_resid_cost<MODEL>(a, 1, sh);
                tick(8);
                if (!(fabs(cand) <= 1.7976931348623157e308)) cand = 1.7976931348623157e308;
                // Bounds-constrained problem and the full step fails the Armijo test: Ceres would now search along the step
                // (line_search.hpp).  That search lives in the host-driven iteration; this kernel hands the solve back untouched
                // (the host re-runs it from the start point: such steps are rare, and a problem this size costs a millisecond).
                if (a.constrained && a.line_search && !(cand <= sh.cost + 1e-4 * (st[2] + st[4]))) { term = CBA_TERM_FAILURE; msg = MSG_LINE_SEARCH; break; }
                const double step_norm = sqrt(st[0]), x_norm = sqrt(st[1]);
                if (step_norm <= eps * (x_norm + eps)) { term = CBA_TERM_CONVERGENCE; msg = MSG_PARAMETER; break; }
                const double cost_change = sh.cost - cand;
                if (fabs(cost_change) <= eps * sh.cost) { term = CBA_TERM_CONVERGENCE; msg = MSG_FUNCTION; break; }
                const double rel = cost_change / model_change;
                invalid = 0;
                if (rel > min_rel_decrease) {  // accept: copy 1 -> copy 0, new linearisation next
                    publish_shared(a, sh, 1, 0);
                    for (int i = tid; i < a.n_views * 7; i += RES_THREADS) a.view[0][i] = a.view[1][i];
                    ++successful;
                    __syncthreads();
                    if (tid == 0) {
                        const double t = 2.0 * rel - 1.0;
                        sh.radius = fmin(max_radius, sh.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
                        sh.decrease_factor = 2.0;
                    }
                    __syncthreads();
                    tick(9);
                    linearise = true;
                } else {  // reject: shrink the radius, eliminate again
                    __syncthreads();
                    if (tid == 0) {
                        sh.radius = sh.radius / sh.decrease_factor;
                        sh.decrease_factor *= 2.0;
                    }
                    __syncthreads();
                    re_eliminate = true;
                }
                continue;
            }
        }
        // invalid step: the linear solve failed or the model did not decrease
        if (++invalid >= 5) { term = CBA_TERM_FAILURE; msg = MSG_INVALID_STEPS; break; }
        __syncthreads();
        if (tid == 0) sh.radius *= 0.5;
        __syncthreads();
        re_eliminate = true;
    }
    __syncthreads();
    phase_consts<CHAIN, MODEL>(a, sh, 0);  // leave bc / sd at the accepted point
    if (tid == 0) {
        a.out[0] = term;
        a.out[1] = iter;
        a.out[2] = successful;
        a.out[3] = initial_cost;
        a.out[4] = sh.cost;
        a.out[5] = msg;
        tick(10);
        for (int k = 0; k < RES_PROF; ++k) a.out[8 + k] = static_cast<double>(sh.prof[k]);
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
bool resident_lm_eligible(const Engine& e, const cba_options& o) {
    const HipLMState* st = reinterpret_cast<const HipLMState*>(e.lm_state);
    if (!st || st->resident_mode == 0) return false;
    const Structure& s = st->s;
    if (e.n_ranks != 1 || e.allreduce || e.rccl_comm) return false;  // the packed all-reduce lives in the host-driven iteration
    if (e.scalar != 0 || o.verbose) return false;
    if (s.n_blocks == 0 || s.nsh > RES_NSH_MAX || s.n_cams > RES_MAX_CAMS) return false;
    if (st->resident_mode == 1) {
        if (st->resident_max_obs >= 0) return s.n_obs <= st->resident_max_obs;  // CBA_LM_RESIDENT_MAX_OBS
        // Measured crossover with the staged iteration (tools/exp.py resident, one MI355X, end of round 3): per LM step the resident
        // kernel costs ~45 + 1.0 n_views + 0.0105 n_obs us on the intrinsic chain against 71 - 80 us staged (flat up to 4e4
        // observations; 86 before the small stages shared launches and the controller kernel kept its pointers' address spaces):
        // 10 x 88 points 61 vs 87, 10 x 144 68 vs 72, 20 x 88 (BASELINE configs[0]) 80 vs 72, 6 x 400 78 vs 80, 30 x 88 110 vs 71.
        // The two-pose chains need six register passes over 276-325 sums on ONE CU: the staged iteration wins at every size
        // measured (extrinsic 200 observations 111 vs 81 us, bundle 360 88 vs 80), so the automatic mode leaves them to it.
        if (s.chain == CBA_CHAIN_INTRINSIC) return 1.0 * s.n_views + 0.0105 * static_cast<double>(s.n_obs) <= 28.0;
        return false;
    }
    return true;
}

static ResidentArgs resident_args(Engine& e, HipLMState& st) {
    const Structure& s = st.s;
    ResidentArgs a{};
    a.n_blocks = s.n_blocks; a.n_cams = s.n_cams; a.n_views = s.n_views; a.nsh = s.nsh;
    a.n_tiles = static_cast<int>(e.n_tilesB); a.sh_base = s.sh_base;
    a.dims = st.dims;
    a.tiles = e.tilesB.p; a.blk_tile_off = e.d_blk_tile_off.p; a.link_off = st.link_off.p; a.cam_off = st.cam_off.p;
    a.blk_cam = e.d_blk_cam.p; a.blk_view = e.d_blk_view.p; a.link_blk = st.link_blk.p; a.view_cam_blk = st.view_cam_blk.p;
    a.cam_blk = st.cam_blk.p; a.view_fixed = e.view_fixed.p;
    a.X = e.X.p; a.Y = e.Y.p; a.u = e.u.p; a.v = e.v.p; a.aux = e.aux.p;
    a.bc = e.bc.p; a.sd = e.sd.p;
    for (int k = 0; k < 2; ++k) { a.intr[k] = e.intr[k].p; a.cam[k] = e.cam[k].p; a.view[k] = e.view[k].p; a.target[k] = e.target[k].p; }
    a.partial = e.partial.p; a.blk_acc = e.blk_acc.p; a.blk_s = e.blk_s.p; a.blk_w = e.blk_w.p; a.cam_acc = e.cam_acc.p;
    a.view_L = e.view_L.p; a.view_y = e.view_y.p; a.view_D = e.view_D.p; a.view_gp = e.view_gp.p; a.view_scale2 = e.view_scale2.p;
    a.blk_Z = e.blk_Z.p; a.view_delta = st.view_delta.p;
    a.Hcc = st.res_Hcc.p; a.Ssch = st.res_Ssch.p;
    a.active = st.res_active.p; a.cam_var = st.res_cam_var.p;
    a.out = st.res_out.p;
    return a;
}

static void resident_launch(Engine& e, const ResidentArgs& a) {
#define CALL(C, M) hipLaunchKernelGGL((k_resident_lm<C, M>), dim3(1), dim3(RES_THREADS), 0, e.stream, a);
    switch (e.chain * 2 + e.model) {
        case 0: CALL(CH_INTRINSIC, CAM_PINHOLE_BC) break;
        case 1: CALL(CH_INTRINSIC, CAM_SCHEIMPFLUG) break;
        case 2: CALL(CH_EXTRINSIC, CAM_PINHOLE_BC) break;
        case 3: CALL(CH_EXTRINSIC, CAM_SCHEIMPFLUG) break;
        case 4: CALL(CH_BUNDLE, CAM_PINHOLE_BC) break;
        case 5: CALL(CH_BUNDLE, CAM_SCHEIMPFLUG) break;
        default: throw std::runtime_error("bad chain/model");
    }
#undef CALL
    CBA_HIP(hipGetLastError());
}

// ROCm loads a code object and sets a kernel up on its first launch: pay for that at handle creation (cf. warm_lm)
void resident_lm_warm(Engine& e) {
    HipLMState& st = *lm_state(e);
    cba_options o{};
    o.max_iterations = 0;
    o.huber_delta = 1.0;
    o.epsilon = 1e-9;
    if (!resident_lm_eligible(e, o)) return;
    cba_summary s{};
    resident_lm_solve(e, o, &s, /*keep_parameters=*/true);
    (void)st;
}

bool resident_lm_solve(Engine& e, const cba_options& o, cba_summary* out, bool keep_parameters) {
    const auto t0 = std::chrono::steady_clock::now();
    HipLMState& st = *lm_state(e);
    const Structure& s = st.s;
    // masks of these options (LMDriver::setup also uploads the fixed-view flags)
    struct NullBackend final : Backend {
        Engine& e;
        explicit NullBackend(Engine& eng) : e(eng) {}
        void set_view_fixed(const std::vector<int32_t>& f) override { if (!f.empty()) e.view_fixed.upload(f.data(), f.size(), e.stream); CBA_HIP(hipStreamSynchronize(e.stream)); }
        void upload_shared(int, const double*, const double*, const double*) override {}
        void normal_eq(double, std::vector<double>&, double[2]) override {}
        void schur(double, bool, bool, std::vector<double>&, std::vector<double>&, double*, int*) override {}
        void trial(const double*, double, TrialStats*) override {}
        void accept() override {}
        void download_private(double*) override {}
        void download_blocks(std::vector<double>&, std::vector<double>&) override {}
        void line_eval(double, double, bool, const PackLayout&, const AllReduce&, int, double*) override {}
    } nb(e);
    LMDriver drv(s, nb, e.h_intr, e.h_cam, e.h_view, e.h_target, [](double*, int64_t) {}, 1, 0);
    const LMDriver::Masks mk = drv.masks(o);
    st.pin_mask.reserve(static_cast<size_t>(s.nsh + s.n_cams));
    for (int i = 0; i < s.nsh; ++i) st.pin_mask.p[i] = mk.active[i];
    for (int c = 0; c < s.n_cams; ++c) st.pin_mask.p[s.nsh + c] = mk.cam_var[c];
    st.res_active.upload(st.pin_mask.p, s.nsh, e.stream);
    st.res_cam_var.upload(st.pin_mask.p + s.nsh, s.n_cams, e.stream);
    // parameters: host state -> copy 0
    e.intr[0].upload(e.h_intr.data(), e.h_intr.size(), e.stream);
    if (e.chain != CBA_CHAIN_INTRINSIC) e.cam[0].upload(e.h_cam.data(), e.h_cam.size(), e.stream);
    if (e.chain == CBA_CHAIN_BUNDLE) e.target[0].upload(e.h_target.data(), 7, e.stream);
    if (!e.h_view.empty()) e.view[0].upload(e.h_view.data(), e.h_view.size(), e.stream);

    ResidentArgs a = resident_args(e, st);
    a.intr_var = mk.intr_var; a.target_var = mk.target_var; a.constrained = mk.constrained;
    a.huber = o.huber_delta; a.eps = o.epsilon; a.max_iterations = o.max_iterations;
    a.line_search = 1;
    if (const char* env = std::getenv("CBA_LM_LINE_SEARCH")) a.line_search = std::atoi(env) != 0;
    resident_launch(e, a);
    // results: [out(8) | intr | cam | target | views] through one pinned buffer, one synchronisation
    const size_t n_intr = e.h_intr.size(), n_cam = e.h_cam.size(), n_view = e.h_view.size();
    st.pin_res.reserve(8 + n_intr + n_cam + 7 + n_view + RES_PROF);
    double* p = st.pin_res.p;
    st.res_out.download(p, 8, e.stream);
    double* prof = p + 8 + n_intr + n_cam + 7 + n_view;
    st.res_out.download(prof, RES_PROF, e.stream, 8);
    e.intr[0].download(p + 8, n_intr, e.stream);
    if (e.chain != CBA_CHAIN_INTRINSIC) e.cam[0].download(p + 8 + n_intr, n_cam, e.stream);
    if (e.chain == CBA_CHAIN_BUNDLE) e.target[0].download(p + 8 + n_intr + n_cam, 7, e.stream);
    if (n_view) e.view[0].download(p + 8 + n_intr + n_cam + 7, n_view, e.stream);
    CBA_HIP(hipStreamSynchronize(e.stream));
    e.active = 0;
    if (static_cast<int>(p[5]) == MSG_LINE_SEARCH) {  // hand the solve to the host-driven iteration: device copy 0 back to the start point
        if (!e.h_view.empty()) e.view[0].upload(e.h_view.data(), e.h_view.size(), e.stream);
        e.intr[0].upload(e.h_intr.data(), e.h_intr.size(), e.stream);
        if (e.chain != CBA_CHAIN_INTRINSIC) e.cam[0].upload(e.h_cam.data(), e.h_cam.size(), e.stream);
        if (e.chain == CBA_CHAIN_BUNDLE) e.target[0].upload(e.h_target.data(), 7, e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
        return false;
    }
    if (!keep_parameters) {
        std::memcpy(e.h_intr.data(), p + 8, sizeof(double) * n_intr);
        if (e.chain != CBA_CHAIN_INTRINSIC) std::memcpy(e.h_cam.data(), p + 8 + n_intr, sizeof(double) * n_cam);
        if (e.chain == CBA_CHAIN_BUNDLE) std::memcpy(e.h_target.data(), p + 8 + n_intr + n_cam, sizeof(double) * 7);
        if (n_view) std::memcpy(e.h_view.data(), p + 8 + n_intr + n_cam + 7, sizeof(double) * n_view);
    }
    if (std::getenv("CBA_LM_RESIDENT_PROFILE")) {
        static const char* const names[RES_PROF] = {"consts", "mode B", "weights + camera sums", "assemble", "schur", "gradient norm",
                                                    "reduced solve", "trial step", "trial cost", "accept", "control", "schur: views",
                                                    "schur: max", "schur: S", "", ""};
        std::fprintf(stderr, "[cba] resident LM phases (us):");
        for (int k = 0; k < RES_PROF - 2; ++k) std::fprintf(stderr, " %s %.1f |", names[k], prof[k] * 0.01);
        std::fprintf(stderr, "\n");
    }
    const int term = static_cast<int>(p[0]), msg = static_cast<int>(p[5]);
    out->termination = term;
    out->success = term == CBA_TERM_CONVERGENCE;
    out->iterations = static_cast<int>(p[1]);
    out->successful_steps = static_cast<int>(p[2]);
    out->initial_cost = p[3];
    out->final_cost = p[4];
    out->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::snprintf(out->report, sizeof(out->report), "calibba(schur LM, resident kernel): %s iters=%d cost %.6e -> %.6e",
                  kResidentMsg[msg >= 0 && msg < 6 ? msg : 3], out->iterations, out->initial_cost, out->final_cost);
    return true;
}

}  // namespace cba
