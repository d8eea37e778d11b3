// kernels_modeb.hip — Mode B (per-block normal equations) with the row evaluation SHARED between the parts.
//
// The packed accumulator vector of a residual block ([H | g | s], or the moment row of the two-pose chains) does not fit one
// lane's registers, so it is split over NP parts.  In kernels_reproj.hip every part is its own launch and re-evaluates the
// residual / Jacobian rows of every observation: from the ISA, 150 of the 263 vector instructions per observation and part
// (130 of 223 in the moment form) are that re-evaluation (profiles/r02_modeb_isa_mix.txt).  Here ONE WORKGROUP of NP wavefronts
// owns a tile; wavefront p accumulates part p for ALL observations of the tile but evaluates the rows of only every NP-th
// 64-observation chunk; the rows (14 / 11 double2 per observation) reach the other wavefronts of the workgroup through LDS:
//
//     per group of NP chunks (NP = 2 .. 4 by form and camera model, see the launcher), wavefront p:   evaluate rows of chunk p -> LDS[p], accumulate them
//                                            barrier
//                                            for q != p: read rows of chunk q from LDS[q] (same lane), accumulate
//                                            barrier
//
// A lane reads back exactly what the same lane of another wavefront wrote: no transposition, unit-stride 16-byte LDS accesses.
// Vector work per observation: R + NP * A instead of NP * (R + A)  (R rows, A accumulate share): 376 instead of 526 instructions
// for the one-pose chain, 409 instead of 669 for the moment form.  Results equal the per-part launches bit for bit: the same
// products are added to the same accumulators in the same observation order (chunks in order, lanes fixed).
#include <cstdlib>

#include "engine.hpp"
#include "mode_b.hpp"
#include "reproj_math.hpp"
#include "wave_reduce.hpp"

#ifndef CBA_NE_MOM4_MINW
#define CBA_NE_MOM4_MINW 1
#endif
#ifndef CBA_NE_DIRECT_MINW
#define CBA_NE_DIRECT_MINW 1
#endif

namespace cba {

typedef double v2f64 __attribute__((ext_vector_type(2)));

// ---- the two accumulation forms behind one interface ------------------------------------------------------------------------
template <int CHAIN, int MODEL, class SPLIT, typename T>
struct DirectForm {
    using D = DirectRows<CHAIN, MODEL>;
    static constexpr int NROW = D::N, NPARTS = SPLIT::parts, NTOT = D::PL * (D::PL + 1) / 2 + D::PL + 1;
    static constexpr int MINW = CBA_NE_DIRECT_MINW;
    static constexpr bool DIRECT = true;  // the row is [H | g | s] itself: its last entry is the block's |r|^2 when the tile is the block
    static constexpr int count(int part) { return SPLIT::count(D::PL, part); }
    static __device__ __forceinline__ int entry(int part, int l) { return SPLIT::entry(D::PL, part, l); }
    static __device__ __forceinline__ void rows(const T* bcp, const T* ip, const T* sp, T x, T y, T u, T v, double* w) {
        direct_rows<CHAIN, MODEL, T>(bcp, ip, sp, x, y, u, v, w);
    }
    template <int PART>
    static __device__ __forceinline__ void accumulate(const double* w, double, double, double* acc) {
        direct_accumulate<CHAIN, MODEL, SPLIT, PART>(w, acc);
    }
};

template <int MODEL, int NP, typename T, int MW = 1>
struct MomentForm {
    static constexpr int PI = IntrSize<MODEL>::value;
    static constexpr int NROW = MomRows<PI>::N, NPARTS = NP, NTOT = MomLayout<PI>::N;
    static constexpr int MINW = MW;
    static constexpr bool DIRECT = false;
    static constexpr int count(int part) { return MomSplit<PI, NP>::T.count[part]; }
    static __device__ __forceinline__ int entry(int part, int l) { return MomSplit<PI, NP>::T.entry[part][l]; }
    static __device__ __forceinline__ void rows(const T* bcp, const T* ip, const T* sp, T x, T y, T u, T v, double* w) {
        mom_rows<MODEL, T>(bcp, ip, sp, x, y, u, v, w);
    }
    template <int PART>
    static __device__ __forceinline__ void accumulate(const double* w, double x, double y, double* acc) {
        mom_accumulate<PI, NP, PART>(w, x, y, acc);
    }
};

template <class FORM> struct ShareDims { static constexpr int NR2 = (FORM::NROW + 1) / 2; };
// NBUF = 2: the row buffer is doubled, group g uses half g & 1, and ONE barrier per group suffices (a wavefront overwrites half h
// only after passing the barrier of the group in between, i.e. after every wavefront has read half h); NBUF = 1: two barriers.

// the work of wavefront PART of the workgroup
// ABL (timing-only ablations, WRONG results): 1 = no barriers, 2 = no LDS reads (the own rows are accumulated NP times), 3 = both
// ONE: every tile of the problem fits ONE group (<= 64 NP observations: the 88-point views of a hand-eye bundle) - no group loop,
// no prefetch of a next group, no second barrier; the registers that frees let a third workgroup live on the CU, and with
// thousands of one-group tiles the kernel is a latency chain per workgroup whose throughput is the number of workgroups in flight.
template <class FORM, int PART, int NBUF, typename T, int ABL = 0, bool ONE = false, bool PF2 = false>
__device__ __forceinline__ void ne_shared_body(const Tile t, int lane, const T* bcp, const T* ip, const T* sp, const T* X, const T* Y,
                                               const T* u, const T* v, v2f64 (*shb)[ShareDims<FORM>::NR2][64], double* out,
                                               double huber_delta, double* w_out, double* s_out, volatile int* flags = nullptr) {
    constexpr int NP = FORM::NPARTS, NROW = FORM::NROW, NR2 = ShareDims<FORM>::NR2;
    // ABL == 4 (experiment builds; results are RIGHT): per-chunk flags in LDS instead of the two workgroup barriers per group.
    // flags[p] = groups whose rows wavefront p has published, flags[NP + p] = reads of chunk p's rows completed so far (all groups).
    // A wavefront waits only for the chunk it is about to read, and before rewriting its own rows for the reads of the last group.
    constexpr bool FLAGS = ABL == 4;
    auto spin_until = [&](int idx, int want) {
        for (int it = 0; it < (1 << 22) && flags[idx] < want; ++it) __builtin_amdgcn_s_sleep(1);  // (bounded: a bug must not hang the chip)
    };
    constexpr int NLOC = FORM::count(PART), NPAD = TransposeSum<16>::pad(NLOC);
    double acc[NPAD];
#pragma unroll
    for (int e = 0; e < NPAD; ++e) acc[e] = 0.0;
    const int n_groups = ONE ? 1 : (t.count + 64 * NP - 1) / (64 * NP);  // the same for every wavefront of the workgroup
    // PF: how many groups ahead the own chunk's x, y, u, v are loaded.  1: into the registers of the current group as soon as those
    // are dead (~0.6 of a pass ahead).  2 (PF2): two register sets used by even / odd groups, each reloaded for the group after
    // next right after its use (~1.6 passes ahead: a pass is ~0.7 us of issue, an HBM round trip under load longer).
    constexpr int PF = PF2 ? 2 : 1;
    T xa = T(0), ya = T(0), ua = T(0), va = T(0), xb = T(0), yb = T(0), ub = T(0), vb = T(0);
    {
        const int j = PART * 64 + lane;
        if (j < t.count) { xa = X[t.xy_start + j]; ya = Y[t.xy_start + j]; ua = u[t.start + j]; va = v[t.start + j]; }
        const int j1 = j + 64 * NP;
        if (PF2 && j1 < t.count) { xb = X[t.xy_start + j1]; yb = Y[t.xy_start + j1]; ub = u[t.start + j1]; vb = v[t.start + j1]; }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see k_normal_eq
    auto step = [&](int g, T& xc, T& yc, T& uc, T& vc) __attribute__((always_inline)) {
        const int j = (g * NP + PART) * 64 + lane;
        v2f64 (*sh)[NR2][64] = shb + (NBUF == 2 ? (g & 1) * NP : 0);
        // The loads of the next group's own chunk are issued INTO the registers of the current one as soon as those are dead
        // (u, v after the rows, x, y after the own chunk's accumulation): the prefetch costs no registers of its own (8 fewer
        // than holding both sets: the general moment-form kernel fits 168 and a third workgroup lives on the CU), and the loads
        // still have the barrier and the other chunks' accumulation to arrive.
        const bool more = !ONE && j + PF * 64 * NP < t.count;
        const int64_t in = t.start + j + PF * 64 * NP, k2n = t.xy_start + j + PF * 64 * NP;
        if (FLAGS && g > 0) spin_until(NP + PART, (NP - 1) * g);  // every reader is done with this wavefront's rows of the last group
        if (j < t.count) {
            double w[2 * NR2];
            FORM::rows(bcp, ip, sp, xc, yc, uc, vc, w);
            if (NROW & 1) w[NROW] = 0.0;
            if (more) { uc = u[in]; vc = v[in]; }
#pragma unroll
            for (int k = 0; k < NR2; ++k) sh[PART][k][lane] = v2f64{w[2 * k], w[2 * k + 1]};
            FORM::template accumulate<PART>(w, static_cast<double>(xc), static_cast<double>(yc), acc);
            if (more) { xc = X[k2n]; yc = Y[k2n]; }
        }
        if (FLAGS) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the rows are in LDS
            if (lane == 0) flags[PART] = g + 1;
        } else if (!(ABL & 1)) {
            __syncthreads();  // every wavefront's rows of this group are in LDS
        }
#pragma unroll
        for (int q = 1; q < NP; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int p = (PART + q) % NP;  // start with the neighbour: the NP wavefronts read NP different LDS regions at a time
            const int jo = (g * NP + p) * 64 + lane;
            if (FLAGS) spin_until(p, g + 1);
            double w[2 * NR2];
            if (jo < t.count) {
#pragma unroll
                for (int k = 0; k < NR2; ++k) { const v2f64 d = sh[(ABL & 2) ? PART : p][k][(ABL & 2) ? 0 : lane]; w[2 * k] = d.x; w[2 * k + 1] = d.y; }
            }
            if (FLAGS) {
                __builtin_amdgcn_s_waitcnt(0xC07F);  // the rows are in registers: the chunk's owner may rewrite them
                if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(flags) + NP + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (jo < t.count) {
                const double xo = static_cast<double>(X[t.xy_start + jo]), yo = static_cast<double>(Y[t.xy_start + jo]);
                FORM::template accumulate<PART>(w, xo, yo, acc);
            }
        }
        if (NBUF == 1 && !(ABL & 1) && !ONE && !FLAGS) __syncthreads();  // before the next group overwrites the rows
    };
    if (PF2) {
#pragma unroll 1
        for (int g = 0; g < n_groups; g += 2) {
            step(g, xa, ya, ua, va);
            if (g + 1 < n_groups) step(g + 1, xb, yb, ub, vb);  // (uniform over the workgroup: the barriers inside stay matched)
        }
    } else {
#pragma unroll 1
        for (int g = 0; g < n_groups; ++g) step(g, xa, ya, ua, va);
    }
    bool owner;
    const int base = wave_transpose_sum<NPAD>(acc, lane, &owner);
#pragma unroll
    for (int jj = 0; jj < TransposeSum<NPAD>::CNT; ++jj) {
        const int l = base + jj;
        if (owner && l < NLOC) {
            const int e = FORM::entry(PART, l);
            if (e < FORM::NTOT) out[e] = acc[jj];
            if constexpr (FORM::DIRECT) {
                // one tile per block: the lane that holds the block's |r|^2 also leaves its robust weight (k_weights' work)
                if (w_out && e == FORM::NTOT - 1) {
                    double rho, wt;
                    huber(acc[jj], huber_delta, &rho, &wt);
                    *w_out = wt;
                    *s_out = acc[jj];
                }
            }
        }
    }
}

// one workgroup of FORM::NPARTS wavefronts per tile
// MINW: wavefronts per SIMD the kernel is compiled for (0: whatever its registers allow).  A form within a few registers of the
// next occupancy step is compiled for that step (FORM::MINW): its row buffers allow the extra workgroup on the CU.
template <class FORM, int NBUF, typename T, int ABL = 0, bool ONE = false, bool PF2 = false>
__global__ __launch_bounds__(64 * FORM::NPARTS, FORM::MINW) void k_ne_shared(const double* __restrict__ gate, const Tile* __restrict__ tiles, int64_t n_tiles, const T* __restrict__ bc,
                                                                  const T* __restrict__ intr, const T* __restrict__ sd,
                                                                  const int32_t* __restrict__ blk_cam, const T* __restrict__ X,
                                                                  const T* __restrict__ Y, const T* __restrict__ u, const T* __restrict__ v,
                                                                  int PI, double* __restrict__ partial, double huber_delta = 0.0,
                                                                  double* __restrict__ blk_w = nullptr, double* __restrict__ blk_s = nullptr) {
    __shared__ v2f64 sh[NBUF * FORM::NPARTS][ShareDims<FORM>::NR2][64];
    __shared__ int sync_flags[ABL == 4 ? 2 * FORM::NPARTS : 1];
    if (ABL == 4) {
        if (threadIdx.x < 2 * FORM::NPARTS) sync_flags[threadIdx.x] = 0;
        __syncthreads();
    }
    if (gate && *gate == 0.0) return;  // (kernels_reproj.hip k_block_consts: a launch queued ahead of the decision it depends on)
    const int64_t w = blockIdx.x;
    if (w >= n_tiles) return;
    const Tile t = tiles[w];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T* bcp = bc + static_cast<int64_t>(t.blk) * BC_SIZE;
    const int cam = blk_cam[t.blk];
    const T* ip = intr + static_cast<int64_t>(cam) * PI;
    const T* sp = sd + static_cast<int64_t>(cam) * SD_SIZE;
    double* out = partial + w * FORM::NTOT;
    double *wo = blk_w ? blk_w + w : nullptr, *so = blk_w ? blk_s + w : nullptr;  // (one tile per block: tile index = block index)
    if (wave == 0) ne_shared_body<FORM, 0, NBUF, T, ABL, ONE, PF2>(t, lane, bcp, ip, sp, X, Y, u, v, sh, out, huber_delta, wo, so, sync_flags);
    if constexpr (FORM::NPARTS > 1) { if (wave == 1) ne_shared_body<FORM, 1, NBUF, T, ABL, ONE, PF2>(t, lane, bcp, ip, sp, X, Y, u, v, sh, out, huber_delta, wo, so, sync_flags); }
    if constexpr (FORM::NPARTS > 2) { if (wave == 2) ne_shared_body<FORM, 2, NBUF, T, ABL, ONE, PF2>(t, lane, bcp, ip, sp, X, Y, u, v, sh, out, huber_delta, wo, so, sync_flags); }
    if constexpr (FORM::NPARTS > 3) { if (wave == 3) ne_shared_body<FORM, 3, NBUF, T, ABL, ONE, PF2>(t, lane, bcp, ip, sp, X, Y, u, v, sh, out, huber_delta, wo, so, sync_flags); }
    if constexpr (FORM::NPARTS > 4) { if (wave == 4) ne_shared_body<FORM, 4, NBUF, T, ABL, ONE, PF2>(t, lane, bcp, ip, sp, X, Y, u, v, sh, out, huber_delta, wo, so, sync_flags); }
}

// ---- launchers ------------------------------------------------------------------------------------------------------------
template <class FORM, int NBUF, typename T>
static void launch_form(Engine& e, const T* bc, const T* intr, const T* sd, const T* X, const T* Y, const T* u, const T* v, double* rows) {
    static const bool one_ok = !(cba_exp_env("CBA_MODEB_ONEGROUP") && std::atoi(cba_exp_env("CBA_MODEB_ONEGROUP")) == 0);
    static const bool pf2 = cba_exp_env("CBA_MODEB_PF2") && std::atoi(cba_exp_env("CBA_MODEB_PF2")) != 0;  // prefetch two groups ahead
    // one tile per block and the caller wants the block weights (Engine::head_huber): the direct form leaves them on its way out
    const bool wts = FORM::DIRECT && e.head_huber >= 0.0 && e.n_tilesB == e.n_blocks && rows == e.blk_acc.p;
    double* wo = wts ? e.blk_w.p : nullptr;
    double* so = wts ? e.blk_s.p : nullptr;
    if (NBUF == 1 && one_ok && e.max_tileB <= 64 * FORM::NPARTS)  // every tile is one group: the single-group kernel
        hipLaunchKernelGGL((k_ne_shared<FORM, 1, T, 0, true>), dim3(static_cast<unsigned>(e.n_tilesB)), dim3(64 * FORM::NPARTS), 0, e.stream,
                           e.gate, e.tilesB.p, e.n_tilesB, bc, intr, sd, e.d_blk_cam.p, X, Y, u, v, e.PI, rows, e.head_huber, wo, so);
    else if (NBUF == 1 && pf2)
        hipLaunchKernelGGL((k_ne_shared<FORM, 1, T, 0, false, true>), dim3(static_cast<unsigned>(e.n_tilesB)), dim3(64 * FORM::NPARTS), 0, e.stream, e.gate,
                           e.tilesB.p, e.n_tilesB, bc, intr, sd, e.d_blk_cam.p, X, Y, u, v, e.PI, rows, e.head_huber, wo, so);
    else
        hipLaunchKernelGGL((k_ne_shared<FORM, NBUF, T>), dim3(static_cast<unsigned>(e.n_tilesB)), dim3(64 * FORM::NPARTS), 0, e.stream, e.gate, e.tilesB.p,
                           e.n_tilesB, bc, intr, sd, e.d_blk_cam.p, X, Y, u, v, e.PI, rows, e.head_huber, wo, so);
    if (wts) e.head_weights = true;
}

template <class F64, class F32, int NBUF = 1>
static void launch_both(Engine& e, double* rows) {
    if (e.scalar) launch_form<F32, NBUF, float>(e, e.bcf.p, e.intrf.p, e.sdf.p, e.Xf.p, e.Yf.p, e.uf.p, e.vf.p, rows);
    else launch_form<F64, NBUF, double>(e, e.bc.p, e.intr[e.active].p, e.sd.p, e.X.p, e.Y.p, e.u.p, e.v.p, rows);
}

// the shared-rows kernel of this engine's chain / model writing one row per tile into `rows` (row stride = the form's NTOT);
// returns false when there is no such kernel (the caller then uses the per-part launches)
bool launch_normal_eq_shared_rows(Engine& e, double* rows) {
    if (e.n_tilesB == 0) return true;
    // experiment knobs (read once): CBA_MODEB_DPARTS = parts of the one-pose direct form (2, 3, 4);
    // CBA_MODEB_VARIANT: bits 0-3 = parts of the moment form (2 .. 5), bit 4 = double-buffered rows, >= 32: timing-only ablations
    // Defaults as measured (profiles/r02_modeb_variants.jsonl, ms per pass): one-pose chain, pinhole 2 parts 0.192 / 3: 0.235 / 4: 0.219;
    // Scheimpflug 2: 0.303 / 3: 0.306 / 4: 0.265; moment form (C3 / 4), pinhole 2: 1.075 / 3: 0.939 / 4: 0.830 / 5: 1.415 (LDS-limited
    // occupancy) with the family-aligned split of MomSplitTable (round-robin entries: 1.002 / 0.955 / 0.907 / 1.618), double-buffered
    // rows 0.961 (no gain: the barrier that remains is the one that costs).
    static const int dparts_env = cba_exp_env("CBA_MODEB_DPARTS") ? std::atoi(cba_exp_env("CBA_MODEB_DPARTS")) : 0;
    static const int variant = cba_exp_env("CBA_MODEB_VARIANT") ? std::atoi(cba_exp_env("CBA_MODEB_VARIANT")) : 4;
    const int dparts = dparts_env ? dparts_env : (e.model == CAM_SCHEIMPFLUG ? 4 : 2);
    if (e.chain == CH_INTRINSIC) {
        if (e.model == CAM_PINHOLE_BC) {
            if (dparts == 4) launch_both<DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<4>, double>, DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<4>, float>>(e, rows);
            else if (dparts == 3) launch_both<DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<3>, double>, DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<3>, float>>(e, rows);
#ifdef CBA_EXPERIMENTS
            else if (variant == 64) {  // per-chunk flags instead of the two barriers per group
                using F = DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<2>, double>;
                hipLaunchKernelGGL((k_ne_shared<F, 1, double, 4>), dim3(static_cast<unsigned>(e.n_tilesB)), dim3(64 * 2), 0, e.stream, e.gate, e.tilesB.p, e.n_tilesB,
                                   e.bc.p, e.intr[e.active].p, e.sd.p, e.d_blk_cam.p, e.X.p, e.Y.p, e.u.p, e.v.p, e.PI, rows);
            }
            else if (variant & 16) launch_both<DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<2>, double>, DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<2>, float>, 2>(e, rows);  // double-buffered rows: one barrier per group
#endif
            else launch_both<DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<2>, double>, DirectForm<CH_INTRINSIC, CAM_PINHOLE_BC, SplitRoundRobin<2>, float>>(e, rows);
        } else {
            if (dparts == 4) launch_both<DirectForm<CH_INTRINSIC, CAM_SCHEIMPFLUG, SplitRoundRobin<4>, double>, DirectForm<CH_INTRINSIC, CAM_SCHEIMPFLUG, SplitRoundRobin<4>, float>>(e, rows);
            else if (dparts == 3) launch_both<DirectForm<CH_INTRINSIC, CAM_SCHEIMPFLUG, SplitRoundRobin<3>, double>, DirectForm<CH_INTRINSIC, CAM_SCHEIMPFLUG, SplitRoundRobin<3>, float>>(e, rows);
            else launch_both<DirectForm<CH_INTRINSIC, CAM_SCHEIMPFLUG, SplitRoundRobin<2>, double>, DirectForm<CH_INTRINSIC, CAM_SCHEIMPFLUG, SplitRoundRobin<2>, float>>(e, rows);
        }
    } else {
        if (!e.modeb_moments) return false;
        if (e.model == CAM_PINHOLE_BC) {
            if ((variant & 15) == 2) launch_both<MomentForm<CAM_PINHOLE_BC, 2, double>, MomentForm<CAM_PINHOLE_BC, 2, float>>(e, rows);
            else if (variant == 36) launch_both<MomentForm<CAM_PINHOLE_BC, 4, double, 3>, MomentForm<CAM_PINHOLE_BC, 4, float>>(e, rows);  // capped at 168 registers
            else if ((variant & 15) == 4) launch_both<MomentForm<CAM_PINHOLE_BC, 4, double>, MomentForm<CAM_PINHOLE_BC, 4, float>>(e, rows);
            else if ((variant & 15) == 5) launch_both<MomentForm<CAM_PINHOLE_BC, 5, double>, MomentForm<CAM_PINHOLE_BC, 5, float>>(e, rows);
            else if (variant & 16) launch_both<MomentForm<CAM_PINHOLE_BC, 3, double>, MomentForm<CAM_PINHOLE_BC, 3, float>, 2>(e, rows);
#ifdef CBA_EXPERIMENTS
            else if (variant == 64 && !e.scalar) {  // per-chunk flags instead of the two barriers per group (results are right)
                using F = MomentForm<CAM_PINHOLE_BC, 4, double>;
                hipLaunchKernelGGL((k_ne_shared<F, 1, double, 4>), dim3(static_cast<unsigned>(e.n_tilesB)), dim3(64 * 4), 0, e.stream, e.gate, e.tilesB.p, e.n_tilesB,
                                   e.bc.p, e.intr[e.active].p, e.sd.p, e.d_blk_cam.p, e.X.p, e.Y.p, e.u.p, e.v.p, e.PI, rows);
            }
            else if (variant >= 32 && variant != 36 && !e.scalar) {  // timing-only ablations (32 + ABL): results are wrong
                using F = MomentForm<CAM_PINHOLE_BC, 3, double>;
                const dim3 g(static_cast<unsigned>(e.n_tilesB)), b(64 * 3);
#define CBA_ABL(A) hipLaunchKernelGGL((k_ne_shared<F, 1, double, A>), g, b, 0, e.stream, e.gate, e.tilesB.p, e.n_tilesB, e.bc.p, e.intr[e.active].p, e.sd.p, \
                                      e.d_blk_cam.p, e.X.p, e.Y.p, e.u.p, e.v.p, e.PI, rows)
                if (variant == 33) CBA_ABL(1); else if (variant == 34) CBA_ABL(2); else CBA_ABL(3);
#undef CBA_ABL
            }
#endif
            else launch_both<MomentForm<CAM_PINHOLE_BC, 3, double>, MomentForm<CAM_PINHOLE_BC, 3, float>>(e, rows);
        } else {
            if ((variant & 15) == 5) launch_both<MomentForm<CAM_SCHEIMPFLUG, 5, double>, MomentForm<CAM_SCHEIMPFLUG, 5, float>>(e, rows);
            else if ((variant & 15) == 3) launch_both<MomentForm<CAM_SCHEIMPFLUG, 3, double>, MomentForm<CAM_SCHEIMPFLUG, 3, float>>(e, rows);
            else launch_both<MomentForm<CAM_SCHEIMPFLUG, 4, double>, MomentForm<CAM_SCHEIMPFLUG, 4, float>>(e, rows);
        }
    }
    CBA_HIP(hipGetLastError());
    return true;
}

}  // namespace cba
