// kernels_reproj.hip — gfx950 kernels of the reprojection hot path.
//
//   k_block_consts / k_scheimpflug_consts   parameters -> per-block chain constants (tiny)
//   k_eval       "Mode A": residual + tangent Jacobian of every observation, written to HBM (SoA)
//   k_resid      "Mode R": per-tile sum of squared residuals (trial-point cost)
//   k_normal_eq  "Mode B": per-tile J^T J / J^T r / |r|^2 accumulated in registers, wave-shuffle
//                reduced, one partial row per tile
//   k_tile_sum   fixed-order sum of a block's tile partials (deterministic: no atomics anywhere)
//
// Work decomposition: one WAVEFRONT (64 lanes) owns one tile = a run of consecutive observations
// of a single residual block.  Everything that is constant over a block (chain matrices, camera
// parameters) is therefore wave-uniform: the tile record is fetched with a scalar load
// (readfirstlane'd wave index) and the constants live in SGPRs / are broadcast, while X,Y,u,v and
// the outputs are unit-stride 16-byte-per-lane (Mode A) or 8-byte-per-lane vector accesses.
// The arithmetic is reproj_math.hpp (shared with the CPU test build).
#include <cstdlib>

#include "engine.hpp"
#include "mode_b.hpp"
#include "reproj_math.hpp"
#include "wave_reduce.hpp"

namespace cba {

__device__ __forceinline__ int64_t wave_index() {
    // wave-uniform by construction; readfirstlane lets the compiler keep it (and everything
    // indexed by it) in scalar registers
    return static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}

template <int CHAIN>
// gate (every kernel that takes one): nullptr, or a flag in device memory; when it reads 0 the launch does nothing.  The LM driver
// queues the head of the NEXT step behind the controller before it knows the controller's decision (lm_core.hpp solve_ctl): the
// controller sets the flag when the step it decided on is the one that was queued.
__global__ void k_block_consts(const double* __restrict__ gate, int n_blocks, const int32_t* __restrict__ blk_cam, const int32_t* __restrict__ blk_view,
                               const double* __restrict__ cam, const double* __restrict__ view,
                               const double* __restrict__ target, const double* __restrict__ aux, double* __restrict__ bc,
                               float* __restrict__ bcf) {
    if (gate && *gate == 0.0) return;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const double *pA, *pB = nullptr, *ax = nullptr;
    if (CHAIN == CH_INTRINSIC) {
        pA = view + 7 * static_cast<int64_t>(blk_view[b]);
    } else if (CHAIN == CH_EXTRINSIC) {
        pA = view + 7 * static_cast<int64_t>(blk_view[b]);
        pB = cam + 7 * static_cast<int64_t>(blk_cam[b]);
    } else {
        pA = target;
        pB = cam + 7 * static_cast<int64_t>(blk_cam[b]);
        ax = aux + 12 * static_cast<int64_t>(b);
    }
    double out[BC_SIZE];
    block_consts<CHAIN>(pA, pB, ax, out);
    for (int i = 0; i < BC_SIZE; ++i) bc[static_cast<int64_t>(b) * BC_SIZE + i] = out[i];
    if (bcf)
        for (int i = 0; i < BC_SIZE; ++i) bcf[static_cast<int64_t>(b) * BC_SIZE + i] = static_cast<float>(out[i]);
}

__global__ void k_scheimpflug_consts(const double* __restrict__ gate, int n_cams, const double* __restrict__ intr, double* __restrict__ sd,
                                     float* __restrict__ sdf) {
    if (gate && *gate == 0.0) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cams) return;
    double out[SD_SIZE];
    for (int i = 0; i < SD_SIZE; ++i) out[i] = 0.0;
    scheimpflug_consts(intr + 12 * static_cast<int64_t>(c), out);
    for (int i = 0; i < SD_SIZE; ++i) sd[static_cast<int64_t>(c) * SD_SIZE + i] = out[i];
    if (sdf)
        for (int i = 0; i < SD_SIZE; ++i) sdf[static_cast<int64_t>(c) * SD_SIZE + i] = static_cast<float>(out[i]);
}

__global__ void k_to_f32(int64_t n, const double* __restrict__ in, float* __restrict__ out) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = static_cast<float>(in[i]);
}

// ---- Mode A -----------------------------------------------------------------------------------
// Algorithmic HBM traffic per observation: 4 loads + 2 residual stores + 2*P Jacobian stores of
// 8 bytes = 304 B (P=16) ... 432 B (P=24).  HBM-bound: ~0.3 kFLOP per observation.
template <typename T> struct Pair;
template <> struct Pair<double> { typedef double vec __attribute__((ext_vector_type(2))); using ld = double2; };
template <> struct Pair<float> { typedef float vec __attribute__((ext_vector_type(2))); using ld = float2; };

// two adjacent observations per lane: one 16-byte (fp64) / 8-byte (fp32) access
template <bool NT, typename T>
__device__ __forceinline__ void store2(T* p, T a, T b) {
    typename Pair<T>::vec v = {a, b};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<typename Pair<T>::vec*>(p));
    else *reinterpret_cast<typename Pair<T>::vec*>(p) = v;
}

// NT: non-temporal (streaming) stores for r / J, which this kernel never re-reads.
// ROWS: consecutive tiles handled by one wavefront.
// BLK: tile-blocked output layout out[tile][2 + 2P][128] (one contiguous 34 KiB region per tile) instead
// of whole-array columns r[2][ld], J[2P][ld].
// ABL (timing-only ablations, wrong outputs): 1 = skip the arithmetic (store the loaded values), 2 = skip the loads
//
// Software pipeline: a wavefront walks ROWS consecutive tiles and issues the 4 observation loads of tile
// k+1 BEFORE it computes and stores tile k, so the ~2 us load latency under a write-saturated memory
// system hides behind 34 KiB of stores instead of stalling the wave (measured: loads are 10 % of the
// bytes but cost 15 % of the time when they sit at the head of every wave).
template <int CHAIN, int MODEL, bool NT, int ROWS, bool BLK, int ABL = 0, typename T = double>
__global__ __launch_bounds__(256) void k_eval(const Tile* __restrict__ tiles, int64_t n_tiles,
                                              const T* __restrict__ bc, const T* __restrict__ intr,
                                              const T* __restrict__ sd, const int32_t* __restrict__ blk_cam,
                                              const T* __restrict__ X, const T* __restrict__ Y,
                                              const T* __restrict__ u, const T* __restrict__ v,
                                              T* __restrict__ r, T* __restrict__ J, int64_t ld) {
    using V2 = typename Pair<T>::ld;
    constexpr int PI = IntrSize<MODEL>::value;
    constexpr int PL = LocalCols<CHAIN, MODEL>::value;
    const int lane = threadIdx.x & 63;
    const int64_t w0 = wave_index() * ROWS;
    if (w0 >= n_tiles) return;
    const int64_t w1 = (w0 + ROWS < n_tiles) ? w0 + ROWS : n_tiles;

    Tile t = tiles[w0];
    V2 Xv, Yv, uv, vv;
    auto load_obs = [&](const Tile& tt, V2& a, V2& b, V2& c, V2& d) {
        if (ABL == 2) {
            a.x = T(1e-3) * lane; a.y = T(2e-3) * lane; b.x = T(-1e-3) * lane; b.y = T(1e-3);
            c.x = T(600); c.y = T(610); d.x = T(300); d.y = T(310);
        } else {
            // lanes past the tile's end read the tile's first pair (in bounds, result unused)
            const int64_t o = (2 * lane < tt.count) ? 2 * lane : 0;
            a = *reinterpret_cast<const V2*>(X + tt.xy_start + o);
            b = *reinterpret_cast<const V2*>(Y + tt.xy_start + o);
            c = *reinterpret_cast<const V2*>(u + tt.start + o);
            d = *reinterpret_cast<const V2*>(v + tt.start + o);
        }
    };
    load_obs(t, Xv, Yv, uv, vv);
#pragma unroll 1
    for (int64_t w = w0; w < w1; ++w) {
        Tile tn = t;
        V2 Xn = Xv, Yn = Yv, un = uv, vn = vv;
        if (ROWS > 1 && w + 1 < w1) {
            tn = tiles[w + 1];
            load_obs(tn, Xn, Yn, un, vn);
        }
        if (2 * lane < t.count) {
            const T* bcp = bc + static_cast<int64_t>(t.blk) * BC_SIZE;
            const int cam = blk_cam[t.blk];
            const T* ip = intr + static_cast<int64_t>(cam) * PI;
            const T* sp = sd + static_cast<int64_t>(cam) * SD_SIZE;
            const int64_t i0 = t.start + 2 * lane;
            T r0[2], r1[2], Ju0[PL], Jv0[PL], Ju1[PL], Jv1[PL];
            if (ABL == 1) {
                r0[0] = Xv.x; r0[1] = Yv.x; r1[0] = uv.x; r1[1] = vv.x;
#pragma unroll
                for (int k = 0; k < PL; ++k) { Ju0[k] = Xv.y + k; Jv0[k] = Yv.y; Ju1[k] = uv.y; Jv1[k] = vv.y + k; }
            } else {
                reproj_point<CHAIN, MODEL, T>(bcp, ip, sp, Xv.x, Yv.x, uv.x, vv.x, r0, Ju0, Jv0);
                reproj_point<CHAIN, MODEL, T>(bcp, ip, sp, Xv.y, Yv.y, uv.y, vv.y, r1, Ju1, Jv1);
            }
            if (BLK) {
                T* o = J + w * static_cast<int64_t>((2 + 2 * PL) * TILE_A) + 2 * lane;
                store2<NT, T>(o, r0[0], r1[0]);
                store2<NT, T>(o + TILE_A, r0[1], r1[1]);
#pragma unroll
                for (int k = 0; k < PL; ++k) {
                    store2<NT, T>(o + (2 + k) * TILE_A, Ju0[k], Ju1[k]);
                    store2<NT, T>(o + (2 + PL + k) * TILE_A, Jv0[k], Jv1[k]);
                }
            } else {
                store2<NT, T>(r + i0, r0[0], r1[0]);
                store2<NT, T>(r + ld + i0, r0[1], r1[1]);
#pragma unroll
                for (int k = 0; k < PL; ++k) {
                    store2<NT, T>(J + static_cast<int64_t>(k) * ld + i0, Ju0[k], Ju1[k]);
                    store2<NT, T>(J + static_cast<int64_t>(PL + k) * ld + i0, Jv0[k], Jv1[k]);
                }
            }
        }
        t = tn; Xv = Xn; Yv = Yn; uv = un; vv = vn;
    }
}

// ---- Mode R -----------------------------------------------------------------------------------
template <int MODEL, typename T>
__global__ __launch_bounds__(256) void k_resid(const Tile* __restrict__ tiles, int64_t n_tiles,
                                               const T* __restrict__ bc, const T* __restrict__ intr,
                                               const T* __restrict__ sd, const int32_t* __restrict__ blk_cam,
                                               const T* __restrict__ X, const T* __restrict__ Y,
                                               const T* __restrict__ u, const T* __restrict__ v,
                                               double* __restrict__ partial_s) {
    constexpr int PI = IntrSize<MODEL>::value;
    const int64_t w = wave_index();
    if (w >= n_tiles) return;
    const Tile t = tiles[w];
    const int lane = threadIdx.x & 63;
    const T* bcp = bc + static_cast<int64_t>(t.blk) * BC_SIZE;
    const int cam = blk_cam[t.blk];
    const T* ip = intr + static_cast<int64_t>(cam) * PI;
    const T* sp = sd + static_cast<int64_t>(cam) * SD_SIZE;
    const double s = resid_tile<MODEL, T>(t, lane, bcp, ip, sp, X, Y, u, v);
    if (lane == 63) partial_s[w] = s;
}

// out[b][e] = sum over the block's tiles (in tile order) of partial[t][e].  s_idx >= 0: the rows are [H | g | s] rows and the thread
// that finishes a block's s = |r|^2 also leaves the block's robust weight and s (k_weights' work, one launch less per LM step)
__global__ void k_tile_sum(const double* __restrict__ gate, int n_blocks, int width, const int64_t* __restrict__ blk_tile_off,
                           const double* __restrict__ partial, double* __restrict__ out, int s_idx = -1, double huber_delta = 0.0,
                           double* __restrict__ blk_w = nullptr, double* __restrict__ blk_s = nullptr) {
    if (gate && *gate == 0.0) return;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= static_cast<int64_t>(n_blocks) * width) return;
    const int b = static_cast<int>(idx / width);
    const int e = static_cast<int>(idx % width);
    double s = 0.0;
    for (int64_t t = blk_tile_off[b]; t < blk_tile_off[b + 1]; ++t) s += partial[t * width + e];
    out[idx] = s;
    if (e == s_idx) {
        double rho, w;
        huber(s, huber_delta, &rho, &w);
        blk_w[b] = w;
        blk_s[b] = s;
    }
}

// scalar_out[0] = 1/2 sum_b rho(s_b), scalar_out[1] = sum_b s_b  (single workgroup, fixed order)
__global__ __launch_bounds__(256) void k_cost(int n_blocks, const double* __restrict__ blk_s, double huber_delta,
                                              double* __restrict__ out) {
    __shared__ double sh[2][256];
    double c = 0.0, ss = 0.0;
    for (int b = static_cast<int>(threadIdx.x); b < n_blocks; b += 256) {
        double rho, w;
        huber(blk_s[b], huber_delta, &rho, &w);
        c += 0.5 * rho;
        ss += blk_s[b];
    }
    sh[0][threadIdx.x] = c;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (static_cast<int>(threadIdx.x) < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sh[0][0]; out[1] = sh[1][0]; }
}

// two-stage form for many blocks (C3: 32 000): every workgroup reduces 2048 blocks to one pair, k_cost_final adds the pairs in
// order (fixed tree: deterministic)
__global__ __launch_bounds__(256) void k_cost_partial(int n_blocks, const double* __restrict__ blk_s, double huber_delta,
                                                      double* __restrict__ part) {
    __shared__ double sh[2][256];
    double c = 0.0, ss = 0.0;
    const int b0 = blockIdx.x * 2048;
    const int b1 = b0 + 2048 < n_blocks ? b0 + 2048 : n_blocks;
    for (int b = b0 + static_cast<int>(threadIdx.x); b < b1; b += 256) {
        double rho, w;
        huber(blk_s[b], huber_delta, &rho, &w);
        c += 0.5 * rho;
        ss += blk_s[b];
    }
    sh[0][threadIdx.x] = c;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (static_cast<int>(threadIdx.x) < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0][0]; part[2 * blockIdx.x + 1] = sh[1][0]; }
}
__global__ void k_cost_final(int n_part, const double* __restrict__ part, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double c = 0.0, ss = 0.0;
    for (int k = 0; k < n_part; ++k) { c += part[2 * k]; ss += part[2 * k + 1]; }
    out[0] = c;
    out[1] = ss;
}

// ---- Mode B -----------------------------------------------------------------------------------
// one tile per wavefront; the body (and the notes on the part split and the row masks) is mode_b.hpp
#ifndef CBA_NE_MINWAVES
#define CBA_NE_MINWAVES 1  // waves per SIMD the Mode B kernels are compiled for (register budget 512 / CBA_NE_MINWAVES).  Measured
                           // (profiles/r02_modeb_variants.jsonl): the pinhole kernels fit 2 waves either way; the Scheimpflug ones need ~270
                           // registers, and one wave with 14-26 values in AGPRs (0.33 ms at C5) beats two waves spilling to scratch (0.49 ms)
#endif
template <int CHAIN, int MODEL, class SPLIT, int PART, typename T>
__global__ __launch_bounds__(256, CBA_NE_MINWAVES) void k_normal_eq(const Tile* __restrict__ tiles, int64_t n_tiles,
                                                   const T* __restrict__ bc, const T* __restrict__ intr,
                                                   const T* __restrict__ sd, const int32_t* __restrict__ blk_cam,
                                                   const T* __restrict__ X, const T* __restrict__ Y,
                                                   const T* __restrict__ u, const T* __restrict__ v,
                                                   double* __restrict__ partial) {
    constexpr int PI = IntrSize<MODEL>::value;
    constexpr int PL = LocalCols<CHAIN, MODEL>::value;
    constexpr int NACC = PL * (PL + 1) / 2 + PL + 1;
    const int64_t w = wave_index();
    if (w >= n_tiles) return;
    const Tile t = tiles[w];
    const int lane = threadIdx.x & 63;
    const int cam = blk_cam[t.blk];
    normal_eq_tile_split<CHAIN, MODEL, SPLIT, PART, T>(t, lane, bc + static_cast<int64_t>(t.blk) * BC_SIZE, intr + static_cast<int64_t>(cam) * PI,
                                                       sd + static_cast<int64_t>(cam) * SD_SIZE, X, Y, u, v, partial + w * NACC);
}

// ---- Mode B, two-pose chains: moments --------------------------------------------------------------------------
// For the EXTRINSIC / BUNDLE chains the 12 pose columns of an observation are d(u,v)/dP times a 3 x 12 matrix affine in the
// target point (reproj_math.hpp, pose_affine_G), so the pose-pose, pose-gradient and pose-intrinsics blocks are linear images
// of a few MOMENTS: sum m_a m_b (du du^T + dv dv^T), sum m_a (du r_u + dv r_v), sum m_a (du Jui^T + dv Jvi^T), m = (1, X, Y).
// Per observation that is 279 fused operations on 201 accumulators (P = 22) instead of 439 on 270, the 12 pose columns are
// never formed, and three launches suffice where the direct form needs four.  k_mom_expand then builds the packed
// [H | g | s] row of every block, so everything downstream (weights, camera sums, Schur step, covariance) is unchanged.
template <int MODEL, int NPARTS, int PART, typename T>
__global__ __launch_bounds__(256) void k_normal_eq_mom(const Tile* __restrict__ tiles, int64_t n_tiles, const T* __restrict__ bc,
                                                       const T* __restrict__ intr, const T* __restrict__ sd,
                                                       const int32_t* __restrict__ blk_cam, const T* __restrict__ X,
                                                       const T* __restrict__ Y, const T* __restrict__ u, const T* __restrict__ v,
                                                       double* __restrict__ partial) {
    constexpr int PI = IntrSize<MODEL>::value;
    constexpr int NMOM = MomLayout<PI>::N;
    constexpr int NLOC = MomSplit<PI, NPARTS>::T.count[PART];
    constexpr int NPAD = TransposeSum<16>::pad(NLOC);
    const int64_t w = wave_index();
    if (w >= n_tiles) return;
    const Tile t = tiles[w];
    const int lane = threadIdx.x & 63;
    const T* bcp = bc + static_cast<int64_t>(t.blk) * BC_SIZE;
    const int cam = blk_cam[t.blk];
    const T* ip = intr + static_cast<int64_t>(cam) * PI;
    const T* sp = sd + static_cast<int64_t>(cam) * SD_SIZE;
    double acc[NPAD];
#pragma unroll
    for (int e = 0; e < NPAD; ++e) acc[e] = 0.0;
    T xc = T(0), yc = T(0), uc = T(0), vc = T(0);
    if (lane < t.count) { xc = X[t.xy_start + lane]; yc = Y[t.xy_start + lane]; uc = u[t.start + lane]; vc = v[t.start + lane]; }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see k_normal_eq
#pragma unroll 1
    for (int k = 0;; ++k) {  // (until the tile is done: the break below)
        const int j = lane + 64 * k;
        if (64 * k >= t.count) break;  // wave-uniform
        T xn = T(0), yn = T(0), un = T(0), vn = T(0);
        if (j + 64 < t.count) {
            const int64_t i = t.start + j + 64, k2 = t.xy_start + j + 64;
            xn = X[k2]; yn = Y[k2]; un = u[i]; vn = v[i];
        }
        if (j < t.count) mom_point<MODEL, NPARTS, PART, T>(bcp, ip, sp, xc, yc, uc, vc, acc);
        xc = xn; yc = yn; uc = un; vc = vn;
    }
    bool owner;
    const int base = wave_transpose_sum<NPAD>(acc, lane, &owner);
    double* out = partial + w * NMOM;
#pragma unroll
    for (int j = 0; j < TransposeSum<NPAD>::CNT; ++j) {
        if (owner && base + j < NLOC) out[MomSplit<PI, NPARTS>::T.entry[PART][base + j]] = acc[j];
    }
}

// blk_acc[b] = packed [H | g | s] of block b from its moment row; one wavefront per block, G and T = Qh Gh in LDS.
// blk_tile_off != nullptr: `blk_mom` holds one row per TILE and the block's row is their sum in tile order (k_tile_sum's work and
// order); blk_w != nullptr: the thread that forms s = |r|^2 also leaves the block's robust weight and s (k_weights' work) - two
// launches less per LM step.
template <int CHAIN, int PI>
__global__ __launch_bounds__(64) void k_mom_expand(const double* __restrict__ gate, int n_blocks, const double* __restrict__ bc, const double* __restrict__ blk_mom,
                                                   double* __restrict__ blk_acc, const int64_t* __restrict__ blk_tile_off = nullptr,
                                                   double huber_delta = 0.0, double* __restrict__ blk_w = nullptr, double* __restrict__ blk_s = nullptr) {
    if (gate && *gate == 0.0) return;
    constexpr int PL = 12 + PI, NH = PL * (PL + 1) / 2, NACC = NH + PL + 1, NMOM = MomLayout<PI>::N;
    static constexpr UpperIndex<PL> UI{};
    __shared__ double G[3][36];
    __shared__ double T[9 * 12];
    __shared__ double mom[NMOM];
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    if (threadIdx.x < 3) pose_affine_G_part<CHAIN>(bc + static_cast<int64_t>(b) * BC_SIZE, threadIdx.x, G[threadIdx.x]);
    if (blk_tile_off) {
        const int64_t t0 = blk_tile_off[b], t1 = blk_tile_off[b + 1];
        for (int e = threadIdx.x; e < NMOM; e += 64) {
            double s = 0.0;
            for (int64_t t = t0; t < t1; ++t) s += blk_mom[t * NMOM + e];
            mom[e] = s;
        }
    } else {
        for (int e = threadIdx.x; e < NMOM; e += 64) mom[e] = blk_mom[static_cast<int64_t>(b) * NMOM + e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * 12; e += 64) T[e] = mom_expand_T<PI>(mom, G, e / 12, e % 12);
    __syncthreads();
    for (int e = threadIdx.x; e < NACC; e += 64) {
        const int i = e < NH ? UI.i[e] : 0, j = e < NH ? UI.j[e] : 0;
        const double val = mom_expand_entry_T<PI>(mom, G, T, e, i, j);
        blk_acc[static_cast<int64_t>(b) * NACC + e] = val;
        if (e == NACC - 1 && blk_w) {
            double rho, w;
            huber(val, huber_delta, &rho, &w);
            blk_w[b] = w;
            blk_s[b] = val;
        }
    }
}

// ---- launchers --------------------------------------------------------------------------------
#define CBA_DISPATCH(e, CALL)                                                                     \
    switch ((e).chain * 2 + (e).model) {                                                          \
        case 0: { CALL(CH_INTRINSIC, CAM_PINHOLE_BC) } break;                                     \
        case 1: { CALL(CH_INTRINSIC, CAM_SCHEIMPFLUG) } break;                                    \
        case 2: { CALL(CH_EXTRINSIC, CAM_PINHOLE_BC) } break;                                     \
        case 3: { CALL(CH_EXTRINSIC, CAM_SCHEIMPFLUG) } break;                                    \
        case 4: { CALL(CH_BUNDLE, CAM_PINHOLE_BC) } break;                                        \
        case 5: { CALL(CH_BUNDLE, CAM_SCHEIMPFLUG) } break;                                       \
        default: throw std::runtime_error("bad chain/model");                                     \
    }

static inline unsigned blocks_for(int64_t n, int per) { return static_cast<unsigned>((n + per - 1) / per); }

// When every residual block is a single Mode B / R tile (views of at most TILE_B observations: every size the reference's tests
// use) the per-tile row IS the per-block row: the tile kernels write the block arrays directly and k_tile_sum is not launched.
static inline bool one_tile_per_block(const Engine& e) { return e.n_tilesB == e.n_blocks; }

void ensure_f32_buffers(Engine& e) {
    if (e.uf.n >= static_cast<size_t>(e.ld) && e.bcf.n > 0) return;
    const DevBuf<double>* src[4] = {&e.X, &e.Y, &e.u, &e.v};
    DevBuf<float>* dst[4] = {&e.Xf, &e.Yf, &e.uf, &e.vf};
    for (int a = 0; a < 4; ++a) {
        const int64_t n = a < 2 ? e.ld_xy : e.ld;
        dst[a]->alloc(static_cast<size_t>(n));
        hipLaunchKernelGGL(k_to_f32, dim3(blocks_for(n, 256)), dim3(256), 0, e.stream, n, src[a]->p, dst[a]->p);
    }
    e.bcf.alloc(static_cast<size_t>(std::max(1, e.n_blocks)) * BC_SIZE);
    e.sdf.alloc(static_cast<size_t>(e.n_cams) * SD_SIZE);
    e.intrf.alloc(static_cast<size_t>(e.n_cams) * e.PI);
    CBA_HIP(hipMemsetAsync(e.sdf.p, 0, e.sdf.n * sizeof(float), e.stream));
    CBA_HIP(hipGetLastError());
    CBA_HIP(hipStreamSynchronize(e.stream));
}

void launch_block_consts(Engine& e, int which) {
    e.active = which;
    if (e.n_blocks == 0) return;
    const unsigned g = blocks_for(e.n_blocks, 128);
    const double* cam = e.cam[which].p;
    const double* view = e.view[which].p;
    const double* target = e.target[which].p;
    float* bcf = e.scalar ? e.bcf.p : nullptr;
    switch (e.chain) {
        case CH_INTRINSIC:
            hipLaunchKernelGGL(k_block_consts<CH_INTRINSIC>, dim3(g), dim3(128), 0, e.stream, e.gate, e.n_blocks, e.d_blk_cam.p,
                               e.d_blk_view.p, cam, view, target, e.aux.p, e.bc.p, bcf);
            break;
        case CH_EXTRINSIC:
            hipLaunchKernelGGL(k_block_consts<CH_EXTRINSIC>, dim3(g), dim3(128), 0, e.stream, e.gate, e.n_blocks, e.d_blk_cam.p,
                               e.d_blk_view.p, cam, view, target, e.aux.p, e.bc.p, bcf);
            break;
        default:
            hipLaunchKernelGGL(k_block_consts<CH_BUNDLE>, dim3(g), dim3(128), 0, e.stream, e.gate, e.n_blocks, e.d_blk_cam.p,
                               e.d_blk_view.p, cam, view, target, e.aux.p, e.bc.p, bcf);
    }
    if (e.model == CAM_SCHEIMPFLUG)
        hipLaunchKernelGGL(k_scheimpflug_consts, dim3(blocks_for(e.n_cams, 64)), dim3(64), 0, e.stream, e.gate, e.n_cams,
                           e.intr[which].p, e.sd.p, e.scalar ? e.sdf.p : nullptr);
    if (e.scalar)
        hipLaunchKernelGGL(k_to_f32, dim3(blocks_for(static_cast<int64_t>(e.n_cams) * e.PI, 64)), dim3(64), 0, e.stream,
                           static_cast<int64_t>(e.n_cams) * e.PI, e.intr[which].p, e.intrf.p);
    CBA_HIP(hipGetLastError());
}

// the per-camera part of launch_block_consts alone (the caller has built the block constants itself: backend_hip.hip k_step_head)
void launch_camera_consts(Engine& e, int which) {
    e.active = which;
    if (e.model == CAM_SCHEIMPFLUG)
        hipLaunchKernelGGL(k_scheimpflug_consts, dim3(blocks_for(e.n_cams, 64)), dim3(64), 0, e.stream, e.gate, e.n_cams,
                           e.intr[which].p, e.sd.p, static_cast<float*>(nullptr));
    CBA_HIP(hipGetLastError());
}

// bc/sd were built from parameter copy e.active; the kernels read the matching intrinsics
static const double* intr_of(Engine& e) { return e.intr[e.active].p; }

template <int C, int M, bool NT, int ROWS>
static void launch_eval_v(Engine& e) {
    // one launch per output segment (engine.hpp Jseg: a blocked output above 4 GiB is a few contiguous blocks of whole tiles): the
    // kernel indexes its output by the tile number relative to the tile table it is given
    const int64_t tw = static_cast<int64_t>(2 + 2 * e.PL) * TILE_A;
    const size_t nseg = e.Jseg.empty() ? 1 : e.Jseg.size();
    for (size_t k = 0; k < nseg; ++k) {
        const int64_t t0 = e.Jseg.empty() ? 0 : static_cast<int64_t>(k) * e.seg_tiles;
        const int64_t nt = e.Jseg.empty() ? e.n_tilesA : std::min<int64_t>(e.seg_tiles, e.n_tilesA - t0);
        double* Jk = e.Jseg.empty() ? e.J.p : e.Jseg[k].p;
        (void)tw;
        const unsigned g = blocks_for(nt, 4 * ROWS);
#define CBA_EVAL_ARGS dim3(g), dim3(256), 0, e.stream, e.tilesA.p + t0, nt, e.bc.p, intr_of(e), e.sd.p, e.d_blk_cam.p, e.X.p, \
                      e.Y.p, e.u.p, e.v.p, e.r.p, Jk, e.ld
#ifdef CBA_EXPERIMENTS  // timing-only ablations of k_eval (outputs are wrong): experiment builds only
        if (ROWS == 1 && NT && e.eval_blocked && e.eval_ablate == 1)
            hipLaunchKernelGGL((k_eval<C, M, NT, 1, true, 1>), CBA_EVAL_ARGS);
        else if (ROWS == 1 && NT && e.eval_blocked && e.eval_ablate == 2)
            hipLaunchKernelGGL((k_eval<C, M, NT, 1, true, 2>), CBA_EVAL_ARGS);
        else
#endif
        if (e.eval_blocked)
            hipLaunchKernelGGL((k_eval<C, M, NT, ROWS, true>), CBA_EVAL_ARGS);
        else
            hipLaunchKernelGGL((k_eval<C, M, NT, ROWS, false>), CBA_EVAL_ARGS);
#undef CBA_EVAL_ARGS
    }
}

template <int C, int M>
static void launch_eval_f32(Engine& e) {  // fp32 study: tile-blocked, non-temporal, one tile per wave
    const unsigned g = blocks_for(e.n_tilesA, 4);
    hipLaunchKernelGGL((k_eval<C, M, true, 1, true, 0, float>), dim3(g), dim3(256), 0, e.stream, e.tilesA.p, e.n_tilesA, e.bcf.p,
                       e.intrf.p, e.sdf.p, e.d_blk_cam.p, e.Xf.p, e.Yf.p, e.uf.p, e.vf.p, static_cast<float*>(nullptr), e.Jf.p, e.ld);
}

void launch_eval(Engine& e) {
    if (e.n_tilesA == 0) return;
    if (e.scalar) {
#define CALL(C, M) launch_eval_f32<C, M>(e);
        CBA_DISPATCH(e, CALL)
#undef CALL
        CBA_HIP(hipGetLastError());
        return;
    }
    // tuning knob (experiments only; read at handle creation): bit 0 = nt stores, bits 1.. = log2(tiles per wave)
    const int variant = e.eval_variant;
#define CALL(C, M)                                                         \
    switch (variant) {                                                     \
        case 1: launch_eval_v<C, M, true, 1>(e); break;                    \
        case 2: launch_eval_v<C, M, false, 2>(e); break;                   \
        case 3: launch_eval_v<C, M, true, 2>(e); break;                    \
        case 4: launch_eval_v<C, M, false, 4>(e); break;                   \
        case 5: launch_eval_v<C, M, true, 4>(e); break;                    \
        case 6: launch_eval_v<C, M, false, 8>(e); break;                   \
        case 7: launch_eval_v<C, M, true, 8>(e); break;                    \
        default: launch_eval_v<C, M, false, 1>(e);                         \
    }
    CBA_DISPATCH(e, CALL)
#undef CALL
    CBA_HIP(hipGetLastError());
}

void launch_resid(Engine& e) {
    if (e.n_tilesB == 0) return;
    const unsigned g = blocks_for(e.n_tilesB, 4);
    double* rows = one_tile_per_block(e) ? e.blk_s.p : e.partial.p;
#define RESID_F64(M) hipLaunchKernelGGL((k_resid<M, double>), dim3(g), dim3(256), 0, e.stream, e.tilesB.p, e.n_tilesB, e.bc.p, \
                                        intr_of(e), e.sd.p, e.d_blk_cam.p, e.X.p, e.Y.p, e.u.p, e.v.p, rows)
#define RESID_F32(M) hipLaunchKernelGGL((k_resid<M, float>), dim3(g), dim3(256), 0, e.stream, e.tilesB.p, e.n_tilesB, e.bcf.p, \
                                        e.intrf.p, e.sdf.p, e.d_blk_cam.p, e.Xf.p, e.Yf.p, e.uf.p, e.vf.p, rows)
    if (e.model == CAM_PINHOLE_BC) { if (e.scalar) RESID_F32(CAM_PINHOLE_BC); else RESID_F64(CAM_PINHOLE_BC); }
    else { if (e.scalar) RESID_F32(CAM_SCHEIMPFLUG); else RESID_F64(CAM_SCHEIMPFLUG); }
#undef RESID_F64
#undef RESID_F32
    if (!one_tile_per_block(e))
        hipLaunchKernelGGL(k_tile_sum, dim3(blocks_for(e.n_blocks, 256)), dim3(256), 0, e.stream, e.gate, e.n_blocks, 1,
                           e.d_blk_tile_off.p, e.partial.p, e.blk_s.p);
    CBA_HIP(hipGetLastError());
}

void launch_cost(Engine& e, double huber_delta, double* out) {
    if (!out) out = e.scalar_out.p;
    if (e.n_blocks > 4096) {
        const int n_part = (e.n_blocks + 2047) / 2048;
        if (e.cost_part.n < static_cast<size_t>(2 * n_part)) e.cost_part.alloc(static_cast<size_t>(2 * n_part));
        hipLaunchKernelGGL(k_cost_partial, dim3(n_part), dim3(256), 0, e.stream, e.n_blocks, e.blk_s.p, huber_delta, e.cost_part.p);
        hipLaunchKernelGGL(k_cost_final, dim3(1), dim3(64), 0, e.stream, n_part, e.cost_part.p, out);
    } else
        hipLaunchKernelGGL(k_cost, dim3(1), dim3(256), 0, e.stream, e.n_blocks, e.blk_s.p, huber_delta, out);
    CBA_HIP(hipGetLastError());
}

template <int C, int M, class SPLIT, int PART>
static void launch_ne_part(Engine& e, unsigned g) {
    double* rows = one_tile_per_block(e) ? e.blk_acc.p : e.partial.p;
    if (e.scalar)
        hipLaunchKernelGGL((k_normal_eq<C, M, SPLIT, PART, float>), dim3(g), dim3(256), 0, e.stream, e.tilesB.p, e.n_tilesB, e.bcf.p,
                           e.intrf.p, e.sdf.p, e.d_blk_cam.p, e.Xf.p, e.Yf.p, e.uf.p, e.vf.p, rows);
    else
        hipLaunchKernelGGL((k_normal_eq<C, M, SPLIT, PART, double>), dim3(g), dim3(256), 0, e.stream, e.tilesB.p, e.n_tilesB, e.bc.p,
                           intr_of(e), e.sd.p, e.d_blk_cam.p, e.X.p, e.Y.p, e.u.p, e.v.p, rows);
}
template <int C, int M>
static void launch_ne(Engine& e, unsigned g) {
    if (e.modeb_shared && launch_normal_eq_shared_rows(e, one_tile_per_block(e) ? e.blk_acc.p : e.partial.p)) return;  // kernels_modeb.hip
    if constexpr (C == CH_INTRINSIC) {  // P = 16 / 18: 153 / 190 accumulators -> pose rows (87 / 99) | intrinsics block (66 / 91)
        // measured (profiles/r02_modeb_variants.jsonl): pose rows | intrinsics block wins for Scheimpflug (C5 0.357 -> 0.329 ms: the
        // intrinsics part drops the pose columns from its row evaluation) and loses for pinhole (C2 0.219 -> 0.222..0.237 ms: its
        // 87-accumulator part needs 262 registers); CBA_MODEB_SPLIT=0|1 forces one form
        const bool split = e.modeb_split < 0 ? M == CAM_SCHEIMPFLUG : e.modeb_split != 0;
        if (split) {
            launch_ne_part<C, M, SplitPoseIntr, 0>(e, g);
            launch_ne_part<C, M, SplitPoseIntr, 1>(e, g);
        } else {
            launch_ne_part<C, M, SplitRoundRobin<2>, 0>(e, g);
            launch_ne_part<C, M, SplitRoundRobin<2>, 1>(e, g);
        }
    } else {  // P = 22 / 24: 276 / 325 accumulators -> 4 parts
        launch_ne_part<C, M, SplitRoundRobin<4>, 0>(e, g);
        launch_ne_part<C, M, SplitRoundRobin<4>, 1>(e, g);
        launch_ne_part<C, M, SplitRoundRobin<4>, 2>(e, g);
        launch_ne_part<C, M, SplitRoundRobin<4>, 3>(e, g);
    }
}

void warm_reproj_kernels() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_tile_sum));
}

template <int M, int NP, int PART>
static void launch_mom_part(Engine& e, unsigned g) {
    constexpr int NMOM = MomLayout<IntrSize<M>::value>::N;
    static_assert(NMOM <= 256, "blk_mom row stride");
    double* rows = one_tile_per_block(e) ? e.blk_mom.p : e.partial.p;  // both with row stride NMOM
    if (e.scalar)
        hipLaunchKernelGGL((k_normal_eq_mom<M, NP, PART, float>), dim3(g), dim3(256), 0, e.stream, e.tilesB.p, e.n_tilesB, e.bcf.p,
                           e.intrf.p, e.sdf.p, e.d_blk_cam.p, e.Xf.p, e.Yf.p, e.uf.p, e.vf.p, rows);
    else
        hipLaunchKernelGGL((k_normal_eq_mom<M, NP, PART, double>), dim3(g), dim3(256), 0, e.stream, e.tilesB.p, e.n_tilesB, e.bc.p,
                           intr_of(e), e.sd.p, e.d_blk_cam.p, e.X.p, e.Y.p, e.u.p, e.v.p, rows);
}

template <int C, int M>
static void launch_mom(Engine& e, unsigned g) {
    constexpr int PI = IntrSize<M>::value;
    if (e.modeb_shared && launch_normal_eq_shared_rows(e, one_tile_per_block(e) ? e.blk_mom.p : e.partial.p)) {
        // one workgroup of 3 / 4 wavefronts per tile, rows evaluated once (kernels_modeb.hip)
    } else if constexpr (M == CAM_PINHOLE_BC) {  // 201 sums -> 3 parts of 67
        launch_mom_part<M, 3, 0>(e, g); launch_mom_part<M, 3, 1>(e, g); launch_mom_part<M, 3, 2>(e, g);
    } else {                    // 244 sums -> 4 parts of 61
        launch_mom_part<M, 4, 0>(e, g); launch_mom_part<M, 4, 1>(e, g); launch_mom_part<M, 4, 2>(e, g); launch_mom_part<M, 4, 3>(e, g);
    }
    // the tile sum and (when the caller gave a Huber parameter, Engine::head_huber) the block weights ride in the expansion
    const bool w = e.head_huber >= 0.0;
    const bool tiles = !one_tile_per_block(e);
    hipLaunchKernelGGL((k_mom_expand<C, PI>), dim3(e.n_blocks), dim3(64), 0, e.stream, e.gate, e.n_blocks, e.bc.p, tiles ? e.partial.p : e.blk_mom.p,
                       e.blk_acc.p, tiles ? e.d_blk_tile_off.p : static_cast<const int64_t*>(nullptr), e.head_huber, w ? e.blk_w.p : static_cast<double*>(nullptr),
                       w ? e.blk_s.p : static_cast<double*>(nullptr));
    e.head_weights = w;
}

void launch_normal_eq(Engine& e) {
    e.head_weights = false;
    if (e.n_tilesB == 0) return;
    const unsigned g = blocks_for(e.n_tilesB, 4);
    if (e.chain != CH_INTRINSIC && e.modeb_moments) {
        if (e.chain == CH_EXTRINSIC) { if (e.model == CAM_PINHOLE_BC) launch_mom<CH_EXTRINSIC, CAM_PINHOLE_BC>(e, g); else launch_mom<CH_EXTRINSIC, CAM_SCHEIMPFLUG>(e, g); }
        else { if (e.model == CAM_PINHOLE_BC) launch_mom<CH_BUNDLE, CAM_PINHOLE_BC>(e, g); else launch_mom<CH_BUNDLE, CAM_SCHEIMPFLUG>(e, g); }
        CBA_HIP(hipGetLastError());
        return;
    }
#define CALL(C, M) launch_ne<C, M>(e, g);
    CBA_DISPATCH(e, CALL)
#undef CALL
    const int64_t tot = static_cast<int64_t>(e.n_blocks) * e.NACC;
    if (!one_tile_per_block(e)) {
        const bool w = e.head_huber >= 0.0;
        hipLaunchKernelGGL(k_tile_sum, dim3(blocks_for(tot, 256)), dim3(256), 0, e.stream, e.gate, e.n_blocks, e.NACC,
                           e.d_blk_tile_off.p, e.partial.p, e.blk_acc.p, w ? e.NACC - 1 : -1, e.head_huber, e.blk_w.p, e.blk_s.p);
        e.head_weights = w;
    }
    CBA_HIP(hipGetLastError());
}

}  // namespace cba
