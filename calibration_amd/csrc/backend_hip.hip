// backend_hip.hip — the HIP implementation of cba::Backend (lm_core.hpp) on an Engine, plus the
// collective glue (host callback or RCCL over xGMI).
//
// Kernels here are the O(#views) / O(#blocks) part of one LM step.  The bodies:
//   weights          rho'(s_b) per residual block (ceres HuberLoss + corrector)
//   cam_partial      weighted per-camera sums of the block normal equations (chunked, fixed order), seg_sum their totals
//   schur_view_wave  per private view, one wavefront: damped H_pp = L L^T, y = L^-1 g_p, Z_b = L^-1 E_b
//   schur_syrk       S_schur = sum_v Z_v^T Z_v (+ g_schur = sum_v Z_v^T y_v), 64x64 output tiles, VALU or fp64 MFMA
//   backsub          delta_p, trial poses, step norms and the views' share of the model-cost terms
// In the LM loop the bodies that do not depend on each other share launches (k_step_head, k_sys_stage2, k_sys_stage3, k_sys_pack:
// "the fused stages" below); the one-kernel-per-body launches remain for the paths off the loop.  The decision itself is the
// controller kernel's (lm_ctl.hip), queued behind the exchange by the ctl_* functions of HipBackend.
// All reductions are two-stage with a fixed summation order (no atomics on fp64), so runs are
// bitwise reproducible and 1/2/4/8-rank runs differ only by the all-reduce's own rounding.
#include <rccl/rccl.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <set>

#include "engine.hpp"
#include "lm_core.hpp"
#include "lm_state.hpp"
#include "schur_math.hpp"
#include "wave_reduce.hpp"

namespace cba {

constexpr int VCHUNK = 8;    // views per syrk / gvec workgroup
constexpr size_t CTL_REC_FETCH = 2 * CS_COUNT + 8;  // HipLMState::ctl_rec = [control record | staged scalars + lmp | fetched parameters]
constexpr int CCHUNK = 16;   // blocks per camera-sum chunk

__global__ void k_weights(int n_blocks, int NACC, int s_idx, const double* __restrict__ blk_acc, double huber_delta,
                          double* __restrict__ blk_w, double* __restrict__ blk_s) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const double s = blk_acc[static_cast<int64_t>(b) * NACC + s_idx];
    double rho, w;
    huber(s, huber_delta, &rho, &w);
    blk_w[b] = w;
    blk_s[b] = s;
}

// k_weights and k_cost (kernels_reproj.hip) in one launch for up to 4096 blocks: the same per-thread strides and the same LDS
// tree as k_cost, so the cost is bit-identical to the two-kernel form.  out = {1/2 sum rho(s_b), sum s_b} (may be pinned host memory)
__global__ __launch_bounds__(256) void k_weights_cost(int n_blocks, int NACC, int s_idx, const double* __restrict__ blk_acc,
                                                      double huber_delta, double* __restrict__ blk_w, double* __restrict__ blk_s,
                                                      double* __restrict__ out) {
    __shared__ double sh[2][256];
    double c = 0.0, ss = 0.0;
    // up to 16 independent loads in flight per thread (each is a cache line of its own: one block's |r|^2), summed in the same order
    constexpr int NQ = 16;
    for (int b0 = static_cast<int>(threadIdx.x); b0 < n_blocks; b0 += NQ * 256) {
        double sv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int b = b0 + q * 256;
            sv[q] = b < n_blocks ? blk_acc[static_cast<int64_t>(b) * NACC + s_idx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int b = b0 + q * 256;
            if (b < n_blocks) {
                double rho, w;
                huber(sv[q], huber_delta, &rho, &w);
                blk_w[b] = w;
                blk_s[b] = sv[q];
                c += 0.5 * rho;
                ss += sv[q];
            }
        }
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

// partial[k][e] = sum over chunk k's blocks (in list order) of w_b * acc[b][e]
__device__ __forceinline__ void cam_partial_body(int k, int NACC, const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ cam_blk,
                                                 const double* __restrict__ blk_w, const double* __restrict__ blk_acc,
                                                 double* __restrict__ partial) {
    const int64_t p0 = chunk_off[k], p1 = chunk_off[k + 1];
    for (int e = threadIdx.x; e < NACC; e += blockDim.x) {
        double s = 0.0;
        for (int64_t p = p0; p < p1; p += CCHUNK) {  // the chunk's block rows in flight together, added in list order
            double a[CCHUNK], w[CCHUNK];
#pragma unroll
            for (int q = 0; q < CCHUNK; ++q) {
                const bool in = p + q < p1;
                const int b = in ? cam_blk[p + q] : 0;
                w[q] = in ? blk_w[b] : 0.0;
                a[q] = in ? blk_acc[static_cast<int64_t>(b) * NACC + e] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < CCHUNK; ++q)
                if (p + q < p1) s += w[q] * a[q];
        }
        partial[static_cast<int64_t>(k) * NACC + e] = s;
    }
}
__global__ void k_cam_partial(int NACC, const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ cam_blk,
                              const double* __restrict__ blk_w, const double* __restrict__ blk_acc,
                              double* __restrict__ partial) {
    cam_partial_body(blockIdx.x, NACC, chunk_off, cam_blk, blk_w, blk_acc, partial);
}

// Column sums of a row-major partial table in a FIXED order, 8 row groups per column: thread (column c, group r)
// adds rows r, r+8, r+16, ... and the 8 group sums are combined in group order through LDS — 8x the parallelism
// and 1/8 the dependent chain of one thread per column (125 chunk rows at 1000 views, 500 at 4000).
constexpr int RS_COLS = 32, RS_GROUPS = 8;

__device__ __forceinline__ double grouped_column_sum(const double* __restrict__ rows, int64_t t0, int64_t t1, int64_t width,
                                                     int64_t e, bool valid, double (*sh)[RS_COLS]) {
    const int c = threadIdx.x % RS_COLS, r = threadIdx.x / RS_COLS;
    double s = 0.0;
    if (valid)
        for (int64_t t = t0 + r; t < t1; t += RS_GROUPS) s += rows[t * width + e];
    sh[r][c] = s;
    __syncthreads();
    double tot = 0.0;
    if (r == 0)
        for (int k = 0; k < RS_GROUPS; ++k) tot += sh[k][c];
    __syncthreads();
    return tot;
}

// out[o][e] = sum_{t in [seg[o], seg[o+1])} rows[t][e]; grid (ceil(width / 32), n_out), 256 threads
__device__ __forceinline__ void seg_sum_body(double* lds, int bx, int o, int width, const int64_t* __restrict__ seg, const double* __restrict__ rows,
                                             double* __restrict__ out) {
    double (*sh)[RS_COLS] = reinterpret_cast<double (*)[RS_COLS]>(lds);
    const int64_t e = static_cast<int64_t>(bx) * RS_COLS + threadIdx.x % RS_COLS;
    const double tot = grouped_column_sum(rows, seg[o], seg[o + 1], width, e, e < width, sh);
    if (threadIdx.x < RS_COLS && e < width) out[static_cast<int64_t>(o) * width + e] = tot;
}
__global__ __launch_bounds__(RS_COLS * RS_GROUPS) void k_seg_sum(int n_out, int width, const int64_t* __restrict__ seg,
                                                                  const double* __restrict__ rows, double* __restrict__ out) {
    __shared__ double lds[RS_GROUPS * RS_COLS];
    (void)n_out;
    seg_sum_body(lds, blockIdx.x, blockIdx.y, width, seg, rows, out);
}

// out[e] = sum_{t < n_rows} rows[t][e]; grid ceil(width / 32), 256 threads
__global__ __launch_bounds__(RS_COLS * RS_GROUPS) void k_row_sum(int64_t n_rows, int64_t width, const double* __restrict__ rows,
                                                                  double* __restrict__ out) {
    __shared__ double sh[RS_GROUPS][RS_COLS];
    const int64_t e = static_cast<int64_t>(blockIdx.x) * RS_COLS + threadIdx.x % RS_COLS;
    const double tot = grouped_column_sum(rows, 0, n_rows, width, e, e < width, sh);
    if (threadIdx.x < RS_COLS && e < width) out[e] = tot;
}

// single workgroup: out[c] = sum_i in[i*w + c] (c < w <= 4); if aux: out[w] = max_i aux[i] and out[w + 1] = #{i : aux[i] < 0}
// (k_schur_view marks a view whose damped H_pp is not positive definite with -1).  `out` may be page-locked host memory.
constexpr int COL_REDUCE_LDS = 6 * 256;  // doubles of LDS scratch (the fused stages hand every body a piece of ONE buffer: the
                                         // compiler does not overlay the static LDS of branches that exclude each other)
__device__ __forceinline__ void col_reduce_body(double* lds, int n, int w, const double* __restrict__ in, const double* __restrict__ aux,
                                                double* __restrict__ out) {
    double (*sh)[256] = reinterpret_cast<double (*)[256]>(lds);
    double acc[4] = {0, 0, 0, 0}, mx = 0.0, bad = 0.0;
    for (int i = static_cast<int>(threadIdx.x); i < n; i += 256) {
        for (int c = 0; c < w; ++c) acc[c] += in[static_cast<int64_t>(i) * w + c];
        if (aux) { mx = fmax(mx, aux[i]); bad += aux[i] < 0.0 ? 1.0 : 0.0; }
    }
    for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] = acc[c];
    sh[4][threadIdx.x] = mx;
    sh[5][threadIdx.x] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (static_cast<int>(threadIdx.x) < o) {
            for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
            sh[4][threadIdx.x] = fmax(sh[4][threadIdx.x], sh[4][threadIdx.x + o]);
            sh[5][threadIdx.x] += sh[5][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int c = 0; c < w; ++c) out[c] = sh[c][0];
        if (aux) { out[w] = sh[4][0]; out[w + 1] = sh[5][0]; }
    }
}
__global__ __launch_bounds__(256) void k_col_reduce(int n, int w, const double* __restrict__ in, const double* __restrict__ aux,
                                                    double* __restrict__ out, const double* __restrict__ gate = nullptr) {
    __shared__ double lds[COL_REDUCE_LDS];
    if (gate && *gate == 0.0) return;  // (kernels_reproj.hip k_block_consts: a launch queued ahead of the decision it depends on)
    col_reduce_body(lds, n, w, in, aux, out);
}

__global__ void k_schur_view(SchurDims d, int n_views, const int64_t* __restrict__ link_off, const int32_t* __restrict__ link_blk,
                             const double* __restrict__ blk_acc, const double* __restrict__ blk_w,
                             const int32_t* __restrict__ fixed, const double* __restrict__ lmp /*[radius, init_scale]*/, int constrained,
                             const double* __restrict__ view, double* __restrict__ scale2, double* __restrict__ L,
                             double* __restrict__ y, double* __restrict__ D, double* __restrict__ gp, double* __restrict__ blk_Z,
                             double* __restrict__ gmax /* -1 marks a failed elimination */) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_views) return;
    const int nb = static_cast<int>(link_off[v + 1] - link_off[v]);
    double gm = 0.0;
    const double radius = lmp[0];
    const bool init_scale = lmp[1] != 0.0;
    const bool ok = schur_view_body(d, nb, link_blk + link_off[v], blk_acc, blk_w, fixed[v] != 0, radius, init_scale,
                                    constrained != 0, view + 7 * static_cast<int64_t>(v), scale2 + 6 * static_cast<int64_t>(v),
                                    L + 36 * static_cast<int64_t>(v), y + 6 * static_cast<int64_t>(v), D + 6 * static_cast<int64_t>(v),
                                    gp + 6 * static_cast<int64_t>(v), blk_Z, &gm);
    gmax[v] = ok ? gm : -1.0;
}

// The same elimination with ONE WAVEFRONT per view (4 views per workgroup).  One thread per view is a ~100 us latency chain however
// few views there are (8-camera rig: 8 blocks x 16 shared columns of forward substitutions per view, 107 us for 500 views and 84 us
// for 4000): here every lane runs the short factor part redundantly and the lanes split the (block, column) pairs of Z.  Same
// operations per value as the serial body: bit-identical results.
__device__ __forceinline__ void schur_view_wave_body(int bx, const SchurDims& d, int n_views, const int64_t* __restrict__ link_off,
                                                     const int32_t* __restrict__ link_blk, const double* __restrict__ blk_acc,
                                                     const double* __restrict__ blk_w, const int32_t* __restrict__ fixed,
                                                     const double* __restrict__ lmp, int constrained, const double* __restrict__ view,
                                                     double* __restrict__ scale2, double* __restrict__ L, double* __restrict__ y,
                                                     double* __restrict__ D, double* __restrict__ gp, double* __restrict__ blk_Z,
                                                     double* __restrict__ gmax) {
    const int v = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_views) return;
    const int nb = static_cast<int>(link_off[v + 1] - link_off[v]);
    const int32_t* blks = link_blk + link_off[v];
    const int n_pairs = nb * d.PSH;
    if (fixed[v] != 0) {
        if (lane == 0) {
            double* Lv = L + 36 * static_cast<int64_t>(v);
            for (int i = 0; i < 36; ++i) Lv[i] = (i % 7 == 0) ? 1.0 : 0.0;
            for (int i = 0; i < 6; ++i) {
                y[6 * static_cast<int64_t>(v) + i] = 0.0; D[6 * static_cast<int64_t>(v) + i] = 0.0; gp[6 * static_cast<int64_t>(v) + i] = 0.0;
                if (lmp[1] != 0.0) scale2[6 * static_cast<int64_t>(v) + i] = 1.0;
            }
            gmax[v] = 0.0;
        }
        for (int p = lane; p < n_pairs; p += 64) {
            const int k = p / d.PSH, c = p - k * d.PSH;
            double* Z = blk_Z + static_cast<int64_t>(blks[k]) * 6 * d.PSH;
            for (int i = 0; i < 6; ++i) Z[i * d.PSH + c] = 0.0;
        }
        return;
    }
    double F[36], rd[6], gm = 0.0;
    const bool ok = schur_view_factor(d, nb, blks, blk_acc, blk_w, lmp[0], lmp[1] != 0.0, constrained != 0, view + 7 * static_cast<int64_t>(v),
                                      scale2 + 6 * static_cast<int64_t>(v), L + 36 * static_cast<int64_t>(v), y + 6 * static_cast<int64_t>(v),
                                      D + 6 * static_cast<int64_t>(v), gp + 6 * static_cast<int64_t>(v), &gm, lane == 0, F, rd);
    if (lane == 0) gmax[v] = ok ? gm : -1.0;
    if (!ok) return;
    for (int p = lane; p < n_pairs; p += 64) {
        const int k = p / d.PSH, c = p - k * d.PSH;
        const int b = blks[k];
        schur_view_zcol(d, F, rd, blk_w[b], blk_acc + static_cast<int64_t>(b) * d.NACC, c, blk_Z + static_cast<int64_t>(b) * 6 * d.PSH);
    }
}
__global__ __launch_bounds__(256) void k_schur_view_wave(SchurDims d, int n_views, const int64_t* __restrict__ link_off,
                                                         const int32_t* __restrict__ link_blk, const double* __restrict__ blk_acc,
                                                         const double* __restrict__ blk_w, const int32_t* __restrict__ fixed,
                                                         const double* __restrict__ lmp, int constrained, const double* __restrict__ view,
                                                         double* __restrict__ scale2, double* __restrict__ L, double* __restrict__ y,
                                                         double* __restrict__ D, double* __restrict__ gp, double* __restrict__ blk_Z,
                                                         double* __restrict__ gmax) {
    schur_view_wave_body(blockIdx.x, d, n_views, link_off, link_blk, blk_acc, blk_w, fixed, lmp, constrained, view, scale2, L, y, D, gp, blk_Z, gmax);
}

// ... and the back-substitution: the lanes split a = Z d_c over the (block, column) pairs (fixed assignment, fixed-order DPP
// wave sums: deterministic; rounding differs from the serial body's summation order), lane 63 finishes
__global__ __launch_bounds__(256) void k_backsub_wave(SchurDims d, int n_views, const int64_t* __restrict__ link_off,
                                                      const int32_t* __restrict__ link_blk, const int32_t* __restrict__ blk_cam,
                                                      const double* __restrict__ blk_Z, const double* __restrict__ delta_sh,
                                                      const int32_t* __restrict__ fixed, const double* __restrict__ L, const double* __restrict__ y,
                                                      const double* __restrict__ D, const double* __restrict__ gp, const double* __restrict__ x,
                                                      double* __restrict__ delta_p, double* __restrict__ xt, double* __restrict__ stats,
                                                      const double* __restrict__ gate = nullptr) {
    if (gate && *gate == 0.0) return;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_views) return;
    const int nb = static_cast<int>(link_off[v + 1] - link_off[v]);
    const int32_t* blks = link_blk + link_off[v];
    const bool fx = fixed[v] != 0;
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (!fx) {
        const int n_pairs = nb * d.PSH;
        for (int p = lane; p < n_pairs; p += 64) {
            const int k = p / d.PSH, c = p - k * d.PSH;
            const int b = blks[k];
            const double dc = delta_sh[blk_cam[b] * d.PC + c];
            const double* Z = blk_Z + static_cast<int64_t>(b) * 6 * d.PSH + c;
            for (int i = 0; i < 6; ++i) a[i] += Z[i * d.PSH] * dc;
        }
        for (int i = 0; i < 6; ++i) a[i] = wave_sum63(a[i]);  // total in lane 63
    }
    if (lane == 63) {
        double o4[4];
        backsub_view_finish(fx, a, L + 36 * static_cast<int64_t>(v), y + 6 * static_cast<int64_t>(v), D + 6 * static_cast<int64_t>(v),
                            gp + 6 * static_cast<int64_t>(v), x + 7 * static_cast<int64_t>(v), delta_p + 6 * static_cast<int64_t>(v),
                            xt + 7 * static_cast<int64_t>(v), o4);
        for (int k = 0; k < 4; ++k) stats[4 * static_cast<int64_t>(v) + k] = o4[k];
    }
}

// Stage the 6 * VCHUNK rows of Z of one view chunk for the 64 columns from c0 (Zs[6 * (v - v0) + k][c]; zero past the shared block,
// past the last view and where the view does not see the column's camera).  Two phases so that no load depends on another one:
// the chunk's (view, camera) -> block table goes to LDS first, then every thread's Z entries are independent, unconditional
// loads (the chain view -> block -> Z row, 24 times in sequence per thread, was most of the syrk kernels' time).
constexpr int SYRK_MAX_CAMS = 64;
__device__ __forceinline__ void stage_block_table(const SchurDims& d, int n_views, int v0, const int32_t* __restrict__ view_cam_blk, int* bsh) {
    if (d.n_cams > SYRK_MAX_CAMS) return;  // (a rig of more than 64 cameras: stage_Z reads the table from global memory)
    for (int t = threadIdx.x; t < VCHUNK * d.n_cams; t += blockDim.x) {
        const int v = v0 + t / d.n_cams;
        bsh[t] = v < n_views ? view_cam_blk[static_cast<int64_t>(v) * d.n_cams + t % d.n_cams] : -1;
    }
}
__device__ __forceinline__ void stage_Z(const SchurDims& d, const int* bsh, int n_views, int v0, const int32_t* __restrict__ view_cam_blk,
                                        const double* __restrict__ blk_Z, int c0, int nsh, double (*Zs)[64]) {
    const bool table = d.n_cams <= SYRK_MAX_CAMS;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < 6 * VCHUNK * 64; idx += 256) {
        const int row = idx >> 6, c = idx & 63, g = c0 + c;
        const int gg = g < nsh ? g : 0;
        const int cam = gg / d.PC, lc = gg - cam * d.PC;
        const int v = v0 + row / 6;
        const int b = table ? bsh[(row / 6) * d.n_cams + cam] : (v < n_views ? view_cam_blk[static_cast<int64_t>(v) * d.n_cams + cam] : -1);
        const double z = blk_Z[(static_cast<int64_t>(b < 0 ? 0 : b) * 6 + row % 6) * d.PSH + lc];
        Zs[row][c] = (g < nsh && b >= 0) ? z : 0.0;
    }
}

// g_schur partial of one view chunk: out[g] = sum_{v in chunk} sum_k Z_v[k][g] y_v[k] (fixed (v, k) order), from the rows of Z the
// syrk kernels have staged in LDS (Zs[6 * views + k][column - c0]) — read from global memory the chain view -> block index -> Z row
// is two dependent loads per view and was most of the kernels' time.  Run by the DIAGONAL tile pair of a chunk for its 64
// columns, so that the whole elimination result is one partial row per chunk and ONE k_row_sum.
__device__ __forceinline__ void schur_gvec_chunk(const double (*Zs)[64], const double* ysh, int nrow, int c0, int nsh, double* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= 64 || c0 + c >= nsh) return;
    double s = 0.0;
    for (int r = 0; r < nrow; ++r) s += Zs[r][c] * ysh[r];
    out[c0 + c] = s;
}

// grid (view chunks, upper tile pairs); 256 threads = 16x16, each a 4x4 micro-tile of a 64x64 tile.
// partial[chunk] = [pair][64*64] then g_schur[nsh]  (row stride n_pairs * 4096 + nsh)
constexpr int SYRK_LDS = 2 * 6 * VCHUNK * 64 + 6 * VCHUNK + VCHUNK * SYRK_MAX_CAMS / 2;  // doubles: Zi | Zj | y | block table (ints)
__device__ __forceinline__ void schur_syrk_body(double* lds, int bx, int by, int gy, const SchurDims& d, int n_views, int nsh, int n_tiles,
        const int32_t* __restrict__ view_cam_blk, const double* __restrict__ blk_Z, const double* __restrict__ y, double* __restrict__ partial) {
    // the chunk's 6 * VCHUNK rows of Z are staged in one go (one barrier per workgroup, not two per view: the staging of a
    // 10-wide shared block is all latency); the products are added in the same (view, k) order as before
    double (*Zi)[64] = reinterpret_cast<double (*)[64]>(lds), (*Zj)[64] = Zi + 6 * VCHUNK;
    double* ysh = lds + 2 * 6 * VCHUNK * 64;
    int* bsh = reinterpret_cast<int*>(ysh + 6 * VCHUNK);
    // decode the upper-triangular tile pair
    int pair = by, ti = 0;
    while (pair >= n_tiles - ti) { pair -= n_tiles - ti; ++ti; }
    const int tj = ti + pair;
    const int i0 = ti * 64, j0 = tj * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    const int v0 = bx * VCHUNK;
    const int nrow = 6 * (min(n_views, v0 + VCHUNK) - v0);
    stage_block_table(d, n_views, v0, view_cam_blk, bsh);
    __syncthreads();
    stage_Z(d, bsh, n_views, v0, view_cam_blk, blk_Z, i0, nsh, Zi);
    if (tj != ti) stage_Z(d, bsh, n_views, v0, view_cam_blk, blk_Z, j0, nsh, Zj);
    if (static_cast<int>(threadIdx.x) < 6 * VCHUNK) ysh[threadIdx.x] = static_cast<int>(threadIdx.x) < nrow ? y[6 * static_cast<int64_t>(v0) + threadIdx.x] : 0.0;
    __syncthreads();
    const double (*Zc)[64] = tj != ti ? Zj : Zi;  // a diagonal tile: both factors are the same 64 columns, staged once
    for (int r = 0; r < nrow; ++r) {
        double a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = Zi[r][ty * 4 + q]; b[q] = Zc[r][tx * 4 + q]; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] += a[p] * b[q];
    }
    double* row = partial + static_cast<int64_t>(bx) * (static_cast<int64_t>(gy) * 4096 + nsh);
    double* out = row + static_cast<int64_t>(by) * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(ty * 4 + p) * 64 + tx * 4 + q] = acc[p][q];
    if (ti == tj) schur_gvec_chunk(Zi, ysh, nrow, i0, nsh, row + static_cast<int64_t>(gy) * 4096);
}
__global__ __launch_bounds__(256) void k_schur_syrk(SchurDims d, int n_views, int nsh, int n_tiles, const int32_t* __restrict__ view_cam_blk,
        const double* __restrict__ blk_Z, const double* __restrict__ y, double* __restrict__ partial) {
    __shared__ double lds[SYRK_LDS];
    schur_syrk_body(lds, blockIdx.x, blockIdx.y, gridDim.y, d, n_views, nsh, n_tiles, view_cam_blk, blk_Z, y, partial);
}

// The same contraction on the matrix cores, used when the shared block is a real contraction (nsh >= 64: the 8-camera rig of
// BASELINE config 3 has nsh = 128, K = 6 x #views).  One workgroup = one 64x64 output tile over a chunk of VCHUNK views: the
// chunk's 6 * VCHUNK = 48 rows of Z are staged in LDS, every wavefront owns a 16 x 64 strip = four 16x16 accumulators and
// issues v_mfma_f64_16x16x4_f64 over the 12 four-row steps (A[i][k]: lane i = l & 15, k = l >> 4; B[k][j] likewise;
// D: col = l & 15, row = (l >> 4) + 4 reg).  fp64 MFMA runs at the vector-FMA rate on MI355X (78.6 TFLOP/s both), so this is
// about USING the matrix pipe where the north-star asks for it, not about speed: the kernel is ~40 us of a 6 ms LM step.
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int SYRK_ROWS = 6 * VCHUNK;  // 48, a multiple of 4

__device__ __forceinline__ void schur_syrk_mfma_body(double* lds, int bx, int by, int gy, const SchurDims& d, int n_views, int nsh, int n_tiles,
        const int32_t* __restrict__ view_cam_blk, const double* __restrict__ blk_Z, const double* __restrict__ y, double* __restrict__ partial) {
    double (*Zi)[64] = reinterpret_cast<double (*)[64]>(lds), (*Zj)[64] = Zi + SYRK_ROWS;
    double* ysh = lds + 2 * SYRK_ROWS * 64;
    int* bsh = reinterpret_cast<int*>(ysh + SYRK_ROWS);
    int pair = by, ti = 0;
    while (pair >= n_tiles - ti) { pair -= n_tiles - ti; ++ti; }
    const int tj = ti + pair;
    const int i0 = ti * 64, j0 = tj * 64;
    const int v0 = bx * VCHUNK;
    stage_block_table(d, n_views, v0, view_cam_blk, bsh);
    __syncthreads();
    stage_Z(d, bsh, n_views, v0, view_cam_blk, blk_Z, i0, nsh, Zi);
    if (tj != ti) stage_Z(d, bsh, n_views, v0, view_cam_blk, blk_Z, j0, nsh, Zj);
    const double (*Zc)[64] = tj != ti ? Zj : Zi;
    const int nrow = 6 * (min(n_views, v0 + VCHUNK) - v0);
    if (static_cast<int>(threadIdx.x) < SYRK_ROWS) ysh[threadIdx.x] = static_cast<int>(threadIdx.x) < nrow ? y[6 * static_cast<int64_t>(v0) + threadIdx.x] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    v4f64 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int step = 0; step < SYRK_ROWS / 4; ++step) {
        const int r = 4 * step + lk;
        const double a = Zi[r][wave * 16 + li];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Zc[r][c * 16 + li], acc[c], 0, 0, 0);
    }
    double* row = partial + static_cast<int64_t>(bx) * (static_cast<int64_t>(gy) * 4096 + nsh);
    double* out = row + static_cast<int64_t>(by) * 4096;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[(wave * 16 + lk + 4 * reg) * 64 + c * 16 + li] = acc[c][reg];
    if (ti == tj) schur_gvec_chunk(Zi, ysh, nrow, i0, nsh, row + static_cast<int64_t>(gy) * 4096);
}
__global__ __launch_bounds__(256) void k_schur_syrk_mfma(SchurDims d, int n_views, int nsh, int n_tiles, const int32_t* __restrict__ view_cam_blk,
        const double* __restrict__ blk_Z, const double* __restrict__ y, double* __restrict__ partial) {
    __shared__ double lds[SYRK_LDS];
    schur_syrk_mfma_body(lds, blockIdx.x, blockIdx.y, gridDim.y, d, n_views, nsh, n_tiles, view_cam_blk, blk_Z, y, partial);
}

__global__ void k_backsub(SchurDims d, int n_views, const int64_t* __restrict__ link_off, const int32_t* __restrict__ link_blk,
                          const int32_t* __restrict__ blk_cam, const double* __restrict__ blk_Z,
                          const double* __restrict__ delta_sh, const int32_t* __restrict__ fixed, const double* __restrict__ L,
                          const double* __restrict__ y, const double* __restrict__ D, const double* __restrict__ gp,
                          const double* __restrict__ x, double* __restrict__ delta_p, double* __restrict__ xt,
                          double* __restrict__ stats /*[n_views][4]*/, const double* __restrict__ gate = nullptr) {
    if (gate && *gate == 0.0) return;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_views) return;
    const int nb = static_cast<int>(link_off[v + 1] - link_off[v]);
    double o4[4];
    backsub_view_body(d, nb, link_blk + link_off[v], blk_cam, blk_Z, delta_sh, fixed[v] != 0, L + 36 * static_cast<int64_t>(v),
                      y + 6 * static_cast<int64_t>(v), D + 6 * static_cast<int64_t>(v), gp + 6 * static_cast<int64_t>(v),
                      x + 7 * static_cast<int64_t>(v), delta_p + 6 * static_cast<int64_t>(v), xt + 7 * static_cast<int64_t>(v), o4);
    for (int k = 0; k < 4; ++k) stats[4 * static_cast<int64_t>(v) + k] = o4[k];
}

// an accepted step: trial copies -> current copies (the shared pack and the private poses), one launch
__global__ void k_accept(int64_t n_shared, const double* __restrict__ shared_trial, double* __restrict__ shared_cur, int64_t n_view,
                         const double* __restrict__ view_trial, double* __restrict__ view_cur) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_shared) shared_cur[i] = shared_trial[i];
    if (i < n_view) view_cur[i] = view_trial[i];
}

// The packed exchange buffer of one linear solve (lm_core.hpp PackLayout), assembled ON THE DEVICE: the camera sums are written
// in place by k_seg_sum; this kernel adds the step statistics, the cost, the dense S_schur unpacked from the syrk tiles, g_schur,
// the failure count and this rank's gradient-max slot (the other ranks' slots are zeroed: the all-reduce is a sum).
struct PackArgs {
    int64_t off_stats, off_cam, off_cost, off_nfail, off_S, off_g, off_gmax;
    int n, n_tiles, n_ranks, rank, n_cam_doubles;
    int has_blocks /* camera sums are in the pack */, has_cost /* stat[4] holds the cost */, has_schur, has_stats;
};
__global__ __launch_bounds__(256) void k_pack(PackArgs a, const double* __restrict__ stat /*[0..4): step2, xnorm2, gd, dHd; [4]: cost*/,
                                              const double* __restrict__ tiles /*[pairs*4096 | g | gmax, nfail]*/, double* __restrict__ pack) {
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t nn = static_cast<int64_t>(a.n) * a.n;
    const int64_t sw = static_cast<int64_t>(a.n_tiles) * (a.n_tiles + 1) / 2 * 4096;
    if (tid < nn) {  // S travels as its upper triangle, packed row-major (PackLayout)
        const int i = static_cast<int>(tid / a.n), j = static_cast<int>(tid % a.n);
        if (i <= j) {
            double val = 0.0;
            if (a.has_schur) {
                const int ti = i >> 6, tj = j >> 6;
                const int pair = ti * a.n_tiles - ti * (ti - 1) / 2 + (tj - ti);  // upper-triangular tile pairs in (ti, tj >= ti) order
                val = tiles[static_cast<int64_t>(pair) * 4096 + (i & 63) * 64 + (j & 63)];
            }
            pack[a.off_S + ctl_sidx(a.n, i, j)] = val;
        }
    }
    if (tid < a.n) pack[a.off_g + tid] = a.has_schur ? tiles[sw + tid] : 0.0;
    if (tid < a.n_ranks) pack[a.off_gmax + tid] = (a.has_schur && tid == a.rank) ? tiles[sw + a.n] : 0.0;
    if (tid == 0) {
        pack[a.off_nfail] = a.has_schur ? tiles[sw + a.n + 1] : 0.0;
        pack[a.off_cost] = a.has_cost ? stat[4] : 0.0;
        // has_stats: 1 = a trial step (stat[2], stat[3] = the views' g^T d, d^T H d), 2 = a line-search sample (stat[2] = their slope)
        pack[a.off_stats + 0] = a.has_stats == 1 ? stat[2] : 0.0;  // PackLayout::GD
        pack[a.off_stats + 1] = a.has_stats == 1 ? stat[3] : 0.0;  // DHD
        pack[a.off_stats + 2] = a.has_stats ? stat[0] : 0.0;       // STEP2
        pack[a.off_stats + 3] = a.has_stats ? stat[1] : 0.0;       // XNORM2
        pack[a.off_stats + 4] = (a.has_stats && a.has_cost) ? stat[4] : 0.0;  // TRIAL_COST
        pack[a.off_stats + 5] = a.has_stats == 2 ? stat[2] : 0.0;  // SLOPE
    }
    if (!a.has_blocks && tid < a.n_cam_doubles) pack[a.off_cam + tid] = 0.0;
}

// ---- the fused stages of one linear solve -----------------------------------------------------------------------------------------
// Between Mode B and the packed exchange an LM step needs ten small dependent reductions (block weights and cost, per-camera sums,
// per-view elimination, the Schur contraction and its sums, the pack).  As launches of their own each costs 4 - 6 us of dispatch
// and drain whatever its work (87 us per step for the 8-camera rig, a sixth of a step when the problem is split over 8 GPUs).
// Here the stages that do not depend on each other share ONE launch (ranges of blockIdx.x run different bodies), and every
// reduction keeps its own fixed order: the results are bit-identical to the one-kernel-per-stage sequence, which the paths off
// the LM loop (covariance, cost queries, the one-thread-per-view form) still use.
//   k_step_head   back-substitution of the views + their blocks' constants at the trial poses      (was 3 launches)
//   k_sys_stage2  per-camera chunk sums | per-view elimination | cost (or its partial sums)         (was 3)
//   k_sys_stage3  camera segment sums | Schur contraction | gradient max | step statistics | cost   (was 4 - 5)
//   k_sys_pack    sums over the view chunks straight into the packed exchange buffer                (was 2)
struct SysArgs {
    SchurDims d;
    int n_views, n_blocks, nsh, n_tiles, n_pairs, n_vchunks, n_cams, NACC, constrained;
    int n_cc, n_vb, n_costp;         // stage 2 ranges: camera chunks | view workgroups (4 views each) | cost workgroups (0, 1 or ceil(n_blocks / 2048))
    int n_seg_x, n_seg, n_syrk;      // stage 3 ranges: segment sums (n_seg_x per camera) | (chunk, tile pair) | then 1 + has_vstats + (n_costp > 1)
    int has_vstats;                  // the views' step statistics (k_step_head) are reduced in stage 3
    double huber;
    const int64_t *link_off, *cchunk_off, *cam_seg;
    const int32_t *link_blk, *cam_blk, *view_fixed, *view_cam_blk;
    const double *blk_acc, *blk_w, *blk_s, *lmp, *view;
    double *view_scale2, *view_L, *view_y, *view_D, *view_gp, *blk_Z, *view_gmax, *cam_partial, *cam_out, *cost_part, *cost_out;
    double *syrk_partial, *tiles_tail /* [gmax, #failed] */, *view_stats, *stat_out;
};

// {1/2 sum rho(s_b), sum s_b} over b = b0 + t, b0 + t + 256, ... < b1 (k_cost / k_cost_partial of kernels_reproj.hip: same strides, same tree)
__device__ __forceinline__ void cost_range_body(double* lds, int b0, int b1, const double* __restrict__ blk_s, double huber_delta, double* __restrict__ out2) {
    double (*sh)[256] = reinterpret_cast<double (*)[256]>(lds);
    double c = 0.0, ss = 0.0;
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
    if (threadIdx.x == 0) { out2[0] = sh[0][0]; out2[1] = sh[1][0]; }
}

__global__ __launch_bounds__(256) void k_sys_stage2(SysArgs a) {
    __shared__ double lds[2 * 256];
    int bx = blockIdx.x;
    if (bx < a.n_cc) {
        cam_partial_body(bx, a.NACC, a.cchunk_off, a.cam_blk, a.blk_w, a.blk_acc, a.cam_partial);
        return;
    }
    bx -= a.n_cc;
    if (bx < a.n_vb) {
        schur_view_wave_body(bx, a.d, a.n_views, a.link_off, a.link_blk, a.blk_acc, a.blk_w, a.view_fixed, a.lmp, a.constrained, a.view,
                             a.view_scale2, a.view_L, a.view_y, a.view_D, a.view_gp, a.blk_Z, a.view_gmax);
        return;
    }
    bx -= a.n_vb;
    if (a.n_costp == 1) cost_range_body(lds, 0, a.n_blocks, a.blk_s, a.huber, a.cost_out);
    else cost_range_body(lds, bx * 2048, min(bx * 2048 + 2048, a.n_blocks), a.blk_s, a.huber, a.cost_part + 2 * bx);
}

template <bool MFMA>
__global__ __launch_bounds__(256) void k_sys_stage3(SysArgs a) {
    __shared__ double lds[SYRK_LDS];
    static_assert(SYRK_LDS >= COL_REDUCE_LDS && SYRK_LDS >= RS_GROUPS * RS_COLS, "one LDS buffer for every body");
    int bx = blockIdx.x;
    if (bx < a.n_seg) {
        seg_sum_body(lds, bx % a.n_seg_x, bx / a.n_seg_x, a.NACC, a.cam_seg, a.cam_partial, a.cam_out);
        return;
    }
    bx -= a.n_seg;
    if (bx < a.n_syrk) {
        const int chunk = bx % a.n_vchunks, pair = bx / a.n_vchunks;
        if (MFMA) schur_syrk_mfma_body(lds, chunk, pair, a.n_pairs, a.d, a.n_views, a.nsh, a.n_tiles, a.view_cam_blk, a.blk_Z, a.view_y, a.syrk_partial);
        else schur_syrk_body(lds, chunk, pair, a.n_pairs, a.d, a.n_views, a.nsh, a.n_tiles, a.view_cam_blk, a.blk_Z, a.view_y, a.syrk_partial);
        return;
    }
    bx -= a.n_syrk;
    if (bx == 0) { col_reduce_body(lds, a.n_views, 0, a.view_gmax, a.view_gmax, a.tiles_tail); return; }
    if (bx == 1 && a.has_vstats) { col_reduce_body(lds, a.n_views, 4, a.view_stats, nullptr, a.stat_out); return; }
    if (threadIdx.x == 0) {  // k_cost_final: the partial pairs in order
        double c = 0.0, ss = 0.0;
        for (int k = 0; k < a.n_costp; ++k) { c += a.cost_part[2 * k]; ss += a.cost_part[2 * k + 1]; }
        a.cost_out[0] = c;
        a.cost_out[1] = ss;
    }
}

// k_row_sum and k_pack in one: workgroup x sums 32 columns of the chunk table [n_vchunks][pairs * 4096 + nsh] (8 row groups, fixed
// order) and writes each straight to its place in the pack (S as its packed upper triangle, g_schur); workgroup 0 adds the scalars
__global__ __launch_bounds__(RS_COLS * RS_GROUPS) void k_sys_pack(PackArgs a, int64_t n_rows, const double* __restrict__ rows,
                                                                   const double* __restrict__ tail /*[gmax, #failed]*/,
                                                                   const double* __restrict__ stat, double* __restrict__ pack) {
    __shared__ double sh[RS_GROUPS][RS_COLS];
    const int64_t sw = static_cast<int64_t>(a.n_tiles) * (a.n_tiles + 1) / 2 * 4096, width = sw + a.n;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * RS_COLS + threadIdx.x % RS_COLS;
    const double tot = grouped_column_sum(rows, 0, n_rows, width, e, e < width, sh);
    if (threadIdx.x < RS_COLS && e < width) {
        if (e < sw) {
            int pair = static_cast<int>(e >> 12), ti = 0;
            while (pair >= a.n_tiles - ti) { pair -= a.n_tiles - ti; ++ti; }
            const int i = ti * 64 + static_cast<int>((e >> 6) & 63), j = (ti + pair) * 64 + static_cast<int>(e & 63);
            if (i <= j && j < a.n) pack[a.off_S + ctl_sidx(a.n, i, j)] = tot;
        } else {
            pack[a.off_g + (e - sw)] = tot;
        }
    }
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid < a.n_ranks) pack[a.off_gmax + tid] = tid == a.rank ? tail[0] : 0.0;
    if (tid == 0) {
        pack[a.off_nfail] = tail[1];
        pack[a.off_cost] = a.has_cost ? stat[4] : 0.0;
        pack[a.off_stats + 0] = a.has_stats == 1 ? stat[2] : 0.0;
        pack[a.off_stats + 1] = a.has_stats == 1 ? stat[3] : 0.0;
        pack[a.off_stats + 2] = a.has_stats ? stat[0] : 0.0;
        pack[a.off_stats + 3] = a.has_stats ? stat[1] : 0.0;
        pack[a.off_stats + 4] = (a.has_stats && a.has_cost) ? stat[4] : 0.0;
        pack[a.off_stats + 5] = a.has_stats == 2 ? stat[2] : 0.0;
    }
    if (!a.has_blocks)
        for (int64_t k = tid; k < a.n_cam_doubles; k += static_cast<int64_t>(gridDim.x) * blockDim.x) pack[a.off_cam + k] = 0.0;
}

// k_backsub_wave, then the constants of the view's residual blocks at its trial pose (k_block_consts' work: a block's constants
// depend on its own view's pose and on the shared blocks the controller has left in copy 1) - the wavefront that has just formed
// the pose hands it to its lanes through registers, lane k takes the view's k-th block.  INTRINSIC / EXTRINSIC chains (the
// bundle chain has no private poses).
template <int CHAIN>
__global__ __launch_bounds__(256) void k_step_head(const double* __restrict__ gate, SchurDims d, int n_views, const int64_t* __restrict__ link_off,
                                                   const int32_t* __restrict__ link_blk, const int32_t* __restrict__ blk_cam,
                                                   const double* __restrict__ blk_Z, const double* __restrict__ delta_sh,
                                                   const int32_t* __restrict__ fixed, const double* __restrict__ L, const double* __restrict__ y,
                                                   const double* __restrict__ D, const double* __restrict__ gp, const double* __restrict__ x,
                                                   double* __restrict__ delta_p, double* __restrict__ xt, double* __restrict__ stats,
                                                   const double* __restrict__ cam_trial, double* __restrict__ bc) {
    if (gate && *gate == 0.0) return;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_views) return;
    const int nb = static_cast<int>(link_off[v + 1] - link_off[v]);
    const int32_t* blks = link_blk + link_off[v];
    const bool fx = fixed[v] != 0;
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (!fx) {
        const int n_pairs = nb * d.PSH;
        for (int p = lane; p < n_pairs; p += 64) {
            const int k = p / d.PSH, c = p - k * d.PSH;
            const int b = blks[k];
            const double dc = delta_sh[blk_cam[b] * d.PC + c];
            const double* Z = blk_Z + static_cast<int64_t>(b) * 6 * d.PSH + c;
            for (int i = 0; i < 6; ++i) a[i] += Z[i * d.PSH] * dc;
        }
        for (int i = 0; i < 6; ++i) a[i] = wave_sum63(a[i]);  // total in lane 63
    }
    double pose[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (lane == 63) {
        double o4[4], dp[6];
        backsub_view_finish(fx, a, L + 36 * static_cast<int64_t>(v), y + 6 * static_cast<int64_t>(v), D + 6 * static_cast<int64_t>(v),
                            gp + 6 * static_cast<int64_t>(v), x + 7 * static_cast<int64_t>(v), dp, pose, o4);
        for (int k = 0; k < 6; ++k) delta_p[6 * static_cast<int64_t>(v) + k] = dp[k];
        for (int k = 0; k < 7; ++k) xt[7 * static_cast<int64_t>(v) + k] = pose[k];
        for (int k = 0; k < 4; ++k) stats[4 * static_cast<int64_t>(v) + k] = o4[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) pose[k] = __shfl(pose[k], 63);
    for (int k = lane; k < nb; k += 64) {
        const int b = blks[k];
        double out[BC_SIZE];
        block_consts<CHAIN>(pose, CHAIN == CH_EXTRINSIC ? cam_trial + 7 * static_cast<int64_t>(blk_cam[b]) : nullptr, nullptr, out);
        for (int i = 0; i < BC_SIZE; ++i) bc[static_cast<int64_t>(b) * BC_SIZE + i] = out[i];
    }
}

// line search sample (line_search.hpp): trial poses at step size a along the last back-substituted step; stats [n_views][4] =
// { |xt - x|^2, |x|^2, slope share (k_view_slope, or 0), 0 }
__global__ void k_scale_step(int n_views, double a, const int32_t* __restrict__ fixed, const double* __restrict__ x,
                             const double* __restrict__ delta_p, double* __restrict__ xt, double* __restrict__ stats) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_views) return;
    double o2[2];
    scale_step_view_body(fixed[v] != 0, a, x + 7 * static_cast<int64_t>(v), delta_p + 6 * static_cast<int64_t>(v), xt + 7 * static_cast<int64_t>(v), o2);
    stats[4 * static_cast<int64_t>(v)] = o2[0];
    stats[4 * static_cast<int64_t>(v) + 1] = o2[1];
    stats[4 * static_cast<int64_t>(v) + 2] = 0.0;
    stats[4 * static_cast<int64_t>(v) + 3] = 0.0;
}
__global__ void k_view_slope(SchurDims d, int n_views, const int64_t* __restrict__ link_off, const int32_t* __restrict__ link_blk,
                             const double* __restrict__ blk_acc, const double* __restrict__ blk_w, const int32_t* __restrict__ fixed,
                             const double* __restrict__ delta_p, double* __restrict__ stats) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_views) return;
    stats[4 * static_cast<int64_t>(v) + 2] = view_slope_body(d, static_cast<int>(link_off[v + 1] - link_off[v]), link_blk + link_off[v], blk_acc,
                                                              blk_w, fixed[v] != 0, delta_p + 6 * static_cast<int64_t>(v));
}

// an accepted SPECULATIVE step: besides the parameter copies, the trial linearisation's block sums and weights become current
__global__ void k_accept_blocks(int64_t n_acc, const double* __restrict__ acc_trial, double* __restrict__ acc_cur, int64_t n_w,
                                const double* __restrict__ w_trial, double* __restrict__ w_cur) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n_acc) acc_cur[i] = acc_trial[i];
    if (i < n_w) w_cur[i] = w_trial[i];
}

static inline unsigned nblk(int64_t n, int per) { return static_cast<unsigned>(std::max<int64_t>(1, (n + per - 1) / per)); }

// ---- Backend on an Engine (state: lm_state.hpp) -------------------------------------------------------
struct HipBackend final : Backend {
    Engine& e;
    HipLMState& st;
    const double* lmp_src;  // [radius, init_scale] as the elimination kernels read it: page-locked host memory written by the host-side
                            // form of the iteration, or the controller's device copy
    explicit HipBackend(Engine& eng, HipLMState& s) : e(eng), st(s), lmp_src(s.pin_lmp.p) { st.current_is_on_device = false; }

    void set_view_fixed(const std::vector<int32_t>& f) override {
        if (!f.empty()) e.view_fixed.upload(f.data(), f.size(), e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
    }
    // Copy 0 goes up at once as ONE copy of the packed blocks (page-locked staging: queued, not blocking).  Copy 1 is only
    // staged here: trial() uploads it together with the shared step.  Right after accept() the device already holds the
    // accepted point in copy 0 (k_accept), so the driver's upload of the same values is skipped.
    void upload_shared(int which, const double* intr, const double* cam, const double* target) override {
        if (which == 0 && st.current_is_on_device) { st.current_is_on_device = false; return; }
        double* pk = st.pin_pack[which].p;
        std::memcpy(pk, intr, sizeof(double) * e.h_intr.size());
        if (e.chain != CBA_CHAIN_INTRINSIC) std::memcpy(pk + e.pk_cam, cam, sizeof(double) * e.h_cam.size());
        if (e.chain == CBA_CHAIN_BUNDLE) std::memcpy(pk + e.pk_target, target, sizeof(double) * 7);
        if (which == 0) {
            // the staging area may still be the source of the previous upload: wait for the stream first (rare path:
            // start of a solve, covariance)
            e.shared_pack[0].upload(pk, e.pk_delta, e.stream);
            CBA_HIP(hipStreamSynchronize(e.stream));
        }
    }
    // ---- stage plumbing: host preparation (every call) / device enqueue (captured once) / result collection ----------
    template <class F>
    void run_stage(HipLMState::GraphSlot& slot, double huber, bool constrained, F&& enqueue) {
        if (!st.graphs_ok) { enqueue(); return; }
        if (slot.huber != huber || slot.constrained != static_cast<int>(constrained) || slot.scalar != e.scalar) {
            if (slot.exec) (void)hipGraphExecDestroy(slot.exec);
            slot.exec = nullptr;
            slot.uses = 0;
            slot.huber = huber; slot.constrained = static_cast<int>(constrained); slot.scalar = e.scalar;
        }
        if (!slot.exec && slot.uses < st.graph_after) {
            ++slot.uses;
            enqueue();
            return;
        }
        if (!slot.exec) {
            if (hipStreamBeginCapture(e.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();
                st.graphs_ok = false;
                enqueue();
                return;
            }
            hipGraph_t graph = nullptr;
            try {
                enqueue();
            } catch (...) {
                (void)hipStreamEndCapture(e.stream, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            const hipError_t ec = hipStreamEndCapture(e.stream, &graph);
            const hipError_t ei = (ec == hipSuccess && graph) ? hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0) : hipErrorUnknown;
            if (graph) (void)hipGraphDestroy(graph);
            if (ei != hipSuccess) {  // no graph support for this sequence: plain launches from now on
                (void)hipGetLastError();
                slot.exec = nullptr;
                st.graphs_ok = false;
                enqueue();
                return;
            }
        }
        CBA_HIP(hipGraphLaunch(slot.exec, e.stream));
    }

    bool prep_normal_eq(std::vector<double>& cam_acc, double cost2[2]) {
        const Structure& s = st.s;
        cam_acc.assign(static_cast<size_t>(s.n_cams) * s.NACC, 0.0);
        cost2[0] = cost2[1] = 0.0;
        return s.n_blocks != 0;
    }
    // which: parameter copy to linearise at; cam_out [n_cams][NACC] and cost_out {cost, sum s} may be device or page-locked host memory
    void enqueue_normal_eq(double huber, int which = 0, double* cam_out = nullptr, double* cost_out = nullptr) {
        enqueue_normal_eq_head(which, huber);
        enqueue_normal_eq_tail(huber, cam_out, cost_out);
    }
    // block constants at copy `which` (unless the caller has built them: k_step_head), Mode B: the per-block [H | g | s]; huber >= 0:
    // the kernel that finishes a block's row also leaves its robust weight where it can (Engine::head_weights says whether it did)
    void enqueue_normal_eq_head(int which, double huber = -1.0, bool consts_done = false) {
        if (!consts_done) launch_block_consts(e, which);
        e.head_huber = st.fuse_small ? huber : -1.0;
        try {
            launch_normal_eq(e);
        } catch (...) {
            e.head_huber = -1.0;
            throw;
        }
        e.head_huber = -1.0;
    }
    void enqueue_normal_eq_tail(double huber, double* cam_out = nullptr, double* cost_out = nullptr) {  // weights, cost, per-camera sums
        const Structure& s = st.s;
        const size_t nca = static_cast<size_t>(s.n_cams) * s.NACC;
        if (!cam_out) cam_out = st.pin_ne.p;
        if (!cost_out) cost_out = st.pin_ne.p + nca;
        const bool fused_cost = s.n_blocks <= 4096 && !e.head_weights;
        if (e.head_weights)
            ;  // the weights came with the block rows; the cost follows below
        else if (fused_cost)
            hipLaunchKernelGGL(k_weights_cost, dim3(1), dim3(256), 0, e.stream, s.n_blocks, s.NACC, s.NH + s.PL, e.blk_acc.p, huber,
                               e.blk_w.p, e.blk_s.p, cost_out);
        else
            hipLaunchKernelGGL(k_weights, dim3(nblk(s.n_blocks, 256)), dim3(256), 0, e.stream, s.n_blocks, s.NACC, s.NH + s.PL,
                               e.blk_acc.p, huber, e.blk_w.p, e.blk_s.p);
        hipLaunchKernelGGL(k_cam_partial, dim3(std::max(1, st.n_cchunks)), dim3(256), 0, e.stream, s.NACC, st.cchunk_off.p,
                           st.cam_blk.p, e.blk_w.p, e.blk_acc.p, st.cam_partial.p);
        // the stage's results are written straight into page-locked host memory (device-visible): no copy command on the stream
        hipLaunchKernelGGL(k_seg_sum, dim3(nblk(s.NACC, RS_COLS), s.n_cams), dim3(RS_COLS * RS_GROUPS), 0, e.stream, s.n_cams,
                           s.NACC, st.cam_seg.p, st.cam_partial.p, cam_out);
        if (!fused_cost) launch_cost(e, huber, cost_out);
        CBA_HIP(hipGetLastError());
    }
    // ---- one linear solve's small stages in three launches (k_sys_stage2 / 3 / k_sys_pack above) -------------------------------------
    bool vstats_pending = false;  // k_step_head has left the views' step statistics unreduced: stage 3 takes them along
    double ctl_huber = 0.0;       // the Huber parameter of the running solve (the head of a step is queued without one at hand)
    bool can_fuse() const { return st.fuse_small && st.schur_wave && st.s.n_views != 0 && !e.scalar; }
    // Everything between Mode B and the exchange: [weights, cost, camera sums] (tail), the elimination of the views at private copy
    // `which`, the contraction and the pack.  has_blocks / has_stats / has_cost: PackArgs.
    void enqueue_system(double huber, bool tail, bool has_blocks, bool constrained, int which, const PackLayout& L, int has_stats, int has_cost = -1) {
        const Structure& s = st.s;
        const bool q2 = s.n_views != 0;
        if (!can_fuse()) {
            if (tail) enqueue_normal_eq_tail(huber, pack_target() + L.cam, st.stat_dev.p + 4);
            if (q2) enqueue_schur(constrained, which, st.sys_tiles.p);
            enqueue_pack(L, has_blocks, q2, has_stats, has_cost);
            return;
        }
        const int n = s.nsh;
        const int64_t sw = static_cast<int64_t>(st.n_pairs) * 4096;
        if (tail && !e.head_weights)
            hipLaunchKernelGGL(k_weights, dim3(nblk(s.n_blocks, 256)), dim3(256), 0, e.stream, s.n_blocks, s.NACC, s.NH + s.PL, e.blk_acc.p,
                               huber, e.blk_w.p, e.blk_s.p);
        SysArgs a{};
        a.d = st.dims; a.n_views = s.n_views; a.n_blocks = s.n_blocks; a.nsh = n; a.n_tiles = st.n_tiles; a.n_pairs = st.n_pairs;
        a.n_vchunks = st.n_vchunks; a.n_cams = s.n_cams; a.NACC = s.NACC; a.constrained = constrained ? 1 : 0;
        a.n_cc = tail ? std::max(1, st.n_cchunks) : 0;
        a.n_vb = static_cast<int>(nblk(s.n_views, 4));
        a.n_costp = tail ? (s.n_blocks <= 4096 ? 1 : (s.n_blocks + 2047) / 2048) : 0;
        a.n_seg_x = static_cast<int>(nblk(s.NACC, RS_COLS));
        a.n_seg = tail ? a.n_seg_x * s.n_cams : 0;
        a.n_syrk = st.n_vchunks * st.n_pairs;
        a.has_vstats = vstats_pending ? 1 : 0;
        a.huber = huber;
        a.link_off = st.link_off.p; a.cchunk_off = st.cchunk_off.p; a.cam_seg = st.cam_seg.p;
        a.link_blk = st.link_blk.p; a.cam_blk = st.cam_blk.p; a.view_fixed = e.view_fixed.p; a.view_cam_blk = st.view_cam_blk.p;
        a.blk_acc = e.blk_acc.p; a.blk_w = e.blk_w.p; a.blk_s = e.blk_s.p; a.lmp = lmp_src; a.view = e.view[which].p;
        a.view_scale2 = e.view_scale2.p; a.view_L = e.view_L.p; a.view_y = e.view_y.p; a.view_D = e.view_D.p; a.view_gp = e.view_gp.p;
        a.blk_Z = e.blk_Z.p; a.view_gmax = st.view_gmax.p; a.cam_partial = st.cam_partial.p; a.cam_out = pack_target() + L.cam;
        if (a.n_costp > 1 && e.cost_part.n < static_cast<size_t>(2 * a.n_costp)) e.cost_part.alloc(static_cast<size_t>(2 * a.n_costp));
        a.cost_part = e.cost_part.p; a.cost_out = st.stat_dev.p + 4;
        a.syrk_partial = st.syrk_partial.p; a.tiles_tail = st.sys_tiles.p + sw + n; a.view_stats = st.view_stats.p; a.stat_out = st.stat_dev.p;
        hipLaunchKernelGGL(k_sys_stage2, dim3(a.n_cc + a.n_vb + a.n_costp), dim3(256), 0, e.stream, a);
        const unsigned g3 = static_cast<unsigned>(a.n_seg + a.n_syrk + 1 + a.has_vstats + (a.n_costp > 1 ? 1 : 0));
        if (n >= 64 && st.syrk_mfma) hipLaunchKernelGGL(k_sys_stage3<true>, dim3(g3), dim3(256), 0, e.stream, a);
        else hipLaunchKernelGGL(k_sys_stage3<false>, dim3(g3), dim3(256), 0, e.stream, a);
        vstats_pending = false;
        hipLaunchKernelGGL(k_sys_pack, dim3(nblk(sw + n, RS_COLS)), dim3(RS_COLS * RS_GROUPS), 0, e.stream,
                           pack_args(L, has_blocks, true, has_stats, has_cost), static_cast<int64_t>(st.n_vchunks), st.syrk_partial.p,
                           st.sys_tiles.p + sw + n, st.stat_dev.p, pack_target());
        CBA_HIP(hipGetLastError());
    }
    void collect_normal_eq(std::vector<double>& cam_acc, double cost2[2]) {
        std::memcpy(cam_acc.data(), st.pin_ne.p, sizeof(double) * cam_acc.size());
        cost2[0] = st.pin_ne.p[cam_acc.size()];
        cost2[1] = st.pin_ne.p[cam_acc.size() + 1];
        e.active = 0;
    }
    bool prep_schur(double radius, bool init_scale, std::vector<double>& S, std::vector<double>& g, double* gmax_priv, int* nfail) {
        const Structure& s = st.s;
        const int n = s.nsh;
        S.assign(static_cast<size_t>(n) * n, 0.0);
        g.assign(n, 0.0);
        *gmax_priv = 0.0;
        *nfail = 0;
        if (s.n_views == 0) return false;
        st.pin_lmp.p[0] = radius;
        st.pin_lmp.p[1] = init_scale ? 1.0 : 0.0;
        return true;
    }
    // which: private pose copy the elimination is made at; tiles_out [syrk tiles | g_schur | gmax, #failed] device or page-locked
    void enqueue_schur(bool constrained, int which = 0, double* tiles_out = nullptr) {
        const Structure& s = st.s;
        const int n = s.nsh;
        if (st.schur_wave)
            hipLaunchKernelGGL(k_schur_view_wave, dim3(nblk(s.n_views, 4)), dim3(256), 0, e.stream, st.dims, s.n_views, st.link_off.p,
                               st.link_blk.p, e.blk_acc.p, e.blk_w.p, e.view_fixed.p, lmp_src, constrained ? 1 : 0,
                               e.view[which].p, e.view_scale2.p, e.view_L.p, e.view_y.p, e.view_D.p, e.view_gp.p, e.blk_Z.p,
                               st.view_gmax.p);
        else
            hipLaunchKernelGGL(k_schur_view, dim3(nblk(s.n_views, 64)), dim3(64), 0, e.stream, st.dims, s.n_views, st.link_off.p,
                               st.link_blk.p, e.blk_acc.p, e.blk_w.p, e.view_fixed.p, lmp_src, constrained ? 1 : 0,
                               e.view[which].p, e.view_scale2.p, e.view_L.p, e.view_y.p, e.view_D.p, e.view_gp.p, e.blk_Z.p,
                               st.view_gmax.p);
        const int64_t sw = static_cast<int64_t>(st.n_pairs) * 4096;
        if (n >= 64 && st.syrk_mfma)
            hipLaunchKernelGGL(k_schur_syrk_mfma, dim3(st.n_vchunks, st.n_pairs), dim3(256), 0, e.stream, st.dims, s.n_views, n, st.n_tiles,
                               st.view_cam_blk.p, e.blk_Z.p, e.view_y.p, st.syrk_partial.p);
        else
            hipLaunchKernelGGL(k_schur_syrk, dim3(st.n_vchunks, st.n_pairs), dim3(256), 0, e.stream, st.dims, s.n_views, n, st.n_tiles,
                               st.view_cam_blk.p, e.blk_Z.p, e.view_y.p, st.syrk_partial.p);
        double* pack = tiles_out ? tiles_out : st.pin.p;  // [syrk tiles | g_schur | gmax, #failed views]; default: page-locked host memory
        hipLaunchKernelGGL(k_row_sum, dim3(nblk(sw + n, RS_COLS)), dim3(RS_COLS * RS_GROUPS), 0, e.stream, static_cast<int64_t>(st.n_vchunks),
                           sw + n, st.syrk_partial.p, pack);
        hipLaunchKernelGGL(k_col_reduce, dim3(1), dim3(256), 0, e.stream, s.n_views, 0, st.view_gmax.p, st.view_gmax.p, pack + sw + n);
        CBA_HIP(hipGetLastError());
    }
    void collect_schur(std::vector<double>& S, std::vector<double>& g, double* gmax_priv, int* nfail) {
        const Structure& s = st.s;
        const int n = s.nsh;
        const int64_t sw = static_cast<int64_t>(st.n_pairs) * 4096;
        const double* tiles = st.pin.p;
        const int32_t nf = static_cast<int32_t>(tiles[static_cast<size_t>(sw) + n + 1] + 0.5);
        for (int i = 0; i < n; ++i) g[i] = tiles[static_cast<size_t>(sw) + i];
        *gmax_priv = tiles[static_cast<size_t>(sw) + n];
        *nfail = nf;
        int pair = 0;
        for (int ti = 0; ti < st.n_tiles; ++ti)
            for (int tj = ti; tj < st.n_tiles; ++tj, ++pair) {
                const double* T = &tiles[static_cast<size_t>(pair) * 4096];
                for (int a = 0; a < 64; ++a) {
                    const int i = ti * 64 + a;
                    if (i >= n) break;
                    for (int b = 0; b < 64; ++b) {
                        const int j = tj * 64 + b;
                        if (j >= n) break;
                        S[static_cast<size_t>(i) * n + j] = T[a * 64 + b];
                        S[static_cast<size_t>(j) * n + i] = T[a * 64 + b];
                    }
                }
            }
    }

    void normal_eq(double huber, std::vector<double>& cam_acc, double cost2[2]) override {  // covariance / cost paths: plain launches
        if (!prep_normal_eq(cam_acc, cost2)) return;
        enqueue_normal_eq(huber);
        CBA_HIP(hipStreamSynchronize(e.stream));
        collect_normal_eq(cam_acc, cost2);
    }
    void schur(double radius, bool init_scale, bool constrained, std::vector<double>& S, std::vector<double>& g, double* gmax_priv,
               int* nfail) override {
        if (!prep_schur(radius, init_scale, S, g, gmax_priv, nfail)) return;
        run_stage(st.g_schur, 0.0, constrained, [&] { enqueue_schur(constrained); });
        CBA_HIP(hipStreamSynchronize(e.stream));
        collect_schur(S, g, gmax_priv, nfail);
    }
    // a new linearisation: every kernel of both stages and both result copies in one graph, ONE stream synchronisation
    void normal_eq_schur(double huber, std::vector<double>& cam_acc, double cost2[2], double radius, bool init_scale, bool constrained,
                         std::vector<double>& S, std::vector<double>& g, double* gmax_priv, int* nfail) override {
        const bool q1 = prep_normal_eq(cam_acc, cost2);
        const bool q2 = prep_schur(radius, init_scale, S, g, gmax_priv, nfail);
        if (!q1 && !q2) return;
        run_stage(st.g_new, huber, constrained, [&] {
            if (q1) enqueue_normal_eq(huber);
            if (q2) enqueue_schur(constrained);
        });
        CBA_HIP(hipStreamSynchronize(e.stream));
        if (q1) collect_normal_eq(cam_acc, cost2);
        if (q2) collect_schur(S, g, gmax_priv, nfail);
    }
    // ---- packed systems (lm_core.hpp): assembled on the device, ONE all-reduce in place on the engine's stream (RCCL), one
    // device-to-host copy of the reduced buffer, one synchronisation ---------------------------------------------------------------
    PackArgs pack_args(const PackLayout& L, bool has_blocks, bool has_schur, int has_stats, int has_cost) const {
        const Structure& s = st.s;
        PackArgs a;
        a.off_stats = L.stats; a.off_cam = L.cam; a.off_cost = L.cost; a.off_nfail = L.nfail; a.off_S = L.S; a.off_g = L.g; a.off_gmax = L.gmax;
        a.n = s.nsh; a.n_tiles = st.n_tiles; a.n_ranks = L.n_ranks; a.rank = e.rank; a.n_cam_doubles = s.n_cams * s.NACC;
        a.has_blocks = has_blocks; a.has_schur = has_schur; a.has_stats = has_stats; a.has_cost = has_cost < 0 ? has_blocks : has_cost;
        return a;
    }
    void ensure_pack(const PackLayout& L) {
        if (st.pack_dev.n < static_cast<size_t>(L.size)) {
            st.pack_dev.alloc(static_cast<size_t>(L.size));
            st.pack_dev.zero(e.stream);
            st.sys_tiles.alloc(static_cast<size_t>(st.n_pairs) * 4096 + st.s.nsh + 8);
            st.stat_dev.alloc(8);
            st.stat_dev.zero(e.stream);
        }
        st.pin_packed.reserve(static_cast<size_t>(L.size));
    }
    // Where the pack is assembled: in device memory, where RCCL reduces it in place and the controller (lm_ctl.hip) reads it;
    // with a host-callback transport (gloo / MPI) the kernels write it straight into page-locked host memory instead.
    bool host_transport() const { return !e.rccl_comm && e.allreduce != nullptr; }
    double* pack_target() { return host_transport() ? st.pin_packed.p : st.pack_dev.p; }
    void enqueue_pack(const PackLayout& L, bool has_blocks, bool has_schur, int has_stats, int has_cost = -1) {
        const int64_t work = std::max<int64_t>({static_cast<int64_t>(st.s.nsh) * st.s.nsh, static_cast<int64_t>(st.s.n_cams) * st.s.NACC, L.n_ranks, 1});
        hipLaunchKernelGGL(k_pack, dim3(nblk(work, 256)), dim3(256), 0, e.stream, pack_args(L, has_blocks, has_schur, has_stats, has_cost),
                           st.stat_dev.p, st.sys_tiles.p, pack_target());
        CBA_HIP(hipGetLastError());
    }
    void wait_step() {
        if (!st.sync_spin) { CBA_HIP(hipStreamSynchronize(e.stream)); return; }
        if (!st.step_done) CBA_HIP(hipEventCreateWithFlags(&st.step_done, hipEventDisableTiming));
        CBA_HIP(hipEventRecord(st.step_done, e.stream));
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0;; ++spins) {
            const hipError_t q = hipEventQuery(st.step_done);
            if (q == hipSuccess) return;
            if (q != hipErrorNotReady) CBA_HIP(q);
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            // a long stage (Mode B over 1e7+ observations) gains nothing from polling: sleep after 300 us
            if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
        }
        CBA_HIP(hipStreamSynchronize(e.stream));
    }
    // Sum [off, off + count) of the packed buffer over the ranks.  host_out != nullptr: the reduced range is also brought to the
    // host (the host-side form of the iteration, line-search samples); to_device: with a host transport the reduced range goes
    // back up for the controller.
    void exchange(int64_t off, int64_t count, const AllReduce& ar, double* host_out, bool to_device = false) {
        if (e.rccl_comm) {  // RCCL over xGMI, in place on the device buffer, on the engine's stream: no host staging
            const ncclResult_t r = ncclAllReduce(st.pack_dev.p + off, st.pack_dev.p + off, static_cast<size_t>(count), ncclDouble, ncclSum,
                                                 reinterpret_cast<ncclComm_t>(e.rccl_comm), e.stream);
            if (r != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
            ++device_allreduce_calls;
            device_allreduce_doubles += count;
        } else if (host_transport()) {  // gloo / MPI callback on the page-locked pack
            wait_step();
            ar(st.pin_packed.p + off, count);
            if (to_device) st.pack_dev.upload(st.pin_packed.p + off, static_cast<size_t>(count), e.stream, static_cast<size_t>(off));
            if (host_out) std::memcpy(host_out + off, st.pin_packed.p + off, sizeof(double) * static_cast<size_t>(count));
            return;
        } else {  // a single rank: the exchange is the identity (counted like one: the protocol does not depend on the rank count)
            ++device_allreduce_calls;
            device_allreduce_doubles += count;
        }
        if (host_out) {
            st.pack_dev.download(st.pin_packed.p + off, static_cast<size_t>(count), e.stream, static_cast<size_t>(off));
            wait_step();
            std::memcpy(host_out + off, st.pin_packed.p + off, sizeof(double) * static_cast<size_t>(count));
        }
    }
    void sys_new(double huber, double radius, bool init_scale, bool constrained, const PackLayout& L, const AllReduce& ar, int rank,
                 double* pack) override {
        (void)rank;
        const Structure& s = st.s;
        ensure_pack(L);
        const bool q1 = s.n_blocks != 0;
        st.pin_lmp.p[0] = radius;
        st.pin_lmp.p[1] = init_scale ? 1.0 : 0.0;
        ctl_huber = huber;
        if (q1) enqueue_normal_eq_head(0, huber);
        enqueue_system(huber, q1, q1, constrained, 0, L, 0);
        exchange(L.cam, L.size - L.cam, ar, pack);
        e.active = 0;
    }
    void sys_resolve(double radius, bool constrained, const PackLayout& L, const AllReduce& ar, int rank, double* pack) override {
        (void)rank;
        const Structure& s = st.s;
        ensure_pack(L);
        st.pin_lmp.p[0] = radius;
        st.pin_lmp.p[1] = 0.0;
        enqueue_system(0.0, false, s.n_blocks != 0, constrained, 0, L, 0, 0);  // only [nfail .. g] travels (the cost slot, outside that range, is not a cost after this)
        exchange(L.nfail, L.gmax - L.nfail, ar, pack);
    }
    bool sys_step(const double* delta_sh, double huber, double radius_next, bool constrained, const PackLayout& L, const AllReduce& ar,
                  int rank, double* pack) override {
        (void)rank;
        const Structure& s = st.s;
        if (e.scalar) return false;  // fp32 study mode keeps the plain sequence
        ensure_pack(L);
        if (e.blk_acc_alt.n < e.blk_acc.n) { e.blk_acc_alt.alloc(e.blk_acc.n); e.blk_w_alt.alloc(e.blk_w.n); }
        std::memcpy(st.pin_pack[1].p + e.pk_delta, delta_sh, sizeof(double) * s.nsh);
        e.shared_pack[1].upload(st.pin_pack[1].p, e.pk_delta + static_cast<size_t>(s.nsh), e.stream);  // trial blocks + step
        // statistics of the step from the CURRENT factors, trial poses into copy 1; then the linearisation at the trial point into the
        // second set of block sums / weights (the current set stays valid for a rejected step)
        st.pin_lmp.p[0] = radius_next;
        st.pin_lmp.p[1] = 0.0;
        ctl_huber = huber;
        step_head_speculative();
        step_tail_enqueue(huber, constrained, L);
        exchange(0, L.size, ar, pack);
        e.active = 1;
        return true;
    }
    void line_eval(double a, double huber, bool want_slope, const PackLayout& L, const AllReduce& ar, int rank, double* pack) override {
        (void)rank;
        const Structure& s = st.s;
        ensure_pack(L);
        if (want_slope && e.blk_acc_alt.n < e.blk_acc.n) { e.blk_acc_alt.alloc(e.blk_acc.n); e.blk_w_alt.alloc(e.blk_w.n); }
        const bool q1 = s.n_blocks != 0, q2 = s.n_views != 0;
        e.shared_pack[1].upload(st.pin_pack[1].p, e.pk_delta, e.stream);  // the shared blocks at this step size (staged by upload_shared(1))
        if (q2)
            hipLaunchKernelGGL(k_scale_step, dim3(nblk(s.n_views, 64)), dim3(64), 0, e.stream, s.n_views, a, e.view_fixed.p, e.view[0].p,
                               st.view_delta.p, e.view[1].p, st.view_stats.p);
        if (q1 && want_slope) {  // linearise at the sample into the second set of block sums (the current set stays valid)
            std::swap(e.blk_acc.p, e.blk_acc_alt.p);
            std::swap(e.blk_w.p, e.blk_w_alt.p);
            try {
                enqueue_normal_eq(huber, 1, pack_target() + L.cam, st.stat_dev.p + 4);
                if (q2)
                    hipLaunchKernelGGL(k_view_slope, dim3(nblk(s.n_views, 64)), dim3(64), 0, e.stream, st.dims, s.n_views, st.link_off.p,
                                       st.link_blk.p, e.blk_acc.p, e.blk_w.p, e.view_fixed.p, st.view_delta.p, st.view_stats.p);
            } catch (...) {
                std::swap(e.blk_acc.p, e.blk_acc_alt.p);
                std::swap(e.blk_w.p, e.blk_w_alt.p);
                throw;
            }
            std::swap(e.blk_acc.p, e.blk_acc_alt.p);
            std::swap(e.blk_w.p, e.blk_w_alt.p);
        } else if (q1) {  // the cost alone (Mode R)
            launch_block_consts(e, 1);
            launch_resid(e);
            launch_cost(e, huber, st.stat_dev.p + 4);
        }
        if (q2)
            hipLaunchKernelGGL(k_col_reduce, dim3(1), dim3(256), 0, e.stream, s.n_views, 4, st.view_stats.p, static_cast<const double*>(nullptr),
                               st.stat_dev.p);
        else
            CBA_HIP(hipMemsetAsync(st.stat_dev.p, 0, 4 * sizeof(double), e.stream));
        CBA_HIP(hipGetLastError());
        enqueue_pack(L, q1 && want_slope, false, 2, q1 ? 1 : 0);
        exchange(0, L.size, ar, pack);
        e.active = 1;
    }
    void accept_step() override {
        const int64_t n_shared = static_cast<int64_t>(e.pk_delta), n_view = static_cast<int64_t>(e.h_view.size());
        hipLaunchKernelGGL(k_accept, dim3(nblk(std::max(n_shared, n_view), 256)), dim3(256), 0, e.stream, n_shared, e.shared_pack[1].p,
                           e.shared_pack[0].p, n_view, e.view[1].p, e.view[0].p);
        const int64_t n_acc = static_cast<int64_t>(st.s.n_blocks) * st.s.NACC, n_w = st.s.n_blocks;
        if (n_acc > 0)
            hipLaunchKernelGGL(k_accept_blocks, dim3(nblk(n_acc, 256)), dim3(256), 0, e.stream, n_acc, e.blk_acc_alt.p, e.blk_acc.p, n_w,
                               e.blk_w_alt.p, e.blk_w.p);
        CBA_HIP(hipGetLastError());
        e.active = 1;  // bc / sd were built from copy 1 = the values copy 0 now holds
    }

    // ---- the controller form of the iteration (lm_core.hpp Backend::ctl_*, lm_ctl.hip) ---------------------------------------------
    // Nothing below waits except ctl_wait(): the launch sequences are queued on the engine's stream, the controller kernel behind
    // the exchange decides, and what the host needs to know arrives in the control record.
    bool ctl_constrained = false;
    int64_t ctl_invocations = 0;
    void ensure_ctl(const PackLayout& L) {
        const Structure& s = st.s;
        const int n = s.nsh;
        const int lda = ctl_lda(n), M8 = ctl_padded(n);
        const bool lds = lm_ctl_fits_lds(n);
        const size_t o_scal = 0, o_lmp = CS_COUNT, o_camc = o_lmp + 8, o_gc = o_camc + static_cast<size_t>(s.n_cams) * s.NACC, o_scale2 = o_gc + n,
                     o_hdiag = o_scale2 + n, o_xs = o_hdiag + n, o_rdiag = o_xs + M8, o_xtmp = o_rdiag + M8, o_Ld = o_xtmp + e.pk_size,
                     o_A = o_Ld + static_cast<size_t>(M8) * CTL_NB, total = o_A + (lds ? 8 : static_cast<size_t>(M8 + 1) * lda);
        if (st.ctl_n != n || st.ctl_buf.n < total) {
            st.ctl_buf.alloc(total);
            st.ctl_buf.zero(e.stream);
            st.ctl_idx.alloc(static_cast<size_t>(3 * std::max(1, n)));  // [effective columns | column -> camera | column -> local column]
            st.ctl_eff.alloc(static_cast<size_t>(std::max(1, n)));
            {   // the column tables (structure.hpp shared_col, inverted), once per problem
                std::vector<int32_t> tab(static_cast<size_t>(3 * std::max(1, n)), 0);
                CtlView T{};
                T.chain = s.chain; T.PC = s.PC;
                for (int i = 0; i < n; ++i) ctl_decode(T, i, &tab[static_cast<size_t>(n) + i], &tab[static_cast<size_t>(2 * n) + i]);
                st.ctl_idx.upload(tab.data(), tab.size(), e.stream);
                CBA_HIP(hipStreamSynchronize(e.stream));
            }
            st.ctl_rec.reserve(CTL_REC_FETCH + e.pk_size + static_cast<size_t>(n) + 8);
            st.ctl_n = n;
        }
        CtlView& V = st.ctl_view;
        V.n = n; V.n_cams = s.n_cams; V.PI = s.PI; V.PL = s.PL; V.NH = s.NH; V.NACC = s.NACC; V.PC = s.PC; V.sh_base = s.sh_base;
        V.chain = s.chain; V.n_ranks = L.n_ranks;
        V.off_stats = L.stats; V.off_cam = L.cam; V.off_cost = L.cost; V.off_nfail = L.nfail; V.off_S = L.S; V.off_g = L.g; V.off_gmax = L.gmax;
        V.pk_cam = static_cast<int64_t>(e.pk_cam); V.pk_target = static_cast<int64_t>(e.pk_target); V.pk_delta = static_cast<int64_t>(e.pk_delta);
        double* b = st.ctl_buf.p;
        V.x_cur = e.shared_pack[0].p; V.x_trial = e.shared_pack[1].p; V.x_tmp = b + o_xtmp;
        V.scal = b + o_scal; V.lmp = b + o_lmp; V.camc = b + o_camc; V.gc = b + o_gc; V.scale2 = b + o_scale2; V.hdiag = b + o_hdiag; V.xs = b + o_xs;
        V.rdiag = b + o_rdiag; V.A = b + o_A; V.Ld = b + o_Ld; V.lda = lda; V.okflag = nullptr;  // (okflag: LDS, set by the kernel)
        V.pack = st.pack_dev.p;
        V.eff = st.ctl_eff.p; V.idx = st.ctl_idx.p; V.colcam = st.ctl_idx.p + n; V.collc = st.ctl_idx.p + 2 * n;
        V.active = st.res_active.p; V.cam_var = st.res_cam_var.p;
        V.rec = st.ctl_rec.p;
    }
    void run_ctl(int mode, int flag) {
        launch_lm_ctl(st.ctl_view, mode, flag, e.stream);
        ++ctl_invocations;
        if (st.ctl_event) {  // (experiment builds: ctl_wait sleeping on an event behind THIS launch; see there why not)
            if (!st.ctl_done) CBA_HIP(hipEventCreateWithFlags(&st.ctl_done, hipEventDisableTiming));
            CBA_HIP(hipEventRecord(st.ctl_done, e.stream));
        }
    }
    bool ctl_begin(const CtlSetup& cs, const PackLayout& L) override {
        if (!st.lm_ctl_mode) return false;
        const Structure& s = st.s;
        ensure_pack(L);
        ensure_ctl(L);
        if (e.blk_acc_alt.n < e.blk_acc.n) { e.blk_acc_alt.alloc(e.blk_acc.n); e.blk_w_alt.alloc(e.blk_w.n); }
        CtlView& V = st.ctl_view;
        V.eps = cs.eps; V.max_iterations = cs.max_iterations; V.constrained = cs.constrained; V.line_search = cs.line_search;
        V.speculate = (cs.speculate && !e.scalar) ? 1 : 0;  // the fp32 study mode keeps the plain sequence
        V.intr_var = cs.intr_var; V.target_var = cs.target_var;
        ctl_constrained = cs.constrained;
        // masks, control scalars, [radius, init_scale] and the start point: page-locked staging, queued copies
        st.pin_mask.reserve(static_cast<size_t>(s.nsh + s.n_cams));
        for (int i = 0; i < s.nsh; ++i) st.pin_mask.p[i] = (*cs.active)[i];
        for (int c = 0; c < s.n_cams; ++c) st.pin_mask.p[s.nsh + c] = (*cs.cam_var)[c];
        st.res_active.upload(st.pin_mask.p, s.nsh, e.stream);
        st.res_cam_var.upload(st.pin_mask.p + s.nsh, s.n_cams, e.stream);
        double* stage = st.ctl_rec.p + CS_COUNT;  // [scal | lmp] then the start point
        ctl_reset(stage);
        stage[CS_COUNT] = 1e4; stage[CS_COUNT + 1] = 1.0;
        st.ctl_buf.upload(stage, CS_COUNT + 2, e.stream);
        for (int k = 0; k < CS_COUNT; ++k) st.ctl_rec.p[k] = 0.0;
        double* pk = st.pin_pack[0].p;
        std::memcpy(pk, cs.intr, sizeof(double) * e.h_intr.size());
        if (e.chain != CBA_CHAIN_INTRINSIC) std::memcpy(pk + e.pk_cam, cs.cam, sizeof(double) * e.h_cam.size());
        if (e.chain == CBA_CHAIN_BUNDLE) std::memcpy(pk + e.pk_target, cs.target, sizeof(double) * 7);
        e.shared_pack[0].upload(pk, e.pk_delta, e.stream);
        e.shared_pack[1].upload(pk, e.pk_delta, e.stream);
        lmp_src = V.lmp;
        ctl_invocations = 0;
        st.current_is_on_device = false;
        return true;
    }
    void ctl_new(double huber, bool first, const PackLayout& L, const AllReduce& ar, int rank) override {
        (void)rank;
        const Structure& s = st.s;
        const bool q1 = s.n_blocks != 0;
        ctl_huber = huber;
        if (q1) enqueue_normal_eq_head(0, huber);
        enqueue_system(huber, q1, q1, ctl_constrained, 0, L, 0);
        exchange(L.cam, L.size - L.cam, ar, nullptr, true);
        run_ctl(CTL_NEW, first ? 1 : 0);
        e.active = 0;
    }
    void ctl_resolve(const PackLayout& L, const AllReduce& ar, int rank) override {
        (void)rank;
        const Structure& s = st.s;
        enqueue_system(0.0, false, s.n_blocks != 0, ctl_constrained, 0, L, 0, 0);  // only [nfail .. g] travels
        exchange(L.nfail, L.gmax - L.nfail, ar, nullptr, true);
        run_ctl(CTL_RESOLVED, 0);
    }
    void enqueue_backsub() {  // delta_p, trial poses (copy 1) and the views' share of the step statistics, from the CURRENT factors
        const Structure& s = st.s;
        hipLaunchKernelGGL(st.schur_wave ? k_backsub_wave : k_backsub, st.schur_wave ? dim3(nblk(s.n_views, 4)) : dim3(nblk(s.n_views, 64)),
                           st.schur_wave ? dim3(256) : dim3(64), 0, e.stream, st.dims, s.n_views, st.link_off.p, st.link_blk.p, e.d_blk_cam.p,
                           e.blk_Z.p, e.delta_sh.p, e.view_fixed.p, e.view_L.p, e.view_y.p, e.view_D.p, e.view_gp.p, e.view[0].p,
                           st.view_delta.p, e.view[1].p, st.view_stats.p, e.gate);
        hipLaunchKernelGGL(k_col_reduce, dim3(1), dim3(256), 0, e.stream, s.n_views, 4, st.view_stats.p, static_cast<const double*>(nullptr),
                           st.stat_dev.p, e.gate);
    }
    // The head of a speculative step: back-substitution, then block constants and Mode B at the trial point into the SECOND set of
    // block sums (the current set stays valid for a rejected step).  Nothing in it needs the host.
    void step_head_speculative() {
        const Structure& s = st.s;
        const bool q1 = s.n_blocks != 0, q2 = s.n_views != 0;
        // one launch for the back-substitution and the block constants where a block's constants hang on its own view's pose
        const bool fused_head = q1 && q2 && can_fuse() && e.chain != CBA_CHAIN_BUNDLE;
        if (fused_head) {
#define CBA_STEP_HEAD(CH)                                                                                                                  \
    hipLaunchKernelGGL(k_step_head<CH>, dim3(nblk(s.n_views, 4)), dim3(256), 0, e.stream, e.gate, st.dims, s.n_views, st.link_off.p,         \
                       st.link_blk.p, e.d_blk_cam.p, e.blk_Z.p, e.delta_sh.p, e.view_fixed.p, e.view_L.p, e.view_y.p, e.view_D.p,          \
                       e.view_gp.p, e.view[0].p, st.view_delta.p, e.view[1].p, st.view_stats.p, e.cam[1].p, e.bc.p)
            if (e.chain == CBA_CHAIN_INTRINSIC) CBA_STEP_HEAD(CH_INTRINSIC); else CBA_STEP_HEAD(CH_EXTRINSIC);
#undef CBA_STEP_HEAD
            launch_camera_consts(e, 1);
            vstats_pending = true;
        } else if (q2) {
            enqueue_backsub();
        } else {
            CBA_HIP(hipMemsetAsync(st.stat_dev.p, 0, 4 * sizeof(double), e.stream));
        }
        if (q1) {
            std::swap(e.blk_acc.p, e.blk_acc_alt.p);
            std::swap(e.blk_w.p, e.blk_w_alt.p);
            try {
                enqueue_normal_eq_head(1, ctl_huber, fused_head);
            } catch (...) {
                std::swap(e.blk_acc.p, e.blk_acc_alt.p);
                std::swap(e.blk_w.p, e.blk_w_alt.p);
                throw;
            }
            std::swap(e.blk_acc.p, e.blk_acc_alt.p);
            std::swap(e.blk_w.p, e.blk_w_alt.p);
        }
    }
    // ... and what follows Mode B, up to the assembled pack (the second set of block sums / weights, the trial poses)
    void step_tail_enqueue(double huber, bool constrained, const PackLayout& L) {
        const Structure& s = st.s;
        const bool q1 = s.n_blocks != 0;
        std::swap(e.blk_acc.p, e.blk_acc_alt.p);
        std::swap(e.blk_w.p, e.blk_w_alt.p);
        try {
            enqueue_system(huber, q1, q1, constrained, 1, L, 1);  // (the elimination with the radius the controller / the driver predicted)
        } catch (...) {
            std::swap(e.blk_acc.p, e.blk_acc_alt.p);
            std::swap(e.blk_w.p, e.blk_w_alt.p);
            throw;
        }
        std::swap(e.blk_acc.p, e.blk_acc_alt.p);
        std::swap(e.blk_w.p, e.blk_w_alt.p);
    }
    void step_tail_speculative(double huber, const PackLayout& L, const AllReduce& ar) {
        step_tail_enqueue(huber, ctl_constrained, L);
        exchange(0, L.size, ar, nullptr, true);
        run_ctl(CTL_STEP, 1);
        e.active = 1;
    }
    void ctl_step(double huber, bool speculative, const PackLayout& L, const AllReduce& ar, int rank) override {
        (void)rank;
        const Structure& s = st.s;
        const bool q1 = s.n_blocks != 0, q2 = s.n_views != 0;
        // the shared trial blocks and the shared step are where the controller left them (copy 1)
        ctl_huber = huber;
        if (speculative) {
            step_head_speculative();
            step_tail_speculative(huber, L, ar);
            return;
        }
        if (q2) enqueue_backsub();
        else CBA_HIP(hipMemsetAsync(st.stat_dev.p, 0, 4 * sizeof(double), e.stream));
        if (q1) {  // the cost alone (Mode R); the block sums / weights of the current point stay
            launch_block_consts(e, 1);
            launch_resid(e);
            launch_cost(e, huber, st.stat_dev.p + 4);
        } else {
            CBA_HIP(hipMemsetAsync(st.stat_dev.p + 4, 0, sizeof(double), e.stream));
        }
        enqueue_pack(L, false, false, 1, 1);
        exchange(L.stats, 6, ar, nullptr, true);
        run_ctl(CTL_STEP, 0);
        e.active = 1;
    }
    // Queue the head of the NEXT speculative step behind the controller invocation that was just queued, before its decision is
    // known: every launch of the head checks the controller's CS_GO flag on the device and does nothing unless the controller
    // accepted the step it decided on with the predicted radius and asks for another speculative step - the usual case, in which
    // the chip goes from the controller straight into the next step (the host's read of the record and its ~15 launches of 3-5 us
    // are off the critical path).  The head assumes the accept has happened: the buffer exchange is made here and undone by
    // ctl_prelaunch_cancel() if the record says otherwise.
    bool ctl_prelaunch() override {
        if (!st.ctl_prelaunch || e.scalar || !e.modeb_shared || (e.chain != CBA_CHAIN_INTRINSIC && !e.modeb_moments)) return false;
        ctl_accept(true);
        e.gate = st.ctl_view.scal + CS_GO;
        try {
            step_head_speculative();
        } catch (...) {
            e.gate = nullptr;
            ctl_accept(true);
            throw;
        }
        e.gate = nullptr;
        return true;
    }
    void ctl_prelaunch_cancel() override { ctl_accept(true); }  // (the gated launches did nothing)
    void ctl_step_tail(double huber, const PackLayout& L, const AllReduce& ar, int rank) override {
        (void)rank;
        step_tail_speculative(huber, L, ar);
    }
    // trial -> current without a copy: the two sets of private poses (and, after a speculative step, of block sums and weights)
    // trade places.  A stage graph captured with the old pointers would be stale: the captured stages belong to the host-side
    // form of the iteration, which is not running; they are dropped and re-captured if it ever runs again.
    void ctl_accept(bool blocks) override {
        std::swap(e.view[0].p, e.view[1].p);
        if (blocks) {
            std::swap(e.blk_acc.p, e.blk_acc_alt.p);
            std::swap(e.blk_w.p, e.blk_w_alt.p);
        }
        for (HipLMState::GraphSlot* g : {&st.g_new, &st.g_schur, &st.g_trial})
            if (g->exec) { (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; g->uses = 0; }
    }
    const double* ctl_wait() override {
        // the controller publishes the record's sequence number last (system-scope release): poll it, fall back to the stream
        volatile const double* seq = st.ctl_rec.p + CS_SEQ;
        const double want = static_cast<double>(ctl_invocations);
        if (st.sync_spin) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int spins = 0; *seq < want; ++spins) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(st.ctl_poll_us)) break;
            }
        }
        if (*seq < want) {
            // Still not there (a Mode B pass over 1e7+ observations is in front of the controller): nap and look again.  No event is
            // recorded behind the controller for this - an event record is a barrier packet between the controller and the head of
            // the next step that is already queued behind it (~6 us of idle chip per step) - and the stream itself cannot be waited
            // on: it holds that next step.  Every millisecond or so: has the stream died or drained without a record, has a peer's
            // collective failed (a rank that threw out of its solve aborts its communicator; this rank must not wait for ever), is
            // the deadline over?
            const auto t0 = std::chrono::steady_clock::now();
            for (int naps = 0; *seq < want; ++naps) {
                std::this_thread::sleep_for(std::chrono::microseconds(st.ctl_event ? 0 : 30));
                if (st.ctl_event) {  // (experiment builds: the event-based wait of the first version)
                    CBA_HIP(hipEventSynchronize(st.ctl_done));
                    break;
                }
                if ((naps & 31) != 31) continue;
                const hipError_t q = hipStreamQuery(e.stream);
                if (q == hipSuccess) break;  // the stream is empty: the record is there, or never will be
                if (q != hipErrorNotReady) CBA_HIP(q);
                if (!e.rccl_comm) continue;
                ncclResult_t async = ncclSuccess;
                (void)ncclCommGetAsyncError(reinterpret_cast<ncclComm_t>(e.rccl_comm), &async);
                const bool late = std::chrono::steady_clock::now() - t0 > std::chrono::seconds(st.rccl_timeout_s);
                if (async != ncclSuccess || late) {
                    rccl_abort(e);
                    throw HipError(late ? "RCCL exchange: no progress within the deadline (a peer rank failed or hung); communicator aborted"
                                        : std::string("RCCL exchange failed on a peer: ") + ncclGetErrorString(async) + "; communicator aborted");
                }
            }
            if (*seq < want) throw HipError("LM controller: the control record did not arrive");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return st.ctl_rec.p;
    }
    void ctl_fetch(double* intr, double* cam, double* target, double* delta) override {
        double* stage = st.ctl_rec.p + CTL_REC_FETCH;
        e.shared_pack[0].download(stage, e.pk_delta, e.stream);
        e.shared_pack[1].download(stage + e.pk_delta, static_cast<size_t>(st.s.nsh), e.stream, e.pk_delta);
        CBA_HIP(hipStreamSynchronize(e.stream));
        std::memcpy(intr, stage, sizeof(double) * e.h_intr.size());
        if (e.chain != CBA_CHAIN_INTRINSIC) std::memcpy(cam, stage + e.pk_cam, sizeof(double) * e.h_cam.size());
        if (e.chain == CBA_CHAIN_BUNDLE) std::memcpy(target, stage + e.pk_target, sizeof(double) * 7);
        std::memcpy(delta, stage + e.pk_delta, sizeof(double) * static_cast<size_t>(st.s.nsh));
    }
    void ctl_line_search_done(const double* scal) override {
        double* stage = st.ctl_rec.p + CS_COUNT;
        std::memcpy(stage, scal, sizeof(double) * CS_COUNT);
        st.ctl_buf.upload(stage, CS_COUNT, e.stream);
        run_ctl(CTL_LS_DONE, 0);
    }

    void trial(const double* delta_sh, double huber, TrialStats* out) override {
        const Structure& s = st.s;
        *out = TrialStats();
        if (s.n_blocks == 0) return;
        std::memcpy(st.pin_pack[1].p + e.pk_delta, delta_sh, sizeof(double) * s.nsh);
        run_stage(st.g_trial, huber, false, [&] {
            e.shared_pack[1].upload(st.pin_pack[1].p, e.pk_delta + static_cast<size_t>(s.nsh), e.stream);  // trial blocks + step
            if (s.n_views > 0) {
                hipLaunchKernelGGL(st.schur_wave ? k_backsub_wave : k_backsub, st.schur_wave ? dim3(nblk(s.n_views, 4)) : dim3(nblk(s.n_views, 64)), st.schur_wave ? dim3(256) : dim3(64), 0, e.stream, st.dims, s.n_views, st.link_off.p,
                                   st.link_blk.p, e.d_blk_cam.p, e.blk_Z.p, e.delta_sh.p, e.view_fixed.p, e.view_L.p, e.view_y.p,
                                   e.view_D.p, e.view_gp.p, e.view[0].p, st.view_delta.p, e.view[1].p, st.view_stats.p, static_cast<const double*>(nullptr));
                hipLaunchKernelGGL(k_col_reduce, dim3(1), dim3(256), 0, e.stream, s.n_views, 4, st.view_stats.p,
                                   static_cast<const double*>(nullptr), st.pin_tr.p + 8);
            }
            // cost at the trial point (Mode R); blk_s / blk_w of the ACCEPTED point stay in blk_acc / blk_w
            launch_block_consts(e, 1);
            launch_resid(e);  // Mode R writes blk_s; blk_acc / blk_w keep the accepted point's values
            launch_cost(e, huber, st.pin_tr.p + 24);
            CBA_HIP(hipGetLastError());
        });
        CBA_HIP(hipStreamSynchronize(e.stream));
        e.active = 1;
        const double* h = st.pin_tr.p;
        const double* c2 = st.pin_tr.p + 24;
        out->step2 = s.n_views > 0 ? h[8] : 0.0;
        out->xnorm2 = s.n_views > 0 ? h[9] : 0.0;
        out->gd = s.n_views > 0 ? h[10] : 0.0;
        out->dHd = s.n_views > 0 ? h[11] : 0.0;
        out->cost = c2[0];
    }
    void accept() override {
        const int64_t n_shared = static_cast<int64_t>(e.pk_delta), n_view = static_cast<int64_t>(e.h_view.size());
        hipLaunchKernelGGL(k_accept, dim3(nblk(std::max(n_shared, n_view), 256)), dim3(256), 0, e.stream, n_shared, e.shared_pack[1].p,
                           e.shared_pack[0].p, n_view, e.view[1].p, e.view[0].p);
        CBA_HIP(hipGetLastError());
        st.current_is_on_device = true;
    }
    void download_private(double* view_pose) override {
        if (e.h_view.empty()) return;
        e.view[0].download(view_pose, e.h_view.size(), e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
    }
    void download_blocks(std::vector<double>& acc, std::vector<double>& w) override {
        acc.resize(static_cast<size_t>(e.n_blocks) * e.NACC);
        w.resize(e.n_blocks);
        e.blk_acc.download(acc.data(), acc.size(), e.stream);
        e.blk_w.download(w.data(), w.size(), e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
    }
};

// ---- engine glue -----------------------------------------------------------------------------------
void destroy_lm_state(Engine& e) {
    delete lm_state(e);
    e.lm_state = nullptr;
}

void init_lm_state(Engine& e, const cba_reproj_problem& d, bool have_records) {
    auto* st = new HipLMState();
    e.lm_state = st;
    build_structure(d, st->s, have_records);
    const Structure& s = st->s;
    st->dims = SchurDims{s.PL, s.NH, s.NACC, s.PSH, s.PC, s.n_cams, s.chain};
    st->n_vchunks = std::max(1, (s.n_views + VCHUNK - 1) / VCHUNK);
    st->n_tiles = (s.nsh + 63) / 64;
    st->n_pairs = st->n_tiles * (st->n_tiles + 1) / 2;
    // camera chunks
    std::vector<int64_t> coff{0}, cseg(s.n_cams + 1, 0);
    for (int c = 0; c < s.n_cams; ++c) {
        for (int64_t p = s.cam_off[c]; p < s.cam_off[c + 1]; p += CCHUNK) coff.push_back(std::min<int64_t>(p + CCHUNK, s.cam_off[c + 1]));
        cseg[c + 1] = static_cast<int64_t>(coff.size()) - 1;
    }
    st->n_cchunks = static_cast<int>(coff.size()) - 1;
    auto up64 = [&](DevBuf<int64_t>& b, const std::vector<int64_t>& v) { b.alloc(v.size()); b.upload(v.data(), v.size(), e.stream); };
    auto up32 = [&](DevBuf<int32_t>& b, const std::vector<int32_t>& v) { b.alloc(v.size()); b.upload(v.data(), v.size(), e.stream); };
    up64(st->cchunk_off, coff);
    up64(st->cam_seg, cseg);
    up32(st->cam_blk, s.cam_blk);
    up64(st->link_off, s.link_off);
    up32(st->link_blk, s.link_blk);
    up32(st->view_cam_blk, s.view_cam_blk);
    st->cam_partial.alloc(static_cast<size_t>(std::max(1, st->n_cchunks)) * s.NACC);
    const size_t nv = std::max(1, s.n_views);
    st->view_gmax.alloc(nv);
    st->view_delta.alloc(nv * 6);
    st->view_delta.zero(e.stream);
    st->view_stats.alloc(nv * 4);
    st->syrk_partial.alloc(static_cast<size_t>(st->n_vchunks) * (static_cast<size_t>(st->n_pairs) * 4096 + s.nsh));  // + g_schur
    // pinned staging of everything a captured stage copies (sizes are fixed per problem: nothing is allocated in a capture)
    st->pin.reserve(static_cast<size_t>(st->n_pairs) * 4096 + s.nsh + 8);
    st->pin_ne.reserve(static_cast<size_t>(s.n_cams) * s.NACC + 2);
    st->pin_tr.reserve(32);
    st->pin_lmp.reserve(2);
    st->pin_pack[0].reserve(e.pk_size);
    st->pin_pack[1].reserve(e.pk_size);
    {   // ROCm loads a translation unit's code object on the first use of one of its kernels (milliseconds for the
        // template-heavy ones): touch both units here so that the first solve does not pay for it
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_schur_view));
        warm_reproj_kernels();
    }
    if (const char* env = std::getenv("CBA_SYRK_MFMA")) st->syrk_mfma = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_SCHUR_WAVE")) st->schur_wave = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_LM_FUSE")) st->fuse_small = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_LM_CTL_EVENT")) st->ctl_event = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_SYNC_SPIN")) st->sync_spin = std::atoi(env);
    if (const char* env = std::getenv("CBA_LM_GRAPH")) {
        const int v = std::atoi(env);
        st->graphs_ok = v != 0;
        st->graph_after = v > 1 ? v : (v == 1 ? 0 : st->graph_after);
    }
    e.blk_w.alloc(std::max(1, s.n_blocks));
    e.cam_acc.alloc(static_cast<size_t>(s.n_cams) * s.NACC);
    e.view_L.alloc(nv * 36);
    e.view_y.alloc(nv * 6);
    e.view_D.alloc(nv * 6);
    e.view_gp.alloc(nv * 6);
    e.view_scale2.alloc(nv * 6);
    e.view_fixed.alloc(nv);
    e.view_fixed.zero(e.stream);
    e.blk_Z.alloc(static_cast<size_t>(std::max(1, s.n_blocks)) * 6 * s.PSH);
    e.blk_Z.zero(e.stream);
    // resident LM (resident_lm.hip)
    up64(st->cam_off, s.cam_off);
    st->res_active.alloc(std::max(1, s.nsh));
    st->res_cam_var.alloc(std::max(1, s.n_cams));
    st->res_Hcc.alloc(static_cast<size_t>(s.nsh) * s.nsh);
    st->res_Ssch.alloc(static_cast<size_t>(s.nsh) * s.nsh);
    st->res_out.alloc(32);
    if (const char* env = cba_exp_env("CBA_LM_CTL")) st->lm_ctl_mode = std::atoi(env) != 0;
    if (const char* env = cba_exp_env("CBA_LM_CTL_POLL_US")) st->ctl_poll_us = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_LM_PRELAUNCH")) st->ctl_prelaunch = std::atoi(env);
    if (const char* env = std::getenv("CBA_RCCL_TIMEOUT_S")) st->rccl_timeout_s = std::max(1, std::atoi(env));
    warm_lm_ctl();
    if (const char* env = std::getenv("CBA_LM_RESIDENT")) st->resident_mode = std::atoi(env);
    if (const char* env = std::getenv("CBA_LM_RESIDENT_MAX_OBS")) st->resident_max_obs = std::atoll(env);
    CBA_HIP(hipStreamSynchronize(e.stream));
}

void solve_stats(const Engine& e, int64_t stats8[8]) {
    const HipLMState* st = reinterpret_cast<const HipLMState*>(e.lm_state);
    for (int k = 0; k < 8; ++k) stats8[k] = st ? st->xs[k] : 0;
}

void set_lm_mode(Engine& e, int mode) {
    if (mode < 0 || mode > 3)
        throw std::invalid_argument("lm mode: 0 chip-wide kernels, 1 automatic, 2 resident kernel whenever possible, 3 chip-wide kernels with the host-side iteration (diagnosis)");
    lm_state(e)->resident_mode = mode == 3 ? 0 : mode;
    lm_state(e)->lm_ctl_mode = mode == 3 ? 0 : 1;
}

void engine_allreduce(Engine& e, double* buf, int64_t n) {
    if (n <= 0) return;
    if (e.rccl_comm) {
        if (e.coll_buf.n < static_cast<size_t>(n)) e.coll_buf.alloc(static_cast<size_t>(n) * 2);
        e.coll_pin.reserve(static_cast<size_t>(n) * 2);
        std::memcpy(e.coll_pin.p, buf, sizeof(double) * static_cast<size_t>(n));
        e.coll_buf.upload(e.coll_pin.p, static_cast<size_t>(n), e.stream);
        const ncclResult_t r = ncclAllReduce(e.coll_buf.p, e.coll_buf.p, static_cast<size_t>(n), ncclDouble, ncclSum,
                                             reinterpret_cast<ncclComm_t>(e.rccl_comm), e.stream);
        if (r != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        e.coll_buf.download(e.coll_pin.p, static_cast<size_t>(n), e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
        std::memcpy(buf, e.coll_pin.p, sizeof(double) * static_cast<size_t>(n));
    } else if (e.allreduce) {
        if (e.allreduce(buf, n, e.allreduce_user) != 0) throw std::runtime_error("allreduce callback failed");
    }
}

void rccl_unique_id(uint8_t* id) {
    static_assert(sizeof(ncclUniqueId) == CBA_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    const ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) throw HipError(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memcpy(id, &u, sizeof(u));
}

void* rccl_comm_create(const uint8_t* id, int n_ranks, int rank) {
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("bad rank / n_ranks");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm;
    const ncclResult_t r = ncclCommInitRank(&comm, n_ranks, u, rank);
    if (r != ncclSuccess) throw HipError(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    return comm;
}
void rccl_comm_destroy(void* comm, bool abort) {
    if (!comm) return;
    if (abort) (void)ncclCommAbort(reinterpret_cast<ncclComm_t>(comm));
    else (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(comm));
}

void rccl_init(Engine& e, const uint8_t* id, int n_ranks, int rank) {
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("bad rank / n_ranks");
    rccl_destroy(e);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm;
    const ncclResult_t r = ncclCommInitRank(&comm, n_ranks, u, rank);
    if (r != ncclSuccess) throw HipError(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    e.rccl_comm = comm;
    e.n_ranks = n_ranks;
    e.rank = rank;
}

// the abort path: this rank cannot go on (an exception in its solve, a peer that no longer answers); ncclCommAbort tears the
// communicator down without waiting for outstanding collectives, which also lets the peers' pending collectives fail instead of
// hanging (they see it through ncclCommGetAsyncError in ctl_wait, or run into their own deadline)
void rccl_abort(Engine& e) {
    if (e.rccl_comm) {
        (void)ncclCommAbort(reinterpret_cast<ncclComm_t>(e.rccl_comm));
        e.rccl_comm = nullptr;
    }
}

void rccl_destroy(Engine& e) {
    if (e.rccl_comm) {
        (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(e.rccl_comm));
        e.rccl_comm = nullptr;
    }
}

static LMDriver make_driver(Engine& e, HipBackend& be) {
    AllReduce ar = [&e](double* buf, int64_t n) { engine_allreduce(e, buf, n); };
    return LMDriver(lm_state(e)->s, be, e.h_intr, e.h_cam, e.h_view, e.h_target, ar, e.n_ranks, e.rank);
}

// One throw-away pass through the three LM stages at handle creation: the first launch of every kernel carries a
// one-time cost on ROCm (code-object load, ~0.1-0.3 ms of per-kernel set-up) that added ~8 ms to the first solve of
// a small problem.  Paid once per process, device and kernel family; results are discarded (every solve re-evaluates from
// the parameters).
void warm_lm(Engine& e) {
    cba_options o{};
    o.max_iterations = 1;
    const bool resident = resident_lm_eligible(e, o);  // default options decide which form this handle will use first
    {   // the cost is per process and device (code objects, kernel set-up), not per handle: pay it once per kernel family
        static std::mutex mu;
        static std::set<uint32_t> warmed;
        const uint32_t key = (static_cast<uint32_t>(e.device) << 8) | (resident ? 0x80u : 0u) | (e.scalar ? 0x40u : 0u) |
                             (static_cast<uint32_t>(e.chain) << 2) | static_cast<uint32_t>(e.model);
        std::lock_guard<std::mutex> lock(mu);
        if (!warmed.insert(key).second) return;
    }
    if (resident) { resident_lm_warm(e); return; }
    HipBackend be(e, *lm_state(e));
    const Structure& s = lm_state(e)->s;
    if (s.n_blocks == 0) return;
    std::vector<double> cam_acc, S, g, d(std::max(1, s.nsh), 0.0);
    double c2[2], gm = 0.0;
    int nf = 0;
    std::vector<int32_t> fixed(s.n_views, 0);
    be.set_view_fixed(fixed);
    be.upload_shared(0, e.h_intr.data(), e.h_cam.data(), e.h_target.data());
    be.normal_eq_schur(1.0, cam_acc, c2, 1e4, true, false, S, g, &gm, &nf);
    std::vector<double> S2, g2;
    be.schur(1e4, false, false, S2, g2, &gm, &nf);
    be.upload_shared(1, e.h_intr.data(), e.h_cam.data(), e.h_target.data());
    TrialStats ts;
    be.trial(d.data(), 1.0, &ts);
    {   // the packed forms of the same stages (k_pack, k_accept_blocks) — without a collective: warm-up is per process, not per group
        const PackLayout L(s, 1);
        std::vector<double> pack(static_cast<size_t>(L.size), 0.0);
        const AllReduce none = [](double*, int64_t) {};
        void* comm = e.rccl_comm;
        e.rccl_comm = nullptr;
        be.sys_new(1.0, 1e4, true, false, L, none, 0, pack.data());
        be.sys_resolve(1e4, false, L, none, 0, pack.data());
        be.upload_shared(1, e.h_intr.data(), e.h_cam.data(), e.h_target.data());
        (void)be.sys_step(d.data(), 1.0, 1e4, false, L, none, 0, pack.data());
        // (the step is NOT accepted: even with a zero shared step the trial poses differ from the current ones; the kernel of
        // accept_step is set up by an empty launch)
        hipLaunchKernelGGL(k_accept_blocks, dim3(1), dim3(64), 0, e.stream, static_cast<int64_t>(0), e.blk_acc_alt.p, e.blk_acc.p,
                           static_cast<int64_t>(0), e.blk_w_alt.p, e.blk_w.p);
        // the controller kernel: one invocation out of turn (the control scalars say the solve has ended: it changes nothing)
        be.ensure_ctl(L);
        double* stage = lm_state(e)->ctl_rec.p + CS_COUNT;
        ctl_reset(stage);
        stage[CS_TERM] = 0.0;
        lm_state(e)->ctl_buf.upload(stage, CS_COUNT, e.stream);
        be.run_ctl(CTL_NEW, 0);
        e.rccl_comm = comm;
    }
    lm_state(e)->g_new.uses = lm_state(e)->g_schur.uses = lm_state(e)->g_trial.uses = 0;
    CBA_HIP(hipStreamSynchronize(e.stream));
}

void solve_lm(Engine& e, const cba_options& o, cba_summary* out) {
    if (resident_lm_eligible(e, o)) {  // small problem: the whole iteration in one kernel launch
        for (int k = 0; k < 8; ++k) lm_state(e)->xs[k] = 0;
        if (resident_lm_solve(e, o, out)) return;
    }
    HipBackend be(e, *lm_state(e));
    LMDriver drv = make_driver(e, be);
    try {
        drv.solve(o, out);
    } catch (...) {
        // this rank leaves the solve: its peers may be inside (or about to enter) the collective of the same step - abort the
        // communicator so that they fail too instead of waiting for a contribution that will never come
        e.gate = nullptr;
        rccl_abort(e);
        throw;
    }
    {
        const ExchangeStats& x = drv.exchange_stats();
        int64_t* xs = lm_state(e)->xs;
        xs[0] = x.allreduce_calls; xs[1] = x.allreduce_doubles; xs[2] = x.speculative_steps; xs[3] = x.speculation_hits;
        xs[4] = x.speculation_misses; xs[5] = x.rejected_steps; xs[6] = x.line_searches; xs[7] = x.line_search_evaluations;
    }
    // leave copy 0 on the device equal to the host state
    e.intr[0].upload(e.h_intr.data(), e.h_intr.size(), e.stream);
    CBA_HIP(hipStreamSynchronize(e.stream));
}

int64_t covariance_dim(const Engine& e) {
    int64_t n = static_cast<int64_t>(e.n_cams) * e.PI;
    if (e.chain != CBA_CHAIN_INTRINSIC) n += 7LL * e.n_cams;
    if (e.chain != CBA_CHAIN_BUNDLE) n += 7LL * e.n_views;
    else n += 7;
    return n;
}

int64_t shared_covariance_dim(const Engine& e) {
    if (e.chain == CBA_CHAIN_BUNDLE) return covariance_dim(e);
    int64_t n = static_cast<int64_t>(e.n_cams) * e.PI;
    if (e.chain != CBA_CHAIN_INTRINSIC) n += 7LL * e.n_cams;
    return n;
}

void compute_covariance(Engine& e, const cba_options& o, double* cov, bool shared_only) {
    HipBackend be(e, *lm_state(e));
    LMDriver drv = make_driver(e, be);
    drv.covariance(o, cov, shared_only);
}
void compute_covariance_views(Engine& e, const cba_options& o, const int32_t* views, int n_sel, double* view_cov) {
    HipBackend be(e, *lm_state(e));
    LMDriver drv = make_driver(e, be);
    drv.covariance(o, nullptr, true, views, n_sel, view_cov);
}

}  // namespace cba
