// mode_b.hpp — the per-tile bodies of Mode B (normal equations) and Mode R (squared residual), device only.
//
// One WAVEFRONT owns one tile = up to TILE_B consecutive observations of one residual block.  k_normal_eq / k_resid
// (kernels_reproj.hip) run one tile per wavefront over the whole chip; the resident LM kernel (resident_lm.hip) walks
// the same tile table with the wavefronts of a single workgroup.  Both produce identical rows.
#pragma once
#include "engine.hpp"
#include "reproj_math.hpp"
#include "wave_reduce.hpp"

namespace cba {

// Per tile: H = sum J^T J (upper triangle, row-major packed), g = sum J^T r, s = sum |r|^2, all
// UNWEIGHTED (the per-block Huber weight is a scalar applied when blocks are assembled).
// The packed accumulator vector [H | g | s] is split round-robin over NPARTS passes so that one
// lane's share stays in registers; every part re-evaluates the (cheap) Jacobian rows.
// T = float: the Jacobian rows are evaluated in fp32 and widened once; every accumulator stays fp64.
//
// The u and v rows of an observation are accumulated as two chained FMAs per entry, and only where the
// row is not STRUCTURALLY zero: of the intrinsics columns [fx fy cx cy skew ...] the u row has no fy / cy
// entry and the v row no fx / cx / skew entry (reproj_math.hpp, both camera models), so about a quarter of
// the products of a naive J^T J vanish at compile time and six entries of H are identically zero.
template <int CHAIN>
struct RowMask {
    static constexpr int OI = CHAIN == CH_INTRINSIC ? 6 : 12;
    static constexpr bool u(int c) { return !(c == OI + 1 || c == OI + 3); }             // u row has an entry in column c
    static constexpr bool v(int c) { return !(c == OI + 0 || c == OI + 2 || c == OI + 4); }
};

// out[e] (e % NPARTS == PART, e < NACC) = this tile's sums; the other entries of the row are left alone
template <int CHAIN, int MODEL, int NPARTS, int PART, typename T>
__device__ __forceinline__ void normal_eq_tile(const Tile t, int lane, const T* bcp, const T* ip,
                                               const T* sp, const T* X, const T* Y,
                                               const T* u, const T* v, double* out) {
    constexpr int PL = LocalCols<CHAIN, MODEL>::value;
    constexpr int NH = PL * (PL + 1) / 2;
    constexpr int NACC = NH + PL + 1;
    constexpr int NLOC = (NACC + NPARTS - 1) / NPARTS;  // this part's share: entries e with e % NPARTS == PART, at e / NPARTS
    constexpr int NPAD = (NLOC + 63) / 64 * 64;
    using M = RowMask<CHAIN>;

    double acc[NPAD];
#pragma unroll
    for (int e = 0; e < NPAD; ++e) acc[e] = 0.0;

    // software pipeline: the four loads of pass k+1 are issued before the ~250 fp64 operations of pass k, so
    // that with only 2 waves per SIMD (the accumulators fill the register file) HBM latency hides behind the math
    T xc = T(0), yc = T(0), uc = T(0), vc = T(0);
    if (lane < t.count) { xc = X[t.xy_start + lane]; yc = Y[t.xy_start + lane]; uc = u[t.start + lane]; vc = v[t.start + lane]; }
    // drain the prologue loads here: with loads pending on loop entry the compiler's wait-count pass makes every
    // pass wait for its own prefetch (s_waitcnt vmcnt(3) before the first use), which defeats the pipeline
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll 1
    for (int k = 0; k < OPL_B; ++k) {
        const int j = lane + 64 * k;
        if (64 * k >= t.count) break;  // wave-uniform
        T xn = T(0), yn = T(0), un = T(0), vn = T(0);
        if (j + 64 < t.count) {
            const int64_t i = t.start + j + 64, k2 = t.xy_start + j + 64;
            xn = X[k2]; yn = Y[k2]; un = u[i]; vn = v[i];
        }
        if (j < t.count) {
            T rt[2], Jut[PL], Jvt[PL];
            reproj_point<CHAIN, MODEL, T>(bcp, ip, sp, xc, yc, uc, vc, rt, Jut, Jvt);
            double rr[2], Ju[PL], Jv[PL];
            rr[0] = rt[0]; rr[1] = rt[1];
#pragma unroll
            for (int a = 0; a < PL; ++a) { Ju[a] = Jut[a]; Jv[a] = Jvt[a]; }
            int e = 0;
#pragma unroll
            for (int a = 0; a < PL; ++a) {
#pragma unroll
                for (int b = a; b < PL; ++b) {
                    if ((e % NPARTS) == PART) {
                        if (M::u(a) && M::u(b)) acc[e / NPARTS] = __builtin_fma(Ju[a], Ju[b], acc[e / NPARTS]);
                        if (M::v(a) && M::v(b)) acc[e / NPARTS] = __builtin_fma(Jv[a], Jv[b], acc[e / NPARTS]);
                    }
                    ++e;
                }
            }
#pragma unroll
            for (int a = 0; a < PL; ++a) {
                if (((NH + a) % NPARTS) == PART) {
                    if (M::u(a)) acc[(NH + a) / NPARTS] = __builtin_fma(Ju[a], rr[0], acc[(NH + a) / NPARTS]);
                    if (M::v(a)) acc[(NH + a) / NPARTS] = __builtin_fma(Jv[a], rr[1], acc[(NH + a) / NPARTS]);
                }
            }
            if (((NH + PL) % NPARTS) == PART)
                acc[(NH + PL) / NPARTS] = __builtin_fma(rr[1], rr[1], __builtin_fma(rr[0], rr[0], acc[(NH + PL) / NPARTS]));
        }
        xc = xn; yc = yn; uc = un; vc = vn;
    }
    // wave totals: lane ends up owning NPAD/64 of this part's entries (wave_reduce.hpp)
    const int base = wave_transpose_sum<NPAD>(acc, lane);
#pragma unroll
    for (int j = 0; j < NPAD / 64; ++j) {
        const int e = (base + j) * NPARTS + PART;
        if (e < NACC) out[e] = acc[j];
    }
}

// the tile's sum of squared residuals; the total lands in LANE 63
template <int MODEL, typename T>
__device__ __forceinline__ double resid_tile(const Tile t, int lane, const T* bcp, const T* ip,
                                             const T* sp, const T* X, const T* Y,
                                             const T* u, const T* v) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < OPL_B; ++k) {
        const int j = lane + 64 * k;
        if (j < t.count) {
            const int64_t i = t.start + j, k2 = t.xy_start + j;
            T rr[2];
            reproj_residual<MODEL, T>(bcp, ip, sp, X[k2], Y[k2], u[i], v[i], rr);
            s += static_cast<double>(rr[0]) * rr[0] + static_cast<double>(rr[1]) * rr[1];
        }
    }
    return wave_sum63(s);
}

}  // namespace cba
