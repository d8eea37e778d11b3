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
// The packed accumulator vector [H | g | s] is split over several launches ("parts", see the split policies below) so that
// one lane's share stays in registers; every part re-evaluates the Jacobian rows it needs.
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

// ---- how the packed accumulator vector [H | g | s] is split over the launches ("parts") ----------------------------------
// A split policy maps a packed entry e to (part, index within the part) and back.
//   SplitRoundRobin<NP>: e -> (e % NP, e / NP): equal shares, every part needs every Jacobian column.
//   SplitPoseIntr (one-pose chain): part 0 = the six pose rows of H (pose-pose and pose-intrinsics) and the pose gradient,
//     part 1 = the intrinsics-intrinsics block, the intrinsics gradient and |r|^2.  Part 1 never touches a pose column, so
//     the compiler drops d(u,v)/dP and the twelve pose entries from its row evaluation (~half of the row cost); the packed
//     order makes both inverses closed forms (H is stored row by row, so the pose rows are a prefix of it).
template <int NP>
struct SplitRoundRobin {
    static constexpr int parts = NP;
    static constexpr int count(int PL, int part) { return (PL * (PL + 1) / 2 + PL + 1 - part + NP - 1) / NP; }
    static constexpr int part_of(int PL, int e) { return e % NP; }
    static constexpr int local(int PL, int e) { return e / NP; }
    static __device__ __forceinline__ int entry(int PL, int part, int l) { return l * NP + part; }
};
struct SplitPoseIntr {
    static constexpr int parts = 2;
    static constexpr int nh(int PL) { return PL * (PL + 1) / 2; }
    static constexpr int e6(int PL) { return 6 * PL - 15; }  // packed entries of H rows 0..5
    static constexpr int count(int PL, int part) { return part == 0 ? e6(PL) + 6 : nh(PL) - e6(PL) + (PL - 6) + 1; }
    static constexpr int part_of(int PL, int e) { return e < e6(PL) ? 0 : e < nh(PL) ? 1 : e < nh(PL) + 6 ? 0 : 1; }
    static constexpr int local(int PL, int e) {
        return e < e6(PL) ? e : e < nh(PL) ? e - e6(PL) : e < nh(PL) + 6 ? e6(PL) + (e - nh(PL)) : (nh(PL) - e6(PL)) + (e - nh(PL) - 6);
    }
    static __device__ __forceinline__ int entry(int PL, int part, int l) {
        if (part == 0) return l < e6(PL) ? l : nh(PL) + (l - e6(PL));
        return l < nh(PL) - e6(PL) ? e6(PL) + l : nh(PL) + 6 + (l - (nh(PL) - e6(PL)));
    }
};

// out[e] (e in this part, e < NACC) = this tile's sums; the other entries of the row are left alone
template <int CHAIN, int MODEL, class SPLIT, int PART, typename T>
__device__ __forceinline__ void normal_eq_tile_split(const Tile t, int lane, const T* bcp, const T* ip,
                                                     const T* sp, const T* X, const T* Y,
                                                     const T* u, const T* v, double* out) {
    constexpr int PL = LocalCols<CHAIN, MODEL>::value;
    constexpr int NH = PL * (PL + 1) / 2;
    constexpr int NACC = NH + PL + 1;
    constexpr int NLOC = SPLIT::count(PL, PART);  // this part's share
    constexpr int NPAD = TransposeSum<16>::pad(NLOC);
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
    for (int k = 0;; ++k) {  // (until the tile is done: the break below)
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
                    if (SPLIT::part_of(PL, e) == PART) {
                        if (M::u(a) && M::u(b)) acc[SPLIT::local(PL, e)] = __builtin_fma(Ju[a], Ju[b], acc[SPLIT::local(PL, e)]);
                        if (M::v(a) && M::v(b)) acc[SPLIT::local(PL, e)] = __builtin_fma(Jv[a], Jv[b], acc[SPLIT::local(PL, e)]);
                    }
                    ++e;
                }
            }
#pragma unroll
            for (int a = 0; a < PL; ++a) {
                if (SPLIT::part_of(PL, NH + a) == PART) {
                    if (M::u(a)) acc[SPLIT::local(PL, NH + a)] = __builtin_fma(Ju[a], rr[0], acc[SPLIT::local(PL, NH + a)]);
                    if (M::v(a)) acc[SPLIT::local(PL, NH + a)] = __builtin_fma(Jv[a], rr[1], acc[SPLIT::local(PL, NH + a)]);
                }
            }
            if (SPLIT::part_of(PL, NH + PL) == PART)
                acc[SPLIT::local(PL, NH + PL)] = __builtin_fma(rr[1], rr[1], __builtin_fma(rr[0], rr[0], acc[SPLIT::local(PL, NH + PL)]));
        }
        xc = xn; yc = yn; uc = un; vc = vn;
    }
    // wave totals: an owning lane ends up with TransposeSum<NPAD>::CNT of this part's entries (wave_reduce.hpp)
    bool owner;
    const int base = wave_transpose_sum<NPAD>(acc, lane, &owner);
#pragma unroll
    for (int j = 0; j < TransposeSum<NPAD>::CNT; ++j) {
        const int l = base + j;
        if (owner && l < NLOC) {
            const int e = SPLIT::entry(PL, PART, l);
            if (e < NACC) out[e] = acc[j];
        }
    }
}

// the round-robin split of the first version (still what the resident kernel and the direct two-pose form use)
template <int CHAIN, int MODEL, int NPARTS, int PART, typename T>
__device__ __forceinline__ void normal_eq_tile(const Tile t, int lane, const T* bcp, const T* ip,
                                               const T* sp, const T* X, const T* Y,
                                               const T* u, const T* v, double* out) {
    normal_eq_tile_split<CHAIN, MODEL, SplitRoundRobin<NPARTS>, PART, T>(t, lane, bcp, ip, sp, X, Y, u, v, out);
}

// ---- rows / accumulate split of both forms: what kernels_modeb.hip hands from the wavefront that evaluated an observation to
// the wavefronts that accumulate the other parts ------------------------------------------------------------------------------
// Direct form rows: w = [ r_u, r_v | Ju pose columns | Jv pose columns | live intrinsics entries of the u row | of the v row ]
// (structural constants of the intrinsics columns are not shipped: see MomRows in reproj_math.hpp).
template <int CHAIN, int MODEL>
struct DirectRows {
    static constexpr int PI = IntrSize<MODEL>::value, OI = CHAIN == CH_INTRINSIC ? 6 : 12, PL = OI + PI;
    static constexpr int N = 2 + 2 * OI + MomRows<PI>::NU + MomRows<PI>::NV;
};

template <int CHAIN, int MODEL, typename T>
__device__ __forceinline__ void direct_rows(const T* bcp, const T* ip, const T* sp, T x, T y, T uo, T vo, double* w) {
    using D = DirectRows<CHAIN, MODEL>;
    using R = MomRows<D::PI>;
    T rt[2], Jut[D::PL], Jvt[D::PL];
    reproj_point<CHAIN, MODEL, T>(bcp, ip, sp, x, y, uo, vo, rt, Jut, Jvt);
    w[0] = rt[0]; w[1] = rt[1];
    int n = 2;
#pragma unroll
    for (int a = 0; a < D::OI; ++a) w[n++] = Jut[a];
#pragma unroll
    for (int a = 0; a < D::OI; ++a) w[n++] = Jvt[a];
#pragma unroll
    for (int j = 0; j < D::PI; ++j)
        if (R::u_live(j)) w[n++] = Jut[D::OI + j];
#pragma unroll
    for (int j = 0; j < D::PI; ++j)
        if (R::v_live(j)) w[n++] = Jvt[D::OI + j];
}

template <int CHAIN, int MODEL, class SPLIT, int PART>
__device__ __forceinline__ void direct_accumulate(const double* w, double* acc) {
    using D = DirectRows<CHAIN, MODEL>;
    using R = MomRows<D::PI>;
    using M = RowMask<CHAIN>;
    constexpr int PL = D::PL, NH = PL * (PL + 1) / 2;
    double rr[2] = {w[0], w[1]}, Ju[PL], Jv[PL];
    {
        int n = 2;
#pragma unroll
        for (int a = 0; a < D::OI; ++a) Ju[a] = w[n++];
#pragma unroll
        for (int a = 0; a < D::OI; ++a) Jv[a] = w[n++];
#pragma unroll
        for (int j = 0; j < D::PI; ++j) Ju[D::OI + j] = R::u_live(j) ? w[n++] : (j == 2 ? 1.0 : 0.0);
#pragma unroll
        for (int j = 0; j < D::PI; ++j) Jv[D::OI + j] = R::v_live(j) ? w[n++] : (j == 3 ? 1.0 : 0.0);
        Jv[D::OI + 1] = Ju[D::OI + 4];  // d v / d fy = d u / d skew (not shipped twice)
    }
    int e = 0;
#pragma unroll
    for (int a = 0; a < PL; ++a) {
#pragma unroll
        for (int b = a; b < PL; ++b) {
            if (SPLIT::part_of(PL, e) == PART) {
                if (M::u(a) && M::u(b)) acc[SPLIT::local(PL, e)] = __builtin_fma(Ju[a], Ju[b], acc[SPLIT::local(PL, e)]);
                if (M::v(a) && M::v(b)) acc[SPLIT::local(PL, e)] = __builtin_fma(Jv[a], Jv[b], acc[SPLIT::local(PL, e)]);
            }
            ++e;
        }
    }
#pragma unroll
    for (int a = 0; a < PL; ++a) {
        if (SPLIT::part_of(PL, NH + a) == PART) {
            if (M::u(a)) acc[SPLIT::local(PL, NH + a)] = __builtin_fma(Ju[a], rr[0], acc[SPLIT::local(PL, NH + a)]);
            if (M::v(a)) acc[SPLIT::local(PL, NH + a)] = __builtin_fma(Jv[a], rr[1], acc[SPLIT::local(PL, NH + a)]);
        }
    }
    if (SPLIT::part_of(PL, NH + PL) == PART)
        acc[SPLIT::local(PL, NH + PL)] = __builtin_fma(rr[1], rr[1], __builtin_fma(rr[0], rr[0], acc[SPLIT::local(PL, NH + PL)]));
}

// the tile's sum of squared residuals; the total lands in LANE 63
template <int MODEL, typename T>
__device__ __forceinline__ double resid_tile(const Tile t, int lane, const T* bcp, const T* ip,
                                             const T* sp, const T* X, const T* Y,
                                             const T* u, const T* v) {
    double s = 0.0;
#pragma unroll 4
    for (int k = 0;; ++k) {  // (until the tile is done: the break below)
        const int j = lane + 64 * k;
        if (64 * k >= t.count) break;  // wave-uniform
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
