// wave_reduce.hpp — fixed-order wave64 reductions in DPP (device only).
#pragma once
#include <hip/hip_runtime.h>

namespace cba {

// DPP move of a 64-bit value (two 32-bit v_mov_b32_dpp); lanes masked off by ROW_MASK, or whose source
// lane is out of range, receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Wave-wide sum in pure VALU/DPP (no LDS round trips; the ds_bpermute butterfly cost ~70 % of Mode B
// when 77+ accumulators are reduced per tile).  Fixed order => bitwise reproducible.  The total lands in
// LANE 63 (other lanes hold partial sums).
__device__ __forceinline__ double wave_sum63(double v) {
    v += dpp_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_f64<0x140, 0xF>(v);  // row_mirror          -> every lane holds its 16-lane row sum
    v += dpp_f64<0x142, 0xA>(v);  // row_bcast15 into rows 1, 3
    v += dpp_f64<0x143, 0xC>(v);  // row_bcast31 into rows 2, 3 -> lane 63 holds the wave sum
    return v;
}

// Transposed wave reduction: every lane enters with N partial sums w[0..N) (N a multiple of 16) and leaves with the
// WAVE TOTALS of TransposeSum<N>::CNT of them in w[0..CNT): entry j of an OWNING lane's result is element (return value + j).
// Each butterfly step halves the working set — a lane keeps the lower or upper half (by one bit of its lane id), hands the
// other half to its partner and adds what the partner hands back — so the whole reduction costs ~7 VALU ops per accumulator
// instead of the 18 of an independent 6-step wave sum per accumulator.  Steps 1, 2 exchange through DPP quad permutes, 4/8/16
// through ds_swizzle, 32 through ds_bpermute (LDS crossbar only, no memory).  Fixed order: bitwise reproducible.
// The four in-row steps (xor 1, 2, 4, 8) always halve (N % 16 == 0).  The two cross-row steps (xor 16, 32) halve while the
// per-lane count is even; once it is odd they become plain exchange-and-add steps, after which both partners hold the same
// totals and only the lane whose bit is clear owns them — so N is padded to a multiple of 16, not 64 (77 sums cost a
// reduction of 80, not 128).
template <int PATTERN>
__device__ __forceinline__ double ds_swizzle_f64(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bpermute_xor32_f64(double v, int lane) {
    const int addr = (lane ^ 32) << 2;
    const int rl = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int rh = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(rh, rl);
}

template <int N>
struct TransposeSum {
    static_assert(N % 16 == 0, "pad the accumulator vector to a multiple of 16");
    static constexpr int R16 = N / 16;            // per lane after the four in-row steps
    static constexpr bool H16 = R16 % 2 == 0;     // the xor-16 step halves
    static constexpr int R32 = H16 ? R16 / 2 : R16;
    static constexpr bool H32 = R32 % 2 == 0;     // the xor-32 step halves
    static constexpr int CNT = H32 ? R32 / 2 : R32;  // finished sums per owning lane
    static constexpr int pad(int n) { return (n + 15) / 16 * 16; }
};

template <int N>
__device__ __forceinline__ int wave_transpose_sum(double* w, int lane, bool* owner) {
    using TS = TransposeSum<N>;
    int base = 0;
#define CBA_XSTEP(H, SEL, XCHG)                                      \
    {                                                                \
        const bool sel = (SEL);                                      \
        _Pragma("unroll") for (int i = 0; i < (H); ++i) {            \
            const double lo = w[i], hi = w[i + (H)];                 \
            const double send = sel ? lo : hi;                       \
            const double keep = sel ? hi : lo;                       \
            w[i] = keep + XCHG(send);                                \
        }                                                            \
        base += sel ? (H) : 0;                                       \
    }
#define CBA_X1(v) dpp_f64<0xB1, 0xF>(v)          /* quad_perm [1,0,3,2]: lane ^ 1 */
#define CBA_X2(v) dpp_f64<0x4E, 0xF>(v)          /* quad_perm [2,3,0,1]: lane ^ 2 */
#define CBA_X4(v) ds_swizzle_f64<0x101F>(v)      /* bit mode, xor 4 */
#define CBA_X8(v) ds_swizzle_f64<0x201F>(v)      /* xor 8 */
#define CBA_X16(v) ds_swizzle_f64<0x401F>(v)     /* xor 16 */
#define CBA_X32(v) bpermute_xor32_f64(v, lane)   /* xor 32 */
    CBA_XSTEP(N / 2, lane & 1, CBA_X1)
    CBA_XSTEP(N / 4, lane & 2, CBA_X2)
    CBA_XSTEP(N / 8, lane & 4, CBA_X4)
    CBA_XSTEP(N / 16, lane & 8, CBA_X8)
    if constexpr (TS::H16) {
        CBA_XSTEP(TS::R16 / 2, lane & 16, CBA_X16)
    } else {
#pragma unroll
        for (int i = 0; i < TS::R16; ++i) w[i] += CBA_X16(w[i]);
    }
    if constexpr (TS::H32) {
        CBA_XSTEP(TS::R32 / 2, lane & 32, CBA_X32)
    } else {
#pragma unroll
        for (int i = 0; i < TS::R32; ++i) w[i] += CBA_X32(w[i]);
    }
#undef CBA_XSTEP
#undef CBA_X1
#undef CBA_X2
#undef CBA_X4
#undef CBA_X8
#undef CBA_X16
#undef CBA_X32
    *owner = (TS::H16 || !(lane & 16)) && (TS::H32 || !(lane & 32));
    return base;
}

}  // namespace cba
