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

// Transposed wave reduction: every lane enters with N_PAD partial sums w[0..N_PAD) (N_PAD a multiple of 64) and
// leaves with the WAVE TOTALS of N_PAD/64 of them in w[0..N_PAD/64): entry j of the lane's result is element
// (return value + j).  Each butterfly step halves the working set — a lane keeps the lower or upper half (by one
// bit of its lane id), hands the other half to its partner and adds what the partner hands back — so the whole
// reduction costs ~7 VALU ops per accumulator instead of the 18 of an independent 6-step wave sum per
// accumulator.  Steps 1, 2 exchange through DPP quad permutes, 4/8/16 through ds_swizzle, 32 through ds_bpermute
// (LDS crossbar only, no memory).  Fixed order: bitwise reproducible.
template <int PATTERN>
__device__ __forceinline__ double ds_swizzle_f64(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
    return __hiloint2double(hi, lo);
}

template <int N_PAD>
__device__ __forceinline__ int wave_transpose_sum(double* w, int lane) {
    static_assert(N_PAD % 64 == 0, "pad the accumulator vector to a multiple of the wave width");
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
    CBA_XSTEP(N_PAD / 2, lane & 1, CBA_X1)
    CBA_XSTEP(N_PAD / 4, lane & 2, CBA_X2)
    CBA_XSTEP(N_PAD / 8, lane & 4, CBA_X4)
    CBA_XSTEP(N_PAD / 16, lane & 8, CBA_X8)
    CBA_XSTEP(N_PAD / 32, lane & 16, CBA_X16)
    {
        constexpr int H = N_PAD / 64;
        const bool sel = lane & 32;
        const int addr = (lane ^ 32) << 2;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const double lo = w[i], hi = w[i + H];
            const double send = sel ? lo : hi;
            const double keep = sel ? hi : lo;
            const int rl = __builtin_amdgcn_ds_bpermute(addr, __double2loint(send));
            const int rh = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(send));
            w[i] = keep + __hiloint2double(rh, rl);
        }
        base += sel ? H : 0;
    }
#undef CBA_XSTEP
#undef CBA_X1
#undef CBA_X2
#undef CBA_X4
#undef CBA_X8
#undef CBA_X16
    return base;
}

}  // namespace cba
