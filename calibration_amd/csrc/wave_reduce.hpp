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

}  // namespace cba
