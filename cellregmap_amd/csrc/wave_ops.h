// Lane-to-lane primitives of a 64-wide wavefront shared by the kernels that reduce across lanes (nullfit.hip, davies.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace crm {

// Sums over the 64 lanes, every lane ending with bitwise the same totals -- the xor butterfly
//   for (off = 32, 16, 8, 4, 2, 1) v += shfl_xor(v, off)
// with the same pairings in the same order, so every bit is the butterfly's -- but level by level for all N values of a
// pass at once (nullfit.hip -- the seven sums of an evaluation: seven dependent chains of six trips through the LDS crossbar each cost
// more than the pass over a short spectrum itself; side by side they wait six times, not forty-two) and without the
// crossbar's address arithmetic where the hardware has the permutation built in: xor 16 / 8 / 4 as ds_swizzle bit masks,
// xor 2 / 1 as DPP quad permutations.
__device__ inline double lane_xor_swizzle(double v, const int level) {   // level 16, 8 or 4: lane ^ level inside 32 lanes
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (level) {
        case 16: lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F); hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F); break;
        case 8: lo = __builtin_amdgcn_ds_swizzle(lo, 0x201F); hi = __builtin_amdgcn_ds_swizzle(hi, 0x201F); break;
        default: lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F); hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F); break;
    }
    return __hiloint2double(hi, lo);
}

__device__ inline double lane_xor_quad(double v, const int level) {      // level 2 or 1: DPP quad_perm [2,3,0,1] / [1,0,3,2]
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (level == 2) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}

__device__ inline double read_lane(double v, const int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// One value over the 64 lanes, the same butterfly: bit for bit  for (off = 32 .. 1) v += __shfl_xor(v, off).
__device__ inline double wave_sum_butterfly(double v) {
    v += __shfl_xor(v, 32, 64);
    v += lane_xor_swizzle(v, 16);
    v += lane_xor_swizzle(v, 8);
    v += lane_xor_swizzle(v, 4);
    v += lane_xor_quad(v, 2);
    v += lane_xor_quad(v, 1);
    return v;
}

// The total of one value over the 64 lanes, in every lane (wave-uniform: it comes back through v_readlane), by the DPP
// row operations of the vector ALU alone -- an inclusive scan inside each row of sixteen (row_shr 1, 2, 4, 8), lane 15 of
// rows 0 and 2 into rows 1 and 3 (row_bcast15), lane 31 into rows 2 and 3 (row_bcast31): the total stands in lane 63.
// No trip through the LDS crossbar: about a third of the butterfly's latency.  NOT the butterfly's order of summation --
// for sums whose bits nothing is pinned to (davies.hip).
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_fetch(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_total(double v) {
    v += dpp_fetch<0x111, 0xF>(v);
    v += dpp_fetch<0x112, 0xF>(v);
    v += dpp_fetch<0x114, 0xF>(v);
    v += dpp_fetch<0x118, 0xF>(v);
    v += dpp_fetch<0x142, 0xA>(v);
    v += dpp_fetch<0x143, 0xC>(v);
    return read_lane(v, 63);
}
__device__ inline int wave_total(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace crm
