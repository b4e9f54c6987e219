// wave_ops.hpp -- cross-lane sums for 64-wide CDNA4 wavefronts.
//
// The reference reduces a warp's partial dot products with a 32-wide
// __shfl_down_sync tree (cuda_src/csr_matrix_cuda.cu:185-187,
// cuda_src/hll_matrix.cu:413-416).  On gfx950 a wavefront has 64 lanes and
// the cheap cross-lane paths are, in order of cost:
//   DPP modifiers on a VALU move (inside a row of 16 lanes, no LDS traffic),
//   ds_swizzle (inside 32 lanes, uses the LDS crossbar but no LDS memory),
//   ds_bpermute (any lane to any lane).
// group_sum<W>() is a butterfly (xor) all-reduce over aligned groups of W
// lanes built from exactly those three: every lane of the group ends with the
// group's sum, so callers may store from any lane.
#pragma once
#include <hip/hip_runtime.h>

namespace spmv {

// DPP control words (GFX9 encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i  inside each 8
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15-i inside each 16
constexpr int SWIZZLE_XOR16 = 0x401F;      // bit-mode: and 0x1f, or 0, xor 0x10

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}

// One butterfly stage: fetch the partner lane's value.  STEP is the xor
// distance (1, 2, 4, 8, 16, 32).  For 4 and 8 the mirror patterns pair lane i
// with a lane of the partner quad/half-row that -- after the earlier stages --
// holds the same partial sum as lane i^STEP, so the result is identical.
template <int STEP>
__device__ __forceinline__ int partner_i32(int v) {
    if constexpr (STEP == 1) return dpp_i32<DPP_QUAD_XOR1>(v);
    else if constexpr (STEP == 2) return dpp_i32<DPP_QUAD_XOR2>(v);
    else if constexpr (STEP == 4) return dpp_i32<DPP_ROW_HALF_MIRROR>(v);
    else if constexpr (STEP == 8) return dpp_i32<DPP_ROW_MIRROR>(v);
    else if constexpr (STEP == 16) return __builtin_amdgcn_ds_swizzle(v, SWIZZLE_XOR16);
    else {
        static_assert(STEP == 32, "xor distance must be a power of two <= 32");
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        return __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, v);
    }
}

template <int STEP>
__device__ __forceinline__ double partner(double v) {
    const int lo = partner_i32<STEP>(__double2loint(v));
    const int hi = partner_i32<STEP>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int STEP>
__device__ __forceinline__ float partner(float v) {
    return __int_as_float(partner_i32<STEP>(__float_as_int(v)));
}

// Sum over aligned groups of W lanes (W = 1, 2, 4, ..., 64); result in every lane.
template <int W, typename T>
__device__ __forceinline__ T group_sum(T v) {
    static_assert(W >= 1 && W <= 64 && (W & (W - 1)) == 0, "group width");
    if constexpr (W >= 2) v += partner<1>(v);
    if constexpr (W >= 4) v += partner<2>(v);
    if constexpr (W >= 8) v += partner<4>(v);
    if constexpr (W >= 16) v += partner<8>(v);
    if constexpr (W >= 32) v += partner<16>(v);
    if constexpr (W >= 64) v += partner<32>(v);
    return v;
}

// Sum over the lanes that share (lane % RW), RW a power of two: the partial sums of one
// row when RW neighbouring rows sit in neighbouring lanes and a row's slices are RW lanes
// apart.  Every step must keep lane % RW, so the 4- and 8-steps rotate inside the 16-lane
// DPP row (row_ror) instead of using the mirror patterns of group_sum.
template <int STEP>
__device__ __forceinline__ int rotate_i32(int v) {
    if constexpr (STEP == 1) return dpp_i32<DPP_QUAD_XOR1>(v);
    else if constexpr (STEP == 2) return dpp_i32<DPP_QUAD_XOR2>(v);
    else if constexpr (STEP == 4) return dpp_i32<0x124>(v);   // row_ror:4
    else if constexpr (STEP == 8) return dpp_i32<0x128>(v);   // row_ror:8
    else return partner_i32<STEP>(v);                         // xor 16 / xor 32
}
template <int STEP>
__device__ __forceinline__ double rotate(double v) {
    return __hiloint2double(rotate_i32<STEP>(__double2hiint(v)), rotate_i32<STEP>(__double2loint(v)));
}
template <int STEP>
__device__ __forceinline__ float rotate(float v) {
    return __int_as_float(rotate_i32<STEP>(__float_as_int(v)));
}
template <typename T>
__device__ __forceinline__ T strided_sum(T v, int rw) {  // rw wave-uniform
    if (rw <= 1) v += rotate<1>(v);
    if (rw <= 2) v += rotate<2>(v);
    if (rw <= 4) v += rotate<4>(v);
    if (rw <= 8) v += rotate<8>(v);
    if (rw <= 16) v += rotate<16>(v);
    if (rw <= 32) v += rotate<32>(v);
    return v;
}

// Runtime-width dispatch for wave-uniform w.
template <typename T>
__device__ __forceinline__ T group_sum_rt(T v, int w) {
    switch (w) {
        case 1: return v;
        case 2: return group_sum<2>(v);
        case 4: return group_sum<4>(v);
        case 8: return group_sum<8>(v);
        case 16: return group_sum<16>(v);
        case 32: return group_sum<32>(v);
        default: return group_sum<64>(v);
    }
}

}  // namespace spmv
