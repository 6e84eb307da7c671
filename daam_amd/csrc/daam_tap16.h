// Lane-level pieces of the 16x16x32-MFMA cross-attention tiling (daam_tap_d64.hip, daam_attend_d64.hip).
//
// "Swapped" product S^T = K Q^T on v_mfma_f32_16x16x32_f16: A = K rows (lane: token row l&15 of a 16-row tile,
// k = 8*(l>>4)..+7 of the 32-wide k-step), B = Q^T (lane: pixel l&15, same k split).  C/D: lane holds pixel l&15 and
// tokens 16*mt + 4*(l>>4) + r (mt = 0..4, r = 0..3) = 20 of the 77 tokens of its pixel; the other 57 sit in the three
// lanes l^16, l^32, l^48.
#pragma once
#include "daam_tap_common.h"

namespace daam {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kD64Row = 160;                       // bytes per K row in LDS (128 + 32 pad)
constexpr int kD64Rows = 80;                       // 5 MFMA row tiles; rows 77..79 stay zero
constexpr int kD64KBuf = kD64Rows * kD64Row;       // 12800
constexpr int kSlots16 = 20;                       // token slots per lane

// token of slot i (= 4*mt + r) for lane quarter h
__device__ __forceinline__ constexpr int slot16_token(int i, int h) { return 16 * (i >> 2) + 4 * h + (i & 3); }

// all-reduce over the four lanes (l, l^16, l^32, l^48) that share a pixel, on the VALU
// (v_permlane16_swap / v_permlane32_swap exchange, no LDS crossbar)
__device__ __forceinline__ float quad_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// value of lane quarter 0 (lanes 0..15 = the lanes holding token 0) of each pixel, in all four of its lanes
__device__ __forceinline__ float quad_bcast0(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);   // r[0] = rows (0, 0, 2, 2)
    r = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);                                    // r[0] = rows (0, 0, 0, 0)
    return __uint_as_float(r[0]);
}
__device__ __forceinline__ float quad_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace daam
