// 64-lane wavefront primitives for gfx950 (CDNA4).  These replace the reference's 32-lane
// __shfl_*_sync trees (scan_kernel.cuh:140-334, sum_reduction.cuh:14-44, ldpc_kernel.cuh:9-30).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msk144
{

// IEEE float operations that must not be fused into FMAs (hipcc's default is -ffp-contract=fast) and a
// correctly rounded square root.  NB: HIP's __fsqrt_rn is the *native* (1 ulp) sqrt and
// __fadd_rn/__fmul_rn are plain operators in this toolchain, hence these helpers.
__device__ __forceinline__ float f32_add(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float f32_sub(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float f32_mul(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float f32_div(float a, float b)
{
    return a / b;  // correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt is hipcc's default
}
// x / D for a small integer constant D, correctly rounded (Markstein: r = RN(1/D), q = RN(x*r),
// e = x - q*D exactly by FMA, result RN(q + e*r)).
template<int D>
__device__ __forceinline__ float div_by_const(float x)
{
    constexpr float d = static_cast<float>(D);
    constexpr float r = 1.0f / d;
    const float q = f32_mul(x, r);
    const float e = fmaf(-q, d, x);
    return fmaf(e, r, q);
}
__device__ __forceinline__ float f32_sqrt(float x)
{
    return __builtin_sqrtf(x);  // v_sqrt_f32 + FMA refinement, correctly rounded
}

__device__ __forceinline__ int lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// DPP move of a 32-bit value; lanes whose source is outside the row/wave keep `v` (bound_ctrl off,
// old = v).
template<int kCtrl>
__device__ __forceinline__ float dpp_f32(float v)
{
    int i = __builtin_bit_cast(int, v);
    int r = __builtin_amdgcn_update_dpp(i, i, kCtrl, 0xF, 0xF, false);
    return __builtin_bit_cast(float, r);
}

constexpr int kDppQuadXor1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;

// max over the 64 lanes, returned wave-uniform.  Six fused v_max_f32_dpp steps: four reduce each
// 16-lane row (quad xor 1, quad xor 2, half mirror, mirror), row_bcast:15 / row_bcast:31 fold the rows
// into lane 63.  hipcc expands the builtin form into mov + nop + mov_dpp + canonicalise + max per step,
// hence the asm; the 2 wait states a DPP read needs after a VALU write are inside the string (hipcc does
// not insert hazard nops for asm statements); the leading s_nop 4 also covers the 5 wait states a DPP op
// needs after a VALU write of EXEC (v_cmpx) that the compiler may have placed right before the statement.
// NaN operands are ignored (v_max_f32 maxNum).
__device__ __forceinline__ float wave_max_f32(float v)
{
    float r;
    asm volatile("s_nop 4\n\t"
                 "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(r)
                 : "v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 63));
}

// min over each aligned group of 8 lanes, result in all 8 (quad xor 1, quad xor 2, half mirror).
__device__ __forceinline__ float oct_min_f32(float v)
{
    float r;
    asm volatile("s_nop 4\n\t"
                 "v_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(r)
                 : "v"(v));
    return r;
}

// max over each aligned group of 8 lanes for two independent values at once (the chains interleave: one s_nop 0 per step completes
// the two wait states a DPP read needs after the VALU write of its source).  NaN operands are ignored (v_max_f32 maxNum).
__device__ __forceinline__ void oct_max2_f32(float& a, float& b)
{
    float ra, rb;
    asm volatile("s_nop 4\n\t"
                 "v_max_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(ra), "=&v"(rb)
                 : "v"(a), "v"(b));
    a = ra;
    b = rb;
}

// v + (its quad/row partners): after the four steps every lane of a 16-lane row holds the row sum, added
// in the order lane^1, lane^2, other quad pair, other half - the reference's shuffle-tree association.
__device__ __forceinline__ float row_sum_f32(float v)
{
    float r;
    asm volatile("s_nop 4\n\t"
                 "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(r)
                 : "v"(v));
    return r;
}

// row_sum_f32 for two independent values at once: the two chains interleave, one s_nop 0 per step completes the two wait
// states a DPP read needs after the VALU write of its source.
__device__ __forceinline__ void row_sum2_f32(float& a, float& b)
{
    float ra, rb;
    asm volatile("s_nop 4\n\t"
                 "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(ra), "=&v"(rb)
                 : "v"(a), "v"(b));
    a = ra;
    b = rb;
}

__device__ __forceinline__ float readlane_f32(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Whole-wave sums of two independent values, (row0 + row1) + (row2 + row3) with the rows summed as in row_sum2_f32: after
// the four in-row steps, row_bcast:15 adds a row's total into the next row (rows 1 and 3 enabled), row_bcast:31 adds row 1's
// pair into row 3, and one readlane of lane 63 per value makes the result wave-uniform: 2 DPP + 1 readlane per value where
// reading the four row totals back costs 4 readlanes + 5 VALU (one SGPR operand per instruction).
__device__ __forceinline__ void wave_sum2_f32(float& a, float& b)
{
    float ra, rb;
    asm volatile("s_nop 4\n\t"
                 "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "=&v"(ra), "=&v"(rb)
                 : "v"(a), "v"(b));
    a = readlane_f32(ra, 63);
    b = readlane_f32(rb, 63);
}

// Sum of a 32-lane half of the wave with the reference's shuffle order (1,2,4,8,16); the result
// is valid in lanes 0 and 32.  Keeping this order makes 144-term block sums bit-compatible with
// sum_reduction_two_cycles on 32-lane warps.
__device__ __forceinline__ float half_tree_sum_f32(float v)
{
    v += __shfl_down(v, 1, 32);
    v += __shfl_down(v, 2, 32);
    v += __shfl_down(v, 4, 32);
    v += __shfl_down(v, 8, 32);
    v += __shfl_down(v, 16, 32);
    return v;
}

}  // namespace msk144
