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

// max over the 64 lanes, returned wave-uniform.  4 DPP stages reduce each 16-lane row, then the
// four row results are combined through readlane.  NaN operands are ignored (v_max_f32 maxNum).
__device__ __forceinline__ float wave_max_f32(float v)
{
    v = __builtin_fmaxf(v, dpp_f32<kDppQuadXor1>(v));
    v = __builtin_fmaxf(v, dpp_f32<kDppQuadXor2>(v));
    v = __builtin_fmaxf(v, dpp_f32<kDppRowHalfMirror>(v));
    v = __builtin_fmaxf(v, dpp_f32<kDppRowMirror>(v));
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}

__device__ __forceinline__ float readlane_f32(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
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
