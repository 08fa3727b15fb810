// Mixing a window down by one frequency hypothesis: cdat2[n] = exp(i*phi_n) * cdat[n] with the
// reference's float phase phi_n = ((float(n)*2pi)*f0)/12000 (scan_kernel.cuh:49-56,
// softbits_kernel.cuh:32-39), shared by the scan and softbits kernels.
#pragma once

#include "wave64.h"

namespace msk144
{

// sin and cos of a float angle |phi| < ~1e4 rad, absolute error <= 1.5e-7 (about 1.3 ulp of 1.0; tools-side check: numpy
// emulation of this float32/FMA sequence against double precision on 2 M points): two-constant Cody-Waite reduction by PI
// with FMA, minimax polynomials on [-pi/2, pi/2] (sin: odd, degree 9; cos: even, degree 10).  Reducing by pi instead of
// pi/2 trades four extra polynomial terms for the whole quadrant fix-up: an odd multiple of pi only flips BOTH signs, one
// shift and two XORs, where the swap of sin and cos by quadrant cost thirteen bit operations.  k = rint(phi/pi) comes from
// the 1.5*2^23 add trick, whose float bits also hold k's parity.  Replaces ocml sincosf (~5x the instructions); the
// reference uses CUDA's sincosf, itself ~2 ulp.
__device__ __forceinline__ void sincos_reduced(float phi, float& sn, float& cs)
{
    constexpr float kRound = 12582912.0f;                             // 1.5 * 2^23: ulp = 1 in [2^23, 2^24)
    const float kb = fmaf(phi, 0.318309886183790672f, kRound);        // 1/pi; low mantissa bit = parity of k
    const float k = kb - kRound;                                      // exact
    float r = fmaf(-k, 3.1415927410125732f, phi);                        // fl(pi)
    r = fmaf(-k, -8.742277657347586e-08f, r);                       // pi - fl(pi)
    const float z = r * r;
    float sp = fmaf(2.635400051076431e-06f, z, -0.00019823024922516197f);
    sp = fmaf(sp, z, 0.008333245292305946f);
    sp = fmaf(sp, z, -0.1666666567325592f);
    const float s_r = fmaf(r * z, sp, r);
    float cp = fmaf(-2.6325329827159294e-07f, z, 2.4776121790637262e-05f);
    cp = fmaf(cp, z, -0.001388867967762053f);
    cp = fmaf(cp, z, 0.0416666604578495f);
    cp = fmaf(cp, z, -0.5f);
    const float c_r = fmaf(cp, z, 1.0f);
    const uint32_t flip = __builtin_bit_cast(uint32_t, kb) << 31;      // k odd: sin(r + k pi) = -sin r, cos(r + k pi) = -cos r
    sn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s_r) ^ flip);
    cs = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, c_r) ^ flip);
}

// x / 12000 correctly rounded without the hardware division sequence (Markstein): with r = RN(1/d),
// q = RN(x*r), e = x - q*d exactly (FMA), RN(q + e*r) is the correctly rounded quotient.
__device__ __forceinline__ float div_sample_rate(float x)
{
    constexpr float r = 1.0f / kSampleRate;  // compile-time, correctly rounded
    const float q = f32_mul(x, r);
    const float e = fmaf(-q, kSampleRate, x);
    return fmaf(e, r, q);
}

// nf = float(n): callers form it as float(tid) + float(i * stride), exact below 2^24 and one plain add per sample
// instead of an integer add plus a v_cvt_f32_u32
__device__ __forceinline__ float mix_phase(float nf, float f0)
{
    const float twopi = 2.0f * 3.14159265358979323846f;
    return div_sample_rate(f32_mul(f32_mul(nf, twopi), f0));
}

__device__ __forceinline__ float2 mix_sample(float2 x, float nf, float f0)
{
    float sn, cs;
    sincos_reduced(mix_phase(nf, f0), sn, cs);
    float2 y;
    y.x = cs * x.x - sn * x.y;
    y.y = cs * x.y + sn * x.x;
    return y;
}

}  // namespace msk144
