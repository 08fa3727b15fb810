// Mixing a window down by one frequency hypothesis: cdat2[n] = exp(i*phi_n) * cdat[n] with the
// reference's float phase phi_n = ((float(n)*2pi)*f0)/12000 (scan_kernel.cuh:49-56,
// softbits_kernel.cuh:32-39), shared by the scan and softbits kernels.
#pragma once

#include "wave64.h"

namespace msk144
{

// sin and cos of a float angle |phi| < ~1e4 rad, about 1.5 ulp: two-constant Cody-Waite reduction by
// pi/2 with FMA, cephes minimax polynomials on [-pi/4, pi/4].  Replaces ocml sincosf (~4x the
// instructions); the reference uses CUDA's sincosf, itself ~2 ulp.
__device__ __forceinline__ void sincos_reduced(float phi, float& sn, float& cs)
{
    const float k = __builtin_rintf(phi * 0.636619772367581343f);  // 2/pi
    float r = fmaf(-k, 1.57079637050628662109375f, phi);           // fl(pi/2)
    r = fmaf(-k, -4.37113900018624283e-8f, r);                      // pi/2 - fl(pi/2)
    const int q = static_cast<int>(k);
    const float z = r * r;
    float sp = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = fmaf(sp, z, -1.6666654611e-1f);
    const float s_r = fmaf(r * z, sp, r);
    float cp = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = fmaf(cp, z, 4.166664568298827e-2f);
    const float c_r = fmaf(z * z, cp, fmaf(-0.5f, z, 1.0f));
    const bool swap = (q & 1) != 0;
    float s = swap ? c_r : s_r;
    float c = swap ? s_r : c_r;
    // quadrant signs: sin negative for q = 2,3; cos negative for q = 1,2
    if(q & 2) s = -s;
    if((q + 1) & 2) c = -c;
    sn = s;
    cs = c;
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

__device__ __forceinline__ float mix_phase(int n, float f0)
{
    const float twopi = 2.0f * 3.14159265358979323846f;
    return div_sample_rate(f32_mul(f32_mul(static_cast<float>(n), twopi), f0));
}

__device__ __forceinline__ float2 mix_sample(float2 x, int n, float f0)
{
    float sn, cs;
    sincos_reduced(mix_phase(n, f0), sn, cs);
    float2 y;
    y.x = cs * x.x - sn * x.y;
    y.y = cs * x.y + sn * x.x;
    return y;
}

}  // namespace msk144
