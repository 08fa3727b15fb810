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
// The quadrant fix-up is pure bit arithmetic (no v_cmp / v_cndmask / v_rndne / v_cvt, all half rate,
// profiles/r02_valu_issue_microbench.txt): k = rint(phi*2/pi) comes from the 1.5*2^23 add trick, whose float bits also hold
// k mod 4, and the swap / sign flips are bit operations.
__device__ __forceinline__ void sincos_reduced(float phi, float& sn, float& cs)
{
    constexpr float kRound = 12582912.0f;                             // 1.5 * 2^23: ulp = 1 in [2^23, 2^24)
    const float kb = fmaf(phi, 0.636619772367581343f, kRound);        // 2/pi; low mantissa bits = k (two's complement)
    const float k = kb - kRound;                                      // exact
    float r = fmaf(-k, 1.57079637050628662109375f, phi);             // fl(pi/2)
    r = fmaf(-k, -4.37113900018624283e-8f, r);                        // pi/2 - fl(pi/2)
    const float z = r * r;
    float sp = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = fmaf(sp, z, -1.6666654611e-1f);
    const float s_r = fmaf(r * z, sp, r);
    float cp = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = fmaf(cp, z, 4.166664568298827e-2f);
    const float c_r = fmaf(z * z, cp, fmaf(-0.5f, z, 1.0f));
    const uint32_t q = __builtin_bit_cast(uint32_t, kb);
    const uint32_t su = __builtin_bit_cast(uint32_t, s_r), cu = __builtin_bit_cast(uint32_t, c_r);
    const uint32_t swap = 0u - (q & 1u);                              // odd quadrant: sin <-> cos
    uint32_t s = (cu & swap) | (su & ~swap);                          // v_bfi_b32
    uint32_t c = (su & swap) | (cu & ~swap);
    // quadrant signs: sin negative for q = 2,3 (bit 1); cos negative for q = 1,2 (bit 1 xor bit 0)
    const uint32_t t = q << 30;
    s ^= t & 0x80000000u;
    c ^= (t ^ (t << 1)) & 0x80000000u;
    sn = __builtin_bit_cast(float, s);
    cs = __builtin_bit_cast(float, c);
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
