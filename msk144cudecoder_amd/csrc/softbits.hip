// softbits: per-candidate frame fold, carrier-phase estimate, matched filter, LLR scaling, sync check.
//
// Replaces softbits_kernel (softbits_kernel.cuh:9-249).  The reference launches one
// 160-thread block per candidate and every block re-mixes the whole 5184-sample window; here one
// workgroup serves all D*8 candidates of a (channel, frequency) pair, mixes the window ONCE into LDS
// (same float phase as the scan) and then each 64-lane wave demodulates candidates on its own, with
// no workgroup barrier after the mix.
//
// Reduction orders follow the reference where that is free:
//   * the 84-term phase sum uses the reference's 42 -> 32 -> 16..1 order (softbits_kernel.cuh:98-124);
//   * the two 144-term sums reproduce sum_reduction_two_cycles on five 32-lane warps
//     (sum_reduction.cuh:14-44): ((w0+w1)+(w2+w3))+w4.
// The phase rotation uses conj(s)/|s| instead of atan2f + sincosf (same unit vector to ~1 ulp).
#include "msk144_kernels.h"
#include "wave64.h"

namespace msk144
{

namespace
{

constexpr int kSbThreads = 256;
constexpr int kSbWaves = kSbThreads / 64;
constexpr int kFoldIters = (kFrameSamples + 63) / 64;  // 14 (13.5)
// rotated frame kept per wave as separate re/im planes, one pad word per 12 samples so that the
// matched-filter reads (lane stride 12 samples) hit distinct banks
constexpr int kPlane = kFrameSamples + kFrameSamples / 12;  // 936

struct SoftbitsArgs
{
    DeviceStore st;
    SyncTemplate tpl;
    int total_tiles;
    int tiles_per_xcd;
};

__device__ __forceinline__ int padded(int n)
{
    return n + n / 12;
}

__global__ __launch_bounds__(kSbThreads) void softbits_kernel(const SoftbitsArgs a)
{
    __shared__ float2 s_x[kWindowSamples];
    __shared__ float s_re[kSbWaves][kPlane];
    __shared__ float s_im[kSbWaves][kPlane];

    const int xcd = blockIdx.x & 7;
    const int tile = xcd * a.tiles_per_xcd + (blockIdx.x >> 3);
    if((blockIdx.x >> 3) >= a.tiles_per_xcd || tile >= a.total_tiles) return;
    const int ch = tile / a.st.F;
    const int b = tile - ch * a.st.F;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- mix (softbits_kernel.cuh:27-52) ----
    const float f0 = -1.0f * a.st.freq[b];
    const float twopi = 2.0f * 3.14159265358979323846f;
    const float2* __restrict__ cdat = a.st.analytic + static_cast<size_t>(ch) * kWindowSamples;
    for(int n = tid; n < kWindowSamples; n += kSbThreads)
    {
        const float phi = f32_div(f32_mul(f32_mul(static_cast<float>(n), twopi), f0), kSampleRate);
        float sn, cs;
        sincosf(phi, &sn, &cs);
        const float2 x = cdat[n];
        float2 y;
        y.x = cs * x.x - sn * x.y;
        y.y = cs * x.y + sn * x.x;
        s_x[n] = y;
    }
    __syncthreads();

    // sync template per lane: cb[lane] for lanes 0..41 (first sync word, samples 0..41) and
    // cb[lane-16] for lanes 16..57 (second sync word: sample 336+t sits in fold slot 5, lane 16+t)
    float cb1r = 0.0f, cb1i = 0.0f, cb2r = 0.0f, cb2i = 0.0f;
#pragma unroll
    for(int k = 0; k < kSyncTaps; k++)
    {
        if(lane == k)
        {
            cb1r = a.tpl.re[k];
            cb1i = a.tpl.im[k];
        }
        if(lane == k + 16)
        {
            cb2r = a.tpl.re[k];
            cb2i = a.tpl.im[k];
        }
    }

    float* re = s_re[wave];
    float* im = s_im[wave];
    const int D = a.st.D;
    const int ncand = D * kSlotsPerPattern;
    const size_t item0 = static_cast<size_t>(ch) * a.st.K + static_cast<size_t>(b) * ncand;

    for(int c = wave; c < ncand; c += kSbWaves)
    {
        const int p = c / kSlotsPerPattern;
        const size_t item = item0 + c;
        uint32_t pos = a.st.pos[item];
        if(pos >= static_cast<uint32_t>(kWindowSamples)) pos -= kWindowSamples;  // scanned positions reach 5375

        // ---- fold the averaged frames (softbits_kernel.cuh:59-82), 864 samples in 14 registers/lane ----
        float fr[kFoldIters], fi[kFoldIters];
#pragma unroll
        for(int i = 0; i < kFoldIters; i++)
        {
            const int n = lane + 64 * i;
            float sr = 0.0f, si = 0.0f;
            if(n < kFrameSamples)
            {
#pragma unroll
                for(int m = 0; m < kPatternBits; m++)
                {
                    if(kPatternMask[p][m])
                    {
                        int idx = static_cast<int>(pos) + n + kFrameSamples * m;  // < 2*5184
                        if(idx >= kWindowSamples) idx -= kWindowSamples;
                        const float2 v = s_x[idx];
                        sr += v.x;
                        si += v.y;
                    }
                }
            }
            fr[i] = sr;
            fi[i] = si;
        }

        // ---- carrier phase from the two sync words (softbits_kernel.cuh:88-128) ----
        // x_t = c3[t]*conj(cb[t]) on lane t (slot 0); x_{42+t} = c3[336+t]*conj(cb[t]) on lane 16+t (slot 5)
        float x1r = fr[0] * cb1r + fi[0] * cb1i;
        float x1i = fi[0] * cb1r - fr[0] * cb1i;
        float x2r = fr[5] * cb2r + fi[5] * cb2i;
        float x2i = fi[5] * cb2r - fr[5] * cb2i;
        x2r = __shfl(x2r, lane + 16);
        x2i = __shfl(x2i, lane + 16);
        float rr = (lane < kSyncTaps) ? x1r + x2r : 0.0f;
        float ri = (lane < kSyncTaps) ? x1i + x2i : 0.0f;
        {
            const float tr = __shfl_down(rr, 32);
            const float ti = __shfl_down(ri, 32);
            if(lane < 10)
            {
                rr += tr;
                ri += ti;
            }
        }
#pragma unroll
        for(int size = 16; size > 0; size >>= 1)
        {
            const float tr = __shfl_down(rr, size);
            const float ti = __shfl_down(ri, size);
            if(lane < size)
            {
                rr += tr;
                ri += ti;
            }
        }
        const float sre = readlane_f32(rr, 0);
        const float sim = readlane_f32(ri, 0);
        // cfac = conj(exp(i*atan2(im,re))) = (re, -im)/|s|
        float cr = 1.0f, ci = 0.0f;
        {
            const float mag = f32_sqrt(fmaf(sre, sre, sim * sim));
            if(mag > 0.0f)
            {
                const float inv = 1.0f / mag;
                cr = sre * inv;
                ci = -sim * inv;
            }
            else if(!(mag == 0.0f))
            {
                cr = mag;  // NaN propagates like the reference's atan2f/sincosf chain
                ci = mag;
            }
        }

        // ---- de-rotate and park the frame in this wave's LDS planes (softbits_kernel.cuh:146-153) ----
#pragma unroll
        for(int i = 0; i < kFoldIters; i++)
        {
            const int n = lane + 64 * i;
            if(n < kFrameSamples)
            {
                const int pn = padded(n);
                re[pn] = fr[i] * cr - fi[i] * ci;
                im[pn] = fr[i] * ci + fi[i] * cr;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- matched filter (softbits_kernel.cuh:157-180): softbit u = 2*piq + sel lives on lane u%64 ----
        float soft[3];
#pragma unroll
        for(int j = 0; j < 3; j++)
        {
            const int u = lane + 64 * j;
            float sb = 0.0f;
            if(u < kSoftBits)
            {
                const int sel = u & 1;
                const int piq = u >> 1;
                int start = 12 * piq + (sel ? 0 : kFrameSamples - 6);
                if(start >= kFrameSamples) start -= kFrameSamples;
                const float* plane = sel ? re : im;
#pragma unroll
                for(int i = 0; i < 12; i++)
                {
                    int k = start + i;
                    if(k >= kFrameSamples) k -= kFrameSamples;
                    sb = fmaf(plane[padded(k)], a.tpl.pp[i], sb);
                }
            }
            soft[j] = sb;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- normalisation (softbits_kernel.cuh:186-201) ----
        float w_s[3], w_q[3];
#pragma unroll
        for(int j = 0; j < 3; j++)
        {
            w_s[j] = half_tree_sum_f32(soft[j]);
            w_q[j] = half_tree_sum_f32(soft[j] * soft[j]);
        }
        const float sum_sav = f32_add(f32_add(f32_add(readlane_f32(w_s[0], 0), readlane_f32(w_s[0], 32)),
                                                  f32_add(readlane_f32(w_s[1], 0), readlane_f32(w_s[1], 32))),
                                        readlane_f32(w_s[2], 0));
        const float sum_s2av = f32_add(f32_add(f32_add(readlane_f32(w_q[0], 0), readlane_f32(w_q[0], 32)),
                                                   f32_add(readlane_f32(w_q[1], 0), readlane_f32(w_q[1], 32))),
                                         readlane_f32(w_q[2], 0));
        const float sav = f32_div(sum_sav, 144.0f);
        const float s2av = f32_div(sum_s2av, 144.0f);
        const float ssig = f32_sqrt(f32_sub(s2av, f32_mul(sav, sav)));
        const float sigma = 0.60f;
        const float scale = f32_div(2.0f, f32_mul(f32_mul(ssig, sigma), sigma));

        // ---- sync-word disagreements (softbits_kernel.cuh:214-241): bits 0..7 and 56..63 ----
        int sync_bit = -1;
        if(lane < 8) sync_bit = lane;
        else if(lane >= kSecondSyncBit) sync_bit = lane - kSecondSyncBit;
        bool disagree = false;
        if(sync_bit >= 0)
        {
            const int hard = (soft[0] < 0.0f) ? -1 : 1;
            disagree = hard != kSync8Pm[sync_bit];
        }
        const int nbad = __popcll(__ballot(disagree));

        // ---- store (softbits_kernel.cuh:204-211,244-247) ----
        float* __restrict__ llr = a.st.llr + item * kCodeBits;
        if(lane >= 8 && lane < 56) llr[lane - 8] = f32_mul(scale, soft[0]);     // u = 8..55   -> 0..47
        llr[48 + lane] = f32_mul(scale, soft[1]);                               // u = 64..127 -> 48..111
        if(lane < 16) llr[112 + lane] = f32_mul(scale, soft[2]);                // u = 128..143 -> 112..127
        if(lane == 0) a.st.nbadsync[item] = nbad;
    }
}

}  // namespace

void launch_softbits(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream)
{
    SoftbitsArgs a;
    a.st = st;
    a.tpl = tpl;
    a.total_tiles = st.channels * st.F;
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
    const int grid = a.tiles_per_xcd * 8;
    hipLaunchKernelGGL(softbits_kernel, dim3(grid), dim3(kSbThreads), 0, stream, a);
}

}  // namespace msk144
