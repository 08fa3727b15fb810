// sync_scan: sync-word correlation over frequency x time offset x averaging pattern.
//
// Replaces scan_kernel (scan_kernel.cuh:27-393).  One workgroup per
// (channel, frequency hypothesis).  The reference evaluates, for each of 5376 positions and each
// pattern, 42 taps x 2*num_avg folded samples; here the linearity of the correlation is used:
//
//   C[n]      = sum_k conj(cdat2[(n+k) mod N]) * cb42[k]          one pass, 5184 x 42 complex MACs
//   S(pos,p)  = sum_{m in mask_p} C[(pos+864m) mod N] + C[(pos+864m+336) mod N]
//   xb        = |S|
//
// cdat2 is the window mixed down with exactly the reference's float phase
// phi = ((float(n)*2pi)*f0)/12000 (scan_kernel.cuh:54), so the only deviations from the reference
// are float re-association of a linear sum and sincos/sqrt ulps (~1e-6 relative on xb).
// Patterns 0..5 are nested prefixes (msk_context.cuh:231-236): S is accumulated across them.
//
// Top-8 rule: per 256-position slice the arg-max (lowest position wins ties), then the 8-slot
// replacement rule in slice order, exactly as scan_kernel.cuh:140-353; the 64-lane arg-max is a DPP
// max + ballot instead of 32-lane shuffle trees.
#include "msk144_kernels.h"
#include "wave64.h"

namespace msk144
{

namespace
{

constexpr int kScanThreads = 512;
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kChunksPerSlice = kSlicePositions / 64;         // 4 wave-chunks per slice
constexpr int kChunks = kScanSlices * kChunksPerSlice;        // 84
constexpr int kWrapPad = kSyncTaps - 1;                       // cdat2 is extended by 41 wrapped samples

struct ScanArgs
{
    DeviceStore st;
    SyncTemplate tpl;
    int total_tiles;
    int tiles_per_xcd;
};

__device__ __forceinline__ int wrap_window(int i)
{
    return i >= kWindowSamples ? i - kWindowSamples : i;
}

__global__ __launch_bounds__(kScanThreads) void scan_kernel(const ScanArgs a)
{
    __shared__ float2 s_x[kWindowSamples + kWrapPad + 7];  // mixed window (+ wrap)
    __shared__ float2 s_c[kWindowSamples];                 // single-frame correlation C[n]
    __shared__ float s_wxb[kScanDepthMax][kChunks];        // per (pattern, wave-chunk) maximum
    __shared__ uint32_t s_wpos[kScanDepthMax][kChunks];

    // XCD-aware tile map: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one
    // contiguous range of tiles - the F tiles of a channel then share one L2 copy of its window.
    const int xcd = blockIdx.x & 7;
    const int tile = xcd * a.tiles_per_xcd + (blockIdx.x >> 3);
    if((blockIdx.x >> 3) >= a.tiles_per_xcd || tile >= a.total_tiles) return;
    const int ch = tile / a.st.F;
    const int b = tile - ch * a.st.F;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- 1. mix down by the hypothesis frequency (scan_kernel.cuh:45-69) ----
    const float f0 = -1.0f * a.st.freq[b];
    const float twopi = 2.0f * 3.14159265358979323846f;
    const float2* __restrict__ cdat = a.st.analytic + static_cast<size_t>(ch) * kWindowSamples;
    for(int n = tid; n < kWindowSamples; n += kScanThreads)
    {
        const float phi = f32_div(f32_mul(f32_mul(static_cast<float>(n), twopi), f0), kSampleRate);
        float sn, cs;
        sincosf(phi, &sn, &cs);
        const float2 x = cdat[n];
        float2 y;
        y.x = cs * x.x - sn * x.y;
        y.y = cs * x.y + sn * x.x;
        s_x[n] = y;
        if(n < kWrapPad) s_x[kWindowSamples + n] = y;
    }
    __syncthreads();

    // ---- 2. C[n] = sum_k conj(x[n+k]) * cb42[k] ----
    for(int n = tid; n < kWindowSamples; n += kScanThreads)
    {
        float cr = 0.0f, ci = 0.0f;
#pragma unroll
        for(int k = 0; k < kSyncTaps; k++)
        {
            const float2 y = s_x[n + k];
            // conj(y)*cb = (y.x*re + y.y*im) + i (y.x*im - y.y*re)
            cr = fmaf(y.x, a.tpl.re[k], cr);
            cr = fmaf(y.y, a.tpl.im[k], cr);
            ci = fmaf(y.x, a.tpl.im[k], ci);
            ci = fmaf(-y.y, a.tpl.re[k], ci);
        }
        s_c[n] = make_float2(cr, ci);
    }
    __syncthreads();

    // ---- 3. fold per pattern, |S|, 64-lane arg-max per wave-chunk ----
    const int D = a.st.D;
    for(int chunk = wave; chunk < kChunks; chunk += kScanWaves)
    {
        const int pos = chunk * 64 + lane;  // 0..5375
        const int q = wrap_window(pos);
        float sr = 0.0f, si = 0.0f;
        for(int p = 0; p < D; p++)
        {
            if(p < kPatternBits)
            {
                // nested prefix masks: add frame p
                const int ia = wrap_window(q + kFrameSamples * p);
                const int ib = wrap_window(ia + kSecondSyncSample);
                const float2 ca = s_c[ia];
                const float2 cb = s_c[ib];
                sr = (sr + ca.x) + cb.x;
                si = (si + ca.y) + cb.y;
            }
            else
            {
                // patterns 7 and 8 (100100, 100110): rebuild from their own masks
                sr = 0.0f;
                si = 0.0f;
                for(int m = 0; m < kPatternBits; m++)
                {
                    if(kPatternMask[p][m])
                    {
                        const int ia = wrap_window(q + kFrameSamples * m);
                        const int ib = wrap_window(ia + kSecondSyncSample);
                        const float2 ca = s_c[ia];
                        const float2 cb = s_c[ib];
                        sr = (sr + ca.x) + cb.x;
                        si = (si + ca.y) + cb.y;
                    }
                }
            }
            const float xb = f32_sqrt(fmaf(sr, sr, si * si));
            const float mx = wave_max_f32(xb);
            const unsigned long long eq = __ballot(xb == mx);
            if(lane == 0)
            {
                const int first = eq ? __builtin_ctzll(eq) : 0;  // lowest position wins ties
                s_wxb[p][chunk] = mx;
                s_wpos[p][chunk] = static_cast<uint32_t>(chunk * 64 + first);
            }
        }
    }
    __syncthreads();

    // ---- 4. per pattern: slice maxima in order, 8-slot replacement rule (scan_kernel.cuh:276-353) ----
    if(tid < D)
    {
        const int p = tid;
        float slot_xb[kSlotsPerPattern];
        uint32_t slot_pos[kSlotsPerPattern];
#pragma unroll
        for(int i = 0; i < kSlotsPerPattern; i++)
        {
            slot_xb[i] = 0.0f;
            slot_pos[i] = 0u;
        }
        for(int s = 0; s < kScanSlices; s++)
        {
            float best = s_wxb[p][s * kChunksPerSlice];
            uint32_t best_pos = s_wpos[p][s * kChunksPerSlice];
#pragma unroll
            for(int w = 1; w < kChunksPerSlice; w++)
            {
                const float o = s_wxb[p][s * kChunksPerSlice + w];
                if(o > best)
                {
                    best = o;
                    best_pos = s_wpos[p][s * kChunksPerSlice + w];
                }
            }
            // arg-min over the stored slots, lowest slot index wins ties
            int worst = 0;
            float worst_xb = slot_xb[0];
#pragma unroll
            for(int i = 1; i < kSlotsPerPattern; i++)
            {
                if(slot_xb[i] < worst_xb)
                {
                    worst_xb = slot_xb[i];
                    worst = i;
                }
            }
            if(best > worst_xb)
            {
#pragma unroll
                for(int i = 0; i < kSlotsPerPattern; i++)
                {
                    if(i == worst)
                    {
                        slot_xb[i] = best;
                        slot_pos[i] = best_pos;
                    }
                }
            }
        }
        const size_t base = static_cast<size_t>(ch) * a.st.K + (static_cast<size_t>(b) * D + p) * kSlotsPerPattern;
#pragma unroll
        for(int i = 0; i < kSlotsPerPattern; i++)
        {
            a.st.pos[base + i] = slot_pos[i];
            a.st.xb[base + i] = slot_xb[i];
        }
    }
}

}  // namespace

void launch_scan(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream)
{
    ScanArgs a;
    a.st = st;
    a.tpl = tpl;
    a.total_tiles = st.channels * st.F;
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
    const int grid = a.tiles_per_xcd * 8;
    hipLaunchKernelGGL(scan_kernel, dim3(grid), dim3(kScanThreads), 0, stream, a);
}

}  // namespace msk144
