// sync_scan: sync-word correlation over frequency x time offset x averaging pattern.
//
// Replaces scan_kernel (scan_kernel.cuh:27-393).  One workgroup (8 waves) per
// (channel, frequency hypothesis).  The reference evaluates, for each of 5376 positions and each
// pattern, 42 taps x 2*num_avg folded samples; here the linearity of the correlation is used:
//
//   C[n]      = sum_k conj(cdat2[(n+k) mod N]) * cb42[k]          one pass over the window
//   S(pos,p)  = sum_{m in mask_p} C[(pos+864m) mod N] + C[(pos+864m+336) mod N]
//   xb        = |S|
//
// cdat2 is the window mixed down with exactly the reference's float phase (scan_kernel.cuh:54), so
// the only deviations from the reference are float re-association of a linear sum and sincos/sqrt ulps
// (~1e-6 relative on xb).  Patterns 0..5 are nested prefixes (msk_context.cuh:231-236): S is
// accumulated across them.
//
// Phases (LDS: one 5184+41 complex buffer, 44 KB per workgroup -> 3 workgroups per CU):
//  1. mix the window into LDS (custom ~25-instruction sincos, mix.h);
//  2. C[n] by pulse decomposition (correlate_pulses below): a thread owns eleven outputs spaced by six, which share
//     seventeen half-pulse sums built from the 102 samples it streams - 187 multiply-adds + 93 adds of complex values per
//     thread instead of one 42-tap sum per output (847).  After a barrier C overwrites the window in place;
//  3. fold + |S|^2 per pattern; each wave takes 128-position half-slices, pre-reduces the lane's two
//     positions, then one DPP max + ballot per (half-slice, pattern).  No barrier in this phase (the
//     reference has 4 per slice).  Lowest position wins exact ties, as the reference's strict-> trees;
//  4. xb = sqrt (correctly rounded) of the half-slice maxima in parallel; then one wave, lane = 8*pattern +
//     slot, runs the reference's 8-slot replacement rule in slice order (scan_kernel.cuh:276-353): slice
//     maximum = better of two halves, arg-min over a pattern's 8 slots by three DPP min steps + ballot
//     (lowest slot wins ties), conditional replace.  21 short steps instead of a 1300-instruction serial tail.
#include "msk144_kernels.h"
#include "mix.h"
#include "wave64.h"

#include <utility>

namespace msk144
{

namespace
{

// Workgroup shape: 8 waves.  A CU holds three 8-wave workgroups with this much LDS but only ~1.5 nine-wave ones
// (tools/ubench/occupancy_probe.hip: 576 threads + 44.5 KB -> 1.47 resident, 512 threads -> 3), so the natural
// 576 x 9 outputs split of the 5184-sample window is the wrong one: 512 threads, eleven outputs each in the correlation
// (79 groups of six threads cover the 864 output columns, the last 38 threads idle through that phase).
constexpr int kOutPerThread = 11;
constexpr int kScanThreads = 512;
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kMixPerThread = (kWindowSamples + kScanThreads - 1) / kScanThreads;  // 11, the last one on 64 threads only
constexpr int kChunk = 128;                                   // positions per wave work unit
constexpr int kChunks = kScanPositions / kChunk;              // 42
constexpr int kWrapPad = kSyncTaps - 1;
static_assert(kScanThreads % 64 == 0 && kScanThreads * kOutPerThread >= kWindowSamples, "one in-place pass");

struct ScanArgs
{
    DeviceStore st;
    float pp[12];
    int total_tiles;
    int tiles_per_xcd;
};

// LDS pointer kept volatile so that the 50 sample loads stay ds_read_b64 (2 LDS cycles each); merged into
// ds_read2_b64 the same bytes take twice as long (MI355X_MICROARCH.md, LDS table).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v2f* lds_f2_ptr;

__device__ __forceinline__ float2 as_float2(v2f v)
{
    return make_float2(v.x, v.y);
}

// ---- correlation by pulse decomposition ----
// The 42-tap template is seven half-sine pulses of 12 samples, alternating between the I and the Q rail and offset by
// half a pulse (msk_context.cuh:188-196): with the half-pulse sums
//     A[m] = sum_{d<6} x[m+d] * pp[d]        B[m] = sum_{d<6} x[m+d] * pp[6+d]        P[m] = A[m] + B[m+6]  (a full pulse)
//     sum_k x[n+k] * cbi[k] = s1 P[n] + s3 P[n+12] + s5 P[n+24] + s7 A[n+36]    =: R
//     sum_k x[n+k] * cbq[k] = s0 B[n] + s2 P[n+6]  + s4 P[n+18] + s6 P[n+30]    =: Q
//     C[n] = sum_k conj(x[n+k]) (cbi[k] + i cbq[k]) = (R.x + Q.y, Q.x - R.y)
// everything C[n] needs lives at offsets n + 6j.  A thread therefore owns R = 11 outputs SPACED BY SIX, n = n0 + 6r: they
// share the R + 6 half-pulse pairs at n0 + 6i, each built from its own six samples, so the 6 (R + 6) samples a thread
// streams are used exactly once per rail: 11 (R + 6) multiply-adds and ~8.5 R adds of complex values per thread instead
// of 77 R + 8 R with one 42-tap sum per output.  Same linear form, different association: ~1e-6 relative on xb, as before.
// Thread t = 6q + c owns n0 = 66 q + c.
constexpr int kPulseHalf = 6;
constexpr int kHalfPulses = kOutPerThread + 6;  // 17 half-pulse positions feed 11 outputs
constexpr int kStream = kHalfPulses * kPulseHalf;  // 102 samples
constexpr int kOutSpan = kOutPerThread * kPulseHalf;  // 66 consecutive outputs per group of 6 threads
constexpr int kOutColumns = kWindowSamples / kPulseHalf;  // 864 columns n = 6k + c
constexpr int kOutGroups = (kOutColumns + kOutPerThread - 1) / kOutPerThread;  // 79 groups of six threads
static_assert(kWindowSamples % kPulseHalf == 0 && kOutGroups * kPulseHalf <= kScanThreads, "thread -> (block of 66, residue) map");
// the last group's stream runs past the wrap pad: those samples only feed outputs beyond the window, which are dropped
constexpr int kStreamPad = kOutSpan * (kOutGroups - 1) + kPulseHalf - 1 + kStream - (kWindowSamples + kWrapPad);
static_assert(kStreamPad > 0 && kStreamPad < 64, "LDS buffer carries kStreamPad readable (unused) samples behind the wrap pad");

template<int kSign>
__device__ __forceinline__ void acc_signed(float2& acc, const float2 v)
{
    if constexpr(kSign > 0)
    {
        acc.x += v.x;
        acc.y += v.y;
    }
    else
    {
        acc.x -= v.x;
        acc.y -= v.y;
    }
}

__device__ __forceinline__ void correlate_pulses(lds_f2_ptr xs, const float (&pp)[12], float2 (&c)[kOutPerThread])
{
    float2 racc[kOutPerThread], qacc[kOutPerThread];
#pragma unroll
    for(int r = 0; r < kOutPerThread; r++)
    {
        racc[r] = make_float2(0.0f, 0.0f);
        qacc[r] = make_float2(0.0f, 0.0f);
    }
    float2 a_prev = make_float2(0.0f, 0.0f);
    // Left alone, the scheduler clusters the 90 loads and sinks the arithmetic behind them (114 VGPRs; capping the registers
    // spills, sched_barrier makes it worse).  The empty asm at the end of every half-pulse takes A_i and B_i as read-write
    // operands together with `off`, an opaque zero in the next loads' address: the half-pulse's arithmetic must be complete -
    // its six samples dead - before the loads of the next one can issue.
    int off = 0;
#pragma unroll
    for(int i = 0; i < kHalfPulses; i++)
    {
        float2 x[kPulseHalf];
#pragma unroll
        for(int d = 0; d < kPulseHalf; d++) x[d] = as_float2(xs[off + i * kPulseHalf + d]);
        // A_i (pp[0] = sin 0 = 0: five terms) and B_i (pp[6] = 1: the first term is the sample)
        float2 av = make_float2(x[1].x * pp[1], x[1].y * pp[1]);
        float2 bv = x[0];  // pp[6] = sin pi/2 = 1 exactly (checked at create)
#pragma unroll
        for(int d = 2; d < kPulseHalf; d++)
        {
            av.x = fmaf(x[d].x, pp[d], av.x);
            av.y = fmaf(x[d].y, pp[d], av.y);
        }
#pragma unroll
        for(int d = 1; d < kPulseHalf; d++)
        {
            bv.x = fmaf(x[d].x, pp[kPulseHalf + d], bv.x);
            bv.y = fmaf(x[d].y, pp[kPulseHalf + d], bv.y);
        }
        if(i < kOutPerThread) acc_signed<kSync8Pm[0]>(qacc[i], bv);                                   // Q_i  += s0 B_i
        if(i >= 1)
        {
            const int j = i - 1;                                                                       // P_j = A_j + B_(j+1)
            const float2 pv = make_float2(a_prev.x + bv.x, a_prev.y + bv.y);
            if(j < kOutPerThread) acc_signed<kSync8Pm[1]>(racc[j], pv);                                // R_j     += s1 P_j
            if(j >= 2 && j - 2 < kOutPerThread) acc_signed<kSync8Pm[3]>(racc[j - 2], pv);              // R_(j-2) += s3 P_j
            if(j >= 4 && j - 4 < kOutPerThread) acc_signed<kSync8Pm[5]>(racc[j - 4], pv);              // R_(j-4) += s5 P_j
            if(j >= 1 && j - 1 < kOutPerThread) acc_signed<kSync8Pm[2]>(qacc[j - 1], pv);              // Q_(j-1) += s2 P_j
            if(j >= 3 && j - 3 < kOutPerThread) acc_signed<kSync8Pm[4]>(qacc[j - 3], pv);              // Q_(j-3) += s4 P_j
            if(j >= 5 && j - 5 < kOutPerThread) acc_signed<kSync8Pm[6]>(qacc[j - 5], pv);              // Q_(j-5) += s6 P_j
        }
        if(i >= 6) acc_signed<kSync8Pm[7]>(racc[i - 6], av);                                           // R_(i-6) += s7 A_i
        a_prev = av;
        asm volatile("" : "+v"(off), "+v"(av.x), "+v"(av.y), "+v"(bv.x), "+v"(bv.y));
    }
#pragma unroll
    for(int r = 0; r < kOutPerThread; r++) c[r] = make_float2(racc[r].x + qacc[r].y, qacc[r].x - racc[r].y);
}

template<int kD>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(const ScanArgs a)
{
    __shared__ float2 s_buf[kWindowSamples + kWrapPad + kStreamPad];  // mixed window, later C[n] in place
    __shared__ float s_wv[kScanDepthMax][kChunks];           // per (pattern, half-slice) max |S|^2
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
    const float2* __restrict__ cdat = a.st.analytic + static_cast<size_t>(ch) * kWindowSamples;
    float2 xin[kMixPerThread];  // all loads in flight before any arithmetic: one L2 latency per tile, not three
#pragma unroll
    for(int i = 0; i < kMixPerThread; i++)
        if((i + 1) * kScanThreads <= kWindowSamples || tid + i * kScanThreads < kWindowSamples) xin[i] = cdat[tid + i * kScanThreads];
    const float tid_f = static_cast<float>(tid);
#pragma unroll
    for(int i = 0; i < kMixPerThread; i++)
    {
        const int n = tid + i * kScanThreads;
        if((i + 1) * kScanThreads <= kWindowSamples || n < kWindowSamples)
        {
            const float2 y = mix_sample(xin[i], tid_f + static_cast<float>(i * kScanThreads), f0);
            s_buf[n] = y;
            if(n < kWrapPad) s_buf[kWindowSamples + n] = y;
        }
    }
    __syncthreads();

    // ---- 2. C[n0 + 6r], r = 0..10, by pulse decomposition (correlate_pulses); C then overwrites the window in place ----
    {
        const int q = tid / kPulseHalf;
        const int n0 = kOutSpan * q + (tid - q * kPulseHalf);
        float2 c[kOutPerThread];
        if(q < kOutGroups) correlate_pulses((lds_f2_ptr)(s_buf + n0), a.pp, c);
        __syncthreads();  // every thread has read its samples: C may overwrite the window
        if(q < kOutGroups)
        {
#pragma unroll
            for(int r = 0; r < kOutPerThread; r++)
                if(n0 + kPulseHalf * r < kWindowSamples) s_buf[n0 + kPulseHalf * r] = c[r];
        }
    }
    __syncthreads();

    // ---- 3. fold per pattern, |S|^2, arg-max per 128-position half-slice ----
    // kD is a template parameter: the pattern loop is straight-line code and all 4*frames LDS reads of a
    // chunk are in flight together.  Patterns 7 and 8 (100100, 100110) reuse the frames already loaded.
    constexpr int D = kD;
    constexpr int kFrames = kD < kPatternBits ? kD : kPatternBits;
    constexpr uint32_t kN8 = kWindowSamples * 8u;  // byte size of the ring
    const char* __restrict__ cbytes = reinterpret_cast<const char*>(s_buf);
    for(int chunk = wave; chunk < kChunks; chunk += kScanWaves)
    {
        float2 ca[2][kFrames], cb[2][kFrames];
        // Ring addresses of the 4*frames samples.  The wrap decision (pos + offset >= 5184) is the same for all 128 positions
        // of a chunk unless one of the 13 wrap points falls inside it (12 of the 42 chunks): the common case is one scalar
        // offset per frame/partner and a single v_add per load; the per-lane add/sub/min chain (72 VALU instructions per
        // chunk) is kept for the chunks that need it.
        const int chunk_u = __builtin_amdgcn_readfirstlane(chunk);
        const uint32_t p0 = static_cast<uint32_t>(chunk_u) * kChunk;  // first position of the chunk, 0..5248
        uint32_t base_a[kFrames], base_b[kFrames];
        bool uniform = p0 + (kChunk - 1) < static_cast<uint32_t>(kWindowSamples) || p0 >= static_cast<uint32_t>(kWindowSamples);
        const uint32_t q0 = p0 >= static_cast<uint32_t>(kWindowSamples) ? p0 - kWindowSamples : p0;
#pragma unroll
        for(int m = 0; m < kFrames; m++)
        {
            uint32_t ta = q0 + static_cast<uint32_t>(kFrameSamples * m);
            if(ta >= static_cast<uint32_t>(kWindowSamples)) ta -= kWindowSamples;
            uint32_t tb = ta + kSecondSyncSample;
            if(tb >= static_cast<uint32_t>(kWindowSamples)) tb -= kWindowSamples;
            uniform = uniform && ta + (kChunk - 1) < static_cast<uint32_t>(kWindowSamples) && tb + (kChunk - 1) < static_cast<uint32_t>(kWindowSamples);
            base_a[m] = ta * 8u;
            base_b[m] = tb * 8u;
        }
        if(uniform)
        {
#pragma unroll
            for(int j = 0; j < 2; j++)
            {
                const uint32_t l8 = static_cast<uint32_t>(j * 64 + lane) * 8u;
#pragma unroll
                for(int m = 0; m < kFrames; m++)
                {
                    ca[j][m] = *reinterpret_cast<const float2*>(cbytes + (l8 + base_a[m]));
                    cb[j][m] = *reinterpret_cast<const float2*>(cbytes + (l8 + base_b[m]));
                }
            }
        }
        else
        {
#pragma unroll
            for(int j = 0; j < 2; j++)
            {
                const uint32_t pos = chunk * kChunk + j * 64 + lane;  // 0..5375
                const uint32_t q8 = (pos >= static_cast<uint32_t>(kWindowSamples) ? pos - kWindowSamples : pos) * 8u;
#pragma unroll
                for(int m = 0; m < kFrames; m++)
                {
                    const uint32_t a8 = q8 + static_cast<uint32_t>(kFrameSamples * 8 * m);
                    const uint32_t ia = min(a8, a8 - kN8);  // a8 mod ring (unsigned wrap trick)
                    const uint32_t b8 = ia + kSecondSyncSample * 8u;
                    const uint32_t ib = min(b8, b8 - kN8);
                    ca[j][m] = *reinterpret_cast<const float2*>(cbytes + ia);
                    cb[j][m] = *reinterpret_cast<const float2*>(cbytes + ib);
                }
            }
        }
        float sr[2] = {0.0f, 0.0f}, si[2] = {0.0f, 0.0f};
#pragma unroll
        for(int p = 0; p < D; p++)
        {
            float v[2];
#pragma unroll
            for(int j = 0; j < 2; j++)
            {
                if(p < kPatternBits)
                {
                    sr[j] = (sr[j] + ca[j][p].x) + cb[j][p].x;  // nested prefix masks: add frame p
                    si[j] = (si[j] + ca[j][p].y) + cb[j][p].y;
                }
                else
                {
                    sr[j] = 0.0f;
                    si[j] = 0.0f;
#pragma unroll
                    for(int m = 0; m < kPatternBits; m++)
                    {
                        if(kPatternMask[p][m])
                        {
                            sr[j] = (sr[j] + ca[j][m].x) + cb[j][m].x;
                            si[j] = (si[j] + ca[j][m].y) + cb[j][m].y;
                        }
                    }
                }
                v[j] = fmaf(sr[j], sr[j], si[j] * si[j]);
            }
            // lane-local best of its two positions (lower position wins ties), then across the wave
            const bool second = v[1] > v[0];
            const float best = second ? v[1] : v[0];
            const float mx = wave_max_f32(best);
            // among lanes holding the maximum, the lowest position: first halves (j=0) come before second halves
            const unsigned long long eq = __ballot(best == mx);
            const unsigned long long sec = __ballot(second);
            const unsigned long long eq0 = eq & ~sec;
            const unsigned long long eq1 = eq & sec;
            if(lane == 0)
            {
                uint32_t off = 0;
                if(eq0) off = __builtin_ctzll(eq0);
                else if(eq1) off = 64 + __builtin_ctzll(eq1);
                s_wv[p][chunk] = mx;
                s_wpos[p][chunk] = static_cast<uint32_t>(chunk * kChunk) + off;
            }
        }
    }
    __syncthreads();

    // ---- 4a. xb = |S| for the 42*D half-slice maxima, in parallel (correctly rounded sqrt) ----
    for(int e = tid; e < D * kChunks; e += kScanThreads)
    {
        const int p = e / kChunks;
        const int c = e - p * kChunks;
        s_wv[p][c] = f32_sqrt(s_wv[p][c]);
    }
    __syncthreads();

    // ---- 4b. 8-slot replacement rule in slice order (scan_kernel.cuh:276-353), one wave: lane = 8*pattern + slot ----
    if(wave == 0)
    {
        const int p = lane >> 3;
        const int slot = lane & 7;
        const int pc = p < D ? p : 0;  // lanes of unused patterns shadow pattern 0 and store nothing
        float my_xb = 0.0f;            // reset(): pos 0, xb 0 (scan_kernel.cuh:78-82)
        uint32_t my_pos = 0u;
        for(int s = 0; s < kScanSlices; s++)
        {
            // slice maximum = better of its two halves, lower position on ties
            const float v0 = s_wv[pc][2 * s], v1 = s_wv[pc][2 * s + 1];
            const bool second = v1 > v0;
            const float best = second ? v1 : v0;
            const uint32_t best_pos = second ? s_wpos[pc][2 * s + 1] : s_wpos[pc][2 * s];
            // arg-min over the 8 stored slots of this pattern, lowest slot index wins ties
            const float mn = oct_min_f32(my_xb);
            const unsigned long long eq = __ballot(my_xb == mn);
            const uint32_t mine = static_cast<uint32_t>(eq >> (8 * p)) & 0xFFu;
            const int worst = __builtin_ctz(mine | 0x100u);
            if(slot == worst && best > my_xb)
            {
                my_xb = best;
                my_pos = best_pos;
            }
        }
        if(p < D)
        {
            const size_t base = static_cast<size_t>(ch) * a.st.K + (static_cast<size_t>(b) * D + p) * kSlotsPerPattern;
            a.st.pos[base + slot] = my_pos;
            a.st.xb[base + slot] = my_xb;
        }
    }
}

}  // namespace

void launch_scan(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream)
{
    ScanArgs a;
    a.st = st;
    for(int i = 0; i < 12; i++) a.pp[i] = tpl.pp[i];
    a.total_tiles = st.channels * st.F;
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
    const dim3 grid(a.tiles_per_xcd * 8), block(kScanThreads);
    switch(st.D)
    {
    case 1: hipLaunchKernelGGL(scan_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(scan_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(scan_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(scan_kernel<4>, grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL(scan_kernel<5>, grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL(scan_kernel<6>, grid, block, 0, stream, a); break;
    case 7: hipLaunchKernelGGL(scan_kernel<7>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(scan_kernel<8>, grid, block, 0, stream, a); break;
    }
}

}  // namespace msk144
