// sync_scan: sync-word correlation over frequency x time offset x averaging pattern.
//
// Replaces scan_kernel (scan_kernel.cuh:27-393).  One workgroup (8 waves) per
// (channel, frequency hypothesis).  The reference evaluates, for each of 5376 positions and each
// pattern, 42 taps x 2*num_avg folded samples; here the linearity of the correlation is used:
//
//   C[n]      = sum_k conj(cdat2[(n+k) mod N]) * cb42[k]          one pass over the window
//   S(pos,p)  = sum_{m in mask_p} C[(pos+864m) mod N] + C[(pos+864m+336) mod N]   = sum_{m in mask_p} E[(pos+864m) mod N]
//   xb        = |S|
//
// cdat2 is the window mixed down with exactly the reference's float phase (scan_kernel.cuh:54), so
// the only deviations from the reference are float re-association of a linear sum and sincos/sqrt ulps
// (~1e-6 relative on xb).  Patterns 0..5 are nested prefixes (msk_context.cuh:231-236): S is
// accumulated across them.
//
// Phases (LDS: one 5184+41 complex buffer, 44 KB per workgroup -> 3 workgroups of 8 waves per CU):
//  1. mix the window into LDS (custom ~25-instruction sincos, mix.h);
//  2. C[n] by pulse decomposition (correlate_pulses below): a thread owns eleven outputs spaced by six, which share
//     seventeen half-pulse sums built from the 102 samples it streams - 187 multiply-adds + 93 adds of complex values per
//     thread instead of one 42-tap sum per output (847).  After a barrier C overwrites the window in place;
//  2b. E[n] = C[n] + C[n + 336]: the two sync words of a frame, once per ring position (in place, one more barrier), so that
//     S(pos, p) = sum_m E[(pos + 864 m) mod N] costs the fold one load and one add per frame;
//  3. fold + |S|^2 per pattern along RUNS: a lane walks eleven consecutive positions of one 256-position slice and keeps a
//     running maximum and its position per pattern (strict >: the lowest position keeps exact ties, as the reference's
//     strict-> trees) - one compare, one max and one select per (position, pattern) and no cross-lane reduction, where
//     the earlier form (positions across lanes, a 6-step DPP max + two ballots per 128 positions and pattern) spent half
//     of the phase on reductions.  24 runs cover a slice = three aligned octets of lanes: each octet reduces in registers
//     (3 DPP steps, first lane at the maximum wins) and one thread per (pattern, slice) merges three octets;
//  4. xb = sqrt (correctly rounded) of the 21 x D slice maxima in parallel; then one wave, lane = 8*pattern + slot, runs the
//     reference's 8-slot replacement rule in slice order (scan_kernel.cuh:276-353): arg-min over a pattern's 8 slots by
//     three DPP min steps + ballot (lowest slot wins ties), conditional replace.
#include "msk144_kernels.h"
#include "mix.h"
#include "phase_stamps.h"
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
constexpr int kMixPerThread = (kWindowSamples + kScanThreads - 1) / kScanThreads;  // 11, the last one on 64 threads only
constexpr int kWrapPad = kSyncTaps - 1;
// Fold / arg-max phase: a lane walks a RUN of eleven consecutive positions of one 256-position slice with running maxima
// (no cross-lane reduction at all); 24 lanes cover a slice (23 x 11 = 253, the 24th run restarts at 245 and overlaps - duplicates
// cannot change a maximum), 21 x 24 = 504 of the 512 lanes are busy.
constexpr int kRun = 11;
constexpr int kRunsPerSlice = (kSlicePositions + kRun - 1) / kRun;  // 24
static_assert(kScanSlices * kRunsPerSlice <= kScanThreads && kRun <= kWrapPad, "one run per lane; a run may cross the ring end inside the pad");
static_assert(kRunsPerSlice % 8 == 0, "a slice is a whole number of aligned 8-lane groups (the octet merge of phase 3b)");
static_assert(kScanThreads % 64 == 0 && kScanThreads * kOutPerThread >= kWindowSamples, "one in-place pass");

struct ScanArgs
{
    DeviceStore st;
    float pp[12];
    int total_tiles;
    int tiles_per_xcd;
    MSK144_STAMP_ARG
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
static_assert(kOutGroups % 3 != 0 && (kOutSpan * 3) % 32 == kPulseHalf, "u -> 3u mod groups is a bijection with consecutive LDS residues");
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
    __shared__ float s_xb[kScanDepthMax][kScanSlices];        // per (pattern, slice): max |S| ...
    __shared__ uint32_t s_xpos[kScanDepthMax][kScanSlices];   // ... and its position
    __shared__ float s_oct_v[kScanDepthMax][kScanThreads / 8];     // per (pattern, octet of runs): max |S|^2 ...
    __shared__ uint32_t s_oct_pos[kScanDepthMax][kScanThreads / 8];  // ... and its position

    // XCD-aware tile map: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one
    // contiguous range of tiles - the F tiles of a channel then share one L2 copy of its window.
    const int xcd = blockIdx.x & 7;
    const int tile = xcd * a.tiles_per_xcd + (blockIdx.x >> 3);
    if((blockIdx.x >> 3) >= a.tiles_per_xcd || tile >= a.total_tiles) return;
    const int ch_rel = tile / a.st.F;  // channel inside the block [ch0, ch0 + nch) this launch covers
    const int b = tile - ch_rel * a.st.F;
    const int ch = a.st.ch0 + ch_rel;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    MSK144_STAMP_ROW(tile);
    MSK144_STAMP(0);
    MSK144_STAMP(11);  // two stamps back to back: the stamp's own cost

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
    MSK144_STAMP(1);
    __syncthreads();
    MSK144_STAMP(2);

    // ---- 2. C[n0 + 6r], r = 0..10, by pulse decomposition (correlate_pulses); C then overwrites the window in place ----
    {
        // Thread 6u + c takes output group q = 3u mod 79 (a bijection, 79 is prime): its stream starts at 66 q + c, and
        // 66 * 3 = 6 mod 32, so the 32 lanes of a ds_read_b64 group read 32 consecutive residues mod 32 - conflict-free.
        // With q = u the group stride is 66 = 2 mod 32 and every read (and every C store) is 3-way conflicted: the phase
        // was LDS-bound (removal experiment: 3.8 ms for 2.5 ms of priced issue).
        const int u = tid / kPulseHalf;
        int q = 3 * u;
        q -= q >= 2 * kOutGroups ? 2 * kOutGroups : (q >= kOutGroups ? kOutGroups : 0);
        if(u >= kOutGroups) q = kOutGroups;  // idle threads
        const int n0 = kOutSpan * q + (tid - u * kPulseHalf);
        float2 c[kOutPerThread];
        if(q < kOutGroups) correlate_pulses((lds_f2_ptr)(s_buf + n0), a.pp, c);
        MSK144_STAMP(3);
        __syncthreads();  // every thread has read its samples: C may overwrite the window
        if(q < kOutGroups)
        {
#pragma unroll
            for(int r = 0; r < kOutPerThread; r++)
            {
                const int n = n0 + kPulseHalf * r;
                if(n < kWindowSamples) s_buf[n] = c[r];
            }
        }
    }
    __syncthreads();
    MSK144_STAMP(4);

    // ---- 2b. E[n] = C[n] + C[n + 336 mod N]: the two sync words of a frame, summed once per ring position ----
    // S(pos, p) = sum over the pattern's frames of E[(pos + 864 m) mod N]: the fold then needs one load and one complex add per
    // frame instead of two (the association (S + a) + b becomes S + (a + b): ~1e-7 relative on xb, like the other re-associations).
    {
        float2 e[kMixPerThread];
#pragma unroll
        for(int i = 0; i < kMixPerThread; i++)
        {
            const int n = tid + i * kScanThreads;
            if((i + 1) * kScanThreads <= kWindowSamples || n < kWindowSamples)
            {
                int m = n + kSecondSyncSample;
                if(m >= kWindowSamples) m -= kWindowSamples;
                const float2 ca = s_buf[n], cb = s_buf[m];
                e[i] = make_float2(ca.x + cb.x, ca.y + cb.y);
            }
        }
        __syncthreads();
#pragma unroll
        for(int i = 0; i < kMixPerThread; i++)
        {
            const int n = tid + i * kScanThreads;
            if((i + 1) * kScanThreads <= kWindowSamples || n < kWindowSamples)
            {
                s_buf[n] = e[i];
                if(n < kRun) s_buf[kWindowSamples + n] = e[i];  // a run of the fold phase may cross the ring end
            }
        }
    }
    __syncthreads();
    MSK144_STAMP(5);

    // ---- 3. fold per pattern, |S|^2, running arg-max along this lane's run of positions ----
    // kD is a template parameter: the pattern loop is straight-line code.  Patterns 7 and 8 (100100, 100110) reuse the frames
    // already loaded.  Lane stride = 11 positions = 22 dwords: the 32 lanes of a ds_read_b64 group tile all 64 banks.
    constexpr int D = kD;
    constexpr int kFrames = kD < kPatternBits ? kD : kPatternBits;
    const bool has_run = tid < kScanSlices * kRunsPerSlice;
    float best[D];
    uint32_t bidx[D];
#pragma unroll
    for(int p = 0; p < D; p++)
    {
        best[p] = 0.0f;  // |S|^2 = 0 everywhere keeps position 0 of the run: lowest position wins ties
        bidx[p] = 0u;
    }
    if(has_run)
    {
        const int slice = tid / kRunsPerSlice;
        const int k = tid - slice * kRunsPerSlice;
        const int in_slice = k * kRun < kSlicePositions - kRun ? k * kRun : kSlicePositions - kRun;  // last run: 245..255
        const uint32_t start = static_cast<uint32_t>(slice * kSlicePositions + in_slice);            // 0..5365
        typedef const volatile __attribute__((address_space(3))) v2f* lds_run_ptr;
        lds_run_ptr pe[kFrames];
#pragma unroll
        for(int m = 0; m < kFrames; m++)
        {
            uint32_t ta = start + static_cast<uint32_t>(kFrameSamples * m);  // < 2 * ring
            ta = min(ta, ta - static_cast<uint32_t>(kWindowSamples));
            ta = min(ta, ta - static_cast<uint32_t>(kWindowSamples));        // start itself may exceed the ring (positions reach 5375)
            pe[m] = (lds_run_ptr)(s_buf + ta);
        }
#pragma unroll
        for(int i = 0; i < kRun; i++)
        {
            float2 ef[kFrames];
#pragma unroll
            for(int m = 0; m < kFrames; m++) ef[m] = as_float2(pe[m][i]);
            float sr = 0.0f, si = 0.0f;
#pragma unroll
            for(int p = 0; p < D; p++)
            {
                if(p == 0)
                {
                    sr = ef[0].x;
                    si = ef[0].y;
                }
                else if(p < kPatternBits)
                {
                    sr += ef[p].x;  // nested prefix masks: add frame p
                    si += ef[p].y;
                }
                else
                {
                    sr = ef[0].x;  // patterns 7 and 8 (100100, 100110) start from frame 0 again
                    si = ef[0].y;
#pragma unroll
                    for(int m = 1; m < kPatternBits; m++)
                    {
                        if(kPatternMask[p][m])
                        {
                            sr += ef[m].x;
                            si += ef[m].y;
                        }
                    }
                }
                const float v = fmaf(sr, sr, si * si);
                // strict >: the lowest position keeps exact ties
                const bool better = v > best[p];
                best[p] = __builtin_fmaxf(v, best[p]);
                bidx[p] = better ? static_cast<uint32_t>(i) : bidx[p];
            }
            // every pattern's running maximum of this position is complete before the next position's loads may issue
            // (patterns 7 and 8 do not depend on all frames, so tying only the last sum would let the others pile up):
            // the window pointers are volatile, so the next position's loads stay behind this statement; its operands make
            // the statement wait for this position's arithmetic
            // (every fourth position rather than every one: two or three positions in flight hide the LDS latency of the next loads
            // without costing a register - 76 VGPRs either way; none at all lets the loads pile up again: 9.42 -> 9.22 / 10.3 ms)
            if(i % 4 == 3) asm volatile("" : "+v"(best[0]), "+v"(best[D - 1]), "+v"(best[D > 6 ? 5 : 0]), "+v"(best[D > 7 ? 6 : 0]));
        }
#pragma unroll
        for(int p = 0; p < D; p++) bidx[p] += start;
    }
    MSK144_STAMP(6);
    // ---- 3b. the 24 runs of a slice meet: octet maxima in registers, three octets per slice through LDS ----
    // Runs sit in lane order (tid = 24 slice + run), so a slice is exactly three aligned groups of eight lanes.  Each octet reduces
    // its eight running maxima with three DPP steps and the FIRST lane holding the maximum (lowest position: the reference's
    // strict-> trees keep the lowest position on exact ties) parks its (|S|^2, position) pair in s_oct - two masked stores per
    // pattern instead of twelve unconditional ones per thread, and a three-element merge per (pattern, slice) instead of a
    // 24-element one that kept six of the eight waves waiting.  s_oct is its own 4 KB of LDS, so no barrier is needed between the
    // fold loop (which still reads the window buffer) and these stores.
    {
        float omax[D + (D & 1)];
#pragma unroll
        for(int p = 0; p < D; p++) omax[p] = best[p];
        if(D & 1) omax[D] = 0.0f;
#pragma unroll
        for(int p = 0; p < D; p += 2) oct_max2_f32(omax[p], omax[p + 1]);
        const uint64_t lower = ((1ull << (lane & 7)) - 1ull) << (lane & 56);  // the lanes of my octet below me
        const int oct = tid >> 3;
#pragma unroll
        for(int p = 0; p < D; p++)
        {
            // >= rather than ==: omax is the DPP maximum of the octet's own values, so the two are the same test - unless the
            // maximum did not survive v_max_f32_dpp bit for bit (a build that flushes denormals would turn a denormal |S|^2 into 0):
            // then every lane at or above the flushed value qualifies and the first of them stores, instead of none (the slice
            // merge below would read an unwritten cell).  tests/test_gpu_parity.py::test_scan_tiny_amplitude_window feeds denormal |S|^2.
            const bool is_max = best[p] >= omax[p];
            const uint64_t eq = __ballot(is_max);
            if(is_max && (eq & lower) == 0ull)
            {
                s_oct_v[p][oct] = best[p];
                s_oct_pos[p][oct] = bidx[p];
            }
        }
    }
    MSK144_STAMP(7);
    __syncthreads();
    MSK144_STAMP(8);

    // ---- 4a. slice maximum = first strict maximum over its three octets in position order, xb = |S| (correctly rounded sqrt) ----
    for(int e = tid; e < D * kScanSlices; e += kScanThreads)
    {
        const int p = e / kScanSlices;
        const int sl = e - p * kScanSlices;
        float bv = s_oct_v[p][3 * sl];
        uint32_t bp = s_oct_pos[p][3 * sl];
#pragma unroll
        for(int k = 1; k < kRunsPerSlice / 8; k++)
        {
            const float v = s_oct_v[p][3 * sl + k];
            const uint32_t q = s_oct_pos[p][3 * sl + k];
            if(v > bv)
            {
                bv = v;
                bp = q;
            }
        }
        s_xb[p][sl] = f32_sqrt(bv);
        s_xpos[p][sl] = bp;
    }
    __syncthreads();
    MSK144_STAMP(9);
    if(wave != 0) MSK144_STAMP_WAVE_END();

    // ---- 4b. 8-slot replacement rule in slice order (scan_kernel.cuh:276-353), one wave: lane = 8*pattern + slot ----
    if(wave == 0)
    {
        const int p = lane >> 3;
        const int slot = lane & 7;
        const int pc = p < D ? p : 0;  // lanes of unused patterns shadow pattern 0 and store nothing
        float my_xb = 0.0f;            // reset(): pos 0, xb 0 (scan_kernel.cuh:78-82)
        uint32_t my_pos = 0u;
        // From the reset state the rule fills slots 0..7 in slice order as long as each of those slice maxima is > 0 (every
        // stored xb is 0, the lowest-numbered slot still at 0 is the arg-min, and best > 0 replaces it): eight steps in one,
        // unless some early maximum is 0 or NaN (all-zero or non-finite window) - then the serial rule runs from the start.
        int s_first = 0;
        {
            const float mine = s_xb[pc][slot];
            if(__ballot(mine > 0.0f) == ~0ull)
            {
                my_xb = mine;
                my_pos = s_xpos[pc][slot];
                s_first = kSlotsPerPattern;
            }
        }
        for(int s = s_first; s < kScanSlices; s++)
        {
            const float best = s_xb[pc][s];
            const uint32_t best_pos = s_xpos[pc][s];
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
        MSK144_STAMP(10);
        MSK144_STAMP_WAVE_END();
    }
}

}  // namespace

void launch_scan(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream)
{
    ScanArgs a;
    a.st = st;
    for(int i = 0; i < 12; i++) a.pp[i] = tpl.pp[i];
    a.total_tiles = st.nch * st.F;  // the whole batch (ch0 = 0, nch = channels) or one channel block of the overlapped schedule
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
#ifdef MSK144_PHASE_STAMPS
    a.stamps = stamp_buffer(0);
#endif
    const dim3 grid(a.tiles_per_xcd * 8), block(kScanThreads);
#ifdef MSK144_SCAN_LDS_PAD_BYTES
    // occupancy probe only (tools/ab_variants.py, DESIGN.md 4.3: what a fused scan -> softbits tile that keeps the mixed window beside C
    // would cost in resident workgroups): unused dynamic LDS that limits a CU to two (+15 KB) or one (+42 KB) workgroups.  Never in the product build.
    constexpr size_t kDynLds = MSK144_SCAN_LDS_PAD_BYTES;
#else
    constexpr size_t kDynLds = 0;
#endif
    switch(st.D)
    {
    case 1: hipLaunchKernelGGL(scan_kernel<1>, grid, block, kDynLds, stream, a); break;
    case 2: hipLaunchKernelGGL(scan_kernel<2>, grid, block, kDynLds, stream, a); break;
    case 3: hipLaunchKernelGGL(scan_kernel<3>, grid, block, kDynLds, stream, a); break;
    case 4: hipLaunchKernelGGL(scan_kernel<4>, grid, block, kDynLds, stream, a); break;
    case 5: hipLaunchKernelGGL(scan_kernel<5>, grid, block, kDynLds, stream, a); break;
    case 6: hipLaunchKernelGGL(scan_kernel<6>, grid, block, kDynLds, stream, a); break;
    case 7: hipLaunchKernelGGL(scan_kernel<7>, grid, block, kDynLds, stream, a); break;
    default: hipLaunchKernelGGL(scan_kernel<8>, grid, block, kDynLds, stream, a); break;
    }
}

}  // namespace msk144
