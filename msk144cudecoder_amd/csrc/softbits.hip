// softbits: per-candidate frame fold, carrier-phase estimate, matched filter, LLR scaling, sync check.
//
// Replaces softbits_kernel (softbits_kernel.cuh:9-249).  The reference launches one
// 160-thread block per candidate and every block re-mixes the whole 5184-sample window; here one
// workgroup (8 waves) serves all D*8 candidates of a (channel, frequency) pair, mixes the window ONCE
// into LDS (same float phase as the scan) and then each 64-lane wave demodulates candidates on its own,
// entirely in registers - no workgroup barrier after the mix and no LDS scratch, so the only LDS is the
// 48.4 KB window (ring + one frame of pad) and three workgroups (24 waves) fit a CU.
//
// A candidate is demodulated in two parts: first the slots its sync check needs (slot 0 carries softbits 0..7 and 56..63,
// slot 2 the partial sum that wraps into softbit 0), the carrier phase and nbadsync; then the middle slot, the
// normalisation and the LLR row.  In blocked staging (no LLR row outlives its channel block) a candidate the index stage
// will drop (nbadsync > threshold) stops after part one; with the LLR store retained every candidate is demodulated in full.
//
// Register layout: the 864-sample folded frame is cut into 144 half-bit groups of 6 samples; group
// h = lane + 64*s (s = 0,1,2) lives in lane `lane`, slot `s`.  Softbit u needs groups u-1 and u
// (softbits_kernel.cuh:158-177: I bits 12j..12j+11, Q bits 12j-6..12j+5), so its 12-tap sum starts in
// the lane of group u-1 (taps pp[0..5]), hops one lane with a DPP wave shift and finishes in lane
// u%64 with taps pp[6..11], and softbit u ends up in lane u%64 of slot u/64.  The tap sums run on the folded
// complex samples; the carrier rotation, common to the whole frame, is applied to the two sums of a group.
// Fold reads are ds_read_b128 (conflict-free at the 48-byte lane stride; see the fold section).
// From there the two 144-term sums (softbits_kernel.cuh:186-194) are one per-lane add over the three slots plus a
// single cross-lane reduction carrying both sums (sum_reduction.cuh:14-44 replaced by two interleaved DPP chains).
// The phase rotation uses conj(s)/|s| instead of atan2f + sincosf (same unit vector to ~1 ulp); the
// 84-term phase sum is accumulated per lane and then across lanes (order differs from the reference's
// 42->32->16 tree: ~1e-7 relative on a rotation angle).
#include "msk144_kernels.h"
#include "mix.h"
#include "phase_stamps.h"
#include "wave64.h"

namespace msk144
{

namespace
{

constexpr int kSbThreads = 512;
constexpr int kSbWaves = kSbThreads / 64;
constexpr int kGroup = 6;                                  // samples per half-bit group
constexpr int kGroups = kFrameSamples / kGroup;            // 144
constexpr int kSlots = (kGroups + 63) / 64;                // 3
// The window is a ring of exactly six frames.  LDS carries one more frame (+ one run of samples) behind it, so a
// folded frame starting anywhere in the ring is 864 CONTIGUOUS samples: its byte address is a wave-uniform frame base
// (scalar arithmetic) plus a per-lane constant - no per-lane wrap arithmetic in the fold.  48.4 KB, three workgroups per CU.
constexpr int kRingPad = kFrameSamples + kGroup - 1;

struct SoftbitsArgs
{
    DeviceStore st;
    SyncTemplate tpl;
    int total_tiles;
    int tiles_per_xcd;
    MSK144_STAMP_ARG
};

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kDppWaveShr1 = 0x138;  // lane l <- lane l-1 across the whole wave (GFX9 DPP)

template<int kCtrl>
__device__ __forceinline__ float dpp_add(float v)
{
    return f32_add(v, dpp_f32<kCtrl>(v));
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;
typedef const volatile __attribute__((address_space(3))) v4f* lds_v4f_ptr;

// Fold of the averaged frames (softbits_kernel.cuh:59-82) for the slots named in kSlotMask.
// A lane reads its group's six samples (48 B).  With ds_read_b64 the 48-byte lane stride makes lanes l and l + 16 of a
// 32-lane access group share a bank pair (6*16 = 0 mod 32): a 2-way conflict on every read.  ds_read_b128 services 16 lanes
// per cycle, and at this stride their sixteen 16-byte pieces tile all 64 banks exactly: conflict-free, 4 cycles per TWO
// samples.  It needs 16-byte alignment, which depends only on the parity of the candidate position (wave-uniform; group
// offsets 6g and frame offsets 864m are even): even -> three b128; odd -> b64, two b128, b64.  volatile keeps the compiler
// from re-merging.  Frame 0 is part of every pattern (msk_context.cuh:231-238): its samples ARE the initial sums.
template<int kSlotMask>
__device__ __forceinline__ void fold_frames(v2f (&acc)[kSlots][kGroup], const char* xbytes, const uint32_t (&lane8)[kSlots], uint32_t pos, int p)
{
#ifdef MSK144_LISTING_EVEN_ONLY
    const bool pos_even = true;  // tools/phase_stamps.py prices ONE of the two alignment paths of a listing (never a run build)
#else
    const bool pos_even = (pos & 1u) == 0u;
#endif
    if(pos_even)
    {
#pragma unroll
        for(int s = 0; s < kSlots; s++)
        {
            if(!((kSlotMask >> s) & 1)) continue;
            lds_v4f_ptr q = (lds_v4f_ptr)(xbytes + (lane8[s] + pos * 8u));
            const v4f q0 = q[0], q1 = q[1], q2 = q[2];
            acc[s][0] = v2f{q0.x, q0.y};
            acc[s][1] = v2f{q0.z, q0.w};
            acc[s][2] = v2f{q1.x, q1.y};
            acc[s][3] = v2f{q1.z, q1.w};
            acc[s][4] = v2f{q2.x, q2.y};
            acc[s][5] = v2f{q2.z, q2.w};
        }
        for(int m = 1; m < kPatternBits; m++)
        {
            if(!kPatternMask[p][m]) continue;  // wave-uniform
            uint32_t fb = pos + static_cast<uint32_t>(kFrameSamples * m);  // frame base in the ring: scalar
            if(fb >= static_cast<uint32_t>(kWindowSamples)) fb -= kWindowSamples;
#pragma unroll
            for(int s = 0; s < kSlots; s++)
            {
                if(!((kSlotMask >> s) & 1)) continue;
                lds_v4f_ptr q = (lds_v4f_ptr)(xbytes + (lane8[s] + fb * 8u));
                const v4f q0 = q[0], q1 = q[1], q2 = q[2];
                acc[s][0] += v2f{q0.x, q0.y};
                acc[s][1] += v2f{q0.z, q0.w};
                acc[s][2] += v2f{q1.x, q1.y};
                acc[s][3] += v2f{q1.z, q1.w};
                acc[s][4] += v2f{q2.x, q2.y};
                acc[s][5] += v2f{q2.z, q2.w};
            }
        }
    }
    else
    {
#pragma unroll
        for(int s = 0; s < kSlots; s++)
        {
            if(!((kSlotMask >> s) & 1)) continue;
            const uint32_t i8 = lane8[s] + pos * 8u;
            lds_v2f_ptr r = (lds_v2f_ptr)(xbytes + i8);
            lds_v4f_ptr q = (lds_v4f_ptr)(xbytes + i8 + 8);
            acc[s][0] = r[0];
            const v4f q0 = q[0], q1 = q[1];
            acc[s][5] = r[5];
            acc[s][1] = v2f{q0.x, q0.y};
            acc[s][2] = v2f{q0.z, q0.w};
            acc[s][3] = v2f{q1.x, q1.y};
            acc[s][4] = v2f{q1.z, q1.w};
        }
        for(int m = 1; m < kPatternBits; m++)
        {
            if(!kPatternMask[p][m]) continue;  // wave-uniform
            uint32_t fb = pos + static_cast<uint32_t>(kFrameSamples * m);
            if(fb >= static_cast<uint32_t>(kWindowSamples)) fb -= kWindowSamples;
#pragma unroll
            for(int s = 0; s < kSlots; s++)
            {
                if(!((kSlotMask >> s) & 1)) continue;
                const uint32_t i8 = lane8[s] + fb * 8u;
                lds_v2f_ptr r = (lds_v2f_ptr)(xbytes + i8);
                lds_v4f_ptr q = (lds_v4f_ptr)(xbytes + i8 + 8);
                const v2f x0 = r[0];
                const v4f q0 = q[0], q1 = q[1];
                const v2f x5 = r[5];
                acc[s][0] += x0;
                acc[s][1] += v2f{q0.x, q0.y};
                acc[s][2] += v2f{q0.z, q0.w};
                acc[s][3] += v2f{q1.x, q1.y};
                acc[s][4] += v2f{q1.z, q1.w};
                acc[s][5] += x5;
            }
        }
    }
}

// Matched filter on the folded complex samples of one slot (softbits_kernel.cuh:157-180 before the rotation of :146-153).
// The rotation is the same for every sample of the frame, so it commutes with the tap sums: filter the folded complex samples
// first (two real FMAs per sample and pulse half) and rotate the two complex sums of a group afterwards - 28 instead of 36
// multiply-adds per group.  Same linear form as rotating every sample first, associated differently (~1e-7 relative).
// pp[0] = sin 0 = 0 and pp[6] = sin pi/2 = 1 exactly (checked at create): the first tap of u1 vanishes, u2's is the sample.
__device__ __forceinline__ void filter_group(const v2f (&x)[kGroup], const float (&pp)[12], v2f& u1, v2f& u2)
{
    u1 = x[1] * pp[1];
    u2 = x[0];
#pragma unroll
    for(int t = 1; t < kGroup; t++)
    {
        if(t > 1)
        {
            u1.x = fmaf(x[t].x, pp[t], u1.x);
            u1.y = fmaf(x[t].y, pp[t], u1.y);
        }
        u2.x = fmaf(x[t].x, pp[kGroup + t], u2.x);
        u2.y = fmaf(x[t].y, pp[kGroup + t], u2.y);
    }
}

// kGateEarly: blocked staging keeps no LLR row of a candidate the index stage will drop (nbadsync > threshold), so such a
// candidate stops after its sync check - a third of the noise candidates at threshold 3.  With the LLR store retained
// (llr_block_channels = channels: dumps, parity tests) every candidate is demodulated in full, as in the reference.
// kHandOver (only with kGateEarly): slots that fold the same frames as a lower slot of their group are handed to it (below).
template<bool kGateEarly, bool kHandOver>
__global__ __launch_bounds__(kSbThreads, 6) void softbits_kernel(const SoftbitsArgs a)
{
    __shared__ __attribute__((aligned(16))) float2 s_x[kWindowSamples + kRingPad + 3];

    const int xcd = blockIdx.x & 7;
    const int tile = xcd * a.tiles_per_xcd + (blockIdx.x >> 3);
    if((blockIdx.x >> 3) >= a.tiles_per_xcd || tile >= a.total_tiles) return;
    const int ch_rel = tile / a.st.F;  // channel inside the block [ch0, ch0 + nch)
    const int b = tile - ch_rel * a.st.F;
    const int ch = a.st.ch0 + ch_rel;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    MSK144_STAMP_ROW(tile);
    MSK144_STAMP(0);
    MSK144_STAMP(11);  // two stamps back to back: the stamp's own cost

    static_assert(kSbWaves == kSlotsPerPattern, "wave w owns slot w of every pattern: candidate c = w + 8 i is (pattern i, slot w)");
    // Candidates of this wave: c = wave + 8 i, i < D.  The scan positions of all the tile's candidates are fetched once, one per lane
    // (the latency hides under the mix phase), and handed out by readlane: the position is a scalar in the candidate loop.
    const int D = a.st.D;
    const int ncand = D * kSlotsPerPattern;
    const size_t item0 = static_cast<size_t>(ch) * a.st.K + static_cast<size_t>(b) * ncand;
    uint32_t pos_of_lane = 0u;  // lane l: scan position of the tile's candidate l (pattern l / 8, slot l % 8)
    if(lane < ncand) pos_of_lane = a.st.pos[item0 + lane];
    // ---- mix (softbits_kernel.cuh:27-52) ----
    const float f0 = -1.0f * a.st.freq[b];
    const float2* __restrict__ cdat = a.st.analytic + static_cast<size_t>(ch) * kWindowSamples;
    {
        constexpr int kPerThread = (kWindowSamples + kSbThreads - 1) / kSbThreads;  // 11 (10.125)
        float2 xin[kPerThread];  // all loads in flight before any arithmetic
#pragma unroll
        for(int i = 0; i < kPerThread; i++)
        {
            const int n = tid + i * kSbThreads;
            xin[i] = n < kWindowSamples ? cdat[n] : make_float2(0.0f, 0.0f);
        }
        const float tid_f = static_cast<float>(tid);
#pragma unroll
        for(int i = 0; i < kPerThread; i++)
        {
            const int n = tid + i * kSbThreads;
            if(n < kWindowSamples)
            {
                const float2 y = mix_sample(xin[i], tid_f + static_cast<float>(i * kSbThreads), f0);
                s_x[n] = y;
                if(n < kRingPad) s_x[kWindowSamples + n] = y;
            }
        }
    }
    MSK144_STAMP(1);
    __syncthreads();
    MSK144_STAMP(2);
#ifdef MSK144_PHASE_STAMPS
    uint64_t st_part1 = 0, st_part2 = 0, st_n2 = 0;  // wave 0: cycles in part one / part two of its candidates, part-two runs
#endif

    // Slots that fold the same frames.  The scan walks 5376 positions of a 5184-sample ring, so positions p and p + 5184 are one
    // place (2.9 % of the slots of a noise window hold such a pair); masks 111111 and 100100 moreover sum the same frames at pos
    // and pos + 864 (+ 2592), so the eight slots of those patterns are mostly the copies of two or three peaks, one per period
    // (exact ties in exact arithmetic: 73 % of the slots of pattern 5 on the bench workload).  The reference demodulates and decodes
    // every copy (softbits_kernel.cuh:56-83 folds the frames of a periodic copy in another order: the same sums up to float
    // association; a wrapped copy is the same computation).  Here, when no LLR row outlives its block, a slot whose position is
    // congruent to a LOWER slot's of its group hands its work to that slot: it stores -1 - slot as its nbadsync, the index stage
    // leaves it out and the collect stage gives it the nbadsync and the decode of the slot it names (index.hip).  With the store
    // retained every slot is computed on its own, as in the reference.
    // Bit 8 i + s of same_frames: slot s of pattern i is congruent to THIS wave's slot of pattern i.
    static_assert(kGateEarly || !kHandOver, "a retained LLR store keeps every slot's own row");
    uint64_t same_frames = 0;
    if(kHandOver)
    {
        uint32_t r = pos_of_lane >= static_cast<uint32_t>(kWindowSamples) ? pos_of_lane - kWindowSamples : pos_of_lane;
        const int pattern_of_lane = lane >> 3;
        if(pattern_of_lane == kFirstPeriodicPattern) r %= static_cast<uint32_t>(kPatternPeriod[0]);
        if(pattern_of_lane == kFirstPeriodicPattern + 1) r %= static_cast<uint32_t>(kPatternPeriod[1]);
        const uint32_t mine = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(4 * ((lane & ~(kSlotsPerPattern - 1)) + wave), static_cast<int>(r)));
        same_frames = __ballot(lane < ncand && r == mine);
    }

    // Phase estimate = sum over the two sync words of folded sample x conj(template) (softbits_kernel.cuh:88-137).  The first
    // sync word covers samples 0..41 = groups 0..6 (lanes 0..6 of slot 0), the second samples 336..377 = groups 56..62.
    // Inside a group the template is one half of the half-sine pulse on each rail, signed by a sync bit
    // (msk_context.cuh:188-196: I carries bits 1,1,3,3,5,5,7 and Q bits 0,2,2,4,4,6,6 over the seven groups; even groups
    // have the rising half on I and the falling half on Q, odd groups the other way round) - so a lane's share of the sum
    // is a signed combination of the two matched-filter sums u1 = sum x[t] pp[t], u2 = sum x[t] pp[6+t] of its group, which
    // the demodulator needs anyway:   even: (sI u1.x + sQ u2.y,  sI u1.y - sQ u2.x)    odd: (sI u2.x + sQ u1.y,  sI u2.y - sQ u1.x)
    // Four per-lane coefficients (0 outside the sync groups) replace the 36 multiply-adds and six table loads per candidate.
    const int cb_group = (lane < 7) ? lane : (lane >= 56 && lane < 63) ? lane - 56 : -1;
    float k_u1x = 0.0f, k_u2x = 0.0f, k_u1y = 0.0f, k_u2y = 0.0f;  // pr = k_u1x u1.x + k_u2x u2.x + k_u1y u1.y + k_u2y u2.y
#pragma unroll
    for(int g = 0; g < 7; g++)
    {
        if(cb_group == g)
        {
            const float s_i = static_cast<float>(kSync8Pm[2 * (g / 2) + 1]);
            const float s_q = static_cast<float>(kSync8Pm[2 * ((g + 1) / 2)]);
            if(g % 2 == 0)
            {
                k_u1x = s_i;
                k_u2y = s_q;
            }
            else
            {
                k_u2x = s_i;
                k_u1y = s_q;
            }
        }
    }
    const bool odd = (lane & 1) != 0;
    float pp[12];
#pragma unroll
    for(int i = 0; i < 12; i++) pp[i] = a.tpl.pp[i];

    const char* __restrict__ xbytes = reinterpret_cast<const char*>(s_x);

    // byte offset of this lane's group inside a frame, per slot.  Slot 2 only has groups 128..143: lanes >= 16 re-read
    // group 143 (harmless, their results are discarded) so the loop stays convergent.
    uint32_t lane8[kSlots];
#pragma unroll
    for(int s = 0; s < kSlots; s++)
    {
        const int last = kGroups - 64 * s - 1;
        const int l = lane < last ? lane : last;
        lane8[s] = static_cast<uint32_t>(kGroup * (l + 64 * s)) * 8u;
    }
    // sync bit this lane checks (softbits 0..7 and 56..63), as +-1; 0 = none
    int sync_pm = 0;
#pragma unroll
    for(int k = 0; k < 8; k++)
        if(lane == k || lane == kSecondSyncBit + k) sync_pm = kSync8Pm[k];

    for(int i = 0; i < D; i++)
    {
        const int p = i, slot = wave;  // wave w owns slot w of every pattern
        const int c = slot + kSbWaves * p;
        const size_t item = item0 + c;
        uint32_t pos = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pos_of_lane), c));
        if(pos >= static_cast<uint32_t>(kWindowSamples)) pos -= kWindowSamples;  // scanned positions reach 5375
        if(kHandOver)
        {
            // a lower slot of this (frequency, pattern) group folds the same frames: it does the work, this slot names it
            const uint32_t lower = static_cast<uint32_t>(same_frames >> (kSlotsPerPattern * p)) & ((1u << slot) - 1u);
            if(lower != 0u)
            {
                if(lane == 0) a.st.nbadsync[item] = -1 - __builtin_ctz(lower);
                continue;
            }
        }
#ifdef MSK144_PHASE_STAMPS
        const uint64_t st_t0 = (stamp_row_ && tid < 64) ? stamp_now() : 0;
#endif

        // ---- part 1: everything the sync check needs.  The sync softbits 0..7 and 56..63 live in slot 0; softbit 0 also takes
        // the partial sum that starts in group 143 (slot 2, lane 15): the frame is circular.  Slot 1 waits. ----
        v2f acc[kSlots][kGroup];  // (re, im) pairs
        v2f u1[kSlots], u2[kSlots];
        fold_frames<0b101>(acc, xbytes, lane8, pos, p);
        filter_group(acc[0], pp, u1[0], u2[0]);
        filter_group(acc[2], pp, u1[2], u2[2]);

        // carrier phase from the two sync words (softbits_kernel.cuh:88-137): sum c3[k]*conj(cb[k])
        float pr = fmaf(k_u2y, u2[0].y, fmaf(k_u1y, u1[0].y, fmaf(k_u2x, u2[0].x, k_u1x * u1[0].x)));
        float pi = fmaf(-k_u2y, u2[0].x, fmaf(-k_u1y, u1[0].x, fmaf(k_u2x, u2[0].y, k_u1x * u1[0].y)));
        wave_sum2_f32(pr, pi);
        const float sre = pr, sim = pi;
        // cfac = conj(exp(i*atan2(im,re))) = (re, -im)/|s|
        float cr = 1.0f, ci = 0.0f;
        {
            const float m2 = fmaf(sre, sre, sim * sim);
            if(m2 > 0.0f)
            {
                const float inv = __builtin_amdgcn_rsqf(m2);  // 1 ulp: the unit phasor only needs ~1e-7
                cr = sre * inv;
                ci = -sim * inv;
            }
            else if(!(m2 == 0.0f))
            {
                cr = m2;  // NaN propagates like the reference's atan2f/sincosf chain
                ci = m2;
            }
        }

        // de-rotate the filtered sums (softbits_kernel.cuh:146-153)
        // va = plane that STARTS a softbit in this lane (even group -> I bit u+1 -> real part,
        // odd group -> Q bit u+1 -> imaginary part); vb = plane that FINISHES softbit u = this group.
        // re = fr*cr - fi*ci, im = fr*ci + fi*cr: pick the coefficient pair per lane once instead of
        // selecting per sample.
        const float a_r = odd ? ci : cr, a_i = odd ? cr : -ci;   // va = fr*a_r + fi*a_i
        const float b_r = odd ? cr : ci, b_i = odd ? -ci : cr;   // vb = fr*b_r + fi*b_i
        float start[kSlots], soft[kSlots];
        start[0] = fmaf(u1[0].y, a_i, u1[0].x * a_r);
        start[2] = fmaf(u1[2].y, a_i, u1[2].x * a_r);
        {
            // incoming partial sum from group h-1: lane-1 of the same slot; group 0 takes group 143 (slot 2, lane 15)
            float in = dpp_f32<kDppWaveShr1>(start[0]);
            const float edge = readlane_f32(start[kSlots - 1], kGroups - 64 * (kSlots - 1) - 1);
            // lane 0 <- edge (one instruction instead of v_mov + v_cndmask).  `edge` comes out of a v_readlane: on the gfx940 family a
            // VALU may read an SGPR written by a VALU only two wait states later, and the compiler does not pad asm statements
            asm("s_nop 1\n\tv_writelane_b32 %0, %1, 0" : "+v"(in) : "s"(edge));
            soft[0] = in + fmaf(u2[0].y, b_i, u2[0].x * b_r);
        }

        // ---- sync-word disagreements (softbits_kernel.cuh:214-241): bits 0..7 and 56..63 ----
        const int hard = (soft[0] < 0.0f) ? -1 : 1;
        const bool disagree = sync_pm != 0 && hard != sync_pm;
        const int nbad = __popcll(__ballot(disagree));
        if(lane == 0) a.st.nbadsync[item] = nbad;
#ifdef MSK144_PHASE_STAMPS
        const uint64_t st_t1 = (stamp_row_ && tid < 64) ? stamp_now() : 0;
        st_part1 += st_t1 - st_t0;
#endif
        if(kGateEarly && nbad > a.st.nbadsync_threshold) continue;  // wave-uniform: the index stage drops this candidate

        // ---- part 2: the middle slot and the rest of the demodulation ----
        fold_frames<0b010>(acc, xbytes, lane8, pos, p);
        filter_group(acc[1], pp, u1[1], u2[1]);
        start[1] = fmaf(u1[1].y, a_i, u1[1].x * a_r);
#pragma unroll
        for(int s = 1; s < kSlots; s++)
        {
            // lane 0 takes lane 63 of the previous slot
            float in = dpp_f32<kDppWaveShr1>(start[s]);
            const float edge = readlane_f32(start[s - 1], 63);
            asm("s_nop 1\n\tv_writelane_b32 %0, %1, 0" : "+v"(in) : "s"(edge));
            const float sb = in + fmaf(u2[s].y, b_i, u2[s].x * b_r);
            soft[s] = (s == kSlots - 1 && lane >= kGroups - 64 * (kSlots - 1)) ? 0.0f : sb;
        }

        // ---- normalisation (softbits_kernel.cuh:186-201) ----
        // sum and sum of squares of the 144 softbits: per lane over its three slots first, then ONE cross-lane reduction
        // for both (two interleaved DPP chains, row sums combined as ((r0+r1)+(r2+r3))).  The reference's
        // sum_reduction_two_cycles order (five 32-lane warp sums) was reproduced term by term until round 2 at the price of
        // six separate reductions; a different association moves sav/s2av by ~1e-7 relative, i.e. every LLR by ~1e-7 of
        // itself - four orders inside the 1e-3 tolerance and below what sincos/sqrt differences already contribute (8e-6).
        float sum_sav, sum_s2av;
        {
            float t = f32_add(f32_add(soft[0], soft[1]), soft[2]);
            float q = fmaf(soft[2], soft[2], fmaf(soft[1], soft[1], f32_mul(soft[0], soft[0])));
            wave_sum2_f32(t, q);
            sum_sav = t;
            sum_s2av = q;
        }
        // sav, s2av, ssig and the scale are wave-uniform numbers formed from two 144-term sums that already differ from the
        // reference's by ~1e-7 relative (different association): rounding them correctly on top (a Markstein division by 144,
        // the library's 15-instruction sqrt, a Newton step on the reciprocal) bought nothing measurable and cost ~35 uniform
        // VALU instructions per candidate.  One-ulp hardware forms: x * (1/144), v_sqrt_f32, 2 * v_rcp_f32.
        const float sav = sum_sav * (1.0f / 144.0f);
        const float s2av = sum_s2av * (1.0f / 144.0f);
        const float ssig = __builtin_amdgcn_sqrtf(fmaf(-sav, sav, s2av));
        const float sigma = 0.60f;
        const float scale = 2.0f * __builtin_amdgcn_rcpf(ssig * (sigma * sigma));

        // ---- store (softbits_kernel.cuh:204-211,244-247) ----
        float* __restrict__ llr = a.st.llr + (item - static_cast<size_t>(a.st.ch0) * a.st.K) * kCodeBits;
        if(lane >= 8 && lane < 56) llr[lane - 8] = f32_mul(scale, soft[0]);      // u = 8..55    -> 0..47
        llr[48 + lane] = f32_mul(scale, soft[1]);                                // u = 64..127  -> 48..111
        if(lane < 16) llr[112 + lane] = f32_mul(scale, soft[2]);                 // u = 128..143 -> 112..127
#ifdef MSK144_PHASE_STAMPS
        if(stamp_row_ && tid < 64)
        {
            st_part2 += stamp_now() - st_t1;
            st_n2++;
        }
#endif
    }
#ifdef MSK144_PHASE_STAMPS
    if(stamp_row_ && tid == 0)
    {
        stamp_row_[3] = st_part1;
        stamp_row_[4] = st_part2;
        stamp_row_[5] = st_n2;
    }
#endif
    MSK144_STAMP(6);
    MSK144_STAMP_WAVE_END();
}

}  // namespace

void launch_softbits(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream)
{
    SoftbitsArgs a;
    a.st = st;
    a.tpl = tpl;
    a.total_tiles = st.nch * st.F;
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
#ifdef MSK144_PHASE_STAMPS
    a.stamps = stamp_buffer(1);
#endif
    const int grid = a.tiles_per_xcd * 8;
    // LLR rows are retained only when one block covers every channel of the handle (msk144_api.cpp: dumps, parity tests)
    if(st.gate_early && st.handover) hipLaunchKernelGGL((softbits_kernel<true, true>), dim3(grid), dim3(kSbThreads), 0, stream, a);
    else if(st.gate_early) hipLaunchKernelGGL((softbits_kernel<true, false>), dim3(grid), dim3(kSbThreads), 0, stream, a);
    else hipLaunchKernelGGL((softbits_kernel<false, false>), dim3(grid), dim3(kSbThreads), 0, stream, a);
}

}  // namespace msk144
