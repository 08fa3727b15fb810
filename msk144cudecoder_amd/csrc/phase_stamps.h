// Opt-in diagnostic build (-DMSK144_PHASE_STAMPS, tools/phase_stamps.py): s_memtime at the phase boundaries of scan_kernel and
// softbits_kernel, written by wave 0 of every kStampEvery-th tile, plus the cycle at which each wave of that workgroup ends.
// In the product build the macros expand to nothing: no stamp executes, no argument is added.  Stamp form per
// cdna_hip_programming.md "In-kernel stamps": ONE asm statement holding the s_memtime and its lgkmcnt(0), fenced by
// sched_barriers; the stamp's own cost (two back to back: ~40 cycles) is measured in slot kStampSelf and subtracted by the tool.
#pragma once

#ifdef MSK144_PHASE_STAMPS
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msk144
{

constexpr int kStampSlots = 24;     // uint64 per sampled workgroup
constexpr int kStampEvery = 61;     // every 61st tile (prime: walks all frequencies and channels)
constexpr int kStampRows = 16384;   // sampled workgroups kept per kernel
constexpr int kStampWaveEnd = 16;   // slots 16..23: cycle at which wave w of the workgroup ended

uint64_t* stamp_buffer(int kernel);  // 0 = scan, 1 = softbits; device memory, allocated and zeroed on first use

__device__ __forceinline__ uint64_t stamp_now()
{
    uint64_t t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

__device__ __forceinline__ uint64_t* stamp_row(uint64_t* buf, int tile)
{
    if(!buf || tile % kStampEvery != 0 || tile / kStampEvery >= kStampRows) return nullptr;
    return buf + static_cast<size_t>(tile / kStampEvery) * kStampSlots;
}

}  // namespace msk144

#define MSK144_STAMP_ARG uint64_t* stamps;
#define MSK144_STAMP_ROW(tile) uint64_t* const stamp_row_ = ::msk144::stamp_row(a.stamps, (tile))
// wave 0, lane 0 writes slot `slot`
#define MSK144_STAMP(slot)                                         \
    do                                                             \
    {                                                              \
        if(stamp_row_ && threadIdx.x < 64)                         \
        {                                                          \
            const uint64_t t_ = ::msk144::stamp_now();             \
            if(threadIdx.x == 0) stamp_row_[(slot)] = t_;          \
        }                                                          \
    } while(0)
// every wave: the cycle it ends at
#define MSK144_STAMP_WAVE_END()                                                                              \
    do                                                                                                       \
    {                                                                                                        \
        if(stamp_row_)                                                                                       \
        {                                                                                                    \
            const uint64_t t_ = ::msk144::stamp_now();                                                       \
            if((threadIdx.x & 63) == 0) stamp_row_[::msk144::kStampWaveEnd + (threadIdx.x >> 6)] = t_;       \
        }                                                                                                    \
    } while(0)
#else
#define MSK144_STAMP_ARG
#define MSK144_STAMP_ROW(tile) \
    do                         \
    {                          \
    } while(0)
#define MSK144_STAMP(slot) \
    do                     \
    {                      \
    } while(0)
#define MSK144_STAMP_WAVE_END() \
    do                          \
    {                           \
    } while(0)
#endif
