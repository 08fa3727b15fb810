// hop_ring: the 50 %-overlap window ring of every stream, kept on the device.
//
// Replaces the host-side ring of the reference's working loop (main.cu:284-288 audio, 349-353 IQ: copy the second half of the
// window over the first, fread the next 2592 samples behind it; the first read fills all 5184, main.cu:271-283).  One workgroup
// per stream that has a hop in this batch: it advances that stream's ring in place and writes the stream's window into the
// compact batch the front end reads (position j of the batch = stream streams[j]), so the host ships 2592 new samples per stream
// and hop instead of a 5184-sample window and keeps no window of its own.  A window is 10 368 bytes in both read modes
// (5184 int16, or 5184 int8 I/Q pairs): the kernel moves 32-bit words and never looks inside them.
#include "msk144_kernels.h"

namespace msk144
{

namespace
{

constexpr int kRingThreads = 256;
constexpr int kWindowWords = kWindowSamples * 2 / 4;  // 2592
constexpr int kHalfWords = kWindowWords / 2;          // 1296

__global__ __launch_bounds__(kRingThreads) void hop_ring_kernel(uint32_t* __restrict__ ring, const uint32_t* __restrict__ hops, const uint32_t* __restrict__ first_halves,
                                                                const int32_t* __restrict__ streams, const uint8_t* __restrict__ is_first, uint32_t* __restrict__ windows)
{
    const int j = blockIdx.x;
    const int s = streams[j];
    const bool first = is_first[j] != 0;
    uint32_t* __restrict__ r = ring + static_cast<size_t>(s) * kWindowWords;
    const uint32_t* __restrict__ hop = hops + static_cast<size_t>(j) * kHalfWords;
    const uint32_t* __restrict__ head = first_halves + static_cast<size_t>(j) * kHalfWords;
    uint32_t* __restrict__ w = windows + static_cast<size_t>(j) * kWindowWords;
    // every thread owns word i of both halves: it reads r[half + i] before anything overwrites it
    for(int i = threadIdx.x; i < kHalfWords; i += kRingThreads)
    {
        const uint32_t lo = first ? head[i] : r[kHalfWords + i];
        const uint32_t hi = hop[i];
        r[i] = lo;
        r[kHalfWords + i] = hi;
        w[i] = lo;
        w[kHalfWords + i] = hi;
    }
}

// Shader clock probe (msk144_clock_probe): spins on the constant 100 MHz counter and reports how many shader cycles went by.  The
// loop is bounded twice - by the tick target and by an iteration cap (each round sleeps 8 x 64 cycles: 2^21 rounds < 0.5 s at any
// clock) - so the wave always ends.
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t* __restrict__ out, uint32_t ticks)
{
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = r0;
    for(int i = 0; i < (1 << 21) && r1 - r0 < ticks; i++)
    {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if(threadIdx.x == 0)
    {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
}

}  // namespace

void launch_clock_probe(uint64_t* out, uint32_t ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, ticks);
}

void launch_hop_ring(void* ring, const void* hops, const void* first_halves, const int32_t* streams, const uint8_t* is_first, void* windows, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(hop_ring_kernel, dim3(n), dim3(kRingThreads), 0, stream, static_cast<uint32_t*>(ring), static_cast<const uint32_t*>(hops),
                       static_cast<const uint32_t*>(first_halves), streams, is_first, static_cast<uint32_t*>(windows));
}

}  // namespace msk144
