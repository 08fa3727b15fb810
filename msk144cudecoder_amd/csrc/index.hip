// index: order-preserving compaction of candidates whose sync words disagree in at most
// `nbadsync_threshold` bits, and collect: the same compaction over accepted decodes.
//
// Replaces index_kernel (index_kernel.cuh:7-76), where thread 0 of a single
// 64-thread block walks every flag serially.  Here one 1024-thread workgroup per channel turns each
// 64-item group into a __ballot mask, prefix-sums the 16 per-wave popcounts through LDS and writes
// idx[ch][0..n) in ascending item order - the same list, built by wave-wide ballots.
#include "msk144_kernels.h"
#include "wave64.h"

#include "../../include/msk144hip.h"

namespace msk144
{

namespace
{

constexpr int kIdxThreads = 1024;
constexpr int kIdxWaves = kIdxThreads / 64;

// Ordered compaction helper: returns this thread's output slot (or -1) for `flag`, advancing `base`
// (workgroup-uniform running count).  All threads of the workgroup must call it.
__device__ __forceinline__ int ordered_slot(bool flag, int& base, int* s_wave_count)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if(lane == 0) s_wave_count[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for(int w = 0; w < kIdxWaves; w++)
    {
        const int c = s_wave_count[w];
        if(w < wave) before += c;
        total += c;
    }
    const int slot = flag ? base + before + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    base += total;
    __syncthreads();
    return slot;
}

// index_kernel.cuh:30-50: a candidate goes to the LDPC when its sync words disagree in at most `threshold` bits.  A negative
// value is not a count: softbits_kernel<true> stores -1 - s for a candidate whose folded frames are those of slot s < its own
// of the same (frequency, pattern) group (positions congruent modulo the ring, or modulo the period of masks 111111 / 100100) -
// that slot is decoded, this one takes its result in the collect stage (source_item below).
__device__ __forceinline__ bool passes_gate(int32_t nbad, int32_t threshold)
{
    return nbad >= 0 && nbad <= threshold;
}

static_assert((kSlotsPerPattern & (kSlotsPerPattern - 1)) == 0, "a group's first item is k with the slot bits cleared");

// the item whose nbadsync and decode item k reports: itself, or the lower slot of its group it was handed to
__device__ __forceinline__ int source_item(const int32_t* __restrict__ nbad, int k)
{
    const int nb = nbad[k];
    return nb < 0 ? (k & ~(kSlotsPerPattern - 1)) + (-1 - nb) : k;
}

__global__ __launch_bounds__(kIdxThreads) void index_kernel(const DeviceStore st)
{
    __shared__ int s_wave_count[kIdxWaves];
    const int ch = st.ch0 + blockIdx.x;
    const size_t off = static_cast<size_t>(ch) * st.K;
    const int32_t* __restrict__ nbad = st.nbadsync + off;
    int32_t* __restrict__ out = st.idx + off;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // Every wave owns one contiguous range of items (a multiple of 64), so the ascending order of the list is wave order, then
    // group order inside the wave: pass 1 counts the range, ONE barrier publishes the sixteen counts, pass 2 re-reads the flags
    // (L2 hits) and writes.  The earlier form walked the channel 1024 items at a time with two barriers per step - 48 barriers for a
    // deep window, 19 us per launch; this one is a chain of two load latencies.
    const int per_wave = ((st.K + kIdxThreads - 1) / kIdxThreads) * 64;
    const int k_begin = wave * per_wave;
    const int k_end = k_begin + per_wave < st.K ? k_begin + per_wave : st.K;
    constexpr int kBatch = 8;  // loads in flight per lane
    int count = 0;
    for(int k0 = k_begin; k0 < k_end; k0 += kBatch * 64)
    {
        int32_t nb[kBatch];
#pragma unroll
        for(int j = 0; j < kBatch; j++)
        {
            const int k = k0 + j * 64 + lane;
            nb[j] = k < k_end ? nbad[k] : 0x7fffffff;
        }
#pragma unroll
        for(int j = 0; j < kBatch; j++)
        {
            const int k = k0 + j * 64 + lane;
            if(k < k_end) st.dec_flag[off + k] = 0;  // clear_result (result_keeper.cuh:61-73) for the fields LDPC may set
            count += __popcll(__ballot(passes_gate(nb[j], st.nbadsync_threshold)));
        }
    }
    if(lane == 0) s_wave_count[wave] = count;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for(int w = 0; w < kIdxWaves; w++)
    {
        const int c = s_wave_count[w];
        if(w < wave) base += c;
        total += c;
    }
    for(int k0 = k_begin; k0 < k_end; k0 += kBatch * 64)
    {
        int32_t nb[kBatch];
#pragma unroll
        for(int j = 0; j < kBatch; j++)
        {
            const int k = k0 + j * 64 + lane;
            nb[j] = k < k_end ? nbad[k] : 0x7fffffff;
        }
#pragma unroll
        for(int j = 0; j < kBatch; j++)
        {
            const int k = k0 + j * 64 + lane;
            const bool flag = passes_gate(nb[j], st.nbadsync_threshold);
            const unsigned long long m = __ballot(flag);
            if(flag) out[base + __popcll(m & ((1ull << lane) - 1ull))] = k;
            base += __popcll(m);
        }
    }
    if(threadIdx.x == 0) st.n_idx[ch] = total;
}

// ---- collect: accepted decodes -> compact msk144_result records, ordered by (channel, item) ----
__global__ __launch_bounds__(kIdxThreads) void collect_count_kernel(const DeviceStore st)
{
    __shared__ int s_wave_count[kIdxWaves];
    __shared__ int s_wave_copies[kIdxWaves];
    const int ch = blockIdx.x;
    const size_t off = static_cast<size_t>(ch) * st.K;
    const int32_t* __restrict__ nbad = st.nbadsync + off;
    int cnt = 0, copies = 0;  // accepted decodes; slots that were handed to a lower slot (nbadsync < 0)
    for(int k = threadIdx.x; k < st.K; k += kIdxThreads)
    {
        cnt += st.dec_flag[off + source_item(nbad, k)] ? 1 : 0;
        copies += nbad[k] < 0 ? 1 : 0;
    }
    // wave reduce
    for(int d = 32; d > 0; d >>= 1)
    {
        cnt += __shfl_down(cnt, d);
        copies += __shfl_down(copies, d);
    }
    if((threadIdx.x & 63) == 0)
    {
        s_wave_count[threadIdx.x >> 6] = cnt;
        s_wave_copies[threadIdx.x >> 6] = copies;
    }
    __syncthreads();
    if(threadIdx.x == 0)
    {
        int total = 0, total_copies = 0;
        for(int w = 0; w < kIdxWaves; w++)
        {
            total += s_wave_count[w];
            total_copies += s_wave_copies[w];
        }
        st.dec_count[ch] = total;
        st.copy_count[ch] = total_copies;
    }
}

__global__ __launch_bounds__(kIdxThreads) void collect_scatter_kernel(const DeviceStore st)
{
    __shared__ int s_wave_count[kIdxWaves];
    __shared__ int s_base;
    const int ch = blockIdx.x;
    const size_t off = static_cast<size_t>(ch) * st.K;

    // records of all lower channels come first
    int before = 0;
    for(int c = threadIdx.x; c < ch; c += kIdxThreads) before += st.dec_count[c];
    for(int d = 32; d > 0; d >>= 1) before += __shfl_down(before, d);
    if((threadIdx.x & 63) == 0) s_wave_count[threadIdx.x >> 6] = before;
    __syncthreads();
    if(threadIdx.x == 0)
    {
        int total = 0;
        for(int w = 0; w < kIdxWaves; w++) total += s_wave_count[w];
        s_base = total;
        if(ch == st.channels - 1) *st.result_count = total + st.dec_count[ch];
    }
    __syncthreads();
    int base = s_base;
    __syncthreads();

    msk144_result* __restrict__ out = static_cast<msk144_result*>(st.results);
    const int32_t* __restrict__ nbad = st.nbadsync + off;
    const int per_freq = st.D * kSlotsPerPattern;
    for(int k0 = 0; k0 < st.K; k0 += kIdxThreads)
    {
        const int k = k0 + threadIdx.x;
        const int src = k < st.K ? source_item(nbad, k) : 0;  // where this slot's nbadsync and decode come from (itself, but for handed-over copies)
        const bool flag = (k < st.K) && st.dec_flag[off + src];
        const int slot = ordered_slot(flag, base, s_wave_count);
        if(slot >= 0 && slot < st.max_results)
        {
            const int b = k / per_freq;
            const int p = (k - b * per_freq) / kSlotsPerPattern;
            msk144_result r;
            r.channel = st.channel_base + ch;
            r.item = k;
            r.f0 = st.freq[b];
            r.pattern_idx = p;
            r.num_avg = kPatternNumAvg[p];
            r.pos = st.pos[off + k];
            r.xb = st.xb[off + k];
            r.nbadsync = nbad[src];
            r.ldpc_iterations = st.dec_iter[off + src];
            r.ldpc_hard_errors = st.dec_nhard[off + src];
            const uint32_t* w = st.dec_msg + (off + src) * 3;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            r.message[0] = w0 >> 24; r.message[1] = w0 >> 16; r.message[2] = w0 >> 8; r.message[3] = w0;
            r.message[4] = w1 >> 24; r.message[5] = w1 >> 16; r.message[6] = w1 >> 8; r.message[7] = w1;
            r.message[8] = w2 >> 24; r.message[9] = w2 >> 16;
            r.reserved[0] = 0;
            r.reserved[1] = 0;
            out[slot] = r;
        }
    }
}

}  // namespace

void launch_index(const DeviceStore& st, hipStream_t stream)
{
    hipLaunchKernelGGL(index_kernel, dim3(st.nch), dim3(kIdxThreads), 0, stream, st);
}

void launch_collect(const DeviceStore& st, hipStream_t stream)
{
    hipLaunchKernelGGL(collect_count_kernel, dim3(st.channels), dim3(kIdxThreads), 0, stream, st);
    hipLaunchKernelGGL(collect_scatter_kernel, dim3(st.channels), dim3(kIdxThreads), 0, stream, st);
}

}  // namespace msk144
