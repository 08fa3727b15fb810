// Front ends: raw samples -> analytic (complex) window in HBM, one workgroup per channel.
//
// Replaces, per channel and fused into one kernel each:
//   audio, method 2: host rms + int16->complex (main.cu:301-307,323), H2D, apply_shift_filter_shift
//                    <<<1,32>>> (analytic2.cuh:235-258), D2H for the SNR tracker (main.cu:327)
//   IQ:              host int8->complex (main.cu:365-371), apply_filter<<<1,32>>> (analytic2.cuh:260-281)
//   audio, method 1: Analytic::execute (analytic_fft.cu:84-157): scale, cuFFT forward, D2H, host mask,
//                    H2D, cuFFT inverse, D2H, H2D  ->  one 8192-point radix-2 FFT pair in LDS
// plus the 8 segment powers SNRTracker::process_data needs (snr_tracker.cu:21-37), so the analytic
// window never has to travel back to the host.
//
// This file is compiled with -ffp-contract=off: the FIR path then performs exactly the float
// operations of the reference expressions (mul, then add), and the rms / segment-power sums are
// accumulated sequentially by one lane in the reference's host order, which makes the method-2 and
// IQ outputs bit-identical to the CPU oracle.  Cost: ~15 us of one lane per channel, hidden behind
// the other channels' workgroups.
#include "msk144_kernels.h"
#include "wave64.h"

namespace msk144
{

namespace
{

constexpr int kFeThreads = 256;

__device__ __forceinline__ float2 cmulf(float2 x, float2 y)
{
    return make_float2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
}

// 8 segment powers in the reference's accumulation order (snr_tracker.cu:23-31)
__device__ __forceinline__ void segment_power(const float2* __restrict__ out, float* __restrict__ seg)
{
    if(threadIdx.x < 8)
    {
        constexpr int block_size = kWindowSamples / 8;
        float acc = 0.0f;
        const float2* p = out + threadIdx.x * block_size;
        for(int i = 0; i < block_size; i++)
        {
            const float2 d = p[i];
            acc = acc + (d.x * d.x + d.y * d.y);
        }
        seg[threadIdx.x] = acc;
    }
}

__device__ __forceinline__ float audio_rms_factor(const float* s_raw, float* s_fac)
{
    // sequential float accumulation, thrust::reduce on a host_vector (main.cu:301-307)
    if(threadIdx.x == 0)
    {
        float acc = 0.0f;
        for(int i = 0; i < kWindowSamples; i++)
        {
            const float b = s_raw[i];
            acc = acc + b * b;
        }
        const float rms = f32_sqrt(f32_div(acc, static_cast<float>(kWindowSamples)));
        *s_fac = f32_div(1.0f, rms);
    }
    __syncthreads();
    return *s_fac;
}

// ---- shift - FIR - FIR - shift (audio) / FIR - FIR (IQ) ----
template<bool kAudio>
__global__ __launch_bounds__(kFeThreads) void frontend_fir_kernel(const DeviceStore st, const void* __restrict__ d_in)
{
    __shared__ float2 s_a[kFirBuffer];
    __shared__ float2 s_b[kFirBuffer];
    __shared__ float s_fac;

    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    float2* __restrict__ out = st.analytic + static_cast<size_t>(ch) * kWindowSamples;

    if(kAudio)
    {
        const int16_t* in = static_cast<const int16_t*>(d_in) + static_cast<size_t>(ch) * kWindowSamples;
        float* s_raw = reinterpret_cast<float*>(s_b);
        for(int n = tid; n < kWindowSamples; n += kFeThreads) s_raw[n] = static_cast<float>(in[n]);
        __syncthreads();
        const float fac = audio_rms_factor(s_raw, &s_fac);
        // a[n] = (fac*x, 0), then c[i] *= w_L[i&7]  (analytic2.cuh:15-48)
        const float2 w_left[8] = {{kSin45, -kSin45}, {0.0f, -1.0f}, {-kSin45, -kSin45}, {-1.0f, 0.0f},
                                  {-kSin45, kSin45}, {0.0f, 1.0f},  {kSin45, kSin45},   {1.0f, 0.0f}};
        for(int i = tid; i < kFirBuffer; i += kFeThreads)
        {
            float2 c = make_float2(0.0f, 0.0f);
            const int n = i - kFirPad;
            if(n >= 0 && n < kWindowSamples) c = make_float2(fac * s_raw[n], 0.0f);
            s_a[i] = cmulf(c, w_left[i & 7]);
        }
    }
    else
    {
        const int8_t* in = static_cast<const int8_t*>(d_in) + static_cast<size_t>(ch) * kWindowSamples * 2;
        for(int i = tid; i < kFirBuffer; i += kFeThreads)
        {
            float2 c = make_float2(0.0f, 0.0f);
            const int n = i - kFirPad;
            if(n >= 0 && n < kWindowSamples)
            {
                const float divider = 128.0f;
                c = make_float2(static_cast<float>(in[2 * n]) / divider, static_cast<float>(in[2 * n + 1]) / divider);
            }
            s_a[i] = c;
        }
    }
    __syncthreads();

    // forward pass, y[i] = sum_k h_k c[i+16-k]  (analytic2.cuh:163-187); outputs 17..5215 are consumed
    for(int i = tid; i < kFirBuffer - 32; i += kFeThreads)
    {
        float2 s = make_float2(0.0f, 0.0f);
#pragma unroll
        for(int t = 0; t < kFirTaps; t++)
        {
            const float h = kFirTapValue[t];
            const float2 c = s_a[i + (16 - kFirTapIndex[t])];
            s.x = s.x + h * c.x;
            s.y = s.y + h * c.y;
        }
        s_b[i] = s;
    }
    __syncthreads();

    // reverse pass, z[i] = sum_k h_k y[i-(16-k)]  (analytic2.cuh:195-219), shift back, store
    const float2 w_right[8] = {{1.0f, 0.0f},  {kSin45, kSin45},   {0.0f, 1.0f},  {-kSin45, kSin45},
                               {-1.0f, 0.0f}, {-kSin45, -kSin45}, {0.0f, -1.0f}, {kSin45, -kSin45}};
    for(int n = tid; n < kWindowSamples; n += kFeThreads)
    {
        const int i = n + kFirPad;
        float2 s = make_float2(0.0f, 0.0f);
#pragma unroll
        for(int t = 0; t < kFirTaps; t++)
        {
            const float h = kFirTapValue[t];
            const float2 c = s_b[i - (16 - kFirTapIndex[t])];
            s.x = s.x + h * c.x;
            s.y = s.y + h * c.y;
        }
        if(kAudio) s = cmulf(s, w_right[i & 7]);
        out[n] = s;
        s_a[n] = s;
    }
    __syncthreads();
    segment_power(s_a, st.seg_power + ch * 8);
}

// ---- FFT analytic signal (analytic_fft.cu) ----
constexpr int kFftLog2 = 13;

__device__ __forceinline__ int brev13(int i)
{
    return static_cast<int>(__brev(static_cast<unsigned>(i)) >> (32 - kFftLog2));
}

// in-place radix-2 DIT over s[8192] (input in bit-reversed order); tw[j] = exp(-2 pi i j / 8192)
template<bool kInverse>
__device__ __forceinline__ void fft8192(float2* s, const float2* __restrict__ tw)
{
    for(int stage = 1; stage <= kFftLog2; stage++)
    {
        const int half = 1 << (stage - 1);
        const int tw_step = kFftSize >> stage;
        for(int t = threadIdx.x; t < kFftSize / 2; t += kFeThreads)
        {
            const int k = t & (half - 1);
            const int i = ((t >> (stage - 1)) << stage) + k;
            float2 w = tw[k * tw_step];
            if(kInverse) w.y = -w.y;
            const float2 u = s[i];
            const float2 v = cmulf(s[i + half], w);
            s[i] = make_float2(u.x + v.x, u.y + v.y);
            s[i + half] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kFeThreads) void frontend_fft_kernel(const DeviceStore st, const int16_t* __restrict__ d_in,
                                                                  const float2* __restrict__ tw, const float* __restrict__ mask)
{
    __shared__ float2 s[kFftSize];
    __shared__ float s_fac;

    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    const int16_t* in = d_in + static_cast<size_t>(ch) * kWindowSamples;
    float2* __restrict__ out = st.analytic + static_cast<size_t>(ch) * kWindowSamples;

    float* s_raw = reinterpret_cast<float*>(s) + kFftSize;  // upper half of the buffer as scratch
    for(int n = tid; n < kWindowSamples; n += kFeThreads) s_raw[n] = static_cast<float>(in[n]);
    __syncthreads();
    const float fac = audio_rms_factor(s_raw, &s_fac);
    const float fac2 = 2.0f / kFftSize;  // analytic_fft.cu:88

    // scaled, zero-padded input in bit-reversed order.  s_raw aliases s: stage through registers.
    float v[kFftSize / kFeThreads];
#pragma unroll
    for(int j = 0; j < kFftSize / kFeThreads; j++)
    {
        const int n = tid + j * kFeThreads;
        v[j] = (n < kWindowSamples) ? fac2 * (fac * s_raw[n]) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for(int j = 0; j < kFftSize / kFeThreads; j++)
    {
        const int n = tid + j * kFeThreads;
        s[brev13(n)] = make_float2(v[j], (n < kWindowSamples) ? fac2 * 0.0f : 0.0f);
    }
    __syncthreads();

    fft8192<false>(s, tw);

    // spectral mask (analytic_fft.cu:118-127), then bit-reverse in place for the inverse transform
    for(int i = tid; i < kFftSize; i += kFeThreads)
    {
        float2 x = s[i];
        if(i < kFftSize / 2)
        {
            const float h = mask[i];
            x = make_float2(x.x * h, x.y * h);
            if(i == 0) x = make_float2(x.x * 0.5f, x.y * 0.5f);
        }
        else
        {
            x = make_float2(0.0f, 0.0f);
        }
        s[i] = x;
    }
    __syncthreads();
    for(int i = tid; i < kFftSize; i += kFeThreads)
    {
        const int r = brev13(i);
        if(i < r)
        {
            const float2 a = s[i];
            s[i] = s[r];
            s[r] = a;
        }
    }
    __syncthreads();

    fft8192<true>(s, tw);

    for(int n = tid; n < kWindowSamples; n += kFeThreads) out[n] = s[n];
    segment_power(s, st.seg_power + ch * 8);
}

}  // namespace

void launch_frontend_audio(const DeviceStore& st, const int16_t* d_in, int analytic_method, const float2* d_twiddle, const float* d_fft_mask,
                           hipStream_t stream)
{
    if(analytic_method == 1)
        hipLaunchKernelGGL(frontend_fft_kernel, dim3(st.channels), dim3(kFeThreads), 0, stream, st, d_in, d_twiddle, d_fft_mask);
    else
        hipLaunchKernelGGL(frontend_fir_kernel<true>, dim3(st.channels), dim3(kFeThreads), 0, stream, st, static_cast<const void*>(d_in));
}

void launch_frontend_iq(const DeviceStore& st, const int8_t* d_in, hipStream_t stream)
{
    hipLaunchKernelGGL(frontend_fir_kernel<false>, dim3(st.channels), dim3(kFeThreads), 0, stream, st, static_cast<const void*>(d_in));
}

}  // namespace msk144
