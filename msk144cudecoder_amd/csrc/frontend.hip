// Front ends: raw samples -> analytic (complex) window in HBM, one workgroup per channel.
//
// Replaces, per channel and fused into one kernel each:
//   audio, method 2: host rms + int16->complex (main.cu:301-307,323), H2D, apply_shift_filter_shift
//                    <<<1,32>>> (analytic2.cuh:235-258), D2H for the SNR tracker (main.cu:327)
//   IQ:              host int8->complex (main.cu:365-371), apply_filter<<<1,32>>> (analytic2.cuh:260-281)
//   audio, method 1: Analytic::execute (analytic_fft.cu:84-157): scale, cuFFT forward, D2H, host mask,
//                    H2D, cuFFT inverse, D2H, H2D  ->  one 8192-point transform pair in LDS, 16 x 16 x 32 mixed radix with
//                    register-resident butterflies and the spectral mask applied between them (frontend_fft_kernel below)
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

// 8 segment powers in the reference's accumulation order (snr_tracker.cu:23-31): acc = acc + norm(d), sample after sample.  The
// norms (d.x * d.x + d.y * d.y, the same two products and one sum) are formed by the threads that produce the samples; what stays
// sequential is the chain of 648 additions per segment, eight lanes side by side.  pow[i * stride] = norm of sample i.
__device__ __forceinline__ void segment_power(const float* pow, int stride, float* __restrict__ seg)
{
    if(threadIdx.x < 8)
    {
        constexpr int block_size = kWindowSamples / 8;
        float acc = 0.0f;
        const float* p = pow + threadIdx.x * block_size * stride;
        for(int i = 0; i < block_size; i++) acc = acc + p[i * stride];
        seg[threadIdx.x] = acc;
    }
}

// 1 / rms of the raw window from its squares s_sq[n] = x[n] * x[n]: sequential float accumulation in sample order, as thrust::reduce
// on a host_vector does (main.cu:301-307).  The products are formed in parallel by the caller - the same float each - so the one
// lane's chain is 5184 dependent additions instead of 5184 multiply-add pairs.
__device__ __forceinline__ float audio_rms_factor(const float* s_sq, float* s_fac)
{
    if(threadIdx.x == 0)
    {
        // four squares per ds_read_b128, 36 reads in flight on one address register: the lane's instruction stream is the 5184
        // dependent additions and little else - 41 instructions per group of 144 samples besides the additions, and one exposed LDS
        // latency per group.  (As scalar reads the loop carried an address move and a load per two additions: 12.6 issue slots per
        // sample for a wave that runs alone while the others wait at the barrier.  Software-pipelining the groups was tried: the
        // register allocator pays for it with a copy per read.)
        constexpr int kGroup = 36;
        static_assert((kWindowSamples / 4) % kGroup == 0, "whole groups");
        const float4* q = reinterpret_cast<const float4*>(s_sq);
        float acc = 0.0f;
        for(int i = 0; i < kWindowSamples / 4; i += kGroup)
        {
            float4 v[kGroup];
#pragma unroll
            for(int j = 0; j < kGroup; j++) v[j] = q[i + j];
#pragma unroll
            for(int j = 0; j < kGroup; j++)
            {
                acc = acc + v[j].x;
                acc = acc + v[j].y;
                acc = acc + v[j].z;
                acc = acc + v[j].w;
            }
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
    __shared__ __align__(16) float2 s_a[kFirBuffer];
    __shared__ __align__(16) float2 s_b[kFirBuffer];
    __shared__ float s_fac;

    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    float2* __restrict__ out = st.analytic + static_cast<size_t>(ch) * kWindowSamples;

    if(kAudio)
    {
        const int16_t* in = static_cast<const int16_t*>(d_in) + static_cast<size_t>(ch) * kWindowSamples;
        float* s_raw = reinterpret_cast<float*>(s_b);
        float* s_sq = s_raw + kWindowSamples;
        static_assert(2 * kWindowSamples <= 2 * kFirBuffer && (kWindowSamples * sizeof(float)) % 16 == 0, "raw samples and their squares fit the second FIR buffer, squares 16-byte aligned");
        for(int n = tid; n < kWindowSamples; n += kFeThreads)
        {
            const float b = static_cast<float>(in[n]);
            s_raw[n] = b;
            s_sq[n] = b * b;
        }
        __syncthreads();
        const float fac = audio_rms_factor(s_sq, &s_fac);
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
    float* s_pow = reinterpret_cast<float*>(s_a);
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
        s_pow[n] = s.x * s.x + s.y * s.y;  // s_a is no longer read: the norms of the output for the segment sums
    }
    __syncthreads();
    segment_power(s_pow, 1, st.seg_power + ch * 8);
}

// ---- FFT analytic signal (analytic_fft.cu:84-157) ----
//
// The reference pads the 5184 scaled samples to 8192, runs cuFFT forward, masks the spectrum ON THE HOST (band weight, DC halved,
// negative frequencies zeroed), runs cuFFT inverse and copies the result back: five PCIe crossings per hop.  Here one workgroup per
// channel does both transforms in LDS as a mixed-radix 16 x 16 x 32 decomposition with register-resident butterflies:
//
//   n = 512 n1 + 32 n2 + n3,   k = k1 + 16 k2 + 256 k3,   W = exp(-2 pi i / 8192):
//   W^(nk) = W16^(n1 k1) . W^((32 n2 + n3) k1) . W16^(n2 k2) . W512^(n3 k2) . W32^(n3 k3)
//
// forward (decimation in frequency): radix-16 over n1, twiddle, radix-16 over n2, twiddle, radix-32 over n3 - bin k then sits at
// position 512 k1 + 32 k2 + k3, so no bit-reversal pass is needed: the mask is applied where the bins lie (k >= 4096, the
// negative frequencies the reference zeroes, is exactly k3 >= 16), and the inverse transform walks the same factorisation
// backwards (radix-32 over k3, radix-16 over k2, radix-16 over k1) and ends in natural order.  The thread that finishes the forward
// transform of a (k1, k2) column is the one that starts its inverse: mask and both radix-32 butterflies stay in registers.  Five
// workgroup barriers for the two transforms (the radix-2 form this replaces: 26 + two reordering passes), every LDS access
// conflict-free (one pad cell per 32: the radix-32 columns are 33 cells apart), twiddles from the 4096-entry table.
constexpr int kFftCells = kFftSize + kFftSize / 32;  // padded: cell(i) = i + i / 32

__device__ __forceinline__ int fft_cell(int i)
{
    return i + (i >> 5);
}

// cos(k pi / 16), k = 0..16: after full unrolling k is a constant and the switch folds to a literal
__device__ __forceinline__ float cos_pi16(int k)
{
    switch(k)
    {
    case 0: return 1.0f;
    case 1: return 0.980785280403230449f;
    case 2: return 0.923879532511286756f;
    case 3: return 0.831469612302545237f;
    case 4: return 0.707106781186547524f;
    case 5: return 0.555570233019602225f;
    case 6: return 0.382683432365089772f;
    case 7: return 0.195090322016128268f;
    case 8: return 0.0f;
    case 9: return -0.195090322016128268f;
    case 10: return -0.382683432365089772f;
    case 11: return -0.555570233019602225f;
    case 12: return -0.707106781186547524f;
    case 13: return -0.831469612302545237f;
    case 14: return -0.923879532511286756f;
    case 15: return -0.980785280403230449f;
    default: return -1.0f;
    }
}

// x . W32^k (forward) or x . conj(W32^k) (inverse), k = 0..15 a compile-time constant after unrolling
template<bool kInverse>
__device__ __forceinline__ float2 mul_w32(float2 x, int k)
{
    if(k == 0) return x;
    if(k == 8) return kInverse ? make_float2(-x.y, x.x) : make_float2(x.y, -x.x);
    const float c = cos_pi16(k);
    const float sn = cos_pi16(k < 8 ? 8 - k : k - 8);  // sin(k pi / 16) = cos(|8 - k| pi / 16) > 0
    const float s = kInverse ? sn : -sn;
    return make_float2(fmaf(x.x, c, -(x.y * s)), fmaf(x.x, s, x.y * c));
}

__device__ __forceinline__ float2 cmul_fma(float2 x, float2 w)
{
    return make_float2(fmaf(x.x, w.x, -(x.y * w.y)), fmaf(x.x, w.y, x.y * w.x));
}

// W^m (forward) or conj(W^m) (inverse), 0 <= m < 8192, from the table tw[j] = exp(-2 pi i j / 8192), j < 4096
template<bool kInverse>
__device__ __forceinline__ float2 twiddle(const float2* __restrict__ tw, int m)
{
    float2 w = tw[m & (kFftSize / 2 - 1)];
    if(m & (kFftSize / 2)) w = make_float2(-w.x, -w.y);
    if(kInverse) w.y = -w.y;
    return w;
}

// R-point DFT (R = 16 or 32) of a register array, natural order in and out: log2(R) radix-2 decimation-in-frequency stages and
// the bit reversal, all indices compile-time constants
template<int R, bool kInverse>
__device__ __forceinline__ void fft_reg(float2 (&a)[R])
{
    constexpr int kLog = R == 32 ? 5 : 4;
#pragma unroll
    for(int stage = 0; stage < kLog; stage++)
    {
        const int len = R >> stage, half = len >> 1;
#pragma unroll
        for(int b = 0; b < R; b += len)
        {
#pragma unroll
            for(int j = 0; j < half; j++)
            {
                const float2 u = a[b + j], v = a[b + j + half];
                a[b + j] = make_float2(u.x + v.x, u.y + v.y);
                a[b + j + half] = mul_w32<kInverse>(make_float2(u.x - v.x, u.y - v.y), j * (32 / len));
            }
        }
    }
#pragma unroll
    for(int i = 0; i < R; i++)
    {
        int r = 0;
#pragma unroll
        for(int bit = 0; bit < kLog; bit++) r |= ((i >> bit) & 1) << (kLog - 1 - bit);
        if(i < r)
        {
            const float2 t = a[i];
            a[i] = a[r];
            a[r] = t;
        }
    }
}

__global__ __launch_bounds__(kFeThreads) void frontend_fft_kernel(const DeviceStore st, const int16_t* __restrict__ d_in,
                                                                  const float2* __restrict__ tw, const float* __restrict__ mask)
{
    __shared__ __align__(16) float2 s[kFftCells];
    __shared__ float s_fac;

    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    const int16_t* in = d_in + static_cast<size_t>(ch) * kWindowSamples;
    float2* __restrict__ out = st.analytic + static_cast<size_t>(ch) * kWindowSamples;

    // raw samples and their squares side by side in the (not yet used) transform buffer; the rms sum stays the reference's
    // sequential float accumulation (main.cu:301-307), only the squares are formed in parallel
    float* s_raw = reinterpret_cast<float*>(s);
    float* s_sq = s_raw + kWindowSamples;
    for(int n = tid; n < kWindowSamples; n += kFeThreads)
    {
        const float b = static_cast<float>(in[n]);
        s_raw[n] = b;
        s_sq[n] = b * b;
    }
    __syncthreads();
    const float fac = audio_rms_factor(s_sq, &s_fac);
    const float fac2 = 2.0f / kFftSize;  // analytic_fft.cu:88

    // ---- forward, radix 16 over n1 (stride 512).  The raw samples alias the buffer: all of a thread's inputs are in registers
    // before anyone writes (padded tail n >= 5184: zeros; imaginary parts: fac2 * 0) ----
    constexpr int kPerThread = kFftSize / 16 / kFeThreads;  // 2 butterflies per thread in the radix-16 passes
    float v[kPerThread][16];
#pragma unroll
    for(int r = 0; r < kPerThread; r++)
#pragma unroll
        for(int n1 = 0; n1 < 16; n1++)
        {
            const int n = n1 * 512 + tid + r * kFeThreads;
            v[r][n1] = (n < kWindowSamples) ? fac2 * (fac * s_raw[n]) : 0.0f;
        }
    __syncthreads();
#pragma unroll
    for(int r = 0; r < kPerThread; r++)
    {
        const int t = tid + r * kFeThreads;
        float2 a[16];
#pragma unroll
        for(int n1 = 0; n1 < 16; n1++) a[n1] = make_float2(v[r][n1], 0.0f);
        fft_reg<16, false>(a);
        s[fft_cell(t)] = a[0];
#pragma unroll
        for(int k1 = 1; k1 < 16; k1++) s[fft_cell(k1 * 512 + t)] = cmul_fma(a[k1], twiddle<false>(tw, t * k1));
    }
    __syncthreads();

    // ---- forward, radix 16 over n2 (stride 32 inside the block of k1), in place ----
#pragma unroll
    for(int r = 0; r < kPerThread; r++)
    {
        const int b = tid + r * kFeThreads, k1 = b >> 5, n3 = b & 31;
        float2 a[16];
#pragma unroll
        for(int n2 = 0; n2 < 16; n2++) a[n2] = s[fft_cell(k1 * 512 + n2 * 32 + n3)];
        fft_reg<16, false>(a);
#pragma unroll
        for(int k2 = 1; k2 < 16; k2++) a[k2] = cmul_fma(a[k2], twiddle<false>(tw, 16 * n3 * k2));
#pragma unroll
        for(int k2 = 0; k2 < 16; k2++) s[fft_cell(k1 * 512 + k2 * 32 + n3)] = a[k2];
    }
    __syncthreads();

    // ---- forward radix 32 over n3, spectral mask (analytic_fft.cu:115-127), inverse radix 32 over k3: one thread per (k1, k2)
    // column, all in registers ----
    {
        const int k1 = tid >> 4, k2 = tid & 15;
        float2 a[32];
#pragma unroll
        for(int n3 = 0; n3 < 32; n3++) a[n3] = s[tid * 33 + n3];
        fft_reg<32, false>(a);
#pragma unroll
        for(int k3 = 0; k3 < 16; k3++)
        {
            const float h = mask[k1 + 16 * k2 + 256 * k3];
            a[k3] = make_float2(a[k3].x * h, a[k3].y * h);
        }
        if(tid == 0) a[0] = make_float2(a[0].x * 0.5f, a[0].y * 0.5f);  // half DC
#pragma unroll
        for(int k3 = 16; k3 < 32; k3++) a[k3] = make_float2(0.0f, 0.0f);  // negative frequencies
        fft_reg<32, true>(a);
        if(k2 != 0)
        {
#pragma unroll
            for(int n3 = 1; n3 < 32; n3++) a[n3] = cmul_fma(a[n3], twiddle<true>(tw, 16 * n3 * k2));
        }
#pragma unroll
        for(int n3 = 0; n3 < 32; n3++) s[tid * 33 + n3] = a[n3];
    }
    __syncthreads();

    // ---- inverse, radix 16 over k2, in place ----
#pragma unroll
    for(int r = 0; r < kPerThread; r++)
    {
        const int b = tid + r * kFeThreads, k1 = b >> 5, n3 = b & 31;
        float2 a[16];
#pragma unroll
        for(int k2 = 0; k2 < 16; k2++) a[k2] = s[fft_cell(k1 * 512 + k2 * 32 + n3)];
        fft_reg<16, true>(a);
        if(k1 != 0)
        {
#pragma unroll
            for(int n2 = 0; n2 < 16; n2++) a[n2] = cmul_fma(a[n2], twiddle<true>(tw, (n2 * 32 + n3) * k1));
        }
#pragma unroll
        for(int n2 = 0; n2 < 16; n2++) s[fft_cell(k1 * 512 + n2 * 32 + n3)] = a[n2];
    }
    __syncthreads();

    // ---- inverse, radix 16 over k1: natural order out.  The first 5184 samples go to HBM; their powers replace them in the
    // buffer (a thread overwrites only cells it has read itself) for the sequential segment sums ----
    float* s_pow = reinterpret_cast<float*>(s);
#pragma unroll
    for(int r = 0; r < kPerThread; r++)
    {
        const int t = tid + r * kFeThreads;
        float2 a[16];
#pragma unroll
        for(int k1 = 0; k1 < 16; k1++) a[k1] = s[fft_cell(k1 * 512 + t)];
        fft_reg<16, true>(a);
#pragma unroll
        for(int n1 = 0; n1 < 16; n1++)
        {
            const int n = n1 * 512 + t;
            if(n < kWindowSamples)
            {
                out[n] = a[n1];
                s_pow[2 * fft_cell(n)] = a[n1].x * a[n1].x + a[n1].y * a[n1].y;
            }
        }
    }
    __syncthreads();
    // the pad cells sit between multiples of 32 samples; a segment is 648 samples, so the chain walks the padded cells itself
    if(tid < 8)
    {
        constexpr int block_size = kWindowSamples / 8;
        float acc = 0.0f;
        for(int i = 0; i < block_size; i++) acc = acc + s_pow[2 * fft_cell(tid * block_size + i)];
        st.seg_power[ch * 8 + tid] = acc;
    }
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
