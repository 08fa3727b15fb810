// Developer microbenchmark: how many workgroups of a given shape does a CU really hold at once?  Every wave of the probe
// kernel sleeps for a fixed wall-clock time T, so a launch of G workgroups takes ceil(G / (256 * resident)) * T and the
// resident count falls out of the launch time.  Run with the scan / softbits shapes and their neighbours.
#include <hip/hip_runtime.h>

#include <cstdio>

template<int kThreads, int kLdsBytes>
__global__ __launch_bounds__(kThreads) void sleep_kernel(float* out, int n, long long ticks)
{
    __shared__ char lds[kLdsBytes > 0 ? kLdsBytes : 4];
    if(threadIdx.x == 0) lds[0] = static_cast<char>(blockIdx.x);
    const long long t0 = wall_clock64();
    while(wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if(threadIdx.x == 63 && blockIdx.x == static_cast<unsigned>(n)) out[0] = lds[0];
}

template<int kThreads, int kLdsBytes>
void run(const char* what)
{
    float* d;
    hipMalloc(&d, 4);
    const int grid = 256 * 24;
    const double t_us = 50.0;
    const long long ticks = static_cast<long long>(t_us * 100.0);  // wall_clock64 runs at 100 MHz
    hipLaunchKernelGGL((sleep_kernel<kThreads, kLdsBytes>), dim3(grid), dim3(kThreads), 0, 0, d, -1, ticks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((sleep_kernel<kThreads, kLdsBytes>), dim3(grid), dim3(kThreads), 0, 0, d, -1, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %4d threads (%2d waves), %6d B LDS: %7.3f ms -> %.2f workgroups resident per CU (%.1f waves per SIMD)\n", what, kThreads, kThreads / 64, kLdsBytes, ms,
           grid * t_us * 1e-3 / (ms * 256.0), grid * t_us * 1e-3 / (ms * 256.0) * (kThreads / 64) / 4.0);
    hipFree(d);
}

int main()
{
    run<576, 44544>("scan shape");
    run<512, 41536>("softbits shape");
    run<576, 0>("9 waves, no LDS");
    run<512, 44544>("8 waves, scan LDS");
    run<640, 44544>("10 waves, scan LDS");
    run<768, 44544>("12 waves, scan LDS");
    run<896, 44544>("14 waves, scan LDS");
    run<1024, 44544>("16 waves, scan LDS");
    run<320, 0>("5 waves, no LDS");
    run<256, 7168>("ldpc shape");
    return 0;
}
