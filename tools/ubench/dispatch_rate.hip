// Developer microbenchmark: how fast can the chip START workgroups?  An (almost) empty kernel is launched with the grid shapes
// of the scan and softbits kernels (513 024 workgroups of 9 / 8 waves holding 44.5 / 41.5 KB of LDS) and with smaller
// workgroups, LDS-free workgroups, etc.  If the empty kernel takes a sizeable part of the real kernel's time, the real
// kernel is bound by workgroup dispatch, not by anything it computes.
#include <hip/hip_runtime.h>

#include <cstdio>

template<int kThreads, int kLdsBytes>
__global__ __launch_bounds__(kThreads) void empty_kernel(float* out, int n)
{
    __shared__ char lds[kLdsBytes > 0 ? kLdsBytes : 4];
    if(threadIdx.x == 0) lds[0] = static_cast<char>(blockIdx.x);
    __syncthreads();
    if(threadIdx.x == 63 && blockIdx.x == static_cast<unsigned>(n)) out[0] = lds[0];
}

template<int kThreads, int kLdsBytes>
void run(const char* what, int grid)
{
    float* d;
    hipMalloc(&d, 4);
    for(int i = 0; i < 3; i++) hipLaunchKernelGGL((empty_kernel<kThreads, kLdsBytes>), dim3(grid), dim3(kThreads), 0, 0, d, -1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for(int i = 0; i < reps; i++) hipLaunchKernelGGL((empty_kernel<kThreads, kLdsBytes>), dim3(grid), dim3(kThreads), 0, 0, d, -1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double waves = static_cast<double>(grid) * (kThreads / 64);
    printf("%-44s grid %7d x %4d threads, %6d B LDS: %8.3f ms = %6.1f ns per workgroup chip-wide, %5.2f G waves/s\n", what, grid, kThreads, kLdsBytes, ms, ms * 1e6 / grid,
           waves / (ms * 1e-3) / 1e9);
    hipFree(d);
}

int main()
{
    const int tiles = 1024 * 501;
    run<576, 44544>("scan shape (9 waves, 44.5 KB)", tiles);
    run<512, 41536>("softbits shape (8 waves, 41.5 KB)", tiles);
    run<576, 0>("9 waves, no LDS", tiles);
    run<512, 0>("8 waves, no LDS", tiles);
    run<256, 0>("4 waves, no LDS", tiles);
    run<64, 0>("1 wave, no LDS", tiles);
    run<256, 7168>("ldpc shape (4 waves, 7 KB), 8192 workgroups", 8192);
    run<1024, 83072>("16 waves, 81 KB", tiles / 2);
    return 0;
}
