// Developer microbenchmark: does the LDS skip a 32-lane group whose lanes are all EXEC-masked?  ds_read_b32 / ds_write_b32 /
// the M0-relative add-TID forms are issued in a loop by N active lanes (N = 64, 38, 32, 6) from enough waves to saturate the
// LDS pipe; cycles per wave-instruction per CU tell whether 38 active lanes cost two array cycles and 32 only one.
#include <hip/hip_runtime.h>

#include <cstdio>

enum Op
{
    READ_B32,
    WRITE_B32,
    READ_ADDTID,
    WRITE_ADDTID
};

template<int OP>
__global__ __launch_bounds__(256) void k(float* out, int active, int iters)
{
    __shared__ float buf[4][2048];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for(int i = lane; i < 2048; i += 64) buf[wave][i] = float(i);
    __syncthreads();
    float acc = 0.0f;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)buf[wave]);
    const unsigned addr = base + 4u * lane;
    if(lane < active)
    {
        for(int it = 0; it < iters; it++)
        {
            float v0, v1, v2, v3, v4, v5, v6, v7;
            if(OP == READ_B32)
            {
                asm volatile("ds_read_b32 %0, %8 offset:0\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                             "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                             : "v"(addr)
                             : "memory");
                acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
            }
            if(OP == READ_ADDTID)
            {
                asm volatile("s_mov_b32 m0, %8\n s_nop 0\n"
                             "ds_read_addtid_b32 %0 offset:0\n ds_read_addtid_b32 %1 offset:256\n ds_read_addtid_b32 %2 offset:512\n ds_read_addtid_b32 %3 offset:768\n"
                             "ds_read_addtid_b32 %4 offset:1024\n ds_read_addtid_b32 %5 offset:1280\n ds_read_addtid_b32 %6 offset:1536\n ds_read_addtid_b32 %7 offset:1792\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                             : "s"(base)
                             : "memory");
                acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
            }
            if(OP == WRITE_B32)
            {
                asm volatile("ds_write_b32 %0, %1 offset:0\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                             "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:1280\n ds_write_b32 %0, %1 offset:1536\n ds_write_b32 %0, %1 offset:1792\n s_waitcnt lgkmcnt(0)"
                             :
                             : "v"(addr), "v"(acc)
                             : "memory");
                acc += 1.0f;
            }
            if(OP == WRITE_ADDTID)
            {
                asm volatile("s_mov_b32 m0, %1\n s_nop 0\n"
                             "ds_write_addtid_b32 %0 offset:0\n ds_write_addtid_b32 %0 offset:256\n ds_write_addtid_b32 %0 offset:512\n ds_write_addtid_b32 %0 offset:768\n"
                             "ds_write_addtid_b32 %0 offset:1024\n ds_write_addtid_b32 %0 offset:1280\n ds_write_addtid_b32 %0 offset:1536\n ds_write_addtid_b32 %0 offset:1792\n s_waitcnt lgkmcnt(0)"
                             :
                             : "v"(acc), "s"(base)
                             : "memory");
                acc += 1.0f;
            }
        }
    }
    if(acc == 12345.678f) out[threadIdx.x] = acc;
}

template<int OP>
void run(const char* name)
{
    float* d;
    (void)hipMalloc(&d, 4096);
    const int iters = 4000;
    const int grid = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    for(int active : {64, 38, 32, 6})
    {
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, active, iters);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, active, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per CU: 32 waves x iters x 8
        const double cyc = ms * 1e-3 * 2.4e9 / (32.0 * iters * 8.0);
        printf("%-22s %2d active lanes: %6.2f cycles per wave-instruction per CU\n", name, active, cyc);
    }
    (void)hipFree(d);
}

int main()
{
    run<READ_B32>("ds_read_b32");
    run<READ_ADDTID>("ds_read_addtid_b32");
    run<WRITE_B32>("ds_write_b32");
    run<WRITE_ADDTID>("ds_write_addtid_b32");
    return 0;
}
