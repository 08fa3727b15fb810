// Developer microbenchmark: issue cost of plain vs packed FP32 FMA, exp2/rcp, DPP add on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template<int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    for(int i = 0; i < iters; i++)
    {
        if(MODE == 0)
        {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
        else if(MODE == 1)
        {
            float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        }
        else if(MODE == 2)
        {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        else if(MODE == 3)
        {
            asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template<int MODE>
void run(const char* name, int blocks_per_cu)
{
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    float* d;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 8 instructions per iteration per wave; waves per SIMD = blocks_per_cu (256 threads = 4 waves = 1 per SIMD)
    const double instr_per_simd = 8.0 * iters * blocks_per_cu;
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}

int main()
{
    for(int w : {1, 2, 4, 8})
    {
        if(w == 1) { run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<2>("exp/rcp", 1); run<3>("add_dpp", 1); }
        if(w == 2) { run<0>("v_fma_f32", 2); run<1>("v_pk_fma_f32", 2); run<2>("exp/rcp", 2); run<3>("add_dpp", 2); }
        if(w == 4) { run<0>("v_fma_f32", 4); run<1>("v_pk_fma_f32", 4); run<2>("exp/rcp", 4); run<3>("add_dpp", 4); }
        if(w == 8) { run<0>("v_fma_f32", 8); run<1>("v_pk_fma_f32", 8); run<2>("exp/rcp", 8); run<3>("add_dpp", 8); }
    }
    return 0;
}
