// Accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) after the mixers' own Cody-Waite reduction, against
// double precision - the question behind a cheaper sincos for mix.h (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -o sincos_hw sincos_hw.hip && ./sincos_hw
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const float* phi, float* sn, float* cs, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    constexpr float kRound = 12582912.0f;
    const float p = phi[i];
    const float kb = fmaf(p, 0.15915494309189535f, kRound);  // 1 / (2 pi)
    const float kk = kb - kRound;
    float r = fmaf(-kk, 6.2831854820251465f, p);              // fl(2 pi)
    r = fmaf(-kk, -1.7484555314695172e-07f, r);              // 2 pi - fl(2 pi)
    const float rev = r * 0.15915494309189535f;
    sn[i] = __builtin_amdgcn_sinf(rev);
    cs[i] = __builtin_amdgcn_cosf(rev);
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> h(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-6000.0, 6000.0);
    for(auto& x : h) x = static_cast<float>(u(g));
    float *d, *s, *c;
    hipMalloc(&d, n * 4);
    hipMalloc(&s, n * 4);
    hipMalloc(&c, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, s, c, n);
    std::vector<float> hs(n), hc(n);
    hipMemcpy(hs.data(), s, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, rs = 0;
    for(int i = 0; i < n; i++)
    {
        const double x = h[i];
        es = std::fmax(es, std::fabs(hs[i] - std::sin(x)));
        ec = std::fmax(ec, std::fabs(hc[i] - std::cos(x)));
        rs += (hs[i] - std::sin(x)) * (hs[i] - std::sin(x));
    }
    printf("v_sin_f32 / v_cos_f32 after reduction by 2 pi: max abs error sin %.3e cos %.3e, rms sin %.3e (n = %d, |phi| < 6000)\n", es, ec, std::sqrt(rs / n), n);
    return 0;
}
