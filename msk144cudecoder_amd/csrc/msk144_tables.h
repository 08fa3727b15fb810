// Host-side table builders of the search context: frequency grid, 42-tap sync template, raised-cosine mask of
// the FFT front end.  Plain C++17 (no HIP), float32 with one rounding per operation (callers compile with
// -ffp-contract=off): the kernels receive these tables as data, so they must carry the same float values the
// reference computes at start-up (msk_context.cuh:95-107,135-196; analytic_fft.cu:32-57).  Shared by the C ABI
// (msk144_api.cpp) and by the CPU test library (host/host_capi.cpp), which is how tests/test_ref_constants.py
// pins them to the reference's constants without a GPU.
#pragma once

#include <cmath>
#include <vector>

#include "msk144_protocol.h"

namespace msk144
{

// Where each run of template taps comes from: `count` taps starting at tap `first` of the real (I) or imaginary (Q)
// rail are the half-sine pulse from sample `pulse_from` on, signed by sync bit `sync_bit`.  The sync word's odd bits
// ride on I, the even bits on Q, offset by half a pulse: Q starts and I ends with a half pulse
// (msk_context.cuh:188-196).
struct TemplateRun
{
    bool imag;
    int first;
    int count;
    int pulse_from;
    int sync_bit;
};
constexpr TemplateRun kTemplateRuns[8] = {
    {true, 0, 6, 6, 0},   {true, 6, 12, 0, 2},   {true, 18, 12, 0, 4},  {true, 30, 12, 0, 6},
    {false, 0, 12, 0, 1}, {false, 12, 12, 0, 3}, {false, 24, 12, 0, 5}, {false, 36, 6, 0, 7},
};
constexpr int kPulseSamples = 12;

inline void half_sine_pulse(float* pp /*[12]*/)
{
    const float pi = 3.14159265358979323846f;
    for(int i = 0; i < kPulseSamples; i++) pp[i] = sinf(static_cast<float>(i) * pi / 12.0f);
}

inline void sync_template(float* re /*[42]*/, float* im /*[42]*/, float* pp /*[12]*/)
{
    half_sine_pulse(pp);
    for(const TemplateRun& r : kTemplateRuns)
    {
        float* rail = r.imag ? im : re;
        const float sign = static_cast<float>(kSync8Pm[r.sync_bit]);
        for(int i = 0; i < r.count; i++) rail[r.first + i] = pp[r.pulse_from + i] * sign;
    }
}

// f_b = center + if1 + b*step, if1 = -half*step (msk_context.cuh:102-107,135)
inline std::vector<float> frequency_grid(float center_hz, float width_hz, float step_hz)
{
    const int half = grid_half_len(width_hz, step_hz);
    const float first_offset = -1 * half * step_hz;
    std::vector<float> f(2 * half + 1);
    for(int b = 0; b < static_cast<int>(f.size()); b++) f[b] = center_hz + first_offset + b * step_hz;
    return f;
}

// Spectral weight of bins 0..nfft/2-1 for the FFT front end: 1 inside +-900 Hz of 1500 Hz, raised-cosine roll-off
// to 0 at +-1100 Hz (symbol time 1/2000 s, roll-off 0.1; analytic_fft.cu:41-57).
inline std::vector<float> fft_band_mask()
{
    const int bins = kFftSize / 2;
    const float bin_hz = 12000.0f / kFftSize;
    const float pi = 3.14159265358979323846f;
    const float symbol_time = 1.0f / 2000.0f;
    const float rolloff = 0.1f;
    const float flat_edge = (1 - rolloff) / (2 * symbol_time);
    const float stop_edge = (1 + rolloff) / (2 * symbol_time);
    std::vector<float> w(bins);
    for(int i = 0; i < bins; i++)
    {
        const float off = fabsf(i * bin_hz - 1500.0f);
        float v = 1.0f;
        if(off > flat_edge && off <= stop_edge) v = v * 0.5f * (1.0f + static_cast<float>(cos((pi * symbol_time / rolloff) * (off - flat_edge))));
        else if(off > stop_edge) v = 0.0f;
        w[i] = v;
    }
    return w;
}

}  // namespace msk144
