// Noise-floor tracker behind the `snr=` field of the output lines.
// Behaviour of SNRTracker (snr_tracker.cu:21-69): 8 segment powers per window -> mean and peak; the
// noise floor starts at the mean, follows a rising mean slowly (0.9/0.1) and a falling mean at once;
// snr = 10 log10(peak/noise - 1), clamped to [-8, 24], printed truncated to int.
// Unlike the reference, the 8 segment powers arrive from the GPU (msk144_segment_power) instead of a
// 41 KB copy of the analytic window.
#pragma once

namespace msk144host
{

class SnrTracker
{
public:
    void update(const float segment_power[8]);
    float snr_db() const { return snr_; }
    int snr_int() const { return static_cast<int>(snr_); }
    float noise_floor() const { return noise_; }

private:
    float noise_ = 0.0f;
    float snr_ = 0.0f;
};

}  // namespace msk144host
