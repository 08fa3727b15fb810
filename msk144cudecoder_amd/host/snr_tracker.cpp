#include "snr_tracker.h"

#include <cmath>

namespace msk144host
{

void SnrTracker::update(const float seg[8])
{
    float total = 0.0f;
    float peak = seg[0];
    for(int i = 0; i < 8; i++)
    {
        total = total + seg[i];
        if(seg[i] > peak) peak = seg[i];
    }
    const float mean = total / 8;

    if(noise_ <= 0.0f) noise_ = mean;                          // first window
    else if(mean > noise_) noise_ = 0.9f * noise_ + 0.1f * mean;  // slow to rise
    else noise_ = mean;                                        // quick to fall

    snr_ = (noise_ > 0.0f) ? 10.0f * std::log10(peak / noise_ - 1.0f) : 0.0f;
    if(snr_ > 24.0f) snr_ = 24.0f;
    if(snr_ < -8.0f) snr_ = -8.0f;
}

}  // namespace msk144host
