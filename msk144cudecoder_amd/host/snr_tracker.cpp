#include "snr_tracker.h"

#include <algorithm>
#include <cmath>

namespace msk144host
{

namespace
{

constexpr float kSnrFloorDb = -8.0f;
constexpr float kSnrCeilDb = 24.0f;

// Asymmetric follower: a louder window raises the floor by a tenth of the difference,
// a quieter one replaces it outright; the very first window seeds it.
float follow_noise(float floor_now, float window_mean)
{
    if(floor_now <= 0.0f) return window_mean;
    if(window_mean > floor_now) return 0.9f * floor_now + 0.1f * window_mean;
    return window_mean;
}

}  // namespace

void SnrTracker::update(const float seg[8])
{
    // mean and peak of the 8 segment powers; the sum runs left to right from 0.0f like the
    // reference's std::accumulate (snr_tracker.cu:33)
    float total = 0.0f;
    for(int i = 0; i < 8; i++) total = total + seg[i];
    const float mean = total / 8;
    const float peak = *std::max_element(seg, seg + 8);

    noise_ = follow_noise(noise_, mean);

    float db = 0.0f;
    if(noise_ > 0.0f) db = 10.0f * std::log10(peak / noise_ - 1.0f);
    // clamp with plain comparisons so that a NaN (all-zero window) passes through as in the reference
    if(db > kSnrCeilDb) db = kSnrCeilDb;
    if(db < kSnrFloorDb) db = kSnrFloorDb;  // also catches the -inf of log10(0)
    snr_ = db;
}

}  // namespace msk144host
