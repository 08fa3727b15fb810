// GPU-independent part of window_decoder (also linked into libmsk144host.so for the CPU tests).
#include "window_decoder.h"

#include <map>

namespace msk144host
{

void unpack_bits(const uint8_t packed[10], uint8_t bits[77])
{
    for(int i = 0; i < 77; i++) bits[i] = (packed[i / 8] >> (7 - (i % 8))) & 1u;
}

std::vector<FilteredResult> postprocess_window(const std::vector<AcceptedCandidate>& accepted, int snr, bool reference_cache_quirk, CallHashTable& table,
                                               ResultFilter& filter)
{
    filter.begin_window();
    struct Cached
    {
        bool found;
        std::string text;
    };
    std::map<std::string, Cached> cache;  // strict mode: keyed by payload
    bool have_first = false;
    Cached first{false, {}};

    for(const AcceptedCandidate& c : accepted)
    {
        Cached res;
        if(reference_cache_quirk)
        {
            if(!have_first)
            {
                first.found = decode_message(c.bits, table, first.text);
                have_first = true;
            }
            res = first;
        }
        else
        {
            const std::string key(reinterpret_cast<const char*>(c.bits), 77);
            auto it = cache.find(key);
            if(it == cache.end())
            {
                Cached d;
                d.found = decode_message(c.bits, table, d.text);
                it = cache.emplace(key, d).first;
            }
            res = it->second;
        }
        if(res.found) filter.put(snr, c.f0, c.num_avg, c.nbadsync, c.pattern_idx, res.text);
    }
    return filter.end_window();
}

}  // namespace msk144host
