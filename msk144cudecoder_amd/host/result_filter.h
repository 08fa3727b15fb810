// Per-window de-duplication of decoded texts (behaviour of ResultFilter, result_filter.cpp:43-74):
// one line per distinct text, in lexicographic order of the text; among candidates with the same text
// the one with the lowest num_avg wins, then the lowest nbadsync.  The reference sorts with std::sort
// (order of exact ties unspecified); here the earliest candidate wins a tie, deterministically.
#pragma once

#include <chrono>
#include <map>
#include <string>
#include <vector>

namespace msk144host
{

struct FilteredResult
{
    int snr = 0;
    float f0 = 0.0f;
    int num_avg = 0;
    int nbadsync = 0;
    int pattern_idx = 0;
    std::string text;
    std::chrono::system_clock::time_point stamp;

    std::string stamp_string() const;  // localtime, YYYYmmddHHMMSS (result_filter.cpp:13-21)
    std::string format_line() const;   // the stdout line of main.cu:409-417, without the newline
};

class ResultFilter
{
public:
    void begin_window() { best_.clear(); }
    void put(int snr, float f0, int num_avg, int nbadsync, int pattern_idx, const std::string& text);
    std::vector<FilteredResult> end_window() const;

private:
    std::map<std::string, FilteredResult> best_;
};

}  // namespace msk144host
