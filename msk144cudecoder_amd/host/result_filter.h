// Per-window de-duplication of decoded texts (behaviour of ResultFilter, result_filter.cpp:43-74):
// one line per distinct text, in lexicographic order of the text; among candidates with the same text
// the one with the lowest num_avg wins, then the lowest nbadsync.  Exact ties are common (one ping is decoded
// by many candidates) and the reference leaves them to std::sort; this class makes the same std::sort call on
// the same sequence, so a build against the same standard library prints the same candidate.  Checked against
// the reference's own compiled result_filter.cpp (oracle/_ref, tests/test_ref_host.py).
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
    void begin_window() { by_text_.clear(); }
    void put(int snr, float f0, int num_avg, int nbadsync, int pattern_idx, const std::string& text);
    std::vector<FilteredResult> end_window() const;

private:
    std::map<std::string, std::vector<FilteredResult>> by_text_;
};

}  // namespace msk144host
