#include "result_filter.h"

#include <algorithm>
#include <ctime>
#include <iomanip>
#include <sstream>
#include <tuple>

namespace msk144host
{

std::string FilteredResult::stamp_string() const
{
    const std::time_t t = std::chrono::system_clock::to_time_t(stamp);
    std::tm tm_local{};
    localtime_r(&t, &tm_local);
    char buf[32];
    std::strftime(buf, sizeof(buf), "%Y%m%d%H%M%S", &tm_local);
    return buf;
}

std::string FilteredResult::format_line() const
{
    // iostream formatting on purpose: setw(2) int, setw(6) float with the default precision
    std::ostringstream os;
    os << "***  "
       << "snr=" << std::setw(2) << snr << "; "
       << "f0=" << std::setw(6) << f0 << "; "
       << "num_avg=" << num_avg << "; "
       << "nbadsync=" << nbadsync << "; "
       << "pattern_idx=" << pattern_idx << "; "
       << "date=" << stamp_string() << "; "
       << "msg='" << text << "'"
       << "; ";
    return os.str();
}

void ResultFilter::put(int snr, float f0, int num_avg, int nbadsync, int pattern_idx, const std::string& text)
{
    FilteredResult r;
    r.snr = snr;
    r.f0 = f0;
    r.num_avg = num_avg;
    r.nbadsync = nbadsync;
    r.pattern_idx = pattern_idx;
    r.text = text;
    r.stamp = std::chrono::system_clock::now();
    by_text_[text].push_back(std::move(r));  // arrival (item) order is kept inside a text group
}

std::vector<FilteredResult> ResultFilter::end_window() const
{
    std::vector<FilteredResult> out;
    out.reserve(by_text_.size());
    for(const auto& kv : by_text_)  // std::map iterates in lexicographic key order, like the reference's std::set
    {
        // The winner of a text group is whatever std::sort leaves in front (result_filter.cpp:61-72).  Exact ties
        // (same num_avg and nbadsync - routine: one ping is decoded by dozens of candidates) are broken by the sort
        // algorithm itself, so the same call on the same sequence is made here: built against the same standard
        // library this picks the same item - and prints the same f0/pattern_idx - as the reference binary.
        std::vector<FilteredResult> group = kv.second;
        std::sort(group.begin(), group.end(),
                  [](const FilteredResult& a, const FilteredResult& b) { return std::tie(a.num_avg, a.nbadsync) < std::tie(b.num_avg, b.nbadsync); });
        out.push_back(group.front());
    }
    return out;
}

}  // namespace msk144host
