#include "result_filter.h"

#include <ctime>
#include <iomanip>
#include <sstream>

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
    auto it = best_.find(text);
    if(it == best_.end())
    {
        best_.emplace(text, std::move(r));
        return;
    }
    const FilteredResult& cur = it->second;
    const bool better = (r.num_avg < cur.num_avg) || (r.num_avg == cur.num_avg && r.nbadsync < cur.nbadsync);
    if(better) it->second = std::move(r);
}

std::vector<FilteredResult> ResultFilter::end_window() const
{
    std::vector<FilteredResult> out;
    out.reserve(best_.size());
    for(const auto& kv : best_) out.push_back(kv.second);  // std::map iterates in lexicographic key order
    return out;
}

}  // namespace msk144host
