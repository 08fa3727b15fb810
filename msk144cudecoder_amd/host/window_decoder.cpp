#include "window_decoder.h"

namespace msk144host
{

WindowDecoder::WindowDecoder(const DecoderOptions& opt)
    : opt_(opt)
{
    msk144_params p;
    msk144_default_params(&p);
    p.center_hz = opt.center_hz;
    p.width_hz = opt.width_hz;
    p.step_hz = opt.step_hz;
    p.scan_depth = opt.scan_depth;
    p.nbadsync_threshold = opt.nbadsync_threshold;
    p.read_mode = opt.read_mode;
    p.analytic_method = opt.analytic_method;
    p.channels = opt.channels < 1 ? 1 : opt.channels;
    opt_.channels = p.channels;
    p.device = opt.device;
    p.max_results = 0x7fffffff;  // clamped by the library to channels * items: the list can never overflow
    if(msk144_create(&p, &handle_) != MSK144_OK)
    {
        error_ = msk144_last_error(nullptr);
        handle_ = nullptr;
        return;
    }
    msk144_geometry(handle_, &F_, &D_, &K_);
    snr_.resize(opt_.channels);
    filter_.resize(opt_.channels);
    calls_.resize(opt_.channels);
    seg_.resize(static_cast<size_t>(opt_.channels) * 8);
}

WindowDecoder::~WindowDecoder()
{
    if(handle_) msk144_destroy(handle_);
}

float WindowDecoder::left_bound() const
{
    float f = 0.0f;
    if(handle_) msk144_frequency(handle_, 0, &f);
    return f;
}

float WindowDecoder::right_bound() const
{
    float f = 0.0f;
    if(handle_) msk144_frequency(handle_, F_ - 1, &f);
    return f;
}

bool WindowDecoder::process(const void* window, std::vector<FilteredResult>& lines)
{
    std::vector<std::vector<FilteredResult>> all;
    const bool ok = process(window, std::vector<bool>(1, true), all);
    lines = all.empty() ? std::vector<FilteredResult>() : std::move(all[0]);
    return ok;
}

bool WindowDecoder::process(const void* windows, const std::vector<bool>& active, std::vector<std::vector<FilteredResult>>& lines)
{
    const int nch = opt_.channels;
    lines.assign(nch, {});
    int rc = (opt_.read_mode == 2) ? msk144_submit_iq(handle_, static_cast<const int8_t*>(windows))
                                   : msk144_submit_audio(handle_, static_cast<const int16_t*>(windows));
    if(rc == MSK144_OK) rc = msk144_decode(handle_);
    if(rc == MSK144_OK) rc = msk144_segment_power(handle_, seg_.data());
    int32_t n = 0;
    if(rc == MSK144_OK) rc = msk144_result_count(handle_, &n);
    if(rc == MSK144_OK)
    {
        results_.resize(n > 0 ? n : 1);
        rc = msk144_results(handle_, results_.data(), static_cast<int32_t>(results_.size()), &n);
        results_.resize(n);
    }
    if(rc != MSK144_OK)
    {
        error_ = msk144_last_error(handle_);
        return false;
    }

    // results arrive ordered by (channel, item): walk them channel by channel
    size_t r = 0;
    for(int c = 0; c < nch; c++)
    {
        std::vector<AcceptedCandidate> accepted;
        for(; r < results_.size() && results_[r].channel == c; r++)
        {
            const msk144_result& res = results_[r];
            AcceptedCandidate a;
            a.f0 = res.f0;
            a.num_avg = res.num_avg;
            a.nbadsync = res.nbadsync;
            a.pattern_idx = res.pattern_idx;
            unpack_bits(res.message, a.bits);
            accepted.push_back(a);
        }
        if(c < static_cast<int>(active.size()) && !active[c]) continue;  // ended stream: leave its state alone
        snr_[c].update(&seg_[static_cast<size_t>(c) * 8]);  // main.cu:388
        lines[c] = postprocess_window(accepted, snr_[c].snr_int(), opt_.reference_cache_quirk, calls_[c], filter_[c]);
        if(opt_.print_bits)
        {
            // debug aid: the checkable artefact is the payload, not the text
            for(FilteredResult& l : lines[c])
            {
                for(const AcceptedCandidate& a : accepted)
                {
                    std::string t;
                    CallHashTable scratch;
                    if(decode_message(a.bits, scratch, t) && t == l.text)
                    {
                        std::string b(77, '0');
                        for(int i = 0; i < 77; i++) b[i] = a.bits[i] ? '1' : '0';
                        l.text += "' bits='" + b;
                        break;
                    }
                }
            }
        }
    }
    return true;
}

}  // namespace msk144host
