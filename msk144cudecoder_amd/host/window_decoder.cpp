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
    p.channels = 1;
    p.device = opt.device;
    if(msk144_create(&p, &handle_) != MSK144_OK)
    {
        error_ = msk144_last_error(nullptr);
        handle_ = nullptr;
        return;
    }
    msk144_geometry(handle_, &F_, &D_, &K_);
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
    lines.clear();
    int rc = (opt_.read_mode == 2) ? msk144_submit_iq(handle_, static_cast<const int8_t*>(window))
                                   : msk144_submit_audio(handle_, static_cast<const int16_t*>(window));
    if(rc == MSK144_OK) rc = msk144_decode(handle_);
    float seg[8];
    if(rc == MSK144_OK) rc = msk144_segment_power(handle_, seg);
    int32_t n = 0;
    if(rc == MSK144_OK) rc = msk144_result_count(handle_, &n);
    if(rc == MSK144_OK)
    {
        results_.resize(n > 0 ? n : 1);
        rc = msk144_results(handle_, results_.data(), static_cast<int32_t>(results_.size()), &n);
        if(rc == MSK144_EOVERFLOW) rc = MSK144_OK;  // list truncated at max_results; keep what we have
        if(static_cast<size_t>(n) < results_.size()) results_.resize(n);
    }
    if(rc != MSK144_OK)
    {
        error_ = msk144_last_error(handle_);
        return false;
    }

    snr_.update(seg);  // main.cu:388

    std::vector<AcceptedCandidate> accepted;
    accepted.reserve(results_.size());
    for(const msk144_result& r : results_)
    {
        AcceptedCandidate c;
        c.f0 = r.f0;
        c.num_avg = r.num_avg;
        c.nbadsync = r.nbadsync;
        c.pattern_idx = r.pattern_idx;
        unpack_bits(r.message, c.bits);
        accepted.push_back(c);
    }
    lines = postprocess_window(accepted, snr_.snr_int(), opt_.reference_cache_quirk, calls_, filter_);
    if(opt_.print_bits)
    {
        // debug aid: the checkable artefact is the payload, not the text
        for(FilteredResult& l : lines)
        {
            for(const AcceptedCandidate& c : accepted)
            {
                std::string t;
                CallHashTable scratch;
                if(decode_message(c.bits, scratch, t) && t == l.text)
                {
                    std::string b(77, '0');
                    for(int i = 0; i < 77; i++) b[i] = c.bits[i] ? '1' : '0';
                    l.text += "' bits='" + b;
                    break;
                }
            }
        }
    }
    return true;
}

}  // namespace msk144host
