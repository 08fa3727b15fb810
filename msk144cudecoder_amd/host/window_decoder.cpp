#include "window_decoder.h"

#include <chrono>
#include <cstring>

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
    // Compact-list capacity: a strong ping is accepted by many of its candidates (neighbouring bins, patterns, slots); 256 per
    // channel on average plus 131072 is far beyond what the synthetic bands here produce (58 records per decoded stream at 0 dB); the
    // library clamps to channels * items.  A hop that does overflow is reported (MSK144_EOVERFLOW -> HopTiming::overflow), its
    // truncated list is processed and the streams keep running - the reference keeps every ResultItem and cannot fail this way.
    const long long cap = opt.max_results > 0 ? opt.max_results : 256ll * p.channels + 131072;  // a one-stream handle can hold every candidate of the widest search grid
    p.max_results = cap > 0x7fffffff ? 0x7fffffff : static_cast<int32_t>(cap);
    if(msk144_create(&p, &handle_) != MSK144_OK)
    {
        error_ = msk144_last_error(nullptr);
        handle_ = nullptr;
        return;
    }
    // The program reads the result list only: no LLR row has to outlive its decode, whatever the number of streams - the kernels
    // then stop a candidate the nbadsync gate drops after its sync check and compute slots that fold the same frames once
    // (include/msk144hip.h), exactly as a large batch does.  --every-slot: every slot on its own, as in the reference.
    msk144_set_llr_retention(handle_, 0);
    if(opt.every_slot) msk144_set_copy_handover(handle_, 0);
    msk144_geometry(handle_, &F_, &D_, &K_);
    snr_.resize(opt_.channels);
    filter_.resize(opt_.channels);
    calls_.resize(opt_.channels);
    window_bytes_ = MSK144_WINDOW_SAMPLES * (opt_.read_mode == 2 ? 2 * sizeof(int8_t) : sizeof(int16_t));
    if(opt_.profile) msk144_set_profiling(handle_, 1);
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
    unsigned char* in = static_cast<unsigned char*>(stage(0));
    if(!in) return false;
    std::vector<int> streams;
    for(int c = 0; c < opt_.channels; c++)
    {
        if(c < static_cast<int>(active.size()) && !active[c]) continue;
        std::memcpy(in + window_bytes_ * streams.size(), static_cast<const unsigned char*>(windows) + window_bytes_ * c, window_bytes_);
        streams.push_back(c);
    }
    if(streams.empty())
    {
        lines.assign(opt_.channels, {});
        return true;
    }
    return submit(0, streams) && collect(0, lines);
}

void* WindowDecoder::stage(int slot)
{
    void* p = nullptr;
    if(msk144_input_slot(handle_, slot, &p, nullptr) != MSK144_OK)
    {
        error_ = msk144_last_error(handle_);
        return nullptr;
    }
    return p;
}

bool WindowDecoder::hop_stage(int slot, HopStage& out)
{
    void *hops = nullptr, *first = nullptr;
    if(msk144_hop_slot(handle_, slot, &hops, &first, &out.streams, &out.is_first) != MSK144_OK)
    {
        error_ = msk144_last_error(handle_);
        return false;
    }
    out.hops = static_cast<unsigned char*>(hops);
    out.first_halves = static_cast<unsigned char*>(first);
    return true;
}

bool WindowDecoder::submit_hops(int slot, int n)
{
    HopStage hs;
    if(!hop_stage(slot, hs)) return false;
    streams_[slot].assign(hs.streams, hs.streams + n);
    int rc = msk144_push_hops(handle_, slot, n);
    if(rc == MSK144_OK) rc = msk144_decode(handle_);
    if(rc == MSK144_OK) rc = msk144_fetch_async(handle_, slot);
    if(rc != MSK144_OK)
    {
        error_ = msk144_last_error(handle_);
        return false;
    }
    return true;
}

bool WindowDecoder::submit(int slot, const std::vector<int>& streams)
{
    streams_[slot] = streams;
    int rc = msk144_submit_slot_n(handle_, slot, static_cast<int32_t>(streams.size()));
    if(rc == MSK144_OK) rc = msk144_decode(handle_);
    if(rc == MSK144_OK) rc = msk144_fetch_async(handle_, slot);
    if(rc != MSK144_OK)
    {
        error_ = msk144_last_error(handle_);
        return false;
    }
    return true;
}

bool WindowDecoder::stage_times(float out[MSK144_T_COUNT])
{
    return msk144_stage_times(handle_, out, nullptr, 0) == MSK144_OK;
}

bool WindowDecoder::collect(int slot, std::vector<std::vector<FilteredResult>>& lines, HopTiming* timing)
{
    using Clock = std::chrono::steady_clock;
    const int nch = opt_.channels;
    lines.assign(nch, {});
    const auto t0 = Clock::now();
    const msk144_result* results = nullptr;
    const float* seg = nullptr;
    int32_t n = 0;
    const int rc = msk144_fetch_wait(handle_, slot, &results, &n, &seg);
    if(rc != MSK144_OK && rc != MSK144_EOVERFLOW)  // on overflow the library hands back the records that fitted
    {
        error_ = "msk144_fetch_wait failed";
        return false;
    }
    last_overflow_ = rc == MSK144_EOVERFLOW;
    const auto t1 = Clock::now();
    const std::vector<int>& streams = streams_[slot];
    const size_t n_results = static_cast<size_t>(n);

    // results arrive ordered by (position in the slot, item): walk them position by position
    size_t r = 0;
    for(size_t j = 0; j < streams.size(); j++)
    {
        const int c = streams[j];
        std::vector<AcceptedCandidate> accepted;
        for(; r < n_results && results[r].channel == static_cast<int32_t>(j); r++)
        {
            const msk144_result& res = results[r];
            AcceptedCandidate a;
            a.f0 = res.f0;
            a.num_avg = res.num_avg;
            a.nbadsync = res.nbadsync;
            a.pattern_idx = res.pattern_idx;
            unpack_bits(res.message, a.bits);
            accepted.push_back(a);
        }
        snr_[c].update(&seg[j * 8]);  // main.cu:388
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
    if(timing)
    {
        timing->wait_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        timing->post_ms = std::chrono::duration<double, std::milli>(Clock::now() - t1).count();
        timing->records = n;
        timing->overflow = rc == MSK144_EOVERFLOW;
    }
    return true;
}

}  // namespace msk144host
