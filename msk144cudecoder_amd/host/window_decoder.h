// One input stream on top of the C ABI: the host part of the reference's working loop
// (main.cu:388-419) and of do_decode (main.cu:474-525) - SNR tracking, payload -> text with the
// per-window decode cache, per-window result filter.
#pragma once

#include "../../include/msk144hip.h"
#include "result_filter.h"
#include "snr_tracker.h"
#include "unpack77.h"

#include <cstdint>
#include <string>
#include <vector>

namespace msk144host
{

struct DecoderOptions
{
    float center_hz = 1500.0f;
    float width_hz = 200.0f;
    float step_hz = 2.0f;
    int scan_depth = 4;
    int nbadsync_threshold = 1;
    int read_mode = 1;
    int analytic_method = 2;
    int device = 0;
    int channels = 1;  // independent input streams decoded together (the reference: 1)
    // The reference keys its per-window decode cache with a comparator that is always false (main.cu:437-445), so every
    // accepted candidate of a window receives the text - or the unpack failure - of the FIRST accepted candidate (and only that
    // one reaches unpack77, i.e. the callsign hash tables): one CRC-13 false positive at a lower frequency bin silences every
    // genuine decode of that window, and a second station is printed with the first one's text.  true (default): reproduce the
    // reference - same arguments, same stdout.  false (--strict-decode): every distinct payload is unpacked on its own.
    bool reference_cache_quirk = true;
    bool print_bits = false;  // append the 77-bit payload to each line (debug)
    bool profile = false;     // record per-stage device times (HIP events; --timing)
    int max_results = 0;      // --max-results: capacity of the compact decode list per hop; 0 = 256 per stream + 131072
    // --every-slot: demodulate and decode every slot on its own as the reference does (softbits_kernel.cuh:56-83, ldpc_kernel.cuh:100-249).
    // Default: the program only reads the result list, so no LLR row has to outlive its decode (msk144_set_llr_retention(h, 0)) and
    // slots that fold the same frames as a lower slot of their group report that slot's result - the same list (include/msk144hip.h)
    bool every_slot = false;
};

// Host wall time of one hop, split the way the pipelined loop spends it.
struct HopTiming
{
    double wait_ms = 0.0;  // blocked until the GPU had finished the hop and its results had arrived (msk144_fetch_wait)
    double post_ms = 0.0;  // payload -> text, SNR, per-window filter
    int records = 0;
    bool overflow = false;  // the hop held more decodes than the compact list: `records` of them were kept and processed
};

// What one accepted candidate contributes, independent of the GPU (unit-testable on the CPU).
struct AcceptedCandidate
{
    float f0;
    int num_avg;
    int nbadsync;
    int pattern_idx;
    uint8_t bits[77];
};

// Host post-processing of one window given the accepted candidates in item order.
std::vector<FilteredResult> postprocess_window(const std::vector<AcceptedCandidate>& accepted, int snr, bool reference_cache_quirk, CallHashTable& table,
                                               ResultFilter& filter);

void unpack_bits(const uint8_t packed[10], uint8_t bits[77]);

class WindowDecoder
{
public:
    explicit WindowDecoder(const DecoderOptions& opt);
    ~WindowDecoder();
    WindowDecoder(const WindowDecoder&) = delete;
    WindowDecoder& operator=(const WindowDecoder&) = delete;

    bool ok() const { return handle_ != nullptr; }
    const std::string& error() const { return error_; }

    int num_freqs() const { return F_; }
    int scan_depth() const { return D_; }
    float left_bound() const;
    float right_bound() const;

    int channels() const { return opt_.channels; }

    // One 5184-sample window per channel, [channels][5184] int16 or [channels][2*5184] int8 I/Q, decoded as one
    // batch; lines[c] = output lines of channel c.  `active[c] == false` marks a stream without a hop this time: it is
    // left out of the batch and its state stays untouched.  Returns false on a library error.
    bool process(const void* windows, const std::vector<bool>& active, std::vector<std::vector<FilteredResult>>& lines);
    // single-stream convenience (channels == 1)
    bool process(const void* window, std::vector<FilteredResult>& lines);

    // The same hop in pipelined form, over the two pinned staging slots the library handle owns: fill stage(slot), submit it
    // (asynchronous: H2D, front end, decode, D2H of count + records + segment powers), and collect it later - possibly on a
    // second thread while the other slot is being filled and submitted.  The reference's loop is strictly serial per hop
    // (main.cu:261-422).
    // A hop covers the streams listed in `streams` (ascending): their windows sit back to back at positions 0, 1, ... of the slot,
    // and the GPU work, the copies and the results of the hop are sized for that many windows - a stream that has no hop this
    // time costs nothing.  lines[c] is indexed by STREAM.
    static constexpr int kSlots = MSK144_SLOTS;
    // The slot's pinned hop-ring inputs (msk144_hop_slot): the device keeps every stream's 50 %-overlap window, so a hop ships
    // 2592 new samples per stream - entry j: hops (and, for a stream's first hop, first_halves) at j * half-window bytes,
    // streams[j] = the stream's number (ascending), is_first[j].
    struct HopStage
    {
        unsigned char* hops = nullptr;
        unsigned char* first_halves = nullptr;
        int32_t* streams = nullptr;
        uint8_t* is_first = nullptr;
    };
    bool hop_stage(int slot, HopStage& out);
    bool submit_hops(int slot, int n);  // the first n entries of the slot's hop stage
    void* stage(int slot);
    bool submit(int slot, const std::vector<int>& streams);
    bool collect(int slot, std::vector<std::vector<FilteredResult>>& lines, HopTiming* timing = nullptr);
    // the hop collect() returned last held more decodes than the compact list (its truncated list was processed)
    bool last_hop_overflowed() const { return last_overflow_; }
    // average device milliseconds per hop of every stage (frontend, scan, softbits, index, ldpc, collect, h2d, d2h); needs
    // DecoderOptions::profile; waits for the GPU
    bool stage_times(float out[MSK144_T_COUNT]);

private:
    DecoderOptions opt_;
    msk144_handle* handle_ = nullptr;
    int F_ = 0, D_ = 0, K_ = 0;
    std::string error_;
    // per-stream host state, as if each stream ran in its own reference process
    std::vector<SnrTracker> snr_;
    std::vector<ResultFilter> filter_;
    std::vector<CallHashTable> calls_;
    std::vector<int> streams_[kSlots];  // streams of the hop in flight on each slot, by position
    size_t window_bytes_ = 0;           // of one stream
    bool last_overflow_ = false;
};

}  // namespace msk144host
