// The multi-stream working loop of msk144hipdecoder, one instance per GPU.
//
// The reference's loop (main.cu:261-422) is one thread, one stream, one device (main.cu:115): fread a hop, decode, print.  Here a
// DeviceLoop owns one library handle (= one device), a contiguous share of the input streams, an ingest+submit thread and a
// post-processing thread, pipelined over the handle's two pinned staging slots; `msk144hipdecoder --devices=0,1,...` runs one
// DeviceLoop per listed device side by side.  Nothing is shared between the loops but the LinePrinter (stdout) and the log lock
// (stderr), so the host work of a hop - poll/read, copy into the pinned slot, text layer - scales with the number of devices
// instead of queueing on one thread.  Stream numbers in the output (`ch=`) and in messages are GLOBAL: first stream of the loop +
// local index.
#pragma once

#include "window_decoder.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace msk144host
{

using Clock = std::chrono::steady_clock;

struct LoopOptions
{
    int hop_timeout_ms = 20;         // --hop-timeout-ms: how long a batch waits for further streams once one has a hop
    int connect_timeout_ms = 10000;  // --connect-timeout-ms: how long a FIFO may stay without a writer
    bool skip_wav = false;           // --skip-wav-header
    bool tag_channels = true;        // lines carry "ch=<global stream>; " (the run has more than one stream)
};

struct Accumulator
{
    double sum = 0.0, worst = 0.0;
    long n = 0;
    void add(double v)
    {
        sum += v;
        if(v > worst) worst = v;
        n++;
    }
    double mean() const { return n ? sum / n : 0.0; }
};

// One line at a time on stderr, whole batches on stdout, whichever loop they come from.
class LinePrinter
{
public:
    // lines of the streams of one batch; ids[j] = global stream number of entry j
    void print(const std::vector<int>& ids, const std::vector<const std::vector<FilteredResult>*>& lines, bool tag_channels);
    void log(const std::string& line);  // stderr, newline appended

private:
    std::mutex out_, err_;
};

struct LoopStats
{
    long batches = 0, hops = 0, late = 0;
    long long worst_ms = 0;
    long overflowed_hops = 0;
    Accumulator ingest, assemble, submit, wait, post, print, latency, records;
    float device_ms[MSK144_T_COUNT] = {};
    bool have_device_ms = false;
    double wall_s = 0.0;
};

class DeviceLoop
{
public:
    // opt.device / opt.channels: the device and the number of streams of THIS loop; first_stream: global number of its stream 0
    DeviceLoop(const DecoderOptions& opt, int first_stream, const LoopOptions& lo, LinePrinter& printer);
    ~DeviceLoop();
    DeviceLoop(const DeviceLoop&) = delete;
    DeviceLoop& operator=(const DeviceLoop&) = delete;

    bool ok() const { return dec_.ok() && error_.empty(); }
    std::string error() const { return error_.empty() ? dec_.error() : error_; }
    WindowDecoder& decoder() { return dec_; }
    int streams() const { return nch_; }
    int first_stream() const { return base_; }
    int device() const { return device_; }

    // --inputs: one file or FIFO per stream (opened non-blocking).  false + error() on failure.
    bool open_inputs(const std::vector<std::string>& paths);
    // --interleaved: the hops arrive through feed() instead (one reader thread splits stdin between the loops)
    void use_feed() { fed_ = true; }

    void start();
    // One hop of every stream of this loop, stream after stream (bytes_per_stream each: a whole window the first time, half a
    // window afterwards).  Blocks while two blocks are already queued (back-pressure into the reader).  false once the loop failed.
    bool feed(const unsigned char* data, size_t bytes_per_stream);
    void feed_end();
    int join();  // 0, or 2 after a library failure (already logged)

    const LoopStats& stats() const { return stats_; }
    // per-stream deadline accounting, printed by the caller
    struct StreamReport
    {
        long hops = 0, late = 0;
        long long worst_ms = 0;
    };
    StreamReport stream_report(int local) const;

private:
    struct Stream
    {
        int fd = -1;
        bool fifo = false;       // a FIFO reads 0 bytes while no writer has connected yet: that is not its end
        bool connected = false;  // a writer is or was there: data seen, a read that would block, or poll() reported anything
        bool eof = false;
        bool readable = true;    // worth a read(): set by poll(), cleared when a read would block or found no writer
        bool first = true;       // next hop is the 5184-sample fill (main.cu:271-283), later ones 2592 (:284-294)
        size_t skip = 0;         // header bytes still to drop
        std::vector<unsigned char> pending;
        bool ready = false;  // a complete hop sits in `pending`
        Clock::time_point ready_at;
        // deadline accounting (owned by the post-processing thread)
        long hops = 0, late = 0;
        long long worst_ms = 0;
    };
    // One submitted hop of every ready stream, handed from the ingest thread to the post-processing thread.
    struct Batch
    {
        int slot = 0;
        std::vector<int> streams;                 // local numbers, ascending
        std::vector<Clock::time_point> ready_at;  // when each of them had its hop complete
        Clock::time_point go;                     // batch released by the policy
        double assemble_ms = 0.0, submit_ms = 0.0;
    };

    void ingest_main();
    void post_main();
    void drain_descriptors(int& open_streams);
    bool take_fed_block(int& open_streams);
    void fail(const std::string& what);

    WindowDecoder dec_;
    const int nch_, base_, device_;
    const LoopOptions lo_;
    LinePrinter& out_;
    std::string error_;
    bool fed_ = false;

    size_t win_bytes_ = 0, half_ = 0, unit_ = 0;
    std::vector<Stream> st_;
    WindowDecoder::HopStage stage_[WindowDecoder::kSlots];
    Clock::time_point opened_at_;

    // hand-over between the ingest thread and the post-processing thread
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Batch> in_flight_;
    std::deque<int> free_slots_;
    bool no_more_ = false, failed_ = false;

    // --interleaved hand-over from the reader
    std::mutex feed_mu_;
    std::condition_variable feed_cv_;
    std::deque<std::vector<unsigned char>> feed_q_;
    std::deque<size_t> feed_bytes_;
    bool feed_eof_ = false;

    std::thread ingest_, post_;
    LoopStats stats_;
};

// Set by the program's SIGINT / SIGTERM handler: every loop then treats its streams as ended - the hops already submitted are
// collected and printed, the summary and "Done" follow and the program exits 0 (the reference has no handler: it dies mid-hop).
extern std::atomic<bool> g_stop_requested;
// Set by DeviceLoop::fail() together with g_stop_requested: the stop was caused by a failing loop, not by the operator.
extern std::atomic<bool> g_loop_failed;

// contiguous split of n streams over `parts` loops, sizes differing by at most one (the rule of sharding.shard_channels)
void split_streams(int n, int parts, int part, int& first, int& count);

}  // namespace msk144host
