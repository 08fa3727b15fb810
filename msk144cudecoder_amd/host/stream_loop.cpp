#include "stream_loop.h"

#include <fcntl.h>
#include <poll.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <iostream>
#include <sstream>

namespace msk144host
{

namespace
{

double ms_between(Clock::time_point a, Clock::time_point b)
{
    return std::chrono::duration<double, std::milli>(b - a).count();
}

constexpr int kSoftLimitMs = 210;  // of the 216 ms a hop lasts (main.cu:398-403)

// A bounded wait for loops that also watch a flag no one notifies for (the signal handler's).  Against the system clock on purpose:
// wait_for() goes through pthread_cond_clockwait, which gcc 11's ThreadSanitizer does not intercept (it then reports the re-acquired
// mutex as a double lock); a wall-clock jump costs at most one early or late wake-up of a loop that re-checks its condition anyway.
void wait_50ms(std::condition_variable& cv, std::unique_lock<std::mutex>& lk)
{
    cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(50));
}

}  // namespace

std::atomic<bool> g_stop_requested{false};
std::atomic<bool> g_loop_failed{false};

void split_streams(int n, int parts, int part, int& first, int& count)
{
    const int base = n / parts, extra = n % parts;
    count = base + (part < extra ? 1 : 0);
    first = part * base + (part < extra ? part : extra);
}

void LinePrinter::print(const std::vector<int>& ids, const std::vector<const std::vector<FilteredResult>*>& lines, bool tag_channels)
{
    std::string text;
    for(size_t j = 0; j < ids.size(); j++)
    {
        for(const FilteredResult& l : *lines[j])
        {
            const std::string line = l.format_line();  // "***  snr=..." -> "***  ch=<c>; snr=..."
            if(tag_channels) text += line.substr(0, 5) + "ch=" + std::to_string(ids[j]) + "; " + line.substr(5);
            else text += line;
            text += '\n';
        }
    }
    if(text.empty()) return;
    std::lock_guard<std::mutex> lk(out_);
    std::cout.write(text.data(), static_cast<std::streamsize>(text.size()));
    std::cout.flush();
}

void LinePrinter::log(const std::string& line)
{
    std::lock_guard<std::mutex> lk(err_);
    std::cerr << line + "\n";
}

DeviceLoop::DeviceLoop(const DecoderOptions& opt, int first_stream, const LoopOptions& lo, LinePrinter& printer)
    : dec_(opt), nch_(opt.channels < 1 ? 1 : opt.channels), base_(first_stream), device_(opt.device), lo_(lo), out_(printer)
{
    const size_t sample_bytes = (opt.read_mode == 1) ? sizeof(int16_t) : 2 * sizeof(int8_t);
    win_bytes_ = MSK144_WINDOW_SAMPLES * sample_bytes;
    half_ = win_bytes_ / 2;
    unit_ = (opt.read_mode == 1) ? sizeof(int16_t) : sizeof(int8_t);  // the reference counts items of this size
    st_.resize(nch_);
    for(Stream& s : st_) s.pending.reserve(win_bytes_);
}

DeviceLoop::~DeviceLoop()
{
    if(ingest_.joinable() || post_.joinable()) join();
    for(Stream& s : st_)
        if(s.fd >= 0) close(s.fd);
}

bool DeviceLoop::open_inputs(const std::vector<std::string>& paths)
{
    for(int c = 0; c < nch_; c++)
    {
        // O_NONBLOCK: opening a FIFO whose writer has not arrived yet returns at once, and read() never parks the batch
        st_[c].fd = open(paths[c].c_str(), O_RDONLY | O_NONBLOCK);
        if(st_[c].fd < 0)
        {
            error_ = "Cannot open input " + paths[c] + ": " + strerror(errno);
            return false;
        }
        st_[c].skip = lo_.skip_wav ? 44 : 0;
        struct stat sb{};
        st_[c].fifo = fstat(st_[c].fd, &sb) == 0 && S_ISFIFO(sb.st_mode);
    }
    return true;
}

void DeviceLoop::fail(const std::string& what)
{
    out_.log("msk144hip: " + what);
    // the sibling loops of a multi-device run must not keep the process alive until their own streams end: they stop reading, finish
    // the hops in flight and the program exits with this loop's status
    g_loop_failed.store(true, std::memory_order_relaxed);
    g_stop_requested.store(true, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(mu_);
        failed_ = true;
    }
    cv_.notify_all();
    {
        std::lock_guard<std::mutex> lk(feed_mu_);  // a feed()/take_fed_block() between its predicate and its wait must not miss this
    }
    feed_cv_.notify_all();
}

void DeviceLoop::start()
{
    opened_at_ = Clock::now();
    for(int k = 0; k < WindowDecoder::kSlots; k++)
    {
        // pinned, owned by the library handle; the stream windows themselves live on the device
        if(!dec_.hop_stage(k, stage_[k]))
        {
            fail(dec_.error());
            return;
        }
        free_slots_.push_back(k);
    }
    post_ = std::thread([this] { post_main(); });
    ingest_ = std::thread([this] { ingest_main(); });
}

bool DeviceLoop::feed(const unsigned char* data, size_t bytes_per_stream)
{
    std::unique_lock<std::mutex> lk(feed_mu_);
    // back-pressure: at most two blocks queued.  The wait wakes every 50 ms for the stop flag (set from a signal handler, which
    // cannot notify a condition variable).
    while(true)
    {
        if(g_stop_requested.load(std::memory_order_relaxed)) return false;
        {
            std::lock_guard<std::mutex> g(mu_);
            if(failed_) return false;
        }
        if(feed_q_.size() < 2) break;
        wait_50ms(feed_cv_, lk);
    }
    feed_q_.emplace_back(data, data + bytes_per_stream * static_cast<size_t>(nch_));
    feed_bytes_.push_back(bytes_per_stream);
    lk.unlock();
    feed_cv_.notify_all();
    return true;
}

void DeviceLoop::feed_end()
{
    {
        std::lock_guard<std::mutex> lk(feed_mu_);
        feed_eof_ = true;
    }
    feed_cv_.notify_all();
}

int DeviceLoop::join()
{
    if(ingest_.joinable()) ingest_.join();
    {
        std::lock_guard<std::mutex> lk(mu_);
        no_more_ = true;
    }
    cv_.notify_all();
    if(post_.joinable()) post_.join();
    bool failed;
    {
        std::lock_guard<std::mutex> lk(mu_);
        failed = failed_;
    }
    stats_.wall_s = ms_between(opened_at_, Clock::now()) * 1e-3;
    stats_.hops = stats_.late = 0;
    stats_.worst_ms = 0;
    for(const Stream& s : st_)
    {
        stats_.hops += s.hops;
        stats_.late += s.late;
        if(s.worst_ms > stats_.worst_ms) stats_.worst_ms = s.worst_ms;
    }
    if(!failed && dec_.ok()) stats_.have_device_ms = dec_.stage_times(stats_.device_ms);
    return failed ? 2 : 0;
}

DeviceLoop::StreamReport DeviceLoop::stream_report(int local) const
{
    StreamReport r;
    r.hops = st_[local].hops;
    r.late = st_[local].late;
    r.worst_ms = st_[local].worst_ms;
    return r;
}

// --interleaved: the reader thread hands over one hop of every stream at a time (blocking, like the reference's fread)
bool DeviceLoop::take_fed_block(int& open_streams)
{
    std::vector<unsigned char> block;
    size_t per = 0;
    bool ended = false;
    {
        std::unique_lock<std::mutex> lk(feed_mu_);
        while(true)
        {
            if(!feed_q_.empty() || feed_eof_) break;
            {
                std::lock_guard<std::mutex> g(mu_);
                if(failed_) break;
            }
            // a stop request reaches this loop through the reader (it stops feeding and calls feed_end()); the timed wait only
            // bounds how long a lost wake-up could last
            wait_50ms(feed_cv_, lk);
        }
        if(!feed_q_.empty())
        {
            block = std::move(feed_q_.front());
            per = feed_bytes_.front();
            feed_q_.pop_front();
            feed_bytes_.pop_front();
        }
        else ended = true;
    }
    feed_cv_.notify_all();
    if(ended)
    {
        for(Stream& s : st_) s.eof = true;
        open_streams = 0;
        return false;
    }
    const auto now = Clock::now();
    for(int c = 0; c < nch_; c++)
    {
        st_[c].pending.assign(block.begin() + static_cast<long>(per * c), block.begin() + static_cast<long>(per * (c + 1)));
        st_[c].ready = true;
        st_[c].ready_at = now;
    }
    open_streams = nch_;
    return true;
}

// --inputs: drain whatever every open stream has, up to one hop each.  Only streams poll() reported (or never asked about) are
// read: at thousands of streams the read() calls that would just say EAGAIN were most of the ingest time.
void DeviceLoop::drain_descriptors(int& open_streams)
{
    static thread_local std::vector<unsigned char> chunk(1 << 16);
    if(g_stop_requested.load(std::memory_order_relaxed))
    {
        // operator stop: no more reads; hops that are complete still go out with the next batch, partial ones are dropped
        for(Stream& s : st_)
            if(!s.eof) s.eof = true;
        open_streams = 0;
        return;
    }
    const bool connect_expired = ms_between(opened_at_, Clock::now()) >= lo_.connect_timeout_ms;
    for(int c = 0; c < nch_; c++)
    {
        Stream& s = st_[c];
        if(s.eof) continue;
        open_streams++;
        // A FIFO nobody has written to yet sits in the poll set without events (Linux reports POLLHUP only after a writer has come
        // and gone); once the connect timeout has passed, one more read() decides: 0 bytes then means the stream never started.
        if(s.fifo && !s.connected && !s.readable && connect_expired) s.readable = true;
        if(!s.readable) continue;
        const size_t need = s.first ? win_bytes_ : half_;
        while(!s.ready)
        {
            const size_t room = s.skip ? (s.skip < chunk.size() ? s.skip : chunk.size()) : need - s.pending.size();
            const ssize_t got = read(s.fd, chunk.data(), room < chunk.size() ? room : chunk.size());
            if(got > 0)
            {
                s.connected = true;
                if(s.skip) s.skip -= static_cast<size_t>(got);
                else s.pending.insert(s.pending.end(), chunk.begin(), chunk.begin() + got);
                if(!s.skip && s.pending.size() == need)
                {
                    s.ready = true;
                    s.ready_at = Clock::now();
                }
                continue;
            }
            if(got == 0 && s.fifo && !s.connected && !connect_expired)
            {
                s.readable = false;  // no writer on this FIFO yet: read() reports 0 bytes, which only means "nobody there so far"
                break;
            }
            if(got == 0)
            {
                // writer closed: what is left is a short read, exactly the reference's end-of-stream message
                out_.log("ch=" + std::to_string(base_ + c) + ": Incomplete read error. rc=" + std::to_string(s.pending.size() / unit_));
                s.eof = true;
                open_streams--;
            }
            else if(errno == EAGAIN || errno == EWOULDBLOCK)
            {
                s.connected = true;   // a read that would block means a writer holds the other end: a later 0-byte read is its end
                s.readable = false;   // drained: wait for poll() to say otherwise
            }
            else if(errno != EINTR)
            {
                out_.log("ch=" + std::to_string(base_ + c) + ": read error: " + strerror(errno));
                s.eof = true;
                open_streams--;
            }
            break;
        }
    }
}

void DeviceLoop::ingest_main()
{
    std::vector<pollfd> pfd(nch_);
    std::vector<int> pfd_stream(nch_);
    double ingest_busy_ms = 0.0;

    while(true)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if(failed_) break;
        }
        // 1. whatever the streams have, up to one hop each
        const auto d0 = Clock::now();
        int open_streams = 0, ready = 0;
        if(fed_)
        {
            bool any_ready = false;
            for(const Stream& s : st_) any_ready = any_ready || s.ready;
            if(!any_ready) take_fed_block(open_streams);
            else open_streams = nch_;
        }
        else drain_descriptors(open_streams);
        for(const Stream& s : st_)
            if(s.ready) ready++;
        ingest_busy_ms += ms_between(d0, Clock::now());
        if(open_streams == 0 && ready == 0) break;

        // 2. batch policy: go when every open stream has its hop, or when the oldest ready hop has waited hop_timeout_ms
        bool go = ready > 0 && ready >= open_streams;
        if(!go && ready > 0)
        {
            Clock::time_point oldest = Clock::now();
            for(const Stream& s : st_)
                if(s.ready && s.ready_at < oldest) oldest = s.ready_at;
            go = std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - oldest).count() >= lo_.hop_timeout_ms;
        }
        if(!go)
        {
            // sleep until more data arrives (or a FIFO's writer arrives or leaves)
            int n = 0;
            for(int c = 0; c < nch_; c++)
                if(!st_[c].eof && !st_[c].ready && !st_[c].readable)
                {
                    pfd[n] = {st_[c].fd, POLLIN, 0};
                    pfd_stream[n++] = c;
                }
            if(poll(pfd.data(), static_cast<nfds_t>(n), ready > 0 ? 5 : (n > 0 ? 50 : 10)) > 0)
                for(int k = 0; k < n; k++)
                    if(pfd[k].revents)
                    {
                        Stream& s = st_[pfd_stream[k]];
                        s.readable = true;  // data, hang-up or error: the next read() tells which
                        if(pfd[k].revents & (POLLIN | POLLHUP)) s.connected = true;  // a writer is, or was, there
                    }
            continue;
        }

        // 3. a free staging slot (back-pressure: with both slots in flight the streams wait in their pipes)
        Batch b;
        b.go = Clock::now();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !free_slots_.empty() || failed_; });
            if(failed_) break;
            b.slot = free_slots_.front();
            free_slots_.pop_front();
        }
        // 4. hand the new samples of every stream that has a hop to the library, packed back to back in the pinned slot: 2592 per
        // stream (all 5184 of a stream's first hop).  The 50 %-overlap window of each stream (main.cu:284-288) lives on the device
        // (msk144_push_hops); streams without a hop sit this batch out and cost nothing on the GPU
        const auto a0 = Clock::now();
        WindowDecoder::HopStage& hs = stage_[b.slot];
        for(int c = 0; c < nch_; c++)
        {
            Stream& s = st_[c];
            if(!s.ready) continue;
            const size_t j = b.streams.size();
            if(s.first)
            {
                memcpy(hs.first_halves + half_ * j, s.pending.data(), half_);
                memcpy(hs.hops + half_ * j, s.pending.data() + half_, half_);
            }
            else memcpy(hs.hops + half_ * j, s.pending.data(), half_);
            hs.streams[j] = c;
            hs.is_first[j] = s.first ? 1 : 0;
            b.streams.push_back(c);
            b.ready_at.push_back(s.ready_at);
            s.first = false;
            s.pending.clear();
            s.ready = false;
        }
        const auto a1 = Clock::now();
        if(!dec_.submit_hops(b.slot, static_cast<int>(b.streams.size())))
        {
            fail(dec_.error());
            break;
        }
        b.assemble_ms = ms_between(a0, a1);
        b.submit_ms = ms_between(a1, Clock::now());
        stats_.ingest.add(ingest_busy_ms);  // read by the caller only after join()
        ingest_busy_ms = 0.0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            in_flight_.push_back(std::move(b));
        }
        cv_.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu_);
        no_more_ = true;
    }
    cv_.notify_all();
}

void DeviceLoop::post_main()
{
    std::vector<std::vector<FilteredResult>> out;
    std::vector<int> ids;
    std::vector<const std::vector<FilteredResult>*> lines;
    while(true)
    {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !in_flight_.empty() || no_more_ || failed_; });
            if(in_flight_.empty()) return;
            b = std::move(in_flight_.front());
            in_flight_.pop_front();
        }
        HopTiming ht;
        if(!dec_.collect(b.slot, out, &ht))
        {
            fail(dec_.error());
            return;
        }
        if(ht.overflow)
        {
            // The reference keeps every ResultItem and cannot fail this way (result_keeper.cuh:85-115); here the compact list is
            // finite: the records that fitted are processed, the hop is reported, the streams keep running.
            stats_.overflowed_hops++;
            out_.log("msk144hipdecoder: device " + std::to_string(device_) + ": a hop of " + std::to_string(b.streams.size()) + " streams held more decodes than the result list (" +
                     std::to_string(ht.records) + " kept; raise --max-results): the list was cut, decoding goes on");
        }
        const auto p0 = Clock::now();
        ids.clear();
        lines.clear();
        for(int c : b.streams)
        {
            ids.push_back(base_ + c);
            lines.push_back(&out[c]);
        }
        out_.print(ids, lines, lo_.tag_channels);
        const auto p1 = Clock::now();
        const long long batch_ms = std::chrono::duration_cast<std::chrono::milliseconds>(p1 - b.go).count();
        if(batch_ms > kSoftLimitMs)
            out_.log("Warning: Working loop takes too much time: " + std::to_string(batch_ms) + " ms of " + std::to_string(kSoftLimitMs) + " ms max.");
        // per-stream deadline: from "hop complete" to "lines printed" a stream has one hop period (216 ms) before its next
        // hop is due; the reference's soft limit of 210 ms (main.cu:398-403) is applied per stream
        for(size_t j = 0; j < b.streams.size(); j++)
        {
            Stream& s = st_[b.streams[j]];
            const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(p1 - b.ready_at[j]).count();
            s.hops++;
            if(ms > kSoftLimitMs) s.late++;
            if(ms > s.worst_ms) s.worst_ms = ms;
        }
        stats_.assemble.add(b.assemble_ms);
        stats_.submit.add(b.submit_ms);
        stats_.wait.add(ht.wait_ms);
        stats_.post.add(ht.post_ms);
        stats_.print.add(ms_between(p0, p1));
        stats_.latency.add(ms_between(b.go, p1));
        stats_.records.add(ht.records);
        stats_.batches++;
        {
            std::lock_guard<std::mutex> lk(mu_);
            free_slots_.push_back(b.slot);
        }
        cv_.notify_all();
    }
}

}  // namespace msk144host
