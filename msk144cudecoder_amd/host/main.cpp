// msk144hipdecoder - stdin -> stdout MSK144 stream decoder for AMD Instinct MI355X.
// Same command-line options, input framing, help text, stderr parameter block and output lines as the reference program
// (main.cu:55-426), with the GPU work behind libmsk144hip.so.  Beyond the reference: several raw
// streams (files or FIFOs) decoded as ONE GPU batch per hop, read without blocking so that a stalled stream never holds
// the others back, with per-stream hop-deadline accounting (the reference's 210 ms watchdog, per batch and per stream).  The
// multi-stream loop is pipelined over the library's two pinned staging slots: this thread reads the streams and submits hop n+1
// while the GPU decodes hop n and a second thread turns the records of hop n-1 into text.
#include "window_decoder.h"

#include <fcntl.h>
#include <getopt.h>
#include <poll.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace msk144host;

namespace
{

using Clock = std::chrono::steady_clock;

// The reference's help, verbatim (main.cu:58-67; its defaults differ from the code's: 100/3/2 here, 200/4/1 in effect -
// kept as printed), followed by what this program adds.
void show_help(const char* prog)
{
    // clang-format off
    std::cout << "Calling conversion: " << prog << " {[--help] | <options> }" << std::endl;
    std::cout << " Where options are: " << std::endl;
    std::cout << "                   --help                      Show this help and exit." << std::endl;
    std::cout << "                   --center-frequency=1500.0   Center frequency in Hz." << std::endl;
    std::cout << "                   --search-step=2.0           Search step in Hz. " << std::endl;
    std::cout << "                   --search-width=100.0        Window in Hz around center frequency to find msk144 signal in. The more Search Width the more GPU resources are needed." << std::endl;
    std::cout << "                   --scan-depth=[1..8]         The more depth the more averagable patterns will be tried. Default=3" << std::endl;
    std::cout << "                   --read-mode=[1|2]           1=Audio,16 bit, mono, 12000sps, 1500Hz-recommended center; 2=IQ,8 bit, 12000sps, 0Hz-center. Default mode = 1." << std::endl;
    std::cout << "                   --analytic-method=[1|2]     How to convert real signal to ananlytyc quadrature signal. 1 = FFT; 2 = Shift-left + LPF + Shift-right. Default=2." << std::endl;
    std::cout << "                   --nbadsync-threshold=[1..4] Specifies how many errors in sync pattern are acceptable to be passed to LDPC decoder. Default=2." << std::endl;
    std::cout << " Additions of msk144hipdecoder (defaults in effect, as in the reference's code: search-width 200, scan-depth 4, nbadsync-threshold 1):" << std::endl;
    std::cout << "                   --inputs=F1,F2,...          Decode several raw streams (files or FIFOs) as one GPU batch per hop instead of stdin; lines then carry ch=<index>." << std::endl;
    std::cout << "                   --inputs-file=PATH          The same, one stream path per line (for hundreds of streams)." << std::endl;
    std::cout << "                   --interleaved=N             Decode N streams that arrive on stdin as ONE interleaved stream - a block of N x 5184 samples (stream after stream), then blocks of N x 2592 per hop - as one GPU batch per hop; lines carry ch=<index>." << std::endl;
    std::cout << "                   --hop-timeout-ms=N          With --inputs: how long a batch waits for further streams once the first one has a hop ready (a batch costs what its streams cost, so small batches are cheap and keep the latency of each stream low). Default=20." << std::endl;
    std::cout << "                   --connect-timeout-ms=N      With --inputs: how long a FIFO may stay without a writer before it counts as ended. Default=10000." << std::endl;
    std::cout << "                   --skip-wav-header           Drop the first 44 bytes of every stream (the reference decodes a RIFF header as 22 samples). Default off." << std::endl;
    std::cout << "                   --reference-decode-cache    Reproduce the reference's per-window text cache, whose comparator is always false: every decode of a window prints the text of the first one. Default: each distinct payload gets its own text." << std::endl;
    std::cout << "                   --strict-decode             Accepted for compatibility (this is the default now)." << std::endl;
    std::cout << "                   --print-bits                Append the 77-bit payload to each output line." << std::endl;
    std::cout << "                   --device=N                  HIP device ordinal. Default=0." << std::endl;
    std::cout << "                   --timing                    With --inputs: per-hop host and device time split (ingest, H2D, GPU, D2H, post-processing) on stderr at the end." << std::endl;
    // clang-format on
}

const char* mode_name(int mode)
{
    if(mode == 1) return "Audio. 16 bits signed.";
    if(mode == 2) return "IQ. 8+8 bits.";
    return "unknown";
}

void split_list(const std::string& list, std::vector<std::string>& out)
{
    size_t a = 0;
    while(a <= list.size())
    {
        const size_t b = list.find(',', a);
        const std::string item = list.substr(a, b == std::string::npos ? std::string::npos : b - a);
        if(!item.empty()) out.push_back(item);
        if(b == std::string::npos) break;
        a = b + 1;
    }
}

void warn_if_late(long long ms)
{
    const int soft_limit_ms = 210;  // of the 216 ms a hop lasts (main.cu:398-403)
    if(ms > soft_limit_ms)
    {
        std::cerr << "Warning: Working loop takes too much time: " << ms << " ms"
                  << " of " << soft_limit_ms << " ms max." << std::endl;
    }
}

void print_lines(int nch, const std::vector<std::vector<FilteredResult>>& lines)
{
    for(int c = 0; c < nch; c++)
    {
        for(const FilteredResult& l : lines[c])
        {
            if(nch == 1)
            {
                std::cout << l.format_line() << std::endl;
            }
            else
            {
                const std::string line = l.format_line();  // "***  snr=..." -> "***  ch=<c>; snr=..."
                std::cout << line.substr(0, 5) << "ch=" << c << "; " << line.substr(5) << std::endl;
            }
        }
    }
}

// One --inputs stream: a non-blocking descriptor and the bytes read so far towards its next hop.
struct Stream
{
    int fd = -1;
    bool fifo = false;            // a FIFO reads 0 bytes while no writer has connected yet: that is not its end
    bool connected = false;       // first byte seen
    bool eof = false;
    bool readable = true;         // worth a read(): set by poll(), cleared when a read would block (a regular file always is)
    bool first = true;            // next hop is the 5184-sample fill (main.cu:271-283), later ones 2592 (:284-294)
    size_t skip = 0;              // header bytes still to drop
    std::vector<unsigned char> pending;
    bool ready = false;           // a complete hop sits in `pending`
    Clock::time_point ready_at;
    // deadline accounting (owned by the post-processing thread)
    long hops = 0, late = 0;
    long long worst_ms = 0;
};

// One submitted hop of every ready stream, handed from the ingest thread to the post-processing thread.
struct Batch
{
    int slot = 0;
    int n = 0;                                 // streams that have a hop in this batch
    std::vector<int> streams;                  // which ones, ascending
    std::vector<Clock::time_point> ready_at;  // when each of them had its hop complete
    Clock::time_point go;                      // batch released by the policy
    double assemble_ms = 0.0, submit_ms = 0.0;
};

struct Accumulator
{
    double sum = 0.0, worst = 0.0;
    long n = 0;
    void add(double v)
    {
        sum += v;
        worst = std::max(worst, v);
        n++;
    }
    double mean() const { return n ? sum / n : 0.0; }
};

double ms_between(Clock::time_point a, Clock::time_point b)
{
    return std::chrono::duration<double, std::milli>(b - a).count();
}

}  // namespace

int main(int argc, char* const argv[])
{
    DecoderOptions opt;
    bool center_set = false;
    bool skip_wav = false;
    int hop_timeout_ms = 20;
    int connect_timeout_ms = 10000;
    int interleaved = 0;
    bool timing = false;
    std::vector<std::string> input_paths;

    static struct option long_options[] = {{"help", no_argument, 0, 0},
                                           {"center-frequency", required_argument, 0, 0},
                                           {"search-step", required_argument, 0, 0},
                                           {"search-width", required_argument, 0, 0},
                                           {"scan-depth", required_argument, 0, 0},
                                           {"read-mode", required_argument, 0, 0},
                                           {"analytic-method", required_argument, 0, 0},
                                           {"nbadsync-threshold", required_argument, 0, 0},
                                           {"strict-decode", no_argument, 0, 0},
                                           {"print-bits", no_argument, 0, 0},
                                           {"device", required_argument, 0, 0},
                                           {"inputs", required_argument, 0, 0},
                                           {"reference-decode-cache", no_argument, 0, 0},
                                           {"skip-wav-header", no_argument, 0, 0},
                                           {"hop-timeout-ms", required_argument, 0, 0},
                                           {"inputs-file", required_argument, 0, 0},
                                           {"timing", no_argument, 0, 0},
                                           {"connect-timeout-ms", required_argument, 0, 0},
                                           {"interleaved", required_argument, 0, 0},
                                           {0, 0, 0, 0}};
    while(true)
    {
        int idx = 0;
        const int c = getopt_long(argc, argv, "", long_options, &idx);
        if(c == -1) break;
        if(c != 0) continue;  // unknown option: getopt has printed its own message, carry on like the reference
        switch(idx)
        {
        case 0: show_help(argv[0]); return 0;
        case 1: opt.center_hz = static_cast<float>(atof(optarg)); center_set = true; break;
        case 2: opt.step_hz = static_cast<float>(atof(optarg)); break;
        case 3: opt.width_hz = static_cast<float>(atof(optarg)); break;
        case 4: opt.scan_depth = atoi(optarg); break;
        case 5: opt.read_mode = atoi(optarg); break;
        case 6: opt.analytic_method = atoi(optarg); break;
        case 7: opt.nbadsync_threshold = atoi(optarg); break;
        case 8: opt.reference_cache_quirk = false; break;
        case 9: opt.print_bits = true; break;
        case 10: opt.device = atoi(optarg); break;
        case 11: split_list(optarg, input_paths); break;
        case 12: opt.reference_cache_quirk = true; break;
        case 13: skip_wav = true; break;
        case 14: hop_timeout_ms = atoi(optarg); break;
        case 15:
        {
            std::ifstream f(optarg);
            if(!f)
            {
                std::cerr << "Cannot open input list " << optarg << std::endl;
                return 2;
            }
            for(std::string line; std::getline(f, line);)
                if(!line.empty()) input_paths.push_back(line);
            break;
        }
        case 16: timing = true; break;
        case 17: connect_timeout_ms = atoi(optarg); break;
        case 18: interleaved = atoi(optarg); break;
        default: show_help(argv[0]); return 0;
        }
    }

    if(!center_set)
    {
        if(opt.read_mode == 1) opt.center_hz = 1500.0f;
        else if(opt.read_mode == 2) opt.center_hz = 0.0f;
        else
        {
            std::cerr << "Wrong read mode " << opt.read_mode << std::endl;
            return 2;
        }
    }
    if(opt.read_mode != 1 && opt.read_mode != 2)
    {
        // reached only with an explicit centre frequency: the reference enters its loop, reports the mode and stops
        std::cerr << "Unsupported mode. Exit." << std::endl;
        std::cout << "Done" << std::endl;
        return 0;
    }

    if(interleaved < 0 || (interleaved > 0 && !input_paths.empty()))
    {
        std::cerr << "--interleaved=N takes N >= 1 streams from stdin and excludes --inputs" << std::endl;
        return 2;
    }
    const bool batched = interleaved > 0 || !input_paths.empty();
    const int nch = interleaved > 0 ? interleaved : (input_paths.empty() ? 1 : static_cast<int>(input_paths.size()));
    opt.channels = nch;
    opt.profile = timing && batched;
    {
        // one descriptor per stream plus what the runtime opens: lift the soft limit when the hard limit allows
        rlimit lim{};
        const rlim_t want = static_cast<rlim_t>(nch) + 256;
        if(getrlimit(RLIMIT_NOFILE, &lim) == 0 && lim.rlim_cur < want)
        {
            lim.rlim_cur = (lim.rlim_max == RLIM_INFINITY || lim.rlim_max > want) ? want : lim.rlim_max;
            setrlimit(RLIMIT_NOFILE, &lim);
        }
    }

    WindowDecoder dec(opt);
    if(!dec.ok())
    {
        std::cerr << "msk144hip: " << dec.error() << std::endl;
        return 2;
    }

    // The reference's parameter block (main.cu:233-252), line for line.  Its four launch-geometry lines are kept with the
    // values the reference would print for these options (F blocks x 256 threads; F*(depth*8) blocks x 160 threads); the
    // geometry actually used here follows under its own names.
    const int F = dec.num_freqs(), D = dec.scan_depth();
    std::cerr << "Actual parameters:" << std::endl
              << "Center Frequency: " << opt.center_hz << "Hz" << std::endl
              << "Search Step: " << opt.step_hz << "Hz" << std::endl
              << "Search Width: " << opt.width_hz << "Hz" << std::endl
              << "Scan Depth: " << D << std::endl
              << "Left Boundary: " << dec.left_bound() << "Hz" << std::endl
              << "Right Boundary: " << dec.right_bound() << "Hz" << std::endl
              << "Read Mode: (" << mode_name(opt.read_mode) << ")" << std::endl;
    if(opt.read_mode == 1) std::cerr << "Analytic Method: " << opt.analytic_method << std::endl;
    std::cerr << "Badsync Threshold: " << opt.nbadsync_threshold << std::endl
              << "Scan-kernel CUDA blocks: " << F << std::endl
              << "Scan-kernel CUDA threads: " << 256 << std::endl
              << "Softbit-kernel CUDA blocks: " << F << "*" << D * 8 << "=" << (F * D * 8) << std::endl
              << "Softbit-kernel CUDA threads: " << 160 << std::endl
              << std::endl;
    std::cerr << "msk144hipdecoder: " << F << " frequency hypotheses x " << D << " patterns x 8 = " << F * D * 8 << " candidates per window; HIP workgroups per window: scan "
              << F << " x 512, softbits " << F << " x 512, LDPC one wave per gated candidate" << std::endl;
    if(batched) std::cerr << "msk144hipdecoder: " << nch << " input streams per GPU batch" << (interleaved > 0 ? " (interleaved on stdin)" : "") << ", hop timeout " << hop_timeout_ms << " ms" << std::endl;

    const size_t sample_bytes = (opt.read_mode == 1) ? sizeof(int16_t) : 2 * sizeof(int8_t);
    const size_t win_bytes = MSK144_WINDOW_SAMPLES * sample_bytes;
    const size_t half = win_bytes / 2;
    const size_t unit = (opt.read_mode == 1) ? sizeof(int16_t) : sizeof(int8_t);  // the reference counts items of this size
    std::vector<std::vector<FilteredResult>> lines;

    if(!batched)
    {
        // ---- the reference's loop: one stream on stdin, blocking reads (main.cu:261-422) ----
        if(skip_wav)
        {
            unsigned char hdr[44];
            if(fread(hdr, 1, sizeof(hdr), stdin) != sizeof(hdr)) std::cerr << "Incomplete read error. rc=0" << std::endl;
        }
        std::vector<unsigned char> ring(win_bytes, 0);
        bool first = true;
        const std::vector<bool> active(1, true);
        while(true)
        {
            unsigned char* w = ring.data();
            size_t want, rc;
            if(first)
            {
                want = win_bytes / unit;
                rc = fread(w, unit, want, stdin);
            }
            else
            {
                memcpy(w, w + half, half);
                want = half / unit;
                rc = fread(w + half, unit, want, stdin);
            }
            first = false;
            if(rc != want)
            {
                std::cerr << "Incomplete read error. rc=" << rc << std::endl;
                break;
            }
            const auto t0 = Clock::now();
            if(!dec.process(ring.data(), active, lines))
            {
                std::cerr << "msk144hip: " << dec.error() << std::endl;
                return 2;
            }
            warn_if_late(std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - t0).count());
            print_lines(1, lines);
        }
        std::cout << "Done" << std::endl;
        return 0;
    }

    // ---- several streams, one GPU batch per hop, non-blocking ingest, pipelined over two staging slots ----
    std::vector<Stream> st(nch);
    for(int c = 0; c < nch && interleaved > 0; c++)
    {
        st[c].pending.reserve(win_bytes);
    }
    for(int c = 0; c < nch && interleaved == 0; c++)
    {
        // O_NONBLOCK: opening a FIFO whose writer has not arrived yet returns at once, and read() never parks the batch
        st[c].fd = open(input_paths[c].c_str(), O_RDONLY | O_NONBLOCK);
        if(st[c].fd < 0)
        {
            std::cerr << "Cannot open input " << input_paths[c] << ": " << strerror(errno) << std::endl;
            return 2;
        }
        st[c].skip = skip_wav ? 44 : 0;
        st[c].pending.reserve(win_bytes);
        struct stat sb{};
        st[c].fifo = fstat(st[c].fd, &sb) == 0 && S_ISFIFO(sb.st_mode);
    }
    const auto opened_at = Clock::now();
    WindowDecoder::HopStage stage[WindowDecoder::kSlots];
    for(int k = 0; k < WindowDecoder::kSlots; k++)
    {
        if(!dec.hop_stage(k, stage[k]))  // pinned, owned by the library handle; the stream windows themselves live on the device
        {
            std::cerr << "msk144hip: " << dec.error() << std::endl;
            return 2;
        }
    }

    // hand-over between this (ingest + submit) thread and the post-processing thread
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Batch> in_flight;
    std::deque<int> free_slots;
    for(int k = 0; k < WindowDecoder::kSlots; k++) free_slots.push_back(k);
    bool no_more = false, failed = false;
    Accumulator t_assemble, t_submit, t_wait, t_post, t_print, t_latency, t_records;
    long batches = 0;

    std::thread post([&]() {
        std::vector<std::vector<FilteredResult>> out;
        while(true)
        {
            Batch b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !in_flight.empty() || no_more; });
                if(in_flight.empty()) return;
                b = std::move(in_flight.front());
                in_flight.pop_front();
            }
            HopTiming ht;
            if(!dec.collect(b.slot, out, &ht))
            {
                std::cerr << "msk144hip: " << dec.error() << std::endl;
                std::lock_guard<std::mutex> lk(mu);
                failed = true;
                cv.notify_all();
                return;
            }
            const auto p0 = Clock::now();
            print_lines(nch, out);
            const auto p1 = Clock::now();
            warn_if_late(std::chrono::duration_cast<std::chrono::milliseconds>(p1 - b.go).count());
            // per-stream deadline: from "hop complete" to "lines printed" a stream has one hop period (216 ms) before its next
            // hop is due; the reference's soft limit of 210 ms (main.cu:398-403) is applied per stream
            for(size_t j = 0; j < b.streams.size(); j++)
            {
                Stream& s = st[b.streams[j]];
                const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(p1 - b.ready_at[j]).count();
                s.hops++;
                if(ms > 210) s.late++;
                if(ms > s.worst_ms) s.worst_ms = ms;
            }
            t_assemble.add(b.assemble_ms);
            t_submit.add(b.submit_ms);
            t_wait.add(ht.wait_ms);
            t_post.add(ht.post_ms);
            t_print.add(ms_between(p0, p1));
            t_latency.add(ms_between(b.go, p1));
            t_records.add(ht.records);
            batches++;
            {
                std::lock_guard<std::mutex> lk(mu);
                free_slots.push_back(b.slot);
            }
            cv.notify_all();
        }
    });
    auto finish = [&](int code) {
        {
            std::lock_guard<std::mutex> lk(mu);
            no_more = true;
        }
        cv.notify_all();
        post.join();
        return code;
    };

    std::vector<pollfd> pfd(nch);
    std::vector<int> pfd_stream(nch);
    std::vector<unsigned char> chunk(1 << 16);
    std::vector<unsigned char> block;  // --interleaved: one hop of every stream
    bool skip_block = skip_wav;
    Accumulator t_ingest;
    auto ingest_since = Clock::now();
    double ingest_busy_ms = 0.0;

    while(true)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if(failed) break;
        }
        // 1. drain whatever every open stream has, up to one hop each
        const auto d0 = Clock::now();
        int open_streams = 0, ready = 0;
        if(interleaved > 0 && !st[0].eof)
        {
            // one block per hop on stdin: the hop of stream 0, then of stream 1, ... (blocking read, like the reference's fread)
            const size_t need = st[0].first ? win_bytes : half;
            block.resize(need * nch);
            if(skip_block)
            {
                unsigned char hdr[44];
                if(fread(hdr, 1, sizeof(hdr), stdin) != sizeof(hdr)) std::cerr << "Incomplete read error. rc=0" << std::endl;
                skip_block = false;
            }
            const size_t got = fread(block.data(), 1, block.size(), stdin);
            if(got != block.size())
            {
                std::cerr << "Incomplete read error. rc=" << got / unit << std::endl;
                for(Stream& s : st) s.eof = true;
            }
            else
            {
                const auto now = Clock::now();
                for(int c = 0; c < nch; c++)
                {
                    st[c].pending.assign(block.begin() + static_cast<long>(need * c), block.begin() + static_cast<long>(need * (c + 1)));
                    st[c].ready = true;
                    st[c].ready_at = now;
                }
            }
        }
        for(int c = 0; c < nch && interleaved > 0; c++)
        {
            if(!st[c].eof) open_streams++;
            if(st[c].ready) ready++;
        }
        for(int c = 0; c < nch && interleaved == 0; c++)
        {
            Stream& s = st[c];
            if(s.eof) continue;
            open_streams++;
            const size_t need = s.first ? win_bytes : half;
            // only streams poll() reported (or never asked about) are read: at thousands of streams the read() calls that would
            // just say EAGAIN were most of the ingest time.  A FIFO still waiting for its writer is probed every round.
            if(!s.readable && !(s.fifo && !s.connected)) continue;
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
                if(got == 0 && s.fifo && !s.connected && ms_between(opened_at, Clock::now()) < connect_timeout_ms)
                    break;  // no writer on this FIFO yet: read() reports 0 bytes, which only means "nobody there so far"
                if(got == 0)
                {
                    // writer closed: what is left is a short read, exactly the reference's end-of-stream message
                    std::cerr << "ch=" << c << ": Incomplete read error. rc=" << s.pending.size() / unit << std::endl;
                    s.eof = true;
                    open_streams--;
                }
                else if(errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)
                {
                    std::cerr << "ch=" << c << ": read error: " << strerror(errno) << std::endl;
                    s.eof = true;
                    open_streams--;
                }
                else if(errno != EINTR) s.readable = false;  // drained: wait for poll() to say otherwise
                break;
            }
        }
        for(int c = 0; c < nch && interleaved == 0; c++)
            if(st[c].ready) ready++;
        ingest_busy_ms += ms_between(d0, Clock::now());
        if(open_streams == 0 && ready == 0) break;

        // 2. batch policy: go when every open stream has its hop, or when the oldest ready hop has waited hop_timeout_ms
        bool go = ready > 0 && ready >= open_streams;
        if(!go && ready > 0)
        {
            Clock::time_point oldest = Clock::now();
            for(const Stream& s : st)
                if(s.ready && s.ready_at < oldest) oldest = s.ready_at;
            go = std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - oldest).count() >= hop_timeout_ms;
        }
        if(!go)
        {
            // sleep until more data arrives; a FIFO nobody writes to yet polls as hung-up at once, so it is left out of the set
            int n = 0;
            for(int c = 0; c < nch; c++)
                if(!st[c].eof && !st[c].ready && !st[c].readable && (st[c].connected || !st[c].fifo))
                {
                    pfd[n] = {st[c].fd, POLLIN, 0};
                    pfd_stream[n++] = c;
                }
            if(poll(pfd.data(), static_cast<nfds_t>(n), ready > 0 ? 5 : (n > 0 ? 50 : 10)) > 0)
                for(int k = 0; k < n; k++)
                    if(pfd[k].revents) st[pfd_stream[k]].readable = true;  // data, hang-up or error: the next read() tells which
            continue;
        }

        // 3. a free staging slot (back-pressure: with both slots in flight the streams wait in their pipes)
        Batch b;
        b.go = Clock::now();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !free_slots.empty() || failed; });
            if(failed) break;
            b.slot = free_slots.front();
            free_slots.pop_front();
        }
        // 4. hand the new samples of every stream that has a hop to the library, packed back to back in the pinned slot: 2592 per
        // stream (all 5184 of a stream's first hop).  The 50 %-overlap window of each stream (main.cu:284-288) lives on the device
        // (msk144_push_hops); streams without a hop sit this batch out and cost nothing on the GPU
        const auto a0 = Clock::now();
        WindowDecoder::HopStage& hs = stage[b.slot];
        for(int c = 0; c < nch; c++)
        {
            Stream& s = st[c];
            if(!s.ready) continue;
            const size_t j = b.streams.size();
            if(s.first)
            {
                memcpy(hs.first_halves + half * j, s.pending.data(), half);
                memcpy(hs.hops + half * j, s.pending.data() + half, half);
            }
            else memcpy(hs.hops + half * j, s.pending.data(), half);
            hs.streams[j] = c;
            hs.is_first[j] = s.first ? 1 : 0;
            b.streams.push_back(c);
            b.ready_at.push_back(s.ready_at);
            s.first = false;
            s.pending.clear();
            s.ready = false;
        }
        b.n = static_cast<int>(b.streams.size());
        const auto a1 = Clock::now();
        if(!dec.submit_hops(b.slot, b.n))
        {
            std::cerr << "msk144hip: " << dec.error() << std::endl;
            return finish(2);
        }
        b.assemble_ms = ms_between(a0, a1);
        b.submit_ms = ms_between(a1, Clock::now());
        t_ingest.add(ingest_busy_ms);
        ingest_busy_ms = 0.0;
        {
            std::lock_guard<std::mutex> lk(mu);
            in_flight.push_back(std::move(b));
        }
        cv.notify_all();
    }
    finish(0);
    if(failed) return 2;

    long total_hops = 0, total_late = 0;
    long long worst = 0;
    for(int c = 0; c < nch; c++)
    {
        total_hops += st[c].hops;
        total_late += st[c].late;
        if(st[c].worst_ms > worst) worst = st[c].worst_ms;
        if(st[c].late) std::cerr << "ch=" << c << ": " << st[c].late << " of " << st[c].hops << " hops answered later than 210 ms (worst " << st[c].worst_ms << " ms)" << std::endl;
        if(st[c].fd >= 0) close(st[c].fd);
    }
    std::cerr << "msk144hipdecoder: " << batches << " batches, " << total_hops << " stream hops, " << total_late << " late, worst latency " << worst << " ms" << std::endl;
    if(timing)
    {
        const double wall_s = ms_between(ingest_since, Clock::now()) * 1e-3;
        auto row = [](const char* name, const Accumulator& a) {
            fprintf(stderr, "msk144hipdecoder timing: %-34s mean %9.3f ms  max %9.3f ms\n", name, a.mean(), a.worst);
        };
        fprintf(stderr, "msk144hipdecoder timing: %d streams, %ld batches in %.2f s wall; per batch (host wall time):\n", nch, batches, wall_s);
        row("ingest (read syscalls, all streams)", t_ingest);
        row("copy new hops into pinned slot", t_assemble);
        row("submit (3 asynchronous calls)", t_submit);
        row("wait for GPU + D2H (post thread)", t_wait);
        row("post-processing (text, SNR, filter)", t_post);
        row("print", t_print);
        row("batch released -> lines printed", t_latency);
        fprintf(stderr, "msk144hipdecoder timing: records per batch mean %.0f max %.0f\n", t_records.mean(), t_records.worst);
        float dev[MSK144_T_COUNT];
        if(dec.stage_times(dev))
            fprintf(stderr, "msk144hipdecoder timing: device per batch (HIP events): H2D %.3f  front end %.3f  scan %.3f  softbits %.3f  index %.3f  LDPC %.3f  collect %.3f  D2H %.3f ms\n",
                    dev[MSK144_T_H2D], dev[MSK144_T_FRONTEND], dev[MSK144_T_SCAN], dev[MSK144_T_SOFTBITS], dev[MSK144_T_INDEX], dev[MSK144_T_LDPC], dev[MSK144_T_COLLECT], dev[MSK144_T_D2H]);
    }
    std::cout << "Done" << std::endl;
    return 0;
}
