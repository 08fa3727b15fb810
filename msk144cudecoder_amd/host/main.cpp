// msk144hipdecoder - stdin -> stdout MSK144 stream decoder for AMD Instinct MI355X.
// Same command-line options, input framing, help text, stderr parameter block and output lines as the reference program
// (main.cu:55-426), with the GPU work behind libmsk144hip.so.  Beyond the reference: several raw
// streams (files or FIFOs) decoded as ONE GPU batch per hop, read without blocking so that a stalled stream never holds
// the others back, with per-stream hop-deadline accounting (the reference's 210 ms watchdog, per batch and per stream).
#include "window_decoder.h"

#include <fcntl.h>
#include <getopt.h>
#include <poll.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
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
    std::cout << "                   --hop-timeout-ms=N          With --inputs: how long a batch waits for lagging streams once the first one has a hop ready. Default=216 (one hop)." << std::endl;
    std::cout << "                   --skip-wav-header           Drop the first 44 bytes of every stream (the reference decodes a RIFF header as 22 samples). Default off." << std::endl;
    std::cout << "                   --reference-decode-cache    Reproduce the reference's per-window text cache, whose comparator is always false: every decode of a window prints the text of the first one. Default: each distinct payload gets its own text." << std::endl;
    std::cout << "                   --strict-decode             Accepted for compatibility (this is the default now)." << std::endl;
    std::cout << "                   --print-bits                Append the 77-bit payload to each output line." << std::endl;
    std::cout << "                   --device=N                  HIP device ordinal. Default=0." << std::endl;
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
    bool eof = false;
    bool first = true;            // next hop is the 5184-sample fill (main.cu:271-283), later ones 2592 (:284-294)
    size_t skip = 0;              // header bytes still to drop
    std::vector<unsigned char> pending;
    bool ready = false;           // a complete hop sits in `pending`
    Clock::time_point ready_at;
    // deadline accounting
    long hops = 0, late = 0;
    long long worst_ms = 0;
};

}  // namespace

int main(int argc, char* const argv[])
{
    DecoderOptions opt;
    bool center_set = false;
    bool skip_wav = false;
    int hop_timeout_ms = 216;
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

    const int nch = input_paths.empty() ? 1 : static_cast<int>(input_paths.size());
    opt.channels = nch;

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
    if(nch > 1) std::cerr << "msk144hipdecoder: " << nch << " input streams per GPU batch, hop timeout " << hop_timeout_ms << " ms" << std::endl;

    const size_t sample_bytes = (opt.read_mode == 1) ? sizeof(int16_t) : 2 * sizeof(int8_t);
    const size_t win_bytes = MSK144_WINDOW_SAMPLES * sample_bytes;
    const size_t half = win_bytes / 2;
    const size_t unit = (opt.read_mode == 1) ? sizeof(int16_t) : sizeof(int8_t);  // the reference counts items of this size
    std::vector<unsigned char> ring(win_bytes * nch, 0);
    std::vector<std::vector<FilteredResult>> lines;

    if(input_paths.empty())
    {
        // ---- the reference's loop: one stream on stdin, blocking reads (main.cu:261-422) ----
        if(skip_wav)
        {
            unsigned char hdr[44];
            if(fread(hdr, 1, sizeof(hdr), stdin) != sizeof(hdr)) std::cerr << "Incomplete read error. rc=0" << std::endl;
        }
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

    // ---- several streams, one GPU batch per hop, non-blocking ingest ----
    std::vector<Stream> st(nch);
    for(int c = 0; c < nch; c++)
    {
        // O_NONBLOCK: opening a FIFO whose writer has not arrived yet returns at once, and read() never parks the batch
        st[c].fd = open(input_paths[c].c_str(), O_RDONLY | O_NONBLOCK);
        if(st[c].fd < 0)
        {
            std::cerr << "Cannot open input " << input_paths[c] << std::endl;
            return 2;
        }
        st[c].skip = skip_wav ? 44 : 0;
        st[c].pending.reserve(win_bytes);
    }
    std::vector<pollfd> pfd(nch);
    std::vector<bool> active(nch, false);
    std::vector<unsigned char> chunk(1 << 16);
    long batches = 0;

    while(true)
    {
        // 1. drain whatever every open stream has, up to one hop each
        int open_streams = 0, ready = 0;
        for(int c = 0; c < nch; c++)
        {
            Stream& s = st[c];
            if(s.eof) continue;
            open_streams++;
            const size_t need = s.first ? win_bytes : half;
            while(!s.ready)
            {
                const size_t room = s.skip ? (s.skip < chunk.size() ? s.skip : chunk.size()) : need - s.pending.size();
                const ssize_t got = read(s.fd, chunk.data(), room < chunk.size() ? room : chunk.size());
                if(got > 0)
                {
                    if(s.skip) s.skip -= static_cast<size_t>(got);
                    else s.pending.insert(s.pending.end(), chunk.begin(), chunk.begin() + got);
                    if(!s.skip && s.pending.size() == need)
                    {
                        s.ready = true;
                        s.ready_at = Clock::now();
                    }
                    continue;
                }
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
                break;
            }
            if(s.ready) ready++;
        }
        if(open_streams == 0 && ready == 0) break;

        // 2. batch policy: go when every open stream has its hop, or when the oldest ready hop has waited hop_timeout_ms
        bool go = ready > 0 && ready == open_streams;
        if(!go && ready > 0)
        {
            Clock::time_point oldest = Clock::now();
            for(const Stream& s : st)
                if(s.ready && s.ready_at < oldest) oldest = s.ready_at;
            go = std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - oldest).count() >= hop_timeout_ms;
        }
        if(!go)
        {
            int n = 0;
            for(int c = 0; c < nch; c++)
                if(!st[c].eof && !st[c].ready) pfd[n++] = {st[c].fd, POLLIN, 0};
            poll(pfd.data(), static_cast<nfds_t>(n), ready > 0 ? 5 : 50);
            continue;
        }

        // 3. advance the window ring of the streams that have a hop; the others keep their state and sit this batch out
        for(int c = 0; c < nch; c++)
        {
            Stream& s = st[c];
            active[c] = s.ready;
            if(!s.ready) continue;
            unsigned char* w = ring.data() + win_bytes * c;
            if(s.first) memcpy(w, s.pending.data(), win_bytes);
            else
            {
                memcpy(w, w + half, half);
                memcpy(w + half, s.pending.data(), half);
            }
            s.first = false;
            s.pending.clear();
        }
        const auto t0 = Clock::now();
        if(!dec.process(ring.data(), active, lines))
        {
            std::cerr << "msk144hip: " << dec.error() << std::endl;
            return 2;
        }
        print_lines(nch, lines);
        const auto t1 = Clock::now();
        warn_if_late(std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
        batches++;
        // per-stream deadline: from "hop complete" to "lines printed" a stream has one hop period (216 ms) before its next
        // hop is due; the reference's soft limit of 210 ms is applied per stream
        for(int c = 0; c < nch; c++)
        {
            Stream& s = st[c];
            if(!s.ready) continue;
            const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(t1 - s.ready_at).count();
            s.hops++;
            if(ms > 210) s.late++;
            if(ms > s.worst_ms) s.worst_ms = ms;
            s.ready = false;
        }
    }

    long total_hops = 0, total_late = 0;
    long long worst = 0;
    for(int c = 0; c < nch; c++)
    {
        total_hops += st[c].hops;
        total_late += st[c].late;
        if(st[c].worst_ms > worst) worst = st[c].worst_ms;
        if(st[c].late) std::cerr << "ch=" << c << ": " << st[c].late << " of " << st[c].hops << " hops answered later than 210 ms (worst " << st[c].worst_ms << " ms)" << std::endl;
        close(st[c].fd);
    }
    std::cerr << "msk144hipdecoder: " << batches << " batches, " << total_hops << " stream hops, " << total_late << " late, worst latency " << worst << " ms" << std::endl;
    std::cout << "Done" << std::endl;
    return 0;
}
