// msk144hipdecoder - stdin -> stdout MSK144 stream decoder for AMD Instinct MI355X.
// Same command-line options, input framing, help text, stderr parameter block and output lines as the reference program
// (main.cu:55-426), with the GPU work behind libmsk144hip.so.  Beyond the reference: several raw
// streams (files or FIFOs) decoded as ONE GPU batch per hop, read without blocking so that a stalled stream never holds
// the others back, with per-stream hop-deadline accounting (the reference's 210 ms watchdog, per batch and per stream).  The
// multi-stream loop is pipelined over the library's two pinned staging slots (stream_loop.h): an ingest thread reads the streams and
// submits hop n+1 while the GPU decodes hop n and a second thread turns the records of hop n-1 into text.  --devices=0,1,... runs
// one such loop per GPU, each on its contiguous share of the streams (the reference binds to one device, main.cu:115).
#include "stream_loop.h"

#include <getopt.h>
#include <poll.h>
#include <signal.h>
#include <sys/resource.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
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
    std::cout << "                   --inputs-file=PATH          The same, one stream path per line (for hundreds of streams)." << std::endl;
    std::cout << "                   --interleaved=N             Decode N streams that arrive on stdin as ONE interleaved stream - a block of N x 5184 samples (stream after stream), then blocks of N x 2592 per hop - as one GPU batch per hop; lines carry ch=<index>." << std::endl;
    std::cout << "                   --hop-timeout-ms=N          With --inputs: how long a batch waits for further streams once the first one has a hop ready (a batch costs what its streams cost, so small batches are cheap and keep the latency of each stream low). Default=20." << std::endl;
    std::cout << "                   --connect-timeout-ms=N      With --inputs: how long a FIFO may stay without a writer before it counts as ended. Default=10000." << std::endl;
    std::cout << "                   --skip-wav-header           Drop the first 44 bytes of every stream (the reference decodes a RIFF header as 22 samples). Default off." << std::endl;
    std::cout << "                   --strict-decode             Unpack every distinct 77-bit payload of a window on its own. Default off: like the reference, whose per-window text cache has a comparator that is always false, every decode of a window prints the text of the window's first accepted candidate (or nothing when that one does not unpack)." << std::endl;
    std::cout << "                   --reference-decode-cache    Accepted for compatibility (the reference's behaviour is the default)." << std::endl;
    std::cout << "                   --print-bits                Append the 77-bit payload to each output line." << std::endl;
    std::cout << "                   --device=N                  HIP device ordinal. Default=0." << std::endl;
    std::cout << "                   --devices=N1,N2,...|all     With --inputs/--interleaved: split the streams contiguously over these devices; each device gets its own ingest and post-processing threads, ch=<index> stays the global stream number. An ordinal may repeat (two independent loops on one GPU)." << std::endl;
    std::cout << "                   --max-results=N             Capacity of the per-hop decode list of a device (default 256 per stream + 131072). A hop that exceeds it is cut and reported; decoding goes on." << std::endl;
    std::cout << "                   --every-slot                Demodulate and decode every candidate slot on its own, as the reference does. Default off: a slot whose position folds the same frames as a lower slot of its group reports that slot's result (same output), and a candidate the nbadsync gate drops is not demodulated beyond its sync check." << std::endl;
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

// SIGINT / SIGTERM in the multi-stream modes: ask the loops to finish what is in flight and leave in order.  The handler only sets the
// flag; every place that can sleep - the --inputs poll, the --interleaved reader below and DeviceLoop::feed()'s back-pressure wait - looks at
// it at least every 50 ms, so the stop does not depend on where the signal lands.  The single-stream loop returns before the handlers
// are installed and keeps the reference's behaviour (default action).
void on_stop_signal(int)
{
    g_stop_requested.store(true, std::memory_order_relaxed);
}

// --interleaved reader: up to n bytes of stdin, waiting in slices of 50 ms so that a stop request is seen whether stdin is a slow pipe
// (poll times out) or a source that never blocks (checked before every read).  Short count = end of input or stop.
size_t read_stdin(unsigned char* dst, size_t n)
{
    size_t got = 0;
    while(got < n && !g_stop_requested.load(std::memory_order_relaxed))
    {
        pollfd p{0, POLLIN, 0};
        const int pr = poll(&p, 1, 50);
        if(pr < 0 && errno != EINTR) break;
        if(pr <= 0) continue;
        const ssize_t r = read(0, dst + got, n - got);
        if(r > 0) got += static_cast<size_t>(r);
        else if(r == 0 || (errno != EINTR && errno != EAGAIN)) break;
    }
    return got;
}

// one share of the input streams: the loop of `device` decodes global streams [first, first + count)
struct Share
{
    int device = 0, first = 0, count = 0;
};

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
    std::vector<std::string> device_list;

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
                                           {"devices", required_argument, 0, 0},
                                           {"max-results", required_argument, 0, 0},
                                           {"every-slot", no_argument, 0, 0},
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
        case 19: split_list(optarg, device_list); break;
        case 20: opt.max_results = atoi(optarg); break;
        case 21: opt.every_slot = true; break;
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

    // which device decodes which streams: contiguous shares, sizes differing by at most one
    std::vector<int> devices;
    if(device_list.size() == 1 && device_list[0] == "all")
    {
        int32_t n = 0;
        if(msk144_device_count(&n) != MSK144_OK)
        {
            std::cerr << "msk144hip: " << msk144_last_error(nullptr) << std::endl;
            return 2;
        }
        for(int d = 0; d < n; d++) devices.push_back(d);
    }
    else
        for(const std::string& d : device_list)
        {
            if(d.empty() || d.find_first_not_of("0123456789") != std::string::npos)
            {
                std::cerr << "--devices takes HIP device ordinals (0,1,...) or 'all', not '" << d << "'" << std::endl;
                return 2;
            }
            devices.push_back(atoi(d.c_str()));
        }
    if(devices.empty()) devices.push_back(opt.device);
    if(devices.size() > 1 && !batched)
    {
        std::cerr << "--devices splits the streams of --inputs / --inputs-file / --interleaved; a single stdin stream runs on one device (--device=N)" << std::endl;
        return 2;
    }
    std::vector<Share> shares(std::min<size_t>(devices.size(), static_cast<size_t>(nch)));  // a device without a stream gets no loop
    for(size_t i = 0; i < shares.size(); i++)
    {
        shares[i].device = devices[i];
        split_streams(nch, static_cast<int>(shares.size()), static_cast<int>(i), shares[i].first, shares[i].count);
    }
    opt.device = shares[0].device;
    opt.channels = shares[0].count;

    LinePrinter printer;
    LoopOptions lo;
    lo.hop_timeout_ms = hop_timeout_ms;
    lo.connect_timeout_ms = connect_timeout_ms;
    lo.skip_wav = skip_wav;
    lo.tag_channels = nch > 1;
    std::unique_ptr<WindowDecoder> single;
    std::unique_ptr<DeviceLoop> first_loop;
    if(batched) first_loop = std::make_unique<DeviceLoop>(opt, 0, lo, printer);
    else single = std::make_unique<WindowDecoder>(opt);
    WindowDecoder& dec = batched ? first_loop->decoder() : *single;
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
    if(shares.size() > 1)
        for(const Share& sh : shares) std::cerr << "msk144hipdecoder: device " << sh.device << " decodes streams " << sh.first << ".." << sh.first + sh.count - 1 << std::endl;

    const size_t sample_bytes = (opt.read_mode == 1) ? sizeof(int16_t) : 2 * sizeof(int8_t);
    const size_t win_bytes = MSK144_WINDOW_SAMPLES * sample_bytes;
    const size_t half = win_bytes / 2;
    const size_t unit = (opt.read_mode == 1) ? sizeof(int16_t) : sizeof(int8_t);  // the reference counts items of this size

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
        std::vector<FilteredResult> lines;
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
            if(!dec.process(ring.data(), lines))
            {
                std::cerr << "msk144hip: " << dec.error() << std::endl;
                return 2;
            }
            if(dec.last_hop_overflowed()) std::cerr << "msk144hipdecoder: the window held more decodes than the result list (raise --max-results): the list was cut, decoding goes on" << std::endl;
            warn_if_late(std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - t0).count());
            for(const FilteredResult& l : lines) std::cout << l.format_line() << std::endl;
        }
        std::cout << "Done" << std::endl;
        return 0;
    }

    // ---- several streams: one DeviceLoop per listed device (stream_loop.h), each with its contiguous share of the streams ----
    std::vector<std::unique_ptr<DeviceLoop>> loops;
    loops.push_back(std::move(first_loop));
    for(size_t i = 1; i < shares.size(); i++)
    {
        DecoderOptions o = opt;
        o.device = shares[i].device;
        o.channels = shares[i].count;
        loops.push_back(std::make_unique<DeviceLoop>(o, shares[i].first, lo, printer));
        if(!loops.back()->ok())
        {
            std::cerr << "msk144hip: device " << o.device << ": " << loops.back()->error() << std::endl;
            return 2;
        }
    }
    for(size_t i = 0; i < loops.size(); i++)
    {
        if(interleaved > 0) loops[i]->use_feed();
        else if(!loops[i]->open_inputs(std::vector<std::string>(input_paths.begin() + shares[i].first, input_paths.begin() + shares[i].first + shares[i].count)))
        {
            std::cerr << loops[i]->error() << std::endl;
            return 2;
        }
    }
    {
        struct sigaction sa{};
        sa.sa_handler = on_stop_signal;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGINT, &sa, nullptr);
        sigaction(SIGTERM, &sa, nullptr);
    }
    const auto run_since = Clock::now();
    for(auto& l : loops) l->start();

    if(interleaved > 0)
    {
        // one block per hop on stdin: the hop of stream 0, then of stream 1, ... (the reference's fread, but interruptible); every
        // loop receives the slice of its own streams.  A stop request ends the reader at the next block boundary at the latest: the
        // blocks already handed over are decoded and printed.
        if(skip_wav)
        {
            unsigned char hdr[44];
            if(read_stdin(hdr, sizeof(hdr)) != sizeof(hdr) && !g_stop_requested.load()) printer.log("Incomplete read error. rc=0");
        }
        std::vector<unsigned char> block;
        bool first_block = true;
        while(!g_stop_requested.load(std::memory_order_relaxed))
        {
            const size_t need = first_block ? win_bytes : half;
            block.resize(need * nch);
            const size_t got = read_stdin(block.data(), block.size());
            if(g_stop_requested.load(std::memory_order_relaxed)) break;  // a partial block is dropped, like a partial hop of --inputs
            if(got != block.size())
            {
                printer.log("Incomplete read error. rc=" + std::to_string(got / unit));
                break;
            }
            bool alive = true;
            for(size_t i = 0; i < loops.size() && alive; i++) alive = loops[i]->feed(block.data() + need * shares[i].first, need);
            if(!alive) break;
            first_block = false;
        }
        for(auto& l : loops) l->feed_end();
    }

    int rc = 0;
    for(auto& l : loops) rc = std::max(rc, l->join());
    if(rc != 0) return rc;

    if(g_stop_requested.load()) std::cerr << "msk144hipdecoder: stopped by signal; hops in flight were finished" << std::endl;
    long total_hops = 0, total_late = 0, batches = 0, overflowed = 0;
    long long worst = 0;
    for(size_t i = 0; i < loops.size(); i++)
    {
        const LoopStats& ls = loops[i]->stats();
        total_hops += ls.hops;
        total_late += ls.late;
        batches += ls.batches;
        overflowed += ls.overflowed_hops;
        worst = std::max(worst, ls.worst_ms);
        for(int c = 0; c < loops[i]->streams(); c++)
        {
            const DeviceLoop::StreamReport r = loops[i]->stream_report(c);
            if(r.late) std::cerr << "ch=" << shares[i].first + c << ": " << r.late << " of " << r.hops << " hops answered later than 210 ms (worst " << r.worst_ms << " ms)" << std::endl;
        }
        if(loops.size() > 1)
            std::cerr << "msk144hipdecoder: device " << shares[i].device << " (streams " << shares[i].first << ".." << shares[i].first + shares[i].count - 1 << "): " << ls.batches << " batches, "
                      << ls.hops << " stream hops, " << ls.late << " late, worst latency " << ls.worst_ms << " ms" << std::endl;
    }
    std::cerr << "msk144hipdecoder: " << batches << " batches, " << total_hops << " stream hops, " << total_late << " late, worst latency " << worst << " ms" << std::endl;
    if(overflowed) std::cerr << "msk144hipdecoder: " << overflowed << " hops overflowed the result list (lists cut, see above)" << std::endl;
    if(timing)
    {
        const double wall_s = std::chrono::duration<double>(Clock::now() - run_since).count();
        auto row = [](const char* name, const Accumulator& a) { fprintf(stderr, "msk144hipdecoder timing: %-34s mean %9.3f ms  max %9.3f ms\n", name, a.mean(), a.worst); };
        for(size_t i = 0; i < loops.size(); i++)
        {
            const LoopStats& ls = loops[i]->stats();
            if(loops.size() > 1) fprintf(stderr, "msk144hipdecoder timing: ---- device %d: %d streams (%d..%d), own ingest and post-processing threads ----\n", shares[i].device, shares[i].count, shares[i].first, shares[i].first + shares[i].count - 1);
            fprintf(stderr, "msk144hipdecoder timing: %d streams, %ld batches in %.2f s wall; per batch (host wall time):\n", shares[i].count, ls.batches, loops.size() > 1 ? ls.wall_s : wall_s);
            row("ingest (read syscalls, all streams)", ls.ingest);
            row("copy new hops into pinned slot", ls.assemble);
            row("submit (3 asynchronous calls)", ls.submit);
            row("wait for GPU + D2H (post thread)", ls.wait);
            row("post-processing (text, SNR, filter)", ls.post);
            row("print", ls.print);
            row("batch released -> lines printed", ls.latency);
            fprintf(stderr, "msk144hipdecoder timing: records per batch mean %.0f max %.0f\n", ls.records.mean(), ls.records.worst);
            const float* dev = ls.device_ms;
            if(ls.have_device_ms)
                fprintf(stderr, "msk144hipdecoder timing: device per batch (HIP events): H2D %.3f  front end %.3f  scan %.3f  softbits %.3f  index %.3f  LDPC %.3f  collect %.3f  D2H %.3f ms\n",
                        dev[MSK144_T_H2D], dev[MSK144_T_FRONTEND], dev[MSK144_T_SCAN], dev[MSK144_T_SOFTBITS], dev[MSK144_T_INDEX], dev[MSK144_T_LDPC], dev[MSK144_T_COLLECT], dev[MSK144_T_D2H]);
        }
    }
    std::cout << "Done" << std::endl;
    return 0;
}
