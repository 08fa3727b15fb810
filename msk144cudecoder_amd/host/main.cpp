// msk144hipdecoder - stdin -> stdout MSK144 stream decoder for AMD Instinct MI355X.
// Same command-line options, input framing and output lines as the reference program
// (main.cu:55-426), with the GPU work behind libmsk144hip.so.
#include "window_decoder.h"

#include <getopt.h>

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

void usage(const char* prog)
{
    std::cout << "Usage: " << prog << " [--help] [options] < samples\n"
              << " Raw samples are read from stdin, decoded messages are written to stdout.\n"
              << " Options:\n"
              << "   --help                      This text.\n"
              << "   --center-frequency=HZ       Centre of the search window. Default 1500 (audio), 0 (IQ).\n"
              << "   --search-step=HZ            Spacing of the frequency hypotheses. Default 2.0.\n"
              << "   --search-width=HZ           Width of the search window around the centre. Default 200.0.\n"
              << "   --scan-depth=[1..8]         Number of frame-averaging patterns tried. Default 4.\n"
              << "   --read-mode=[1|2]           1 = audio, 16 bit signed mono, 12000 sps; 2 = IQ, 8+8 bit signed, 12000 sps. Default 1.\n"
              << "   --analytic-method=[1|2]     Audio only: 1 = FFT, 2 = shift + low-pass + shift. Default 2.\n"
              << "   --nbadsync-threshold=N      Sync-word bit errors tolerated before LDPC decoding. Default 1.\n"
              << " Additions of this implementation:\n"
              << "   --strict-decode             Unpack every distinct payload of a window (the reference reuses the first one).\n"
              << "   --print-bits                Append the 77-bit payload to each output line.\n"
              << "   --device=N                  HIP device ordinal. Default 0.\n"
              << "   --inputs=F1,F2,...          Decode several raw streams (files or FIFOs) as one GPU batch instead of stdin;\n"
              << "                               output lines then carry ch=<index> after the leading stars.\n";
}

const char* mode_name(int mode)
{
    if(mode == 1) return "Audio. 16 bits signed.";
    if(mode == 2) return "IQ. 8+8 bits.";
    return "unknown";
}

}  // namespace

int main(int argc, char* const argv[])
{
    DecoderOptions opt;
    bool center_set = false;
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
                                           {0, 0, 0, 0}};
    while(true)
    {
        int idx = 0;
        const int c = getopt_long(argc, argv, "", long_options, &idx);
        if(c == -1) break;
        if(c != 0) continue;  // unknown option: getopt has printed its own message, carry on like the reference
        switch(idx)
        {
        case 0: usage(argv[0]); return 0;
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
        case 11:
        {
            std::string list(optarg);
            size_t a = 0;
            while(a <= list.size())
            {
                const size_t b = list.find(',', a);
                const std::string item = list.substr(a, b == std::string::npos ? std::string::npos : b - a);
                if(!item.empty()) input_paths.push_back(item);
                if(b == std::string::npos) break;
                a = b + 1;
            }
            break;
        }
        default: usage(argv[0]); return 0;
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

    // input streams: stdin (the reference's only mode) or --inputs files/FIFOs, one channel each
    std::vector<FILE*> streams;
    if(input_paths.empty())
    {
        streams.push_back(stdin);
    }
    else
    {
        for(const std::string& path : input_paths)
        {
            FILE* f = fopen(path.c_str(), "rb");
            if(!f)
            {
                std::cerr << "Cannot open input " << path << std::endl;
                return 2;
            }
            streams.push_back(f);
        }
    }
    const int nch = static_cast<int>(streams.size());
    opt.channels = nch;

    WindowDecoder dec(opt);
    if(!dec.ok())
    {
        std::cerr << "msk144hip: " << dec.error() << std::endl;
        return 2;
    }

    std::cerr << "Actual parameters:" << std::endl
              << "Center Frequency: " << opt.center_hz << "Hz" << std::endl
              << "Search Step: " << opt.step_hz << "Hz" << std::endl
              << "Search Width: " << opt.width_hz << "Hz" << std::endl
              << "Scan Depth: " << dec.scan_depth() << std::endl
              << "Left Boundary: " << dec.left_bound() << "Hz" << std::endl
              << "Right Boundary: " << dec.right_bound() << "Hz" << std::endl
              << "Read Mode: (" << mode_name(opt.read_mode) << ")" << std::endl;
    if(opt.read_mode == 1) std::cerr << "Analytic Method: " << opt.analytic_method << std::endl;
    std::cerr << "Badsync Threshold: " << opt.nbadsync_threshold << std::endl
              << "Frequency hypotheses: " << dec.num_freqs() << std::endl
              << "Candidates per window: " << dec.num_freqs() * dec.scan_depth() * 8 << std::endl;
    if(nch > 1) std::cerr << "Input streams: " << nch << std::endl;
    std::cerr << std::endl;

    // window ring per stream: first read fills 5184 samples, every later read replaces the older half
    // (main.cu:271-294 audio, :337-359 IQ)
    const size_t sample_bytes = (opt.read_mode == 1) ? sizeof(int16_t) : 2 * sizeof(int8_t);
    const size_t win_bytes = MSK144_WINDOW_SAMPLES * sample_bytes;
    const size_t unit = (opt.read_mode == 1) ? sizeof(int16_t) : sizeof(int8_t);  // the reference counts items of this size
    std::vector<unsigned char> ring(win_bytes * nch, 0);
    std::vector<bool> active(nch, true);
    bool first = true;
    std::vector<std::vector<FilteredResult>> lines;

    while(true)
    {
        int alive = 0;
        for(int c = 0; c < nch; c++)
        {
            if(!active[c]) continue;
            unsigned char* w = ring.data() + win_bytes * c;
            size_t want, rc;
            if(first)
            {
                want = win_bytes / unit;
                rc = fread(w, unit, want, streams[c]);
            }
            else
            {
                const size_t half = win_bytes / 2;
                memcpy(w, w + half, half);
                want = half / unit;
                rc = fread(w + half, unit, want, streams[c]);
            }
            if(rc != want)
            {
                if(nch > 1) std::cerr << "ch=" << c << ": ";
                std::cerr << "Incomplete read error. rc=" << rc << std::endl;
                active[c] = false;
                memset(w, 0, win_bytes);
                continue;
            }
            alive++;
        }
        first = false;
        if(alive == 0) break;

        const auto t0 = std::chrono::steady_clock::now();
        if(!dec.process(ring.data(), active, lines))
        {
            std::cerr << "msk144hip: " << dec.error() << std::endl;
            return 2;
        }
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        const int soft_limit_ms = 210;  // of the 216 ms a hop lasts (main.cu:398-403)
        if(ms > soft_limit_ms)
        {
            std::cerr << "Warning: Working loop takes too much time: " << ms << " ms"
                      << " of " << soft_limit_ms << " ms max." << std::endl;
        }
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

    for(FILE* f : streams)
        if(f != stdin) fclose(f);
    std::cout << "Done" << std::endl;
    return 0;
}
