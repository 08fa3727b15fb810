// C entry points of the GPU-independent host logic, for the CPU test-suite (libmsk144host.so).
#include "window_decoder.h"

#include "../csrc/msk144_tables.h"

#include <cstring>

using namespace msk144host;

extern "C" {

void* msk144host_table_new() { return new CallHashTable(); }
void msk144host_table_free(void* t) { delete static_cast<CallHashTable*>(t); }
void msk144host_table_clear(void* t) { static_cast<CallHashTable*>(t)->clear(); }
unsigned msk144host_hash(const char* call, int bits) { return CallHashTable::hash(call, bits); }

int msk144host_message_gate(const unsigned char* bits77) { return message_gate(bits77) ? 1 : 0; }

// out: at least 64 bytes
int msk144host_decode_message(void* table, const unsigned char* bits77, char* out)
{
    std::string text;
    const bool ok = decode_message(bits77, *static_cast<CallHashTable*>(table), text);
    std::strncpy(out, text.c_str(), 63);
    out[63] = 0;
    return ok ? 1 : 0;
}

void* msk144host_snr_new() { return new SnrTracker(); }
void msk144host_snr_free(void* s) { delete static_cast<SnrTracker*>(s); }
int msk144host_snr_update(void* s, const float* seg8)
{
    static_cast<SnrTracker*>(s)->update(seg8);
    return static_cast<SnrTracker*>(s)->snr_int();
}
float msk144host_snr_db(void* s) { return static_cast<SnrTracker*>(s)->snr_db(); }

// Post-process one window.  accepted: n records of {f0, num_avg, nbadsync, pattern_idx, bits[77]} in
// item order.  Lines are written as a '\n'-joined string of format_line() outputs with the date field
// blanked (tests mask it anyway).  Returns the number of lines.
struct msk144host_accepted
{
    float f0;
    int num_avg;
    int nbadsync;
    int pattern_idx;
    unsigned char bits[77];
};

int msk144host_postprocess(void* table, const msk144host_accepted* acc, int n, int snr, int quirk, char* out, int out_cap)
{
    std::vector<AcceptedCandidate> v(n);
    for(int i = 0; i < n; i++)
    {
        v[i].f0 = acc[i].f0;
        v[i].num_avg = acc[i].num_avg;
        v[i].nbadsync = acc[i].nbadsync;
        v[i].pattern_idx = acc[i].pattern_idx;
        std::memcpy(v[i].bits, acc[i].bits, 77);
    }
    ResultFilter filter;
    std::vector<FilteredResult> lines = postprocess_window(v, snr, quirk != 0, *static_cast<CallHashTable*>(table), filter);
    std::string joined;
    for(size_t i = 0; i < lines.size(); i++)
    {
        if(i) joined += "\n";
        joined += lines[i].format_line();
    }
    std::strncpy(out, joined.c_str(), out_cap - 1);
    out[out_cap - 1] = 0;
    return static_cast<int>(lines.size());
}

// ---- ResultFilter alone (tests/test_ref_host.py drives it side by side with the reference's compiled class) ----
void* msk144host_filter_new() { return new ResultFilter(); }
void msk144host_filter_free(void* f) { delete static_cast<ResultFilter*>(f); }
void msk144host_filter_begin(void* f) { static_cast<ResultFilter*>(f)->begin_window(); }
void msk144host_filter_put(void* f, int snr, float f0, int num_avg, int nbadsync, int pattern_idx, const char* text)
{
    static_cast<ResultFilter*>(f)->put(snr, f0, num_avg, nbadsync, pattern_idx, text);
}
// writes the window's lines as records {snr, f0, num_avg, nbadsync, pattern_idx, text[64]}; returns their number
struct msk144host_filtered
{
    int snr;
    float f0;
    int num_avg;
    int nbadsync;
    int pattern_idx;
    char text[64];
};
int msk144host_filter_end(void* f, msk144host_filtered* out, int cap)
{
    const std::vector<FilteredResult> r = static_cast<ResultFilter*>(f)->end_window();
    for(int i = 0; i < static_cast<int>(r.size()) && i < cap; i++)
    {
        out[i].snr = r[i].snr;
        out[i].f0 = r[i].f0;
        out[i].num_avg = r[i].num_avg;
        out[i].nbadsync = r[i].nbadsync;
        out[i].pattern_idx = r[i].pattern_idx;
        std::strncpy(out[i].text, r[i].text.c_str(), sizeof(out[i].text) - 1);
        out[i].text[sizeof(out[i].text) - 1] = 0;
    }
    return static_cast<int>(r.size());
}

// ---- the search-context tables libmsk144hip.so hands to its kernels (csrc/msk144_tables.h), for CPU-side pinning ----
void msk144host_sync_template(float* re42, float* im42, float* pp12) { msk144::sync_template(re42, im42, pp12); }
int msk144host_frequency_grid(float center, float width, float step, float* out, int cap)
{
    const std::vector<float> f = msk144::frequency_grid(center, width, step);
    for(int i = 0; i < static_cast<int>(f.size()) && i < cap; i++) out[i] = f[i];
    return static_cast<int>(f.size());
}
int msk144host_fft_band_mask(float* out, int cap)
{
    const std::vector<float> w = msk144::fft_band_mask();
    for(int i = 0; i < static_cast<int>(w.size()) && i < cap; i++) out[i] = w[i];
    return static_cast<int>(w.size());
}

}  // extern "C"
