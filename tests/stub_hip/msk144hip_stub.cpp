// TEST-ONLY stand-in for libmsk144hip.so: the entry points msk144hipdecoder's host loop calls, with no GPU behind them, so that
// the loop's own logic (non-blocking ingest, batch policy, two-slot pipeline, post-processing thread, deadline accounting, end of
// stream) runs under `pytest -m "not gpu"`.  It decodes nothing: a "decode" takes MSK144_STUB_DECODE_MS milliseconds of wall time on
// a worker thread (the asynchronous GPU) and yields one record per channel whose window's first sample is 0x7777.  Its payload is a
// telemetry message (i3 = 0, n3 = 5: 71 bits printed as hex by the text layer) holding the channel and the second sample of each
// window half - enough to check from stdout that every hop of every stream reached the decoder once, in order, in its own slot.
// MSK144_STUB_DEVICES=N makes it report N devices (msk144_device_count; a handle on ordinal >= N is refused), and the device ordinal
// of the handle that decoded a window rides in the top byte of the payload, so a test of `--devices=...` can tell which loop served
// which stream.  A hop with more records than msk144_params.max_results is cut and reported as MSK144_EOVERFLOW, like the library.
#include "../../include/msk144hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>
#include <thread>
#include <vector>

struct msk144_handle
{
    msk144_params p;
    std::vector<int16_t> in[MSK144_SLOTS];
    // hop ring (msk144_hop_slot / msk144_push_hops): per-slot inputs and every stream's window
    std::vector<int16_t> hops[MSK144_SLOTS], first_halves[MSK144_SLOTS];
    std::vector<int32_t> streams[MSK144_SLOTS];
    std::vector<uint8_t> is_first[MSK144_SLOTS];
    std::vector<int16_t> ring;
    std::vector<msk144_result> out[MSK144_SLOTS];
    std::vector<float> seg[MSK144_SLOTS];
    std::future<void> job[MSK144_SLOTS];
    bool pending[MSK144_SLOTS] = {false, false};
    std::future<void> last;  // decodes run one after the other, like kernels on one stream
    int cur = 0;
    int active[MSK144_SLOTS] = {0, 0};  // windows the hop in each slot covers
    int decode_ms = 5;
    int fail_after = -1;  // MSK144_STUB_FAIL_DEVICE / MSK144_STUB_FAIL_AFTER: the n-th hop pushed on that device fails
    int pushed = 0;
    bool overflow[MSK144_SLOTS] = {false, false};
    std::string error;
};

static int stub_devices()
{
    const char* e = std::getenv("MSK144_STUB_DEVICES");
    const int n = e ? std::atoi(e) : 1;
    return n < 1 ? 1 : n;
}

static thread_local std::string g_create_error = "stub";

extern "C" {

void msk144_default_params(msk144_params* p)
{
    std::memset(p, 0, sizeof(*p));
    p->center_hz = 1500.0f;
    p->width_hz = 200.0f;
    p->step_hz = 2.0f;
    p->scan_depth = 4;
    p->nbadsync_threshold = 1;
    p->read_mode = 1;
    p->analytic_method = 2;
    p->channels = 1;
}

int msk144_device_count(int32_t* n)
{
    *n = stub_devices();
    return MSK144_OK;
}

int msk144_clock_probe(msk144_handle*, int32_t, float* mhz)
{
    *mhz = 0.0f;
    return MSK144_OK;
}

int msk144_create(const msk144_params* p, msk144_handle** out)
{
    if(p->device < 0 || p->device >= stub_devices())
    {
        g_create_error = "device ordinal out of range";
        *out = nullptr;
        return MSK144_EINVAL;
    }
    auto* h = new msk144_handle();
    h->p = *p;
    if(const char* e = std::getenv("MSK144_STUB_DECODE_MS")) h->decode_ms = std::atoi(e);
    if(const char* d = std::getenv("MSK144_STUB_FAIL_DEVICE"))
        if(std::atoi(d) == p->device)
        {
            const char* n = std::getenv("MSK144_STUB_FAIL_AFTER");
            h->fail_after = n ? std::atoi(n) : 0;
        }
    *out = h;
    return MSK144_OK;
}

void msk144_destroy(msk144_handle* h)
{
    if(!h) return;
    for(auto& j : h->job)
        if(j.valid()) j.wait();
    delete h;
}

const char* msk144_last_error(const msk144_handle* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int msk144_geometry(const msk144_handle* h, int32_t* f, int32_t* d, int32_t* k)
{
    const int F = 2 * static_cast<int>((h->p.width_hz / 2) / h->p.step_hz) + 1;
    if(f) *f = F;
    if(d) *d = h->p.scan_depth;
    if(k) *k = F * h->p.scan_depth * 8;
    return MSK144_OK;
}

int msk144_frequency(const msk144_handle* h, int32_t b, float* hz)
{
    *hz = h->p.center_hz - static_cast<int>((h->p.width_hz / 2) / h->p.step_hz) * h->p.step_hz + b * h->p.step_hz;
    return MSK144_OK;
}

int msk144_set_profiling(msk144_handle*, int32_t) { return MSK144_OK; }
int msk144_llr_block_channels(const msk144_handle* h, int32_t* n)
{
    if(!h || !n) return MSK144_EINVAL;
    *n = h->p.channels;
    return MSK144_OK;
}
// MSK144_STUB_LOG_MODES=1: report on stderr what the program asks of its handles
int msk144_set_llr_retention(msk144_handle*, int32_t retain)
{
    if(getenv("MSK144_STUB_LOG_MODES")) fprintf(stderr, "stub: msk144_set_llr_retention(%d)\n", retain);
    return MSK144_OK;
}
int msk144_set_copy_handover(msk144_handle*, int32_t enable)
{
    if(getenv("MSK144_STUB_LOG_MODES")) fprintf(stderr, "stub: msk144_set_copy_handover(%d)\n", enable);
    return MSK144_OK;
}

int msk144_stage_times(msk144_handle*, float* ms, int32_t*, int32_t)
{
    for(int i = 0; i < MSK144_T_COUNT; i++) ms[i] = 0.0f;
    return MSK144_OK;
}

int msk144_input_slot(msk144_handle* h, int32_t s, void** w, size_t* bytes)
{
    if(s < 0 || s >= MSK144_SLOTS) return MSK144_EINVAL;
    const size_t n = static_cast<size_t>(h->p.channels) * MSK144_WINDOW_SAMPLES;
    if(h->in[s].size() != n) h->in[s].assign(n, 0);
    *w = h->in[s].data();
    if(bytes) *bytes = n * sizeof(int16_t);
    return MSK144_OK;
}

int msk144_submit_slot_n(msk144_handle* h, int32_t s, int32_t n)
{
    if(n < 1 || n > h->p.channels) return MSK144_EINVAL;
    if(h->pending[s])
    {
        h->error = "stub: slot submitted again before its results were fetched";
        return MSK144_ESTATE;
    }
    h->cur = s;
    h->active[s] = n;
    return MSK144_OK;
}

int msk144_submit_slot(msk144_handle* h, int32_t s) { return msk144_submit_slot_n(h, s, h->p.channels); }

int msk144_hop_slot(msk144_handle* h, int32_t s, void** hops, void** first_halves, int32_t** streams, uint8_t** is_first)
{
    if(s < 0 || s >= MSK144_SLOTS) return MSK144_EINVAL;
    const size_t nch = static_cast<size_t>(h->p.channels);
    if(h->ring.empty())
    {
        h->ring.assign(nch * MSK144_WINDOW_SAMPLES, 0);
        for(int k = 0; k < MSK144_SLOTS; k++)
        {
            h->hops[k].assign(nch * MSK144_HOP_SAMPLES, 0);
            h->first_halves[k].assign(nch * MSK144_HOP_SAMPLES, 0);
            h->streams[k].assign(nch, 0);
            h->is_first[k].assign(nch, 0);
            h->in[k].assign(nch * MSK144_WINDOW_SAMPLES, 0);
        }
    }
    *hops = h->hops[s].data();
    *first_halves = h->first_halves[s].data();
    *streams = h->streams[s].data();
    *is_first = h->is_first[s].data();
    return MSK144_OK;
}

// the device-side ring, on the CPU: advance the listed streams' windows and lay them out as the compact batch of the slot
int msk144_push_hops(msk144_handle* h, int32_t s, int32_t n)
{
    if(h->ring.empty() || n < 1 || n > h->p.channels) return MSK144_EINVAL;
    if(h->pending[s])
    {
        h->error = "stub: slot submitted again before its results were fetched";
        return MSK144_ESTATE;
    }
    if(h->fail_after >= 0 && h->pushed++ >= h->fail_after)
    {
        h->error = "stub: injected device failure";
        return MSK144_EHIP;
    }
    const size_t half = MSK144_HOP_SAMPLES;
    for(int j = 0; j < n; j++)
    {
        const int c = h->streams[s][j];
        if(c < 0 || c >= h->p.channels || (j > 0 && c <= h->streams[s][j - 1]))
        {
            h->error = "stub: streams must be ascending";
            return MSK144_EINVAL;
        }
        int16_t* r = h->ring.data() + static_cast<size_t>(c) * MSK144_WINDOW_SAMPLES;
        if(h->is_first[s][j]) std::memcpy(r, h->first_halves[s].data() + j * half, half * sizeof(int16_t));
        else std::memmove(r, r + half, half * sizeof(int16_t));
        std::memcpy(r + half, h->hops[s].data() + j * half, half * sizeof(int16_t));
        std::memcpy(h->in[s].data() + static_cast<size_t>(j) * MSK144_WINDOW_SAMPLES, r, MSK144_WINDOW_SAMPLES * sizeof(int16_t));
    }
    h->cur = s;
    h->active[s] = n;
    return MSK144_OK;
}

int msk144_decode(msk144_handle*) { return MSK144_OK; }

int msk144_fetch_async(msk144_handle* h, int32_t s)
{
    if(s != h->cur || h->pending[s])
    {
        h->error = "stub: fetch of the wrong slot";
        return MSK144_ESTATE;
    }
    h->pending[s] = true;
    std::shared_future<void> before = h->last.valid() ? h->last.share() : std::shared_future<void>();
    std::promise<void> done;
    h->last = done.get_future();
    h->job[s] = std::async(std::launch::async, [h, s, before, done = std::move(done)]() mutable {
        if(before.valid()) before.wait();
        std::this_thread::sleep_for(std::chrono::milliseconds(h->decode_ms));
        h->out[s].clear();
        h->seg[s].assign(static_cast<size_t>(h->p.channels) * 8, 1.0f);
        for(int c = 0; c < h->active[s]; c++)
        {
            const int16_t* w = h->in[s].data() + static_cast<size_t>(c) * MSK144_WINDOW_SAMPLES;
            if(w[0] != 0x7777) continue;
            msk144_result r{};
            r.channel = c;
            r.item = c;
            r.f0 = 1500.0f;
            const uint64_t v = (static_cast<uint64_t>(h->p.device & 0xff) << 56) | (static_cast<uint64_t>(c) << 32) | (static_cast<uint64_t>(static_cast<uint16_t>(w[1])) << 16) |
                               static_cast<uint16_t>(w[MSK144_HOP_SAMPLES + 1]);
            uint8_t bits[80] = {0};
            for(int i = 0; i < 64; i++) bits[7 + i] = (v >> (63 - i)) & 1u;  // 71-bit field, MSB first: 7 leading zeros + 64 bits
            bits[71] = 1;                                                     // n3 = 5 (101), i3 = 0 (000)
            bits[73] = 1;
            for(int i = 0; i < 80; i++) r.message[i / 8] |= static_cast<uint8_t>(bits[i] << (7 - (i % 8)));
            h->out[s].push_back(r);
        }
        h->overflow[s] = h->p.max_results > 0 && h->out[s].size() > static_cast<size_t>(h->p.max_results);
        if(h->overflow[s]) h->out[s].resize(static_cast<size_t>(h->p.max_results));
        done.set_value();
    });
    return MSK144_OK;
}

int msk144_fetch_wait(msk144_handle* h, int32_t s, const msk144_result** rec, int32_t* n, const float** seg)
{
    if(!h->pending[s]) return MSK144_ESTATE;
    h->job[s].wait();
    h->pending[s] = false;
    *rec = h->out[s].data();
    *n = static_cast<int32_t>(h->out[s].size());
    if(seg) *seg = h->seg[s].data();
    return h->overflow[s] ? MSK144_EOVERFLOW : MSK144_OK;
}

}  // extern "C"
