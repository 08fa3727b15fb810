// C ABI of libmsk144hip.so (include/msk144hip.h): handle lifetime, buffers, launch order.
// Host-side restatement of the reference's setup code; compiled with -ffp-contract=off so that the
// frequency grid, the sync template and the FFT mask are computed with the reference's float ops.
#include "../../include/msk144hip.h"

#include "msk144_kernels.h"
#include "msk144_tables.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace msk144;

static_assert(sizeof(msk144_candidate) == kReferenceResultItemBytes, "msk144_candidate must mirror the reference ResultItem");
static_assert(sizeof(msk144_result) == 52, "msk144_result layout");

struct msk144_handle
{
    msk144_params params{};
    DeviceStore st{};
    SyncTemplate tpl{};
    std::vector<float> freq_host;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // device allocations
    float* d_freq = nullptr;
    float2* d_cb42 = nullptr;
    void* d_input = nullptr;  // staging for host-submitted windows
    float2* d_twiddle = nullptr;
    float* d_fft_mask = nullptr;
    std::vector<void*> allocs;

    int llr_block = 1;  // channels per softbits->index->LDPC block
    bool retained = true;  // every LLR row of a decode stays readable (one block covers all channels and msk144_set_llr_retention was not switched off)
    bool rows_valid = true;  // the LLR store holds every row of the last softbits run (false after a decode without retention, until the next retained one)
    int active = 1;     // channels the current hop covers (msk144_submit_slot_n: the first n of the slot); <= st.channels
    bool have_window = false;
    bool decoded = false;
    bool profiling = false;
    // HIP-event pairs recorded around each stage; a pool so that many steps can be in flight
    // before the times are harvested at the next synchronisation point
    struct Span
    {
        int stage;
        int call;  // spans of one decode/submit call are summed into one sample
        hipEvent_t e0, e1;
    };
    int call_id = 0;
    int last_call[MSK144_T_COUNT];
    std::vector<Span> spans_pending;
    std::vector<hipEvent_t> ev_free;
    hipEvent_t ev_open = nullptr;
    double t_sum[MSK144_T_COUNT]{};
    int t_cnt[MSK144_T_COUNT]{};

    // Pinned staging slots (msk144_input_slot .. msk144_fetch_wait), allocated on first use.  Slot 0's device record list is
    // st.results of the plain calls; slot 1 has its own, so the list of one slot survives the decode of the other.
    struct Slot
    {
        void* in = nullptr;                 // pinned windows, input_bytes()
        msk144_result* d_records = nullptr; // device record list this slot's decode writes
        msk144_result* out = nullptr;       // pinned records, max_results
        int32_t* out_count = nullptr;       // pinned
        float* out_seg = nullptr;           // pinned [channels][8]
        int32_t copied = 0;                 // records covered by the asynchronous copy
        hipEvent_t done = nullptr;
        bool pending = false;
        // hop-ring inputs (msk144_hop_slot), pinned, allocated on first use
        void* hops = nullptr;          // [channels][half window]
        void* first_halves = nullptr;  // [channels][half window]
        int32_t* streams = nullptr;    // [channels]
        uint8_t* is_first = nullptr;   // [channels]
    };
    Slot slots[MSK144_SLOTS];
    bool slots_ready = false;
    int cur_slot = 0;                        // the slot whose record list the next decode writes
    std::atomic<int32_t> last_total{0};      // record count of the last fetched hop: sizes the next asynchronous copy
    hipStream_t copy_stream = nullptr;       // remainder copies of msk144_fetch_wait (may run on a second thread)
    // device side of the hop ring: every stream's current window, and the staging of one batch of hops
    void* d_ring = nullptr;
    void* d_hops = nullptr;
    void* d_first = nullptr;
    int32_t* d_streams = nullptr;
    uint8_t* d_isfirst = nullptr;
    bool ring_ready = false;
    // msk144_clock_probe: its own stream, so that the probe wave runs beside the decode kernels
    hipStream_t probe_stream = nullptr;
    uint64_t* d_probe = nullptr;

    std::string error;
};

namespace
{

thread_local std::string g_create_error;

// Channels per softbits -> index -> LDPC block when the caller leaves msk144_params.llr_block_channels at 0 (include/msk144hip.h): the
// bottom of a flat valley on the 1024-channel bench step (profiles/r06_sweep_ldpc_grid_and_block.txt, r06_sweep_block.txt)
constexpr int kDefaultLlrBlockChannels = 128;

int fail(msk144_handle* h, int code, const std::string& msg)
{
    if(h) h->error = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                                   \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t e_ = (expr);                                                                            \
        if(e_ != hipSuccess) return fail(h, MSK144_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while(0)

template<typename T>
int dev_alloc(msk144_handle* h, T** p, size_t count)
{
    void* q = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
    if(e != hipSuccess)
    {
        char buf[160];
        snprintf(buf, sizeof(buf), "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        return fail(h, MSK144_ENOMEM, buf);
    }
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return MSK144_OK;
}

size_t window_bytes(const msk144_handle* h)
{
    return (h->params.read_mode == 2) ? 2 * kWindowSamples : kWindowSamples * sizeof(int16_t);
}

size_t input_bytes(const msk144_handle* h)
{
    return window_bytes(h) * h->params.channels;
}

// the store as the kernels of the current hop see it: the first `active` channels
DeviceStore active_store(const msk144_handle* h)
{
    DeviceStore st = h->st;
    st.channels = h->active;
    st.nch = h->active;
    return st;
}

hipEvent_t ev_take(msk144_handle* h)
{
    if(!h->ev_free.empty())
    {
        hipEvent_t e = h->ev_free.back();
        h->ev_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if(hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void ev_begin(msk144_handle* h, int)
{
    if(!h->profiling) return;
    h->ev_open = ev_take(h);
    if(h->ev_open) (void)hipEventRecord(h->ev_open, h->stream);
}

void ev_end(msk144_handle* h, int stage)
{
    if(!h->profiling || !h->ev_open) return;
    hipEvent_t e1 = ev_take(h);
    if(!e1)
    {
        h->ev_free.push_back(h->ev_open);
        h->ev_open = nullptr;
        return;
    }
    (void)hipEventRecord(e1, h->stream);
    h->spans_pending.push_back({stage, h->call_id, h->ev_open, e1});
    h->ev_open = nullptr;
}

// fold finished event pairs into the running sums; requires the stream to be idle
void harvest_times(msk144_handle* h)
{
    for(const auto& sp : h->spans_pending)
    {
        float ms = 0.0f;
        if(hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess)
        {
            h->t_sum[sp.stage] += ms;
            if(h->last_call[sp.stage] != sp.call)
            {
                h->last_call[sp.stage] = sp.call;
                h->t_cnt[sp.stage]++;
            }
        }
        h->ev_free.push_back(sp.e0);
        h->ev_free.push_back(sp.e1);
    }
    h->spans_pending.clear();
}

int host_alloc(msk144_handle* h, void** p, size_t bytes)
{
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if(e != hipSuccess)
    {
        char buf[160];
        snprintf(buf, sizeof(buf), "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        return fail(h, MSK144_ENOMEM, buf);
    }
    return MSK144_OK;
}

int ensure_slots(msk144_handle* h)
{
    if(h->slots_ready) return MSK144_OK;
    HIP_TRY(h, hipSetDevice(h->params.device));
    if(!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for(int s = 0; s < MSK144_SLOTS; s++)
    {
        msk144_handle::Slot& sl = h->slots[s];
        int rc = MSK144_OK;
        if(!sl.in) rc = host_alloc(h, &sl.in, input_bytes(h));
        if(rc == MSK144_OK && !sl.out) rc = host_alloc(h, reinterpret_cast<void**>(&sl.out), sizeof(msk144_result) * static_cast<size_t>(h->st.max_results));
        if(rc == MSK144_OK && !sl.out_count) rc = host_alloc(h, reinterpret_cast<void**>(&sl.out_count), sizeof(int32_t));
        if(rc == MSK144_OK && !sl.out_seg) rc = host_alloc(h, reinterpret_cast<void**>(&sl.out_seg), sizeof(float) * 8 * h->st.channels);
        if(rc != MSK144_OK) return rc;
        if(!sl.d_records)
        {
            if(s == 0) sl.d_records = static_cast<msk144_result*>(h->st.results);
            else if((rc = dev_alloc(h, &sl.d_records, h->st.max_results)) != MSK144_OK) return rc;
        }
        // blocking-sync event: the waiting thread sleeps instead of spinning on a core the ingest thread needs
        if(!sl.done) HIP_TRY(h, hipEventCreateWithFlags(&sl.done, hipEventBlockingSync | hipEventDisableTiming));
    }
    h->slots_ready = true;
    return MSK144_OK;
}

int ensure_ring(msk144_handle* h)
{
    if(h->ring_ready) return MSK144_OK;
    int rc = ensure_slots(h);
    if(rc != MSK144_OK) return rc;
    const size_t nch = static_cast<size_t>(h->st.channels);
    const size_t half = window_bytes(h) / 2;
    uint8_t *ring = nullptr, *hops = nullptr, *first = nullptr;
    if((rc = dev_alloc(h, &ring, nch * window_bytes(h))) != MSK144_OK) return rc;
    if((rc = dev_alloc(h, &hops, nch * half)) != MSK144_OK) return rc;
    if((rc = dev_alloc(h, &first, nch * half)) != MSK144_OK) return rc;
    if((rc = dev_alloc(h, &h->d_streams, nch)) != MSK144_OK) return rc;
    if((rc = dev_alloc(h, &h->d_isfirst, nch)) != MSK144_OK) return rc;
    h->d_ring = ring;
    h->d_hops = hops;
    h->d_first = first;
    HIP_TRY(h, hipMemsetAsync(h->d_ring, 0, nch * window_bytes(h), h->stream));
    for(auto& sl : h->slots)
    {
        if((rc = host_alloc(h, &sl.hops, nch * half)) != MSK144_OK) return rc;
        if((rc = host_alloc(h, &sl.first_halves, nch * half)) != MSK144_OK) return rc;
        if((rc = host_alloc(h, reinterpret_cast<void**>(&sl.streams), nch * sizeof(int32_t))) != MSK144_OK) return rc;
        if((rc = host_alloc(h, reinterpret_cast<void**>(&sl.is_first), nch)) != MSK144_OK) return rc;
        std::memset(sl.is_first, 0, nch);
    }
    h->ring_ready = true;
    return MSK144_OK;
}

int copy_windows_in(msk144_handle* h, const void* host_windows)
{
    ev_begin(h, MSK144_T_H2D);
    hipError_t e = hipMemcpyAsync(h->d_input, host_windows, window_bytes(h) * h->active, hipMemcpyHostToDevice, h->stream);
    ev_end(h, MSK144_T_H2D);
    if(e != hipSuccess) return fail(h, MSK144_EHIP, std::string("hipMemcpyAsync(windows): ") + hipGetErrorString(e));
    return MSK144_OK;
}

// Long profiled runs (msk144hipdecoder --timing) never reach a synchronisation point: fold the spans whose end event has already
// completed, oldest first, so that the pending list stays short without waiting for anything.
void harvest_finished(msk144_handle* h)
{
    if(h->spans_pending.size() < 4096) return;
    size_t done = 0;
    while(done < h->spans_pending.size() && hipEventQuery(h->spans_pending[done].e1) == hipSuccess) done++;
    if(done == 0) return;
    std::vector<msk144_handle::Span> rest(h->spans_pending.begin() + static_cast<long>(done), h->spans_pending.end());
    h->spans_pending.resize(done);
    harvest_times(h);
    h->spans_pending = std::move(rest);
}

int run_frontend(msk144_handle* h, const void* d_in)
{
    ev_begin(h, MSK144_T_FRONTEND);
    const DeviceStore st = active_store(h);
    if(h->params.read_mode == 2) launch_frontend_iq(st, static_cast<const int8_t*>(d_in), h->stream);
    else launch_frontend_audio(st, static_cast<const int16_t*>(d_in), h->params.analytic_method, h->d_twiddle, h->d_fft_mask, h->stream);
    ev_end(h, MSK144_T_FRONTEND);
    HIP_TRY(h, hipGetLastError());
    h->have_window = true;
    h->decoded = false;
    return MSK144_OK;
}

}  // namespace

extern "C" {

void msk144_default_params(msk144_params* p)
{
    if(!p) return;
    std::memset(p, 0, sizeof(*p));
    p->center_hz = 1500.0f;  // main.cu:124 (audio); IQ callers set 0 (main.cu:125)
    p->width_hz = 200.0f;    // main.cu:130
    p->step_hz = 2.0f;       // main.cu:129
    p->scan_depth = 4;       // main.cu:131
    p->nbadsync_threshold = 1;  // main.cu:133
    p->read_mode = 1;
    p->analytic_method = 2;  // main.cu:132
    p->channels = 1;
    p->device = 0;
    p->max_results = 0;
    p->llr_block_channels = 0;
}

const char* msk144_last_error(const msk144_handle* h)
{
    return h ? h->error.c_str() : g_create_error.c_str();
}

int msk144_device_count(int32_t* n)
{
    if(!n) return fail(nullptr, MSK144_EINVAL, "null argument");
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        *n = 0;
        return fail(nullptr, MSK144_EHIP, "no HIP device available (libmsk144hip has no CPU fallback)");
    }
    *n = ndev;
    return MSK144_OK;
}

int msk144_create(const msk144_params* params, msk144_handle** out)
{
    if(!params || !out) return fail(nullptr, MSK144_EINVAL, "null argument");
    *out = nullptr;
    if(!(params->step_hz > 0.0f)) return fail(nullptr, MSK144_EINVAL, "search step must be > 0");  // assert at msk_context.cuh:97
    if(!(params->width_hz >= 0.0f)) return fail(nullptr, MSK144_EINVAL, "search width must be >= 0");
    if(params->channels < 1) return fail(nullptr, MSK144_EINVAL, "channels must be >= 1");
    if(params->llr_block_channels < 0) return fail(nullptr, MSK144_EINVAL, "llr_block_channels must be >= 0");
    if(params->read_mode != 1 && params->read_mode != 2) return fail(nullptr, MSK144_EINVAL, "read_mode must be 1 (audio) or 2 (IQ)");
    if(params->read_mode == 1 && params->analytic_method != 1 && params->analytic_method != 2)
        return fail(nullptr, MSK144_EINVAL, "analytic_method must be 1 (FFT) or 2 (shift-filter-shift)");

    msk144_handle* h = new(std::nothrow) msk144_handle();
    if(!h) return fail(nullptr, MSK144_ENOMEM, "out of host memory");
    h->params = *params;
    h->params.scan_depth = clamp_scan_depth(params->scan_depth);
    h->llr_block = params->llr_block_channels > 0 ? params->llr_block_channels : (params->channels <= kDefaultLlrBlockChannels ? params->channels : kDefaultLlrBlockChannels);
    if(h->llr_block > params->channels) h->llr_block = params->channels;
    h->params.llr_block_channels = h->llr_block;
    for(int s = 0; s < MSK144_T_COUNT; s++) h->last_call[s] = -1;

    auto bail = [&](int code) {
        g_create_error = h->error;
        msk144_destroy(h);
        return code;
    };

    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    {
        h->error = "no HIP device available (libmsk144hip has no CPU fallback)";
        return bail(MSK144_EHIP);
    }
    if(params->device < 0 || params->device >= ndev)
    {
        h->error = "device ordinal out of range";
        return bail(MSK144_EINVAL);
    }
    if(hipSetDevice(params->device) != hipSuccess)
    {
        h->error = "hipSetDevice failed";
        return bail(MSK144_EHIP);
    }

    h->freq_host = frequency_grid(params->center_hz, params->width_hz, params->step_hz);
    const int F = static_cast<int>(h->freq_host.size());

    DeviceStore& st = h->st;
    st.channels = params->channels;
    st.F = F;
    st.D = h->params.scan_depth;
    st.K = F * st.D * kSlotsPerPattern;
    st.nbadsync_threshold = params->nbadsync_threshold;
    st.ch0 = 0;
    st.nch = st.channels;
    h->retained = h->llr_block >= st.channels;
    st.gate_early = h->retained ? 0 : 1;
    st.handover = st.gate_early;  // msk144_set_copy_handover
    h->active = st.channels;
    const long long total = static_cast<long long>(st.channels) * st.K;
    long long maxr = params->max_results > 0 ? params->max_results : (1 << 20);
    if(maxr > total) maxr = total;
    st.max_results = static_cast<int32_t>(maxr);
    h->params.max_results = st.max_results;

    sync_template(h->tpl.re, h->tpl.im, h->tpl.pp);
    // the kernels skip the multiplications by pp[0] and pp[6] (softbits.hip, scan.hip): exact only for these values
    if(h->tpl.pp[0] != 0.0f || h->tpl.pp[kPulseSamples / 2] != 1.0f)
    {
        h->error = "half-sine pulse table: pp[0] != 0 or pp[6] != 1";
        return bail(MSK144_EINVAL);
    }

    if(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
    {
        h->error = "hipStreamCreate failed";
        return bail(MSK144_EHIP);
    }
    h->stream = h->own_stream;

    const size_t ck = static_cast<size_t>(total);
    int rc = MSK144_OK;
    auto A = [&](int r) { if(rc == MSK144_OK) rc = r; };
    A(dev_alloc(h, &h->d_freq, F));
    A(dev_alloc(h, &h->d_cb42, kSyncTaps));
    A(dev_alloc(h, &st.analytic, static_cast<size_t>(st.channels) * kWindowSamples));
    A(dev_alloc(h, &st.seg_power, static_cast<size_t>(st.channels) * 8));
    A(dev_alloc(h, &st.pos, ck));
    A(dev_alloc(h, &st.xb, ck));
    A(dev_alloc(h, &st.nbadsync, ck));
    A(dev_alloc(h, &st.llr, static_cast<size_t>(h->llr_block) * st.K * kCodeBits));
    A(dev_alloc(h, &st.idx, ck));
    A(dev_alloc(h, &st.n_idx, st.channels));
    A(dev_alloc(h, &st.dec_flag, ck));
    A(dev_alloc(h, &st.dec_iter, ck));
    A(dev_alloc(h, &st.dec_nhard, ck));
    A(dev_alloc(h, &st.dec_msg, ck * 3));
    A(dev_alloc(h, &st.dec_count, st.channels));
    A(dev_alloc(h, &st.copy_count, st.channels));
    A(dev_alloc(h, &st.result_count, 1));
    {
        msk144_result* r = nullptr;
        A(dev_alloc(h, &r, st.max_results));
        st.results = r;
    }
    {
        uint8_t* in = nullptr;
        A(dev_alloc(h, &in, input_bytes(h)));
        h->d_input = in;
    }
    if(rc != MSK144_OK) return bail(rc);
    st.freq = h->d_freq;
    st.cb42 = h->d_cb42;

    bool ok = hipMemcpy(h->d_freq, h->freq_host.data(), sizeof(float) * F, hipMemcpyHostToDevice) == hipSuccess;
    {
        float2 cb[kSyncTaps];
        for(int k = 0; k < kSyncTaps; k++) cb[k] = make_float2(h->tpl.re[k], h->tpl.im[k]);
        ok = ok && hipMemcpy(h->d_cb42, cb, sizeof(cb), hipMemcpyHostToDevice) == hipSuccess;
    }
    // initial state on the handle's own (non-blocking) stream, which does not synchronise with the null stream; create
    // returns only after these have completed
    hipStream_t s0 = h->own_stream;
    ok = ok && hipMemsetAsync(st.dec_flag, 0, ck, s0) == hipSuccess;
    ok = ok && hipMemsetAsync(st.n_idx, 0, sizeof(int32_t) * st.channels, s0) == hipSuccess;
    ok = ok && hipMemsetAsync(st.dec_count, 0, sizeof(int32_t) * st.channels, s0) == hipSuccess;
    ok = ok && hipMemsetAsync(st.copy_count, 0, sizeof(int32_t) * st.channels, s0) == hipSuccess;
    ok = ok && hipMemsetAsync(st.result_count, 0, sizeof(int32_t), s0) == hipSuccess;
    ok = ok && hipMemsetAsync(st.pos, 0, ck * sizeof(uint32_t), s0) == hipSuccess;
    ok = ok && hipMemsetAsync(st.nbadsync, 0, ck * sizeof(int32_t), s0) == hipSuccess;
    ok = ok && hipStreamSynchronize(s0) == hipSuccess;

    if(ok && params->read_mode == 1 && params->analytic_method == 1)
    {
        const std::vector<float> mask = fft_band_mask();
        std::vector<float2> tw(kFftSize / 2);
        for(int j = 0; j < kFftSize / 2; j++)
        {
            const double ang = -2.0 * M_PI * j / kFftSize;
            tw[j] = make_float2(static_cast<float>(cos(ang)), static_cast<float>(sin(ang)));
        }
        if(dev_alloc(h, &h->d_twiddle, tw.size()) != MSK144_OK || dev_alloc(h, &h->d_fft_mask, mask.size()) != MSK144_OK) return bail(MSK144_ENOMEM);
        ok = ok && hipMemcpy(h->d_twiddle, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(h->d_fft_mask, mask.data(), mask.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    if(!ok)
    {
        h->error = std::string("device initialisation failed: ") + hipGetErrorString(hipGetLastError());
        return bail(MSK144_EHIP);
    }
    *out = h;
    return MSK144_OK;
}

void msk144_destroy(msk144_handle* h)
{
    if(!h) return;
    (void)hipSetDevice(h->params.device);
    if(h->probe_stream) (void)hipStreamSynchronize(h->probe_stream);
    if(h->stream) (void)hipStreamSynchronize(h->stream);
    for(void* p : h->allocs) (void)hipFree(p);
    for(auto& sl : h->slots)
    {
        if(sl.in) (void)hipHostFree(sl.in);
        if(sl.out) (void)hipHostFree(sl.out);
        if(sl.out_count) (void)hipHostFree(sl.out_count);
        if(sl.out_seg) (void)hipHostFree(sl.out_seg);
        if(sl.hops) (void)hipHostFree(sl.hops);
        if(sl.first_halves) (void)hipHostFree(sl.first_halves);
        if(sl.streams) (void)hipHostFree(sl.streams);
        if(sl.is_first) (void)hipHostFree(sl.is_first);
        if(sl.done) (void)hipEventDestroy(sl.done);
    }
    if(h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if(h->probe_stream) (void)hipStreamDestroy(h->probe_stream);
    for(const auto& sp : h->spans_pending)
    {
        (void)hipEventDestroy(sp.e0);
        (void)hipEventDestroy(sp.e1);
    }
    for(hipEvent_t e : h->ev_free) (void)hipEventDestroy(e);
    if(h->ev_open) (void)hipEventDestroy(h->ev_open);
    if(h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int msk144_geometry(const msk144_handle* h, int32_t* num_freqs, int32_t* scan_depth, int32_t* items_per_channel)
{
    if(!h) return MSK144_EINVAL;
    if(num_freqs) *num_freqs = h->st.F;
    if(scan_depth) *scan_depth = h->st.D;
    if(items_per_channel) *items_per_channel = h->st.K;
    return MSK144_OK;
}

int msk144_frequency(const msk144_handle* h, int32_t block_idx, float* hz)
{
    if(!h || !hz || block_idx < 0 || block_idx >= h->st.F) return MSK144_EINVAL;
    *hz = h->freq_host[block_idx];
    return MSK144_OK;
}

int msk144_set_stream(msk144_handle* h, void* hip_stream)
{
    if(!h) return MSK144_EINVAL;
    HIP_TRY(h, hipSetDevice(h->params.device));  // every entry that touches the device selects the handle's own first: a caller may hold handles on several
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    harvest_times(h);
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return MSK144_OK;
}

int msk144_submit_audio(msk144_handle* h, const int16_t* windows)
{
    if(!h || !windows) return fail(h, MSK144_EINVAL, "null argument");
    if(h->params.read_mode != 1) return fail(h, MSK144_ESTATE, "handle was created for IQ input (read_mode 2)");
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    h->cur_slot = 0;
    h->active = h->st.channels;
    int rc = copy_windows_in(h, windows);
    return rc == MSK144_OK ? run_frontend(h, h->d_input) : rc;
}

int msk144_submit_iq(msk144_handle* h, const int8_t* windows)
{
    if(!h || !windows) return fail(h, MSK144_EINVAL, "null argument");
    if(h->params.read_mode != 2) return fail(h, MSK144_ESTATE, "handle was created for audio input (read_mode 1)");
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    h->cur_slot = 0;
    h->active = h->st.channels;
    int rc = copy_windows_in(h, windows);
    return rc == MSK144_OK ? run_frontend(h, h->d_input) : rc;
}

int msk144_submit_audio_device(msk144_handle* h, const int16_t* d_windows)
{
    if(!h || !d_windows) return fail(h, MSK144_EINVAL, "null argument");
    if(h->params.read_mode != 1) return fail(h, MSK144_ESTATE, "handle was created for IQ input (read_mode 2)");
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    h->cur_slot = 0;
    h->active = h->st.channels;
    return run_frontend(h, d_windows);
}

int msk144_submit_iq_device(msk144_handle* h, const int8_t* d_windows)
{
    if(!h || !d_windows) return fail(h, MSK144_EINVAL, "null argument");
    if(h->params.read_mode != 2) return fail(h, MSK144_ESTATE, "handle was created for audio input (read_mode 1)");
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    h->cur_slot = 0;
    h->active = h->st.channels;
    return run_frontend(h, d_windows);
}

int msk144_submit_analytic(msk144_handle* h, const float* windows)
{
    if(!h || !windows) return fail(h, MSK144_EINVAL, "null argument");
    HIP_TRY(h, hipSetDevice(h->params.device));
    HIP_TRY(h, hipMemcpyAsync(h->st.analytic, windows, sizeof(float2) * kWindowSamples * h->st.channels, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->cur_slot = 0;
    h->active = h->st.channels;
    h->have_window = true;
    h->decoded = false;
    return MSK144_OK;
}

int msk144_decode_stages(msk144_handle* h, uint32_t stages)
{
    if(!h) return MSK144_EINVAL;
    if((stages & (MSK144_STAGE_SCAN | MSK144_STAGE_SOFTBITS)) && !h->have_window) return fail(h, MSK144_ESTATE, "decode before any window was submitted");
    const bool blocked = !h->retained;
    const uint32_t mid = MSK144_STAGE_SOFTBITS | MSK144_STAGE_INDEX | MSK144_STAGE_LDPC;
    if(blocked && (stages & mid) != 0 && (stages & mid) != mid)
        return fail(h, MSK144_ENOTRETAINED, "LLR rows are not retained (blocked staging, or msk144_set_llr_retention(h, 0)): softbits, index and LDPC run together per channel block; a partial stage run needs llr_block_channels = channels");
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    if(h->profiling) harvest_finished(h);
    const DeviceStore cur = active_store(h);
    if(stages & MSK144_STAGE_SCAN)
    {
        ev_begin(h, MSK144_T_SCAN);
        launch_scan(cur, h->tpl, h->stream);
        ev_end(h, MSK144_T_SCAN);
    }
    // softbits -> index -> LDPC, one channel block at a time (one block = everything unless llr_block_channels says otherwise)
    if(stages & MSK144_STAGE_SOFTBITS) h->rows_valid = h->retained;
    if(stages & mid)
    {
        DeviceStore blk = cur;
        for(int ch0 = 0; ch0 < cur.channels; ch0 += h->llr_block)
        {
            blk.ch0 = ch0;
            blk.nch = cur.channels - ch0 < h->llr_block ? cur.channels - ch0 : h->llr_block;
            if(stages & MSK144_STAGE_SOFTBITS)
            {
                ev_begin(h, MSK144_T_SOFTBITS);
                launch_softbits(blk, h->tpl, h->stream);
                ev_end(h, MSK144_T_SOFTBITS);
            }
            if(stages & MSK144_STAGE_INDEX)
            {
                ev_begin(h, MSK144_T_INDEX);
                launch_index(blk, h->stream);
                ev_end(h, MSK144_T_INDEX);
            }
            if(stages & MSK144_STAGE_LDPC)
            {
                ev_begin(h, MSK144_T_LDPC);
                launch_ldpc(blk, h->stream);
                ev_end(h, MSK144_T_LDPC);
            }
        }
    }
    if(stages & MSK144_STAGE_COLLECT)
    {
        DeviceStore out = cur;
        if(h->slots_ready) out.results = h->slots[h->cur_slot].d_records;  // the record list of the slot being decoded
        ev_begin(h, MSK144_T_COLLECT);
        launch_collect(out, h->stream);
        ev_end(h, MSK144_T_COLLECT);
    }
    HIP_TRY(h, hipGetLastError());
    h->decoded = true;
    return MSK144_OK;
}

int msk144_decode(msk144_handle* h)
{
    return msk144_decode_stages(h, MSK144_STAGE_ALL);
}

int msk144_synchronize(msk144_handle* h)
{
    if(!h) return MSK144_EINVAL;
    HIP_TRY(h, hipSetDevice(h->params.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    harvest_times(h);
    return MSK144_OK;
}

int msk144_result_count(msk144_handle* h, int32_t* n)
{
    if(!h || !n) return fail(h, MSK144_EINVAL, "null argument");
    if(!h->decoded) return fail(h, MSK144_ESTATE, "no decode has been run");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    HIP_TRY(h, hipMemcpy(n, h->st.result_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    return MSK144_OK;
}

int msk144_results(msk144_handle* h, msk144_result* out, int32_t cap, int32_t* n)
{
    if(!h || !n || (cap > 0 && !out)) return fail(h, MSK144_EINVAL, "null argument");
    int32_t total = 0;
    int rc = msk144_result_count(h, &total);
    if(rc != MSK144_OK) return rc;
    *n = total;
    int32_t avail = total < h->st.max_results ? total : h->st.max_results;
    int32_t ncopy = avail < cap ? avail : cap;
    const void* list = h->slots_ready ? h->slots[h->cur_slot].d_records : h->st.results;  // the list the last decode wrote
    if(ncopy > 0) HIP_TRY(h, hipMemcpy(out, list, sizeof(msk144_result) * ncopy, hipMemcpyDeviceToHost));
    if(total > h->st.max_results) return fail(h, MSK144_EOVERFLOW, "more decodes than max_results; list truncated");
    return MSK144_OK;
}

int msk144_results_device(msk144_handle* h, const msk144_result** d_records, const int32_t** d_count)
{
    if(!h || !d_records || !d_count) return fail(h, MSK144_EINVAL, "null argument");
    *d_records = h->slots_ready ? h->slots[h->cur_slot].d_records : static_cast<const msk144_result*>(h->st.results);
    *d_count = h->st.result_count;
    return MSK144_OK;
}

int msk144_set_channel_base(msk144_handle* h, int32_t base)
{
    if(!h) return MSK144_EINVAL;
    if(base < 0) return fail(h, MSK144_EINVAL, "channel base must be >= 0");
    h->st.channel_base = base;  // kernel argument by value: takes effect at the next decode
    return MSK144_OK;
}

int msk144_set_llr_retention(msk144_handle* h, int32_t retain)
{
    if(!h) return MSK144_EINVAL;
    if(retain && h->llr_block < h->st.channels)
        return fail(h, MSK144_ENOTRETAINED, "this handle decodes in blocks of fewer channels than it holds: an LLR row never outlives its block; create it with llr_block_channels = channels");
    h->retained = retain != 0;
    h->st.gate_early = h->retained ? 0 : 1;
    h->st.handover = h->st.gate_early;
    return MSK144_OK;
}

int msk144_llr_block_channels(const msk144_handle* h, int32_t* channels_per_block)
{
    if(!h || !channels_per_block) return MSK144_EINVAL;
    *channels_per_block = h->llr_block;
    return MSK144_OK;
}

int msk144_set_copy_handover(msk144_handle* h, int32_t enable)
{
    if(!h) return MSK144_EINVAL;
    if(enable && h->retained)
        return fail(h, MSK144_ENOTRETAINED, "every LLR row of this handle is retained (llr_block_channels = channels): every slot is computed on its own; copies are handed over only when no row outlives its block (blocked staging, or msk144_set_llr_retention(h, 0))");
    h->st.handover = enable ? 1 : 0;  // kernel argument by value: takes effect at the next decode
    return MSK144_OK;
}

int msk144_copy_handover(const msk144_handle* h, int32_t* enabled)
{
    if(!h || !enabled) return MSK144_EINVAL;
    *enabled = h->st.handover;
    return MSK144_OK;
}

int msk144_copy_count(msk144_handle* h, int64_t* slots)
{
    if(!h || !slots) return fail(h, MSK144_EINVAL, "null argument");
    if(!h->decoded) return fail(h, MSK144_ESTATE, "no decode has been run");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    std::vector<int32_t> per_channel(static_cast<size_t>(h->active));
    HIP_TRY(h, hipMemcpy(per_channel.data(), h->st.copy_count, sizeof(int32_t) * per_channel.size(), hipMemcpyDeviceToHost));
    int64_t total = 0;
    for(int32_t c : per_channel) total += c;
    *slots = total;
    return MSK144_OK;
}

int msk144_segment_power(msk144_handle* h, float* out)
{
    if(!h || !out) return fail(h, MSK144_EINVAL, "null argument");
    if(!h->have_window) return fail(h, MSK144_ESTATE, "no window submitted");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    HIP_TRY(h, hipMemcpy(out, h->st.seg_power, sizeof(float) * 8 * h->st.channels, hipMemcpyDeviceToHost));
    return MSK144_OK;
}

int msk144_input_slot(msk144_handle* h, int32_t slot, void** host_windows, size_t* bytes)
{
    if(!h || !host_windows || slot < 0 || slot >= MSK144_SLOTS) return fail(h, MSK144_EINVAL, "bad argument");
    int rc = ensure_slots(h);
    if(rc != MSK144_OK) return rc;
    *host_windows = h->slots[slot].in;
    if(bytes) *bytes = input_bytes(h);
    return MSK144_OK;
}

int msk144_submit_slot(msk144_handle* h, int32_t slot)
{
    return msk144_submit_slot_n(h, slot, h ? h->st.channels : 0);
}

int msk144_submit_slot_n(msk144_handle* h, int32_t slot, int32_t n_channels)
{
    if(!h || slot < 0 || slot >= MSK144_SLOTS) return fail(h, MSK144_EINVAL, "bad argument");
    if(n_channels < 1 || n_channels > h->st.channels) return fail(h, MSK144_EINVAL, "n_channels must be 1..channels");
    int rc = ensure_slots(h);
    if(rc != MSK144_OK) return rc;
    if(h->slots[slot].pending) return fail(h, MSK144_ESTATE, "slot submitted again before its results were fetched (msk144_fetch_wait)");
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    h->cur_slot = slot;
    h->active = n_channels;
    rc = copy_windows_in(h, h->slots[slot].in);
    return rc == MSK144_OK ? run_frontend(h, h->d_input) : rc;
}

int msk144_hop_slot(msk144_handle* h, int32_t slot, void** hops, void** first_halves, int32_t** streams, uint8_t** is_first)
{
    if(!h || !hops || !first_halves || !streams || !is_first || slot < 0 || slot >= MSK144_SLOTS) return fail(h, MSK144_EINVAL, "bad argument");
    HIP_TRY(h, hipSetDevice(h->params.device));
    int rc = ensure_ring(h);
    if(rc != MSK144_OK) return rc;
    msk144_handle::Slot& sl = h->slots[slot];
    *hops = sl.hops;
    *first_halves = sl.first_halves;
    *streams = sl.streams;
    *is_first = sl.is_first;
    return MSK144_OK;
}

int msk144_push_hops(msk144_handle* h, int32_t slot, int32_t n)
{
    if(!h || slot < 0 || slot >= MSK144_SLOTS) return fail(h, MSK144_EINVAL, "bad argument");
    if(!h->ring_ready) return fail(h, MSK144_ESTATE, "msk144_push_hops before msk144_hop_slot");
    if(n < 1 || n > h->st.channels) return fail(h, MSK144_EINVAL, "n must be 1..channels");
    msk144_handle::Slot& sl = h->slots[slot];
    if(sl.pending) return fail(h, MSK144_ESTATE, "slot submitted again before its results were fetched (msk144_fetch_wait)");
    bool any_first = false;
    for(int32_t j = 0; j < n; j++)
    {
        if(sl.streams[j] < 0 || sl.streams[j] >= h->st.channels || (j > 0 && sl.streams[j] <= sl.streams[j - 1]))
            return fail(h, MSK144_EINVAL, "streams must be ascending stream numbers below channels");
        any_first = any_first || sl.is_first[j] != 0;
    }
    HIP_TRY(h, hipSetDevice(h->params.device));
    h->call_id++;
    h->cur_slot = slot;
    h->active = n;
    const size_t half = window_bytes(h) / 2;
    ev_begin(h, MSK144_T_H2D);
    hipError_t e = hipMemcpyAsync(h->d_hops, sl.hops, half * n, hipMemcpyHostToDevice, h->stream);
    if(e == hipSuccess && any_first) e = hipMemcpyAsync(h->d_first, sl.first_halves, half * n, hipMemcpyHostToDevice, h->stream);
    if(e == hipSuccess) e = hipMemcpyAsync(h->d_streams, sl.streams, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream);
    if(e == hipSuccess) e = hipMemcpyAsync(h->d_isfirst, sl.is_first, n, hipMemcpyHostToDevice, h->stream);
    ev_end(h, MSK144_T_H2D);
    if(e != hipSuccess) return fail(h, MSK144_EHIP, std::string("msk144_push_hops: ") + hipGetErrorString(e));
    launch_hop_ring(h->d_ring, h->d_hops, h->d_first, h->d_streams, h->d_isfirst, h->d_input, n, h->stream);
    return run_frontend(h, h->d_input);
}

int msk144_fetch_async(msk144_handle* h, int32_t slot)
{
    if(!h || slot < 0 || slot >= MSK144_SLOTS) return fail(h, MSK144_EINVAL, "bad argument");
    if(!h->decoded) return fail(h, MSK144_ESTATE, "no decode has been run");
    int rc = ensure_slots(h);
    if(rc != MSK144_OK) return rc;
    if(slot != h->cur_slot) return fail(h, MSK144_ESTATE, "fetch of a slot other than the one just decoded");
    msk144_handle::Slot& sl = h->slots[slot];
    if(sl.pending) return fail(h, MSK144_ESTATE, "slot already has a fetch in flight");
    HIP_TRY(h, hipSetDevice(h->params.device));
    long long guess = 2ll * h->last_total.load(std::memory_order_relaxed) + 256;
    if(guess < 1024) guess = 1024;
    if(guess > h->st.max_results) guess = h->st.max_results;
    sl.copied = static_cast<int32_t>(guess);
    ev_begin(h, MSK144_T_D2H);
    hipError_t e = hipMemcpyAsync(sl.out_count, h->st.result_count, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if(e == hipSuccess) e = hipMemcpyAsync(sl.out_seg, h->st.seg_power, sizeof(float) * 8 * h->active, hipMemcpyDeviceToHost, h->stream);
    if(e == hipSuccess) e = hipMemcpyAsync(sl.out, sl.d_records, sizeof(msk144_result) * static_cast<size_t>(sl.copied), hipMemcpyDeviceToHost, h->stream);
    ev_end(h, MSK144_T_D2H);
    if(e == hipSuccess) e = hipEventRecord(sl.done, h->stream);
    if(e != hipSuccess) return fail(h, MSK144_EHIP, std::string("msk144_fetch_async: ") + hipGetErrorString(e));
    sl.pending = true;
    return MSK144_OK;
}

// May run on a second thread (the other slot being submitted meanwhile): touches only this slot, last_total and copy_stream.
int msk144_fetch_wait(msk144_handle* h, int32_t slot, const msk144_result** records, int32_t* n, const float** seg_power)
{
    if(!h || !records || !n || slot < 0 || slot >= MSK144_SLOTS) return MSK144_EINVAL;
    msk144_handle::Slot& sl = h->slots[slot];
    if(!h->slots_ready || !sl.pending) return MSK144_ESTATE;
    if(hipSetDevice(h->params.device) != hipSuccess || hipEventSynchronize(sl.done) != hipSuccess) return MSK144_EHIP;
    const int32_t total = *sl.out_count;
    const int32_t avail = total < h->st.max_results ? total : h->st.max_results;
    if(avail > sl.copied)
    {
        // the estimate was short: the slot's own device list is still intact (the other slot decodes into its own)
        hipError_t e = hipMemcpyAsync(sl.out + sl.copied, sl.d_records + sl.copied, sizeof(msk144_result) * static_cast<size_t>(avail - sl.copied), hipMemcpyDeviceToHost,
                                      h->copy_stream);
        if(e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
        if(e != hipSuccess) return MSK144_EHIP;
    }
    h->last_total.store(total, std::memory_order_relaxed);
    sl.pending = false;
    *records = sl.out;
    *n = avail;
    if(seg_power) *seg_power = sl.out_seg;
    return total > h->st.max_results ? MSK144_EOVERFLOW : MSK144_OK;
}

int msk144_dump_analytic(msk144_handle* h, int32_t channel, float* out)
{
    if(!h || !out || channel < 0 || channel >= h->st.channels) return fail(h, MSK144_EINVAL, "bad argument");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    HIP_TRY(h, hipMemcpy(out, h->st.analytic + static_cast<size_t>(channel) * kWindowSamples, sizeof(float2) * kWindowSamples, hipMemcpyDeviceToHost));
    return MSK144_OK;
}

int msk144_dump_candidates(msk144_handle* h, int32_t channel, msk144_candidate* out)
{
    if(!h || !out || channel < 0 || channel >= h->st.channels) return fail(h, MSK144_EINVAL, "bad argument");
    if(!h->retained) return fail(h, MSK144_ENOTRETAINED, "LLR rows are not retained (blocked staging, or msk144_set_llr_retention(h, 0)); create the handle with llr_block_channels = channels for candidate dumps");
    if(!h->rows_valid) return fail(h, MSK144_ENOTRETAINED, "the last decode ran without LLR retention: its rows are incomplete; decode again now that retention is on");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    const DeviceStore& st = h->st;
    const size_t K = st.K;
    const size_t off = static_cast<size_t>(channel) * K;
    std::vector<uint32_t> pos(K), msg(K * 3);
    std::vector<float> xb(K), llr(K * kCodeBits);
    std::vector<int32_t> nbad(K);
    std::vector<uint8_t> flag(K), iter(K), nhard(K);
    HIP_TRY(h, hipMemcpy(pos.data(), st.pos + off, K * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(xb.data(), st.xb + off, K * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(nbad.data(), st.nbadsync + off, K * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(llr.data(), st.llr + off * kCodeBits, K * kCodeBits * 4, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(flag.data(), st.dec_flag + off, K, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(iter.data(), st.dec_iter + off, K, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(nhard.data(), st.dec_nhard + off, K, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(msg.data(), st.dec_msg + off * 3, K * 12, hipMemcpyDeviceToHost));
    const int per_freq = st.D * kSlotsPerPattern;
    std::memset(out, 0, sizeof(msk144_candidate) * K);
    for(size_t k = 0; k < K; k++)
    {
        msk144_candidate& c = out[k];
        const int b = static_cast<int>(k) / per_freq;
        const int p = (static_cast<int>(k) - b * per_freq) / kSlotsPerPattern;
        c.block_idx = b;
        c.pattern_idx = p;
        c.pos = pos[k];
        c.f0 = h->freq_host[b];
        c.nbadsync = nbad[k];
        c.xb = xb[k];
        c.num_avg = kPatternNumAvg[p];
        std::memcpy(c.softbits_wo_sync, &llr[k * kCodeBits], sizeof(float) * kCodeBits);
        if(flag[k])
        {
            c.is_message_present = 1;
            c.ldpc_num_iterations = iter[k];
            c.ldpc_num_hard_errors = nhard[k];
            for(int i = 0; i < kMessageBits; i++) c.message[i] = static_cast<char>((msg[k * 3 + i / 32] >> (31 - (i % 32))) & 1u);
        }
    }
    return MSK144_OK;
}

int msk144_dump_indexes(msk144_handle* h, int32_t channel, int32_t* out, int32_t* n)
{
    if(!h || !out || !n || channel < 0 || channel >= h->st.channels) return fail(h, MSK144_EINVAL, "bad argument");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    HIP_TRY(h, hipMemcpy(n, h->st.n_idx + channel, sizeof(int32_t), hipMemcpyDeviceToHost));
    if(*n > 0) HIP_TRY(h, hipMemcpy(out, h->st.idx + static_cast<size_t>(channel) * h->st.K, sizeof(int32_t) * (*n), hipMemcpyDeviceToHost));
    return MSK144_OK;
}

int msk144_load_candidates(msk144_handle* h, int32_t channel, const msk144_candidate* items)
{
    if(!h || !items || channel < 0 || channel >= h->st.channels) return fail(h, MSK144_EINVAL, "bad argument");
    if(!h->retained) return fail(h, MSK144_ENOTRETAINED, "a handle that does not retain its LLR rows cannot take loaded candidates; create the handle with llr_block_channels = channels");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    const DeviceStore& st = h->st;
    const size_t K = st.K;
    const size_t off = static_cast<size_t>(channel) * K;
    std::vector<uint32_t> pos(K);
    std::vector<float> xb(K), llr(K * kCodeBits);
    std::vector<int32_t> nbad(K);
    for(size_t k = 0; k < K; k++)
    {
        pos[k] = items[k].pos;
        xb[k] = items[k].xb;
        nbad[k] = items[k].nbadsync;
        std::memcpy(&llr[k * kCodeBits], items[k].softbits_wo_sync, sizeof(float) * kCodeBits);
    }
    HIP_TRY(h, hipMemcpy(st.pos + off, pos.data(), K * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(st.xb + off, xb.data(), K * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(st.nbadsync + off, nbad.data(), K * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(st.llr + off * kCodeBits, llr.data(), K * kCodeBits * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemsetAsync(st.dec_flag + off, 0, K, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MSK144_OK;
}

int msk144_clock_probe(msk144_handle* h, int32_t spin_us, float* shader_mhz)
{
    if(!h || !shader_mhz) return fail(h, MSK144_EINVAL, "null argument");
    if(spin_us < 1 || spin_us > 100000) return fail(h, MSK144_EINVAL, "spin_us must be 1..100000");
    HIP_TRY(h, hipSetDevice(h->params.device));
    if(!h->probe_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->probe_stream, hipStreamNonBlocking));
    if(!h->d_probe)
    {
        int rc = dev_alloc(h, &h->d_probe, 2);
        if(rc != MSK144_OK) return rc;
    }
    launch_clock_probe(h->d_probe, static_cast<uint32_t>(spin_us) * 100u, h->probe_stream);
    HIP_TRY(h, hipGetLastError());
    uint64_t v[2] = {0, 0};
    HIP_TRY(h, hipMemcpyAsync(v, h->d_probe, sizeof(v), hipMemcpyDeviceToHost, h->probe_stream));
    HIP_TRY(h, hipStreamSynchronize(h->probe_stream));
    if(v[1] == 0) return fail(h, MSK144_EHIP, "clock probe: the 100 MHz counter did not advance");
    *shader_mhz = static_cast<float>(static_cast<double>(v[0]) / static_cast<double>(v[1]) * 100.0);
    return MSK144_OK;
}

int msk144_set_profiling(msk144_handle* h, int32_t enable)
{
    if(!h) return MSK144_EINVAL;
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    h->profiling = enable != 0;
    return MSK144_OK;
}

int msk144_stage_times(msk144_handle* h, float* avg_ms, int32_t* samples, int32_t reset)
{
    if(!h || !avg_ms) return fail(h, MSK144_EINVAL, "null argument");
    int rc = msk144_synchronize(h);
    if(rc != MSK144_OK) return rc;
    for(int s = 0; s < MSK144_T_COUNT; s++) avg_ms[s] = h->t_cnt[s] ? static_cast<float>(h->t_sum[s] / h->t_cnt[s]) : 0.0f;
    if(samples)
        for(int s = 0; s < MSK144_T_COUNT; s++) samples[s] = h->t_cnt[s];
    if(reset)
    {
        for(int s = 0; s < MSK144_T_COUNT; s++)
        {
            h->t_sum[s] = 0.0;
            h->t_cnt[s] = 0;
        }
    }
    return MSK144_OK;
}

}  // extern "C"
