/*
 * msk144hip - C ABI of the MI355X-native MSK144 hot path (libmsk144hip.so).
 *
 * The reference (alexander-sholohov/msk144cudecoder @ 2024_10_08) has no plugin/FFI interface: its
 * only external boundary is the process (raw samples on stdin, text on stdout).  This ABI is cut
 * along the reference's internal seams so that its main loop can call it instead of its own CUDA:
 *
 *   msk144_create            MSK144SearchContext ctor + ResultKeeper::init + LDPCContext::init +
 *                            Analytic(8192)                       msk_context.cuh:23-38, main.cu:211-226
 *   msk144_submit_audio      rms normalise, int16->complex, apply_shift_filter_shift<<<1,32>>> or
 *                            Analytic::execute                    main.cu:300-332
 *   msk144_submit_iq         int8 I/Q -> complex, apply_filter<<<1,32>>>   main.cu:365-380
 *   msk144_decode            clear_result + scan_kernel + softbits_kernel + index_kernel +
 *                            ldpc_kernel                          main.cu:461-468
 *   msk144_results           get_all_results() + the is_message_present filter of the host loop
 *                                                                 main.cu:477-484
 *   msk144_segment_power     the 8 segment powers SNRTracker::process_data sums from the analytic
 *                            window                               snr_tracker.cu:21-37, main.cu:388
 *   msk144_hop_slot,         the 50 %-overlap window ring kept on the device: 2592 new samples per stream and hop travel, not
 *   msk144_push_hops         5184-sample windows             main.cu:271-294, 337-359
 *   msk144_input_slot ..     the same hop, pipelined over two pinned staging slots (fread buffer -> H2D -> kernels ->
 *   msk144_fetch_wait        D2H of what the host loop consumes)  main.cu:261-422, 474-525
 *   msk144_dump_candidates   the raw ResultItem array (parity/debug)     result_keeper.cuh:17-32,123-130
 *   msk144_destroy           ~MSK144SearchContext / deinit        msk_context.cuh:81-120
 *   msk144_device_count      cudaGetDeviceCount behind cudaSetDeviceFlags (the reference drives device 0 only; the multi-device
 *                            stream program asks how many it may split its streams over)   main.cu:115
 *   msk144_llr_block_channels how many channels' softbits the handle keeps at a time (the reference keeps 512 B in every ResultItem of its one
 *                            stream, result_keeper.cuh:17-32; here a block of channels shares one LLR store)
 *   msk144_set_llr_retention whether the 128 softbits of every candidate stay readable after the decode, as in the reference's
 *                            ResultItem array (result_keeper.cuh:17-32, 105-115) - the stream program never reads them
 *   msk144_set_copy_handover whether slots that fold the same frames as a lower slot of their group are computed again, as
 *   msk144_copy_count        softbits_kernel / ldpc_kernel do for every slot (softbits_kernel.cuh:56-83, ldpc_kernel.cuh:100-249)
 *   msk144_clock_probe       gpu_timer.h's role for the one figure HIP events cannot give: the shader clock a running batch
 *                            actually gets (s_memtime / s_memrealtime), read beside it on a side stream
 *
 * One handle = one device + one HIP stream + `channels` independent input streams decoded per call
 * (the reference decodes one).  Plain pointers and sizes only; no exceptions cross the boundary:
 * every entry returns 0 or a negative MSK144_E* code, msk144_last_error() gives the text.
 * Handles are independent of each other; a handle is not re-entrant.
 */
#ifndef MSK144HIP_H
#define MSK144HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSK144_WINDOW_SAMPLES 5184 /* 6 frames of 864 samples (common.h:15) */
#define MSK144_HOP_SAMPLES 2592    /* 50 % overlap (main.cu:284) */
#define MSK144_CODE_BITS 128
#define MSK144_MESSAGE_BITS 77

enum
{
    MSK144_OK = 0,
    MSK144_EINVAL = -1,   /* bad argument / parameter */
    MSK144_EHIP = -2,     /* HIP runtime error (text in msk144_last_error) */
    MSK144_ENOMEM = -3,   /* device or host allocation failed */
    MSK144_ESTATE = -4,   /* call out of order (e.g. decode before submit) */
    MSK144_EOVERFLOW = -5, /* more decodes than max_results; results truncated */
    MSK144_ENOTRETAINED = -6 /* blocked staging does not retain what was asked for (LLR rows of an earlier block, or a partial
                                stage run); create the handle with llr_block_channels = channels */
};

/* msk144_decode_stages bits, in launch order (main.cu:463-467) */
enum
{
    MSK144_STAGE_SCAN = 1,
    MSK144_STAGE_SOFTBITS = 2,
    MSK144_STAGE_INDEX = 4,
    MSK144_STAGE_LDPC = 8,
    MSK144_STAGE_COLLECT = 16,
    MSK144_STAGE_ALL = 31
};

/* indices into msk144_stage_times */
enum
{
    MSK144_T_FRONTEND = 0,
    MSK144_T_SCAN = 1,
    MSK144_T_SOFTBITS = 2,
    MSK144_T_INDEX = 3,
    MSK144_T_LDPC = 4,
    MSK144_T_COLLECT = 5,
    MSK144_T_H2D = 6,  /* host -> device copy of the submitted windows (msk144_submit_audio/_iq/_slot) */
    MSK144_T_D2H = 7,  /* device -> host copy of count, records and segment powers (msk144_fetch_async) */
    MSK144_T_COUNT = 8
};

#define MSK144_SLOTS 2 /* pinned staging slots per handle (msk144_input_slot .. msk144_fetch_wait) */

typedef struct msk144_params
{
    float center_hz;        /* --center-frequency (1500 audio / 0 IQ) */
    float width_hz;         /* --search-width  (200) */
    float step_hz;          /* --search-step   (2)   */
    int32_t scan_depth;     /* --scan-depth    (4), clamped to 1..8 like msk_context.cuh:29-33 */
    int32_t nbadsync_threshold; /* --nbadsync-threshold (1) */
    int32_t read_mode;      /* --read-mode: 1 = int16 audio, 2 = int8 I/Q */
    int32_t analytic_method;/* --analytic-method: 1 = FFT, 2 = shift-filter-shift (audio only) */
    int32_t channels;       /* independent streams decoded per call (reference: 1) */
    int32_t device;         /* HIP device ordinal */
    int32_t max_results;    /* capacity of the compact result list; 0 = default */
    int32_t llr_block_channels; /* channels per softbits->index->LDPC block (reference: result_keeper.cuh:105-115 keeps every
                                   candidate's 128 softbits; ldpc_kernel.cuh:116-142 reads them back).  The LLR rows of a block
                                   (block x items x 512 B) are produced and consumed back to back, so the LLR store never grows
                                   with the batch (1.58 GB per 128-channel block at the deep config instead of 12.6 GB per 1024
                                   channels).  0 = automatic: min(channels, 128), the bottom of a flat valley on the 1024-channel bench
                                   step (48: +1.4 %, 64: +0.8 %, 96: +0.3 %, 160: +0.1 %, 192: +0.3 %, 256: +0.9 % step time).  With fewer channels per block than channels no row outlives
                                   its block, and a candidate the nbadsync gate drops (index_kernel.cuh:7-76) is not demodulated
                                   beyond its sync check.  Candidate dumps need every row retained:
                                   = channels: retain everything, every candidate in full (parity-dump mode) */
} msk144_params;

/* One accepted decode (CRC ok, < 18 hard errors), fields as the reference's host loop consumes them
 * (main.cu:484-522). */
typedef struct msk144_result
{
    int32_t channel;
    int32_t item;            /* k = block_idx*D*8 + pattern_idx*8 + slot inside the channel */
    float f0;                /* Hz */
    int32_t pattern_idx;
    int32_t num_avg;
    uint32_t pos;
    float xb;
    int32_t nbadsync;
    int32_t ldpc_iterations;
    int32_t ldpc_hard_errors;
    uint8_t message[10];     /* 77 payload bits, MSB first, zero padded */
    uint8_t reserved[2];
} msk144_result;

/* Same layout as the reference's ResultKeeper::ResultItem (result_keeper.cuh:17-32), 632 bytes. */
typedef struct msk144_candidate
{
    uint32_t block_idx;
    uint32_t pattern_idx;
    uint32_t pos;
    float f0;
    int32_t nbadsync;
    float xb;
    int32_t num_avg;
    float softbits_wo_sync[128];
    uint8_t is_message_present;
    int32_t ldpc_num_iterations;
    int32_t ldpc_num_hard_errors;
    char message[77];
} msk144_candidate;

typedef struct msk144_handle msk144_handle;

/* defaults exactly as main.cu:122-133 (NOT the help text) */
void msk144_default_params(msk144_params* p);

/* HIP devices visible to this process (0 and MSK144_EHIP when there is none): what msk144_params.device may range over.  The
 * reference binds the process to one device (main.cu:115); msk144hipdecoder --devices=... creates one handle per listed ordinal. */
int msk144_device_count(int32_t* n);

int msk144_create(const msk144_params* params, msk144_handle** out);
void msk144_destroy(msk144_handle* h);
/* h may be NULL: error text of the last failed msk144_create on this thread */
const char* msk144_last_error(const msk144_handle* h);

/* F frequency hypotheses, D patterns, items per channel = F*D*8 */
int msk144_geometry(const msk144_handle* h, int32_t* num_freqs, int32_t* scan_depth, int32_t* items_per_channel);
int msk144_frequency(const msk144_handle* h, int32_t block_idx, float* hz);

/* Run on a caller-owned HIP stream (hipStream_t passed as void*); NULL = the handle's own stream. */
int msk144_set_stream(msk144_handle* h, void* hip_stream);

/* Front end.  Host buffers are copied H2D on the handle's stream; *_device variants take device
 * pointers that must stay valid until the front-end kernel has run.
 *   audio: int16 [channels][5184];   iq: int8 [channels][2*5184] interleaved I,Q. */
int msk144_submit_audio(msk144_handle* h, const int16_t* windows);
int msk144_submit_iq(msk144_handle* h, const int8_t* windows);
int msk144_submit_audio_device(msk144_handle* h, const int16_t* d_windows);
int msk144_submit_iq_device(msk144_handle* h, const int8_t* d_windows);
/* bypass the front end: complex64 (re,im) [channels][5184], host memory (parity tests) */
int msk144_submit_analytic(msk144_handle* h, const float* windows);

/* Asynchronous: enqueues the kernels and returns. */
int msk144_decode(msk144_handle* h);
int msk144_decode_stages(msk144_handle* h, uint32_t stages);
int msk144_synchronize(msk144_handle* h);

/* Waits for the decode, copies the compact result list.  *n = number of decodes (<= cap copied). */
int msk144_results(msk144_handle* h, msk144_result* out, int32_t cap, int32_t* n);
int msk144_result_count(msk144_handle* h, int32_t* n);
/* device-side list for callers that gather on the GPU (RCCL): records + count stay valid until the
 * next decode */
int msk144_results_device(msk144_handle* h, const msk144_result** d_records, const int32_t** d_count);
/* Multi-GPU sharding (no reference counterpart: the reference decodes one stream on one GPU, main.cu:115):
 * msk144_result.channel = base + local channel, so that the records of a rank that owns channels
 * [base, base+channels) carry global channel ids when they are gathered.  Default 0. */
int msk144_set_channel_base(msk144_handle* h, int32_t base);

/* Copies (no reference counterpart: the reference demodulates and decodes every one of its F*D*8 slots, softbits_kernel.cuh:56-83,
 * ldpc_kernel.cuh:100-249).  Two slots of one (frequency, pattern) group whose scan positions are congruent modulo the 5184-sample
 * ring - or, for masks 111111 / 100100, modulo their period 864 / 2592 - fold the SAME frames.  In blocked staging
 * (llr_block_channels < channels: no LLR row outlives its block) such a slot is by default not computed again: it reports the
 * nbadsync, iterations, hard errors and payload of the LOWEST congruent slot of its group, with its own position and xb.
 *   msk144_set_copy_handover(h, 0)  every slot is demodulated and decoded on its own, as in the reference (still blocked staging,
 *                                   still no demodulation beyond the sync check for candidates the nbadsync gate drops);
 *   msk144_set_copy_handover(h, 1)  the default of a blocked handle; MSK144_ENOTRETAINED on a handle that retains every LLR row
 *                                   (such a handle always computes every slot: its dumps show each slot's own row).
 * Takes effect at the next decode.  msk144_copy_count: slots of the last decode that were handed over (0 when switched off). */
int msk144_set_copy_handover(msk144_handle* h, int32_t enable);
/* A handle whose one block covers all its channels (llr_block_channels = channels; the default up to 128 channels, e.g. the single
 * stream of the reference's program) keeps every candidate's 128 softbits readable after the decode, as the reference's ResultItem
 * array does (result_keeper.cuh:17-32): msk144_dump_candidates, partial stage runs and msk144_load_candidates work, and every slot is
 * demodulated in full.  A caller that only reads the result list - the stream program - says so with
 *   msk144_set_llr_retention(h, 0)  the handle behaves like a blocked one: a candidate the nbadsync gate drops stops after its sync
 *                                   check, copies are handed over (above), dumps / partial stage runs / loaded candidates are refused
 *                                   with MSK144_ENOTRETAINED - the kernels bench.py times at 1024 channels;
 *   msk144_set_llr_retention(h, 1)  back to retaining; MSK144_ENOTRETAINED on a handle created with fewer channels per block than
 *                                   channels.
 * Takes effect at the next decode. */
int msk144_set_llr_retention(msk144_handle* h, int32_t retain);
/* channels per softbits -> index -> LDPC block this handle decodes in (msk144_params.llr_block_channels after the automatic choice) */
int msk144_llr_block_channels(const msk144_handle* h, int32_t* channels_per_block);
int msk144_copy_handover(const msk144_handle* h, int32_t* enabled);
int msk144_copy_count(msk144_handle* h, int64_t* slots);

int msk144_segment_power(msk144_handle* h, float* out /*[channels][8]*/);

/* Pinned staging, two slots ("handle owns device + pinned host buffers").  The reference's loop is strictly serial
 * per hop - fread, H2D, kernels, cudaDeviceSynchronize, 15 MB D2H, host loop, print (main.cu:261-422, 461-525).  With the slots a
 * caller fills the windows of hop n+1 and reads the results of hop n-1 while the GPU decodes hop n, and one hop costs one
 * asynchronous H2D and one asynchronous D2H of exactly what the host loop consumes (count, compact records, 8 segment powers per
 * channel) instead of five blocking calls:
 *
 *     msk144_input_slot(h, s, &win, &bytes);   fill win[channels][5184] (int16) or [channels][2*5184] (int8), pinned
 *     msk144_submit_slot(h, s);                H2D + front end                      (asynchronous)
 *     msk144_decode(h);                        kernels; the record list of slot s   (asynchronous)
 *     msk144_fetch_async(h, s);                D2H into the slot's pinned output    (asynchronous)
 *     ... other work, e.g. the same four calls for slot 1 - s ...
 *     msk144_fetch_wait(h, s, &rec, &n, &seg); blocks until slot s has arrived; rec/seg point into the slot (valid until the slot
 *                                              is submitted again)
 *
 * Each slot has its own device-side record list, so the list of slot s stays intact while slot 1 - s decodes.  The asynchronous
 * copy covers a running estimate of the record count (twice the last count, at least 1024); msk144_fetch_wait copies a
 * remainder synchronously.  The buffers are allocated at the first msk144_input_slot / msk144_fetch_async call of a handle.
 * msk144_fetch_wait may be called from a second thread while the first thread submits the OTHER slot; everything else about a
 * handle stays single-threaded.  msk144_fetch_wait returns MSK144_EOVERFLOW (records truncated to max_results) like
 * msk144_results. */
int msk144_input_slot(msk144_handle* h, int32_t slot, void** host_windows, size_t* bytes);
int msk144_submit_slot(msk144_handle* h, int32_t slot);
/* The same for a hop that covers only the first n_channels windows of the slot (1..channels): every kernel, copy and result of
 * the hop is sized for n_channels, so a partial batch - streams that lag sit it out - costs what its streams cost, not what the
 * handle's capacity costs.  Record channel numbers are positions in the slot. */
int msk144_submit_slot_n(msk144_handle* h, int32_t slot, int32_t n_channels);
/* Library-side hop ring (the reference's host keeps the 50 %-overlap window itself: main.cu:284-288 / 349-353 copy the second half
 * over the first and fread 2592 new samples behind it; the first read fills all 5184, main.cu:271-283).  With these two calls the
 * DEVICE keeps every stream's window: per hop the caller writes, for each of the n streams that have one, the 2592 new samples into
 * hops[j], the stream's number (0 .. channels-1, ascending) into streams[j] and is_first[j] = 0 - or, for a stream's very first
 * hop, its first 2592 samples into first_halves[j], the second 2592 into hops[j] and is_first[j] = 1.  msk144_push_hops copies only
 * what the n hops need (half of what msk144_submit_slot_n copies), advances the rings of those streams on the device and runs the
 * front end on their n windows; msk144_decode / msk144_fetch_async / msk144_fetch_wait follow as after msk144_submit_slot_n (record
 * channel numbers are positions j).  A stream that is not listed keeps its window.  All four arrays are pinned, owned by the handle,
 * sized for `channels` entries, allocated at the first msk144_hop_slot call; sample units as in msk144_submit_audio / _iq. */
int msk144_hop_slot(msk144_handle* h, int32_t slot, void** hops /*[channels][2592 samples]*/, void** first_halves /*[channels][2592 samples]*/,
                    int32_t** streams /*[channels]*/, uint8_t** is_first /*[channels]*/);
int msk144_push_hops(msk144_handle* h, int32_t slot, int32_t n);
int msk144_fetch_async(msk144_handle* h, int32_t slot);
int msk144_fetch_wait(msk144_handle* h, int32_t slot, const msk144_result** records, int32_t* n, const float** seg_power /*[channels][8]*/);

/* parity / debug */
int msk144_dump_analytic(msk144_handle* h, int32_t channel, float* out /*[5184][2]*/);
int msk144_dump_candidates(msk144_handle* h, int32_t channel, msk144_candidate* out /*[items_per_channel]*/);
int msk144_dump_indexes(msk144_handle* h, int32_t channel, int32_t* out /*[items_per_channel]*/, int32_t* n);
/* overwrite a channel's candidate store (pos, nbadsync, softbits) so later stages can be run on
 * known inputs */
int msk144_load_candidates(msk144_handle* h, int32_t channel, const msk144_candidate* items);

/* per-stage device time (HIP events recorded on the decode stream around every launch), summed over the launches of one
 * msk144_decode / msk144_submit_* call and averaged over the calls since the last reset; samples[s] = calls measured */
int msk144_set_profiling(msk144_handle* h, int32_t enable);
int msk144_stage_times(msk144_handle* h, float* avg_ms /*[MSK144_T_COUNT]*/, int32_t* samples /*[MSK144_T_COUNT] or NULL*/, int32_t reset);

/* Shader clock the handle's device runs at right now, in MHz: a one-wave kernel on a side stream of the handle spins for spin_us
 * microseconds of the constant 100 MHz counter (s_memrealtime) and divides the shader cycles that passed (s_memtime) by it.  It runs
 * BESIDE whatever the handle's stream is executing (one wave, no LDS), so calling it right after msk144_decode reads the clock the
 * decode kernels get - the number a sustained-throughput claim needs next to its step time (DVFS: a chip that has idled for a hop
 * period starts its next batch at a lower clock).  Blocks the caller for about spin_us.  spin_us 1..100000. */
int msk144_clock_probe(msk144_handle* h, int32_t spin_us, float* shader_mhz);

#ifdef __cplusplus
}
#endif

#endif /* MSK144HIP_H */
