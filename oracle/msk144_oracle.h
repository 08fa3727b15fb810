/*
 * CPU ORACLE for the MSK144 hot path - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain C++ restatement, in float32 and in the reference's own operation order, of
 * alexander-sholohov/msk144cudecoder @ 2024_10_08 (front ends -> scan -> softbits -> index ->
 * LDPC/CRC).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * shipped decoder (libmsk144hip.so and the host code above it) never does.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path, its only
 * sample input (demo/0001.wav) is a missing blob, and its CUDA sources cannot be built or run in
 * this environment (no nvcc, no NVIDIA GPU; oracle/_ref is therefore absent).  What pins this
 * oracle instead: structural constants checked independently (sync word, CRC-13 polynomial, Tanner
 * graph weights/rank), and encode -> modulate -> decode round trips (tests/test_oracle_*.py).
 *
 * Build: `make -C oracle` (g++ -O2 -ffp-contract=off, libm transcendentals).
 * Each function cites the reference lines it follows (paths relative to /root/reference/src).
 */
#pragma once

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_complex
{
    float re, im;
} orc_complex;

/* Field-for-field image of the reference's ResultKeeper::ResultItem (result_keeper.cuh:17-32);
 * sizeof == 632, offsets as listed in SURVEY.md section 8(a) row a10. */
typedef struct orc_item
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
} orc_item;

/* Search context (msk_context.cuh:23-38,95-113). */
typedef struct orc_ctx
{
    float center_freq;
    float step;
    float if1;
    int num_blocks;          /* F = number of frequency hypotheses */
    int scan_depth;          /* D, clamped 1..8 */
    int nbadsync_threshold;
    int total_items;         /* F * D * 8 */
    int num_threads;         /* OpenMP threads used over frequency hypotheses (>=1) */
} orc_ctx;

/* which build of this file the library is: "parity" (-ffp-contract=off, what the parity tests compare with), "contract-fast" or
 * "forced-fma" (the two FMA-contracting builds of tests/test_oracle_fma_bracket.py; see the note in msk144_oracle.cpp) */
const char* orc_build_variant(void);
int orc_sizeof_item(void);
void orc_ctx_init(orc_ctx* ctx, float center_freq, float search_width, float search_step, int scan_depth, int nbadsync_threshold);
void orc_set_threads(orc_ctx* ctx, int n);
float orc_frequency(const orc_ctx* ctx, int block_idx);

/* constants as the reference builds them (msk_context.cuh:137-196) */
void orc_get_cb42(orc_complex* cb42 /*[42]*/);
void orc_get_pp12(float* pp /*[12]*/);

/* front ends: main.cu:300-332 (audio), :365-380 (IQ), analytic2.cuh, analytic_fft.cu */
void orc_normalize_audio(const int16_t* win /*[5184]*/, orc_complex* out /*[5184]*/);
void orc_convert_iq(const int8_t* win /*[2*5184]*/, orc_complex* out /*[5184]*/);
void orc_analytic2(const orc_complex* in /*[5184]*/, orc_complex* out /*[5184]*/, int with_shift);
void orc_analytic_fft(const orc_complex* in /*[5184]*/, orc_complex* out /*[5184]*/);
/* method: 1 = FFT, 2 = shift-filter-shift */
void orc_frontend_audio(const int16_t* win, int analytic_method, orc_complex* out);
void orc_frontend_iq(const int8_t* win, orc_complex* out);

/* hot kernels, one window */
void orc_clear_items(const orc_ctx* ctx, orc_item* items);
void orc_scan(const orc_ctx* ctx, const orc_complex* cdat, orc_item* items);
void orc_softbits(const orc_ctx* ctx, const orc_complex* cdat, orc_item* items);
int orc_index(const orc_ctx* ctx, const orc_item* items, int32_t* indexes);
void orc_ldpc(const orc_ctx* ctx, orc_item* items, const int32_t* indexes, int n_indexed);
/* clear + scan + softbits + index + ldpc (main.cu:461-468); returns N_idx */
int orc_decode_window(const orc_ctx* ctx, const orc_complex* cdat, orc_item* items, int32_t* indexes);

/* debug/analysis helpers used by tolerance-aware parity tests */
/* xb of every scanned position (5376) for one (frequency, pattern) */
void orc_scan_xb(const orc_ctx* ctx, const orc_complex* cdat, int block_idx, int pattern_idx, float* xb /*[5376]*/);
/* softbits of an arbitrary (frequency, pattern, pos): all 144 raw softbits, 128 LLRs, nbadsync */
void orc_softbits_at(const orc_ctx* ctx, const orc_complex* cdat, int block_idx, int pattern_idx, uint32_t pos, float* soft144, float* llr128,
                     int32_t* nbadsync);
/* BP decode of one LLR vector; returns 1 when accepted */
int orc_ldpc_one(const float* llr128, char* message77, int32_t* iters, int32_t* nhard);

/* CRC-13 exactly as ldpc_kernel.cuh:32-63 + ldpc_context.cuh:185-213 */
uint16_t orc_crc13(const uint8_t* buf, int length);
int orc_check_crc_bits(const char* cw /*[>=90] values 0/1*/);

/* host post-processing (next rows f-2): snr_tracker.cu:21-69, decode_softbits.cpp:25-30 */
typedef struct orc_snr_tracker
{
    float noise_power;
    float snr;
} orc_snr_tracker;
void orc_snr_init(orc_snr_tracker* t);
void orc_snr_process(orc_snr_tracker* t, const orc_complex* data, unsigned length);
int orc_snr_int(const orc_snr_tracker* t);
void orc_segment_power(const orc_complex* data, unsigned length, float* seg8);
int orc_message_gate(const char* message77);

#ifdef __cplusplus
}
#endif
