// Internal interface between the C-ABI host code (msk144_api.cpp) and the gfx950 kernels.
// One launcher per reference kernel; every launcher only enqueues work on `stream`.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msk144_protocol.h"

namespace msk144
{

// 42-tap sync template and the 12-sample half-sine, built on the host exactly like
// msk_context.cuh:137-196 and handed to kernels by value (kernarg -> SGPRs).
struct SyncTemplate
{
    float re[kSyncTaps];
    float im[kSyncTaps];
    float pp[12];
};

// Candidate store, structure-of-arrays, [channel][item] with item k = b*D*8 + p*8 + slot
// (result_keeper.cuh:85-91).  Replaces the reference's 632-byte array-of-structs ResultItem.
struct DeviceStore
{
    int32_t channels;
    int32_t F;                // frequency hypotheses
    int32_t D;                // scan depth (patterns)
    int32_t K;                // items per channel = F*D*8
    int32_t nbadsync_threshold;
    int32_t max_results;
    int32_t channel_base;     // added to the channel number in result records (multi-GPU sharding)
    // Blocked staging: softbits / index / LDPC are launched per block of channels [ch0, ch0 + nch), so the LLR store is
    // llr_block x K x 512 B however large the batch is (a 128-channel block: 1.58 GB at the deep config; the reference keeps
    // 512 B of softbits in every 632-byte item).  llr is indexed by (channel - ch0), every other array by the absolute
    // channel.  scan, front ends and collect always cover all `channels`.
    int32_t ch0;
    int32_t nch;
    // LLR rows do not outlive their block (the handle's block is smaller than its channel capacity): a candidate the nbadsync
    // gate is going to drop is not demodulated beyond its sync check (softbits_kernel<true>)
    int32_t gate_early;
    // Copies handed over (only with gate_early; msk144_set_copy_handover): a slot whose position folds the same frames as a LOWER slot
    // of its (frequency, pattern) group is not demodulated or decoded, it reports that slot's result.  0: every slot is computed on its
    // own, as the reference does (softbits_kernel.cuh:56-83, ldpc_kernel.cuh:100-249)
    int32_t handover;

    const float* freq;        // [F] Hz, host-computed as msk_context.cuh:135
    const float2* cb42;       // [42] sync template (re, im), for kernels that index it per lane
    float2* analytic;         // [channels][5184] front-end output
    float* seg_power;         // [channels][8]

    uint32_t* pos;            // [channels][K]
    float* xb;                // [channels][K]
    int32_t* nbadsync;        // [channels][K]
    float* llr;               // [llr_block][K][128], row of (channel, item) = ((channel - ch0) * K + item) * 128
    int32_t* idx;             // [channels][K] gated item numbers, ascending
    int32_t* n_idx;           // [channels]
    uint8_t* dec_flag;        // [channels][K] is_message_present
    uint8_t* dec_iter;        // [channels][K]
    uint8_t* dec_nhard;       // [channels][K]
    uint32_t* dec_msg;        // [channels][K][3]  77 bits MSB first in 96

    int32_t* dec_count;       // [channels] decodes per channel
    int32_t* copy_count;      // [channels] slots of the last decode that were handed to a lower slot (collect stage)
    int32_t* result_count;    // [1]
    void* results;            // [max_results] msk144_result
};

// front ends (frontend.hip)
void launch_frontend_audio(const DeviceStore& st, const int16_t* d_in, int analytic_method, const float2* d_twiddle, const float* d_fft_mask,
                           hipStream_t stream);
void launch_frontend_iq(const DeviceStore& st, const int8_t* d_in, hipStream_t stream);

// device-side window ring of every stream (hopring.hip): ring[streams[j]] advances by the hop at position j (or is filled from
// first_halves[j] + hops[j] when is_first[j]); windows[j] = the stream's new window.  Halves and windows in raw input bytes.
void launch_hop_ring(void* ring, const void* hops, const void* first_halves, const int32_t* streams, const uint8_t* is_first, void* windows, int n, hipStream_t stream);

// one wave that spins for `ticks` of the 100 MHz counter; out[0] = shader cycles elapsed, out[1] = 100 MHz ticks elapsed (hopring.hip)
void launch_clock_probe(uint64_t* out, uint32_t ticks, hipStream_t stream);

// hot kernels
void launch_scan(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream);
void launch_softbits(const DeviceStore& st, const SyncTemplate& tpl, hipStream_t stream);
void launch_index(const DeviceStore& st, hipStream_t stream);
void launch_ldpc(const DeviceStore& st, hipStream_t stream);
void launch_collect(const DeviceStore& st, hipStream_t stream);

}  // namespace msk144
