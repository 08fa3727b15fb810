// TEST INFRASTRUCTURE - C entry points over the REFERENCE's own host-side classes.
//
// This file contains no reference code: it #includes the reference headers where they lie
// (/root/reference/src, -I on the command line, see oracle/ref/Makefile) and is linked with objects compiled
// from the reference's UNMODIFIED sources result_filter.cpp (plain g++) and snr_tracker.cu (clang, host side
// only: `hipcc -x hip --offload-host-only`, which supplies __host__/__device__ and the image's rocThrust
// <thrust/host_vector.h>; no hipify, nothing is translated and no device code is produced).  Output:
// oracle/_ref/libmsk144_ref_host.so, git-ignored.  Used only by tests/ and tests/golden/make_ref_host_fixtures.py
// to pin host/result_filter.cpp, host/snr_tracker.cpp, the oracle's SNR tracker and the Tanner-graph table to
// the reference's compiled behaviour.  Everything else on the hot path is CUDA (cuda* runtime calls, 32-lane
// __shfl_*_sync, cuFFT) and cannot be built here - see DESIGN.md section 2.
#include <cstring>
#include <string>
#include <vector>

#include <thrust/copy.h>
#include <thrust/device_free.h>
#include <thrust/device_malloc.h>
#include <thrust/host_vector.h>

#include "result_filter.h"   // /root/reference/src
#include "snr_tracker.h"     // /root/reference/src (pulls common.h, smath_complex.h)
#include "ldpc_context.cuh"  // /root/reference/src: the ldpc_reverse_map table as the compiler sees it

extern "C" {

void* ref_filter_new() { return new ResultFilter(); }
void ref_filter_free(void* f) { delete static_cast<ResultFilter*>(f); }
void ref_filter_block_begin(void* f) { static_cast<ResultFilter*>(f)->blockBegin(); }
void ref_filter_put(void* f, int snr, float f0, int num_avg, int nbadsync, int pattern_idx, const char* msg)
{
    static_cast<ResultFilter*>(f)->putMessage(snr, f0, num_avg, nbadsync, pattern_idx, std::string(msg));
}
void ref_filter_block_end(void* f) { static_cast<ResultFilter*>(f)->blockEnd(); }
int ref_filter_count(void* f) { return static_cast<int>(static_cast<ResultFilter*>(f)->getBlockResult().size()); }
int ref_filter_get(void* f, int i, int* snr, float* f0, int* num_avg, int* nbadsync, int* pattern_idx, char* msg, int cap)
{
    const auto& r = static_cast<ResultFilter*>(f)->getBlockResult();
    if(i < 0 || i >= static_cast<int>(r.size())) return -1;
    *snr = r[i].snr;
    *f0 = r[i].f0;
    *num_avg = r[i].num_avg;
    *nbadsync = r[i].nbadsync;
    *pattern_idx = r[i].pattern_idx;
    std::strncpy(msg, r[i].message.c_str(), cap - 1);
    msg[cap - 1] = 0;
    return static_cast<int>(r[i].updateStampAsString().size());  // 14 = YYYYmmddHHMMSS
}

void* ref_snr_new() { return new SNRTracker(); }
void ref_snr_free(void* s) { delete static_cast<SNRTracker*>(s); }
// iq: `length` interleaved (re, im) float pairs = the analytic window the reference copies back (main.cu:327,388)
int ref_snr_process(void* s, const float* iq, unsigned length)
{
    static_assert(sizeof(Complex) == 2 * sizeof(float), "smath::Complex is two floats");
    static_cast<SNRTracker*>(s)->process_data(reinterpret_cast<const Complex*>(iq), length);
    return static_cast<SNRTracker*>(s)->getSNRI();
}
float ref_snr_float(void* s) { return static_cast<SNRTracker*>(s)->getSNRF(); }

// [128][3][2] chars: (slot row, check) of edge k of bit n (ldpc_context.cuh:10-139)
const char* ref_ldpc_reverse_map(int* bytes)
{
    *bytes = static_cast<int>(sizeof(ldpc_reverse_map));
    return &ldpc_reverse_map[0][0][0];
}
int ref_crc13_poly() { return CRC13_POLY; }
int ref_geometry(int which)
{
    switch(which)
    {
    case 0: return Num864;
    case 1: return Num6x864;
    case 2: return Num42;
    case 3: return SecondSyncBase;
    case 4: return NumberOfLDPCIterations;
    case 5: return NumCandidatesPerPattern;
    case 6: return ScanDepthMax;
    case 7: return NumScanThreads;
    default: return -1;
    }
}

}  // extern "C"
