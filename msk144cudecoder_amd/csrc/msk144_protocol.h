// MSK144 protocol constants shared by the HIP kernels, the host code and the CPU oracle.
//
// Everything here is a property of the MSK144 air interface / of the search grid the reference
// decoder uses, restated in this project's own layout (check-major Tanner graph instead of the
// reference's bit-major edge map).  Reference locations (relative to /root/reference/src):
//   frame geometry, thread/slot counts ............ common.h:14-47
//   sync word, half-sine pulse, 42-tap template ... msk_context.cuh:137-154,176-227
//   averaging patterns ............................ msk_context.cuh:229-248
//   (128,90) LDPC Tanner graph, CRC-13 polynomial . ldpc_context.cuh:7-139,160-163
//   FIR taps of the default front end ............. analytic2.cuh:145-159
//
// Plain C++17, no HIP dependency: usable from g++ (oracle, host) and hipcc (device code may
// index the constexpr tables at run time).
#pragma once

#include <cstdint>

namespace msk144
{

// ---- frame geometry (common.h:14-27) ----
constexpr int kFrameSamples = 864;               // 144 bits x 6 samples, 72 ms at 12 kHz
constexpr int kWindowSamples = 6 * 864;          // 5184 samples = one decode window
constexpr int kHopSamples = kWindowSamples / 2;  // 50 % overlap between windows (main.cu:284)
constexpr int kSyncTaps = 42;                    // 7 half-sine pulses x 6 samples
constexpr int kSecondSyncSample = (8 + 48) * 6;  // 336: second sync word inside a frame
constexpr int kSecondSyncBit = 8 + 48;           // 56
constexpr int kSoftBits = 144;
constexpr int kCodeBits = 128;
constexpr int kMessageBits = 77;
constexpr int kCrcBits = 13;
constexpr int kChecks = 38;
constexpr int kMaxCheckDegree = 11;
constexpr int kEdgesPerBit = 3;
constexpr int kLdpcIterations = 10;              // common.h:27
constexpr int kMaxHardErrors = 18;               // accept iff nhard < 18 (ldpc_kernel.cuh:209)
constexpr float kSampleRate = 12000.0f;

// ---- scan geometry (common.h:29-45, scan_kernel.cuh:85-89) ----
constexpr int kSlicePositions = 256;                    // one slice = 256 consecutive positions
constexpr int kScanSlices = 21;                         // ceil(5184/256): positions 0..5375
constexpr int kScanPositions = kScanSlices * kSlicePositions;  // 5376 (192 past the window)
constexpr int kSlotsPerPattern = 8;                     // candidates kept per (frequency, pattern)
constexpr int kScanDepthMax = 8;
constexpr int kPatternBits = 6;

// Frame-averaging masks, bit m set => frame m of the window takes part (msk_context.cuh:231-238).
constexpr uint8_t kPatternMask[kScanDepthMax][kPatternBits] = {
    {1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 0, 0},
    {1, 1, 1, 1, 1, 0}, {1, 1, 1, 1, 1, 1}, {1, 0, 0, 1, 0, 0}, {1, 0, 0, 1, 1, 0},
};
constexpr int kPatternNumAvg[kScanDepthMax] = {1, 2, 3, 4, 5, 6, 2, 3};
// Patterns 5 (111111) and 6 (100100) sum the same frames at pos and at pos + 864 (pattern 5) / pos + 2592 (pattern 6): their
// correlation and their folded frame are periodic in the position.
constexpr int kFirstPeriodicPattern = 5;
constexpr int kPatternPeriod[2] = {864, 2592};

// MSK144 sync word (msk_context.cuh:149); kSync8Pm = 2*s8-1.
constexpr int kSync8[8] = {0, 1, 1, 1, 0, 0, 1, 0};
constexpr int kSync8Pm[8] = {-1, 1, 1, 1, -1, -1, 1, -1};

// CRC-13 generator polynomial (ldpc_context.cuh:7).
constexpr uint16_t kCrc13Poly = 0x15D7;

// ---- default front end: 15-tap FIR, taps 3 and 13 are zero and skipped (analytic2.cuh:145-159) ----
constexpr int kFirTaps = 13;
constexpr int kFirTapIndex[kFirTaps] = {1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15};
constexpr float kFirTapValue[kFirTaps] = {
    -0.04225694f, -0.03046893f, 0.04570339f, 0.09859952f, 0.14789927f, 0.18281356f, 0.19542026f,
    0.18281356f,  0.14789927f,  0.09859952f, 0.04570339f, -0.03046893f, -0.04225694f,
};
constexpr int kFirPad = 32;                                  // zero guard on each side of the window
constexpr int kFirBuffer = kWindowSamples + 2 * kFirPad;     // 5248
constexpr float kSin45 = 0.707106781f;                       // analytic2.cuh:9

// ---- FFT front end (analytic_fft.cu:18-57) ----
constexpr int kFftSize = 8192;

// ---- (128,90) LDPC code, check-major ----
// kCheckBits[c][j] = codeword bit sitting in slot j of parity check c, -1 = empty slot.
// Bits inside a check are ascending; the three checks of a bit, taken in ascending check order,
// are its edges k = 0,1,2.  Both orderings were verified against ldpc_context.cuh:10-139
// (oracle/tools/inspect_ref_graph.py), so slot/edge numbering - and with it the order of every
// floating-point sum and product in the BP decoder - is the reference's.
constexpr int8_t kCheckBits[kChecks][kMaxCheckDegree] = {
    {1, 14, 26, 39, 52, 64, 76, 90, 93, 114, -1},
    {2, 5, 27, 40, 53, 65, 77, 91, 93, 119, -1},
    {3, 15, 28, 41, 54, 66, 76, 89, 92, 105, 117},
    {4, 16, 29, 42, 51, 63, 78, 91, 101, 118, -1},
    {5, 17, 30, 43, 55, 67, 79, 88, 94, 107, 124},
    {6, 13, 31, 44, 56, 67, 78, 89, 95, 115, 120},
    {3, 18, 32, 42, 57, 68, 80, 96, 106, 120, -1},
    {1, 19, 29, 38, 53, 69, 79, 97, 106, 127, -1},
    {2, 20, 33, 45, 58, 66, 78, 98, 106, 122, -1},
    {7, 14, 28, 46, 55, 70, 81, 99, 110, 127, -1},
    {8, 21, 33, 43, 51, 71, 82, 100, 102, 125, -1},
    {9, 16, 25, 47, 59, 72, 83, 90, 109, 120, -1},
    {6, 22, 34, 37, 54, 72, 81, 100, 108, 119, -1},
    {10, 18, 35, 48, 52, 69, 84, 101, 103, 125, -1},
    {9, 19, 36, 45, 57, 70, 84, 104, 108, 121, -1},
    {4, 22, 36, 46, 56, 73, 85, 92, 109, 124, -1},
    {11, 12, 26, 40, 60, 67, 86, 96, 108, 112, -1},
    {10, 15, 37, 44, 57, 71, 77, 98, 107, 114, -1},
    {3, 21, 30, 40, 59, 73, 81, 104, 111, 114, -1},
    {11, 23, 31, 38, 61, 62, 87, 98, 101, 117, -1},
    {0, 18, 24, 44, 61, 74, 76, 99, 111, 118, -1},
    {5, 24, 32, 49, 58, 70, 82, 97, 116, 117, -1},
    {9, 15, 38, 50, 52, 65, 82, 94, 110, 123, -1},
    {8, 23, 34, 41, 58, 75, 88, 93, 113, 121, -1},
    {6, 16, 36, 49, 53, 74, 87, 110, 113, 122, -1},
    {10, 24, 34, 47, 60, 64, 87, 104, 115, 124, -1},
    {8, 20, 31, 48, 54, 68, 83, 85, 94, 112, 118},
    {1, 17, 27, 46, 62, 72, 86, 95, 105, 125, -1},
    {11, 25, 30, 48, 63, 71, 80, 99, 105, 119, -1},
    {12, 14, 27, 47, 50, 75, 84, 92, 102, 122, -1},
    {7, 19, 35, 43, 56, 74, 77, 90, 112, 116, -1},
    {4, 20, 28, 39, 50, 69, 80, 95, 116, 121, -1},
    {7, 25, 33, 39, 61, 73, 79, 91, 115, 126, -1},
    {0, 12, 13, 35, 41, 63, 65, 83, 100, 107, -1},
    {13, 21, 26, 49, 62, 68, 88, 103, 109, 127, -1},
    {0, 17, 29, 45, 59, 64, 89, 96, 113, 126, -1},
    {2, 22, 32, 51, 55, 75, 86, 103, 111, 123, -1},
    {23, 37, 42, 60, 66, 85, 97, 102, 123, 126, -1},
};

// Per-candidate record layout of the reference's device result array
// (result_keeper.cuh:17-32, sizeof == 632).  Mirrored by msk144_candidate in include/msk144hip.h
// and by the oracle so parity dumps compare field by field.
constexpr int kReferenceResultItemBytes = 632;

// Number of frequency hypotheses and the offset of the first one, as the reference derives them
// (msk_context.cuh:95-107): half = int((width/2)/step), F = 2*half+1, if1 = -half*step.
inline int grid_half_len(float search_width, float search_step)
{
    const float half_len_in_hz = search_width / 2;
    const float half_len_cnt = half_len_in_hz / search_step;
    return static_cast<int>(half_len_cnt);
}

// Scan depth clamp (msk_context.cuh:29-33): 1..8 (NumPatternBitsToScan == FixedNumBitsInPattern
// so the third clamp never fires).
inline int clamp_scan_depth(int d)
{
    if(d < 1) d = 1;
    if(d > kScanDepthMax) d = kScanDepthMax;
    return d;
}

}  // namespace msk144
