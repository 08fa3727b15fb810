// GENERATED offline (simulated annealing over the bit -> lane, check -> lane and first-edge-order assignments; the search script
// is described in DESIGN.md section 4) - tests/test_ldpc_layout.py re-derives the conflict counts from these tables with the
// LDS bank rules of MI355X_MICROARCH.md (ds_read_b32 / ds_write_b32: two 32-lane groups, 32 banks of 4 bytes).
//
// LDS layout of the BP message tile (ldpc.hip): lane l owns codeword bits kBitOfLane[0][l] and kBitOfLane[1][l]; check c is
// processed by lane kLaneOfCheck[c]; the cell of slot j of check c is j*kTileRowStride + kLaneOfCheck[c];
// kSwapFirstEdges[n] = 1: instruction 0 takes bit n's second edge and instruction 1 its first - free, their messages are only
// ever added to each other first ((tov0 + tov1) + tov2).
// Edge-side extra LDS cycles per iteration and access direction: 9 (worst instruction 2-way); the natural layout
// (bit n in lane n % 64, stride 40) has 18 (3-way).  The check side is conflict-free by construction.
#pragma once

#include <cstdint>

namespace msk144
{

constexpr int kTileRowStride = 38;
constexpr uint8_t kBitOfLane[2][64] = {
    {17, 3, 127, 36, 51, 64, 113, 35, 123, 15, 83, 56, 53, 23, 76, 94, 73, 115, 62, 109, 37, 71, 14, 26, 46, 72, 33, 89, 102, 30, 92, 90, 118, 20, 93, 126, 105, 87, 13, 1, 19, 84, 4, 60, 55, 111, 8, 85, 6, 122, 18, 121, 120, 77, 80, 29, 32, 54, 28, 78, 124, 0, 81, 50},
    {119, 7, 104, 57, 59, 70, 2, 63, 22, 68, 106, 125, 88, 116, 10, 47, 99, 74, 48, 75, 100, 12, 27, 103, 86, 101, 49, 82, 108, 44, 117, 5, 95, 43, 42, 39, 96, 97, 16, 69, 107, 91, 58, 45, 52, 79, 25, 98, 34, 21, 40, 67, 65, 41, 38, 112, 66, 61, 9, 110, 114, 31, 11, 24},
};
constexpr uint8_t kSwapFirstEdges[128] = {1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0, 1, 1, 0, 1, 0, 0, 1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 0, 1, 0, 1, 1, 0, 1, 1, 0, 0, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1, 0};
constexpr uint8_t kLaneOfCheck[38] = {25, 8, 1, 29, 3, 0, 31, 10, 4, 7, 27, 30, 22, 26, 11, 12, 5, 6, 23, 20, 13, 35, 37, 14, 17, 28, 2, 21, 36, 32, 34, 33, 15, 18, 9, 16, 19, 24};

}  // namespace msk144
