// 77-bit payload -> message text ("unpack77" of the WSJT-X 77-bit protocol).
//
// The reference links WSJT-X's Fortran for this (decode_softbits.cpp:23-55, f_interop.cpp:61-71);
// that source is absent here (empty submodule), so this is a native implementation written from the
// published protocol description (Franke/Somerville/Taylor, "The FT4 and FT8 Communication Protocols",
// QEX 2020).  CONFORMANCE IS BEST-EFFORT AND UNPINNED: the checkable contract of this project is the
// 77-bit payload; use --print-bits to see it next to the text.
//
// Covered: the message types the reference's pre-gate lets through (decode_softbits.cpp:25-30):
//   i3=0: n3=0 free text, n3=5 telemetry (n3=2 is reported as "not decodable")
//   i3=1/2 standard messages (calls, /R, /P, grid, report, RRR/RR73/73)
//   i3=4 one non-standard call + one hashed call
//   i3=5 EU VHF contest (two hashed calls, report+serial, 6-character grid)
// State: like unpack77(nrx=1), successfully unpacked calls enter 10/12/22-bit hash tables that live
// as long as the CallHashTable object, so later hashed references resolve.
#pragma once

#include <cstdint>
#include <map>
#include <string>

namespace msk144host
{

class CallHashTable
{
public:
    void save(const std::string& call);
    std::string lookup12(uint32_t h) const;  // "<CALL>" or "<...>"
    std::string lookup22(uint32_t h) const;
    static uint32_t hash(const std::string& call, int bits);  // 10, 12 or 22
    void clear();
    size_t size() const { return h22_.size(); }

private:
    std::map<uint32_t, std::string> h10_, h12_, h22_;
};

// i3/n3 pre-gate of the reference (decode_softbits.cpp:25-30): true = hand the payload to unpack77
bool message_gate(const uint8_t bits[77]);

// bits[i] in {0,1}, MSB first.  Returns false when the payload is not a decodable message.
bool unpack77(const uint8_t bits[77], CallHashTable& table, std::string& text);

// gate + unpack + right trim, exactly the reference's decode_message()
bool decode_message(const uint8_t bits[77], CallHashTable& table, std::string& text);

}  // namespace msk144host
