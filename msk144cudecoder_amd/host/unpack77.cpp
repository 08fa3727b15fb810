#include "unpack77.h"

#include <algorithm>
#include <cstdio>

namespace msk144host
{

namespace
{

constexpr uint32_t kNTokens = 2063592;
constexpr uint32_t kMax22 = 4194304;
constexpr uint32_t kMaxGrid4 = 32400;

const char* const kA1 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";   // 37: first callsign character
const char* const kA2 = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ";    // 36
const char* const kA3 = "0123456789";                              // 10
const char* const kA4 = " ABCDEFGHIJKLMNOPQRSTUVWXYZ";             // 27
const char* const kA38 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ/"; // 38: hashed / non-standard calls
const char* const kA42 = " 0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ+-./?";  // 42: free text

uint64_t field(const uint8_t* bits, int first, int count)
{
    uint64_t v = 0;
    for(int i = 0; i < count; i++) v = (v << 1) | (bits[first + i] & 1u);
    return v;
}

std::string ltrim(const std::string& s)
{
    size_t i = s.find_first_not_of(' ');
    return i == std::string::npos ? std::string() : s.substr(i);
}

std::string rtrim(const std::string& s)
{
    size_t i = s.find_last_not_of(' ');
    return i == std::string::npos ? std::string() : s.substr(0, i + 1);
}

std::string trim(const std::string& s)
{
    return rtrim(ltrim(s));
}

// 28-bit field -> token, hashed call or standard call
bool unpack28(uint32_t n28, const CallHashTable& table, std::string& call)
{
    if(n28 < kNTokens)
    {
        if(n28 == 0) call = "DE";
        else if(n28 == 1) call = "QRZ";
        else if(n28 == 2) call = "CQ";
        else if(n28 <= 1002)
        {
            char buf[16];
            std::snprintf(buf, sizeof(buf), "CQ %03u", n28 - 3);
            call = buf;
        }
        else if(n28 <= 532443)
        {
            uint32_t n = n28 - 1003;
            char c[5] = {0, 0, 0, 0, 0};
            for(int i = 3; i >= 0; i--)
            {
                c[i] = kA4[n % 27];
                n /= 27;
            }
            call = "CQ " + ltrim(std::string(c, 4));
        }
        else
        {
            return false;
        }
        return true;
    }
    n28 -= kNTokens;
    if(n28 < kMax22)
    {
        call = table.lookup22(n28);
        return true;
    }
    uint32_t n = n28 - kMax22;
    char c[7];
    c[5] = kA4[n % 27];
    n /= 27;
    c[4] = kA4[n % 27];
    n /= 27;
    c[3] = kA4[n % 27];
    n /= 27;
    c[2] = kA3[n % 10];
    n /= 10;
    c[1] = kA2[n % 36];
    n /= 36;
    if(n >= 37) return false;
    c[0] = kA1[n];
    c[6] = 0;
    std::string s = trim(std::string(c, 6));
    if(s.find(' ') != std::string::npos) return false;  // embedded blank: not a callsign
    // Swaziland and Guinea prefixes travel in a shortened form
    if(s.size() >= 4 && s.compare(0, 3, "3D0") == 0) s = "3DA0" + s.substr(3);
    else if(s.size() >= 2 && s[0] == 'Q' && s[1] >= 'A' && s[1] <= 'Z') s = "3X" + s.substr(1);
    call = s;
    return !call.empty();
}

bool to_grid4(uint32_t n, std::string& g)
{
    const uint32_t j1 = n / (18 * 10 * 10);
    if(j1 > 17) return false;
    n -= j1 * 18 * 10 * 10;
    const uint32_t j2 = n / 100;
    if(j2 > 17) return false;
    n -= j2 * 100;
    const uint32_t j3 = n / 10;
    const uint32_t j4 = n % 10;
    g.clear();
    g += static_cast<char>('A' + j1);
    g += static_cast<char>('A' + j2);
    g += static_cast<char>('0' + j3);
    g += static_cast<char>('0' + j4);
    return true;
}

bool to_grid6(uint32_t n, std::string& g)
{
    const uint32_t d1 = 18 * 10 * 10 * 24 * 24, d2 = 10 * 10 * 24 * 24, d3 = 10 * 24 * 24, d4 = 24 * 24;
    const uint32_t j1 = n / d1;
    if(j1 > 17) return false;
    n -= j1 * d1;
    const uint32_t j2 = n / d2;
    if(j2 > 17) return false;
    n -= j2 * d2;
    const uint32_t j3 = n / d3;
    n -= j3 * d3;
    const uint32_t j4 = n / d4;
    n -= j4 * d4;
    const uint32_t j5 = n / 24;
    const uint32_t j6 = n % 24;
    g.clear();
    g += static_cast<char>('A' + j1);
    g += static_cast<char>('A' + j2);
    g += static_cast<char>('0' + j3);
    g += static_cast<char>('0' + j4);
    g += static_cast<char>('A' + j5);
    g += static_cast<char>('A' + j6);
    return true;
}

bool is_plain_call(const std::string& c)
{
    return c.size() >= 3 && c[0] != '<' && c.compare(0, 3, "CQ ") != 0;
}

// 71 bits -> 13 characters base 42 (most significant first)
std::string unpack_text71(const uint8_t* bits)
{
    // 71-bit integer in two limbs, repeated division by 42
    uint8_t work[71];
    for(int i = 0; i < 71; i++) work[i] = bits[i] & 1u;
    char out[13];
    for(int pos = 12; pos >= 0; pos--)
    {
        // long division of the bit string by 42
        uint32_t rem = 0;
        for(int i = 0; i < 71; i++)
        {
            rem = rem * 2 + work[i];
            if(rem >= 42)
            {
                work[i] = 1;
                rem -= 42;
            }
            else
            {
                work[i] = 0;
            }
        }
        out[pos] = kA42[rem];
    }
    return std::string(out, 13);
}

}  // namespace

uint32_t CallHashTable::hash(const std::string& call, int bits)
{
    std::string c = call;
    c.resize(11, ' ');
    uint64_t n8 = 0;
    for(int i = 0; i < 11; i++)
    {
        const char ch = c[i];
        int j = 0;
        for(int k = 0; k < 38; k++)
            if(kA38[k] == ch)
            {
                j = k;
                break;
            }
        n8 = 38 * n8 + static_cast<uint64_t>(j);
    }
    const uint64_t prod = 47055833459ull * n8;  // 64-bit wrap-around
    return static_cast<uint32_t>(prod >> (64 - bits));
}

void CallHashTable::save(const std::string& call)
{
    const std::string c = trim(call);
    if(c.size() < 3 || c[0] == '<') return;
    h10_[hash(c, 10)] = c;
    h12_[hash(c, 12)] = c;
    h22_[hash(c, 22)] = c;
}

std::string CallHashTable::lookup12(uint32_t h) const
{
    auto it = h12_.find(h);
    return it == h12_.end() ? "<...>" : "<" + it->second + ">";
}

std::string CallHashTable::lookup22(uint32_t h) const
{
    auto it = h22_.find(h);
    return it == h22_.end() ? "<...>" : "<" + it->second + ">";
}

void CallHashTable::clear()
{
    h10_.clear();
    h12_.clear();
    h22_.clear();
}

bool message_gate(const uint8_t bits[77])
{
    const int n3 = static_cast<int>(field(bits, 71, 3));
    const int i3 = static_cast<int>(field(bits, 74, 3));
    if((i3 == 0 && (n3 == 1 || n3 == 3 || n3 == 4 || n3 > 5)) || i3 == 3 || i3 > 5) return false;
    return true;
}

bool unpack77(const uint8_t bits[77], CallHashTable& table, std::string& text)
{
    text.clear();
    const int n3 = static_cast<int>(field(bits, 71, 3));
    const int i3 = static_cast<int>(field(bits, 74, 3));

    if(i3 == 0 && n3 == 0)
    {
        text = ltrim(unpack_text71(bits));
        text = rtrim(text);
        return !text.empty();
    }
    if(i3 == 0 && n3 == 5)
    {
        char buf[32];
        std::snprintf(buf, sizeof(buf), "%06X%06X%06X", static_cast<unsigned>(field(bits, 0, 23)), static_cast<unsigned>(field(bits, 23, 24)),
                      static_cast<unsigned>(field(bits, 47, 24)));
        std::string s(buf);
        size_t i = s.find_first_not_of('0');
        text = (i == std::string::npos) ? std::string() : s.substr(i);
        return !text.empty();
    }
    if(i3 == 1 || i3 == 2)
    {
        const uint32_t n28a = static_cast<uint32_t>(field(bits, 0, 28));
        const int ipa = bits[28] & 1;
        const uint32_t n28b = static_cast<uint32_t>(field(bits, 29, 28));
        const int ipb = bits[57] & 1;
        const int ir = bits[58] & 1;
        const uint32_t igrid4 = static_cast<uint32_t>(field(bits, 59, 15));
        std::string c1, c2;
        if(!unpack28(n28a, table, c1) || !unpack28(n28b, table, c2)) return false;
        const char* suffix = (i3 == 1) ? "/R" : "/P";
        if(is_plain_call(c1))
        {
            table.save(c1);
            if(ipa) c1 += suffix;
        }
        if(is_plain_call(c2))
        {
            table.save(c2);
            if(ipb) c2 += suffix;
        }
        if(igrid4 <= kMaxGrid4)
        {
            std::string g;
            if(!to_grid4(igrid4, g)) return false;
            text = c1 + " " + c2 + (ir ? " R " : " ") + g;
            if(text.compare(0, 3, "CQ ") == 0 && ir) return false;
        }
        else
        {
            const uint32_t irpt = igrid4 - kMaxGrid4;
            text = c1 + " " + c2;
            if(irpt == 2) text += " RRR";
            else if(irpt == 3) text += " RR73";
            else if(irpt == 4) text += " 73";
            else if(irpt >= 5)
            {
                int isnr = static_cast<int>(irpt) - 35;
                if(isnr > 50) isnr -= 101;
                char buf[16];
                std::snprintf(buf, sizeof(buf), "%c%02d", isnr < 0 ? '-' : '+', isnr < 0 ? -isnr : isnr);
                text += ir ? " R" : " ";
                text += buf;
            }
            if(text.compare(0, 3, "CQ ") == 0 && irpt >= 2) return false;
        }
        if(text.compare(0, 4, "CQ <") == 0) return false;
        return true;
    }
    if(i3 == 4)
    {
        const uint32_t n12 = static_cast<uint32_t>(field(bits, 0, 12));
        uint64_t n58 = field(bits, 12, 58);
        const int iflip = bits[70] & 1;
        const int nrpt = static_cast<int>(field(bits, 71, 2));
        const int icq = bits[73] & 1;
        char c[11];
        for(int i = 10; i >= 0; i--)
        {
            c[i] = kA38[n58 % 38];
            n58 /= 38;
        }
        const std::string c11 = trim(std::string(c, 11));
        if(c11.empty()) return false;
        const std::string hashed = table.lookup12(n12);
        table.save(c11);
        std::string c1 = iflip ? c11 : hashed;
        std::string c2 = iflip ? hashed : c11;
        if(icq)
        {
            text = "CQ " + c11;
        }
        else
        {
            text = c1 + " " + c2;
            if(nrpt == 1) text += " RRR";
            else if(nrpt == 2) text += " RR73";
            else if(nrpt == 3) text += " 73";
        }
        return true;
    }
    if(i3 == 5)
    {
        const uint32_t n12 = static_cast<uint32_t>(field(bits, 0, 12));
        const uint32_t n22 = static_cast<uint32_t>(field(bits, 12, 22));
        const int ir = bits[34] & 1;
        const int irpt = static_cast<int>(field(bits, 35, 3));
        const int iserial = static_cast<int>(field(bits, 38, 11));
        const uint32_t igrid6 = static_cast<uint32_t>(field(bits, 49, 25));
        if(igrid6 > 18662399u) return false;
        std::string g;
        if(!to_grid6(igrid6, g)) return false;
        char exch[16];
        std::snprintf(exch, sizeof(exch), "%2d%04d", 52 + irpt, iserial);
        text = table.lookup12(n12) + " " + table.lookup22(n22) + (ir ? " R " : " ") + exch + " " + g;
        return true;
    }
    return false;  // 0.2 and everything the gate should have stopped
}

bool decode_message(const uint8_t bits[77], CallHashTable& table, std::string& text)
{
    if(!message_gate(bits)) return false;
    if(!unpack77(bits, table, text)) return false;
    text = rtrim(text);  // the reference trims the 37-character Fortran buffer on the right
    if(text.size() > 37) text.resize(37);
    return true;
}

}  // namespace msk144host
