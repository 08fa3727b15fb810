/*
 * CPU ORACLE (test infrastructure) - see msk144_oracle.h for the rules that govern this file.
 * float32 throughout, reference expression order, no FMA contraction (-ffp-contract=off).
 */
#include "msk144_oracle.h"

#include "../msk144cudecoder_amd/csrc/msk144_protocol.h"

#include <cmath>
#include <cstring>
#include <vector>

using namespace msk144;

static_assert(sizeof(orc_item) == kReferenceResultItemBytes, "orc_item must mirror the reference ResultItem");

namespace
{

// ---------------------------------------------------------------------------------------------
// complex helpers: smath_complex.h:16-87
// ---------------------------------------------------------------------------------------------
struct C
{
    float re, im;
};
inline C cadd(C x, C y) { return {x.re + y.re, x.im + y.im}; }                                           // :49-52
inline C cmul(C x, C y) { return {x.re * y.re - x.im * y.im, x.re * y.im + x.im * y.re}; }               // :54-57
inline C smul(float m, C z) { return {m * z.re, m * z.im}; }                                             // :59-62
inline C cconj(C z) { return {z.re, -z.im}; }                                                            // :64-67

// ---- contraction bracket (tests/test_oracle_fma_bracket.py) ----
// The reference's CUDA build contracts a*b+c into FMAs in DEVICE code (nvcc's default -fmad=true; CMakeLists.txt:130-132 adds only
// an optional fast-math switch); its host code is plain x86-64 without FMA.  The parity build of this oracle contracts nothing
// (-ffp-contract=off), so "bit-exact with the oracle" means bit-exact with a build that never ran on NVIDIA hardware.  Two more
// builds of this same file bracket what the real binary may compute (oracle/Makefile, target `fma`):
//   * libmsk144_oracle_contract.so: -ffp-contract=fast -mfma - gcc fuses every single-use product into the add or subtract that
//     consumes it, everywhere (host-side restatements included: an over-approximation);
//   * libmsk144_oracle_fmaf.so: -DORC_FORCE_FMA - the device-side complex products and the three accumulate hot spots
//     (analytic2.cuh:163-219 FIR, scan_kernel.cuh:106-122 correlation, softbits_kernel.cuh:157-180 matched filter) written as the
//     FULLY fused chains an aggressively fusing backend emits (NVPTX enables aggressive FMA fusion: s + x*y over complex numbers
//     becomes fma(x.re, y.re, fma(-x.im, y.im, s.re))).
// The *_dev helpers mark the device-side sites; host-side sites (SNR tracker, FFT stand-in) keep cmul/smul.
#ifdef ORC_FORCE_FMA
inline C cmul_dev(C x, C y) { return {fmaf(x.re, y.re, -(x.im * y.im)), fmaf(x.re, y.im, x.im * y.re)}; }
inline C cmac_dev(C s, C x, C y) { return {fmaf(x.re, y.re, fmaf(-x.im, y.im, s.re)), fmaf(x.re, y.im, fmaf(x.im, y.re, s.im))}; }
inline C smac_dev(C s, float m, C z) { return {fmaf(m, z.re, s.re), fmaf(m, z.im, s.im)}; }
inline float mac_dev(float s, float a, float b) { return fmaf(a, b, s); }
#else
inline C cmul_dev(C x, C y) { return cmul(x, y); }
inline C cmac_dev(C s, C x, C y) { return cadd(s, cmul(x, y)); }
inline C smac_dev(C s, float m, C z) { return cadd(s, smul(m, z)); }
inline float mac_dev(float s, float a, float b) { return s + a * b; }
#endif
#ifndef ORC_VARIANT_NAME
#define ORC_VARIANT_NAME "parity"
#endif
const char* const kBuildVariant = ORC_VARIANT_NAME;  // named by the Makefile target that set the flags

// ---- math-library bracket (tests/test_oracle_fma_bracket.py, third build "cuda-libm") ----
// The reference's device code calls CUDA's sincosf, atan2f, hypotf and tanhf (smath_complex.h:79-85 and abs(), softbits_kernel.cuh:137,
// ldpc_kernel.cuh:236); this oracle calls glibc's, which are correctly rounded or within 1 ulp.  CUDA documents larger errors (CUDA C
// Programming Guide, "Mathematical Functions": sincosf 2 ulp, atan2f 3 ulp, hypotf 3 ulp, tanhf 2 ulp; sqrtf and division are correctly
// rounded by default) and its results are not available here.  With -DORC_PERTURB_ULP the four calls return glibc's value moved by a
// deterministic pseudo-random number of ulps within those bounds (a hash of the argument bits: the same argument always gives the same
// result, as a real library would) - not CUDA's values, but values as far from the true ones as CUDA's may be.
#ifdef ORC_PERTURB_ULP
inline uint32_t hash_bits(uint32_t a, uint32_t b)
{
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u + (a << 6) + (a >> 2));
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}
inline float nudge(float v, int max_ulp, uint32_t h)
{
    if(!std::isfinite(v) || v == 0.0f) return v;
    const int k = static_cast<int>(h % static_cast<uint32_t>(2 * max_ulp + 1)) - max_ulp;  // -max_ulp .. +max_ulp
    uint32_t u;
    std::memcpy(&u, &v, 4);
    u = static_cast<uint32_t>(static_cast<int32_t>(u) + (v > 0.0f ? k : -k));  // k ulps away from zero for k > 0, either sign of v
    float r;
    std::memcpy(&r, &u, 4);
    return std::isfinite(r) ? r : v;
}
inline uint32_t fbits(float x)
{
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return u;
}
inline void lib_sincosf(float x, float* s, float* c)
{
    sincosf(x, s, c);
    *s = nudge(*s, 2, hash_bits(fbits(x), 1u));
    *c = nudge(*c, 2, hash_bits(fbits(x), 2u));
}
inline float lib_atan2f(float y, float x) { return nudge(atan2f(y, x), 3, hash_bits(fbits(y), fbits(x))); }
inline float lib_hypotf(float x, float y) { return nudge(hypotf(x, y), 3, hash_bits(fbits(x), fbits(y) ^ 0x55u)); }
inline float lib_tanhf(float x) { return nudge(tanhf(x), 2, hash_bits(fbits(x), 3u)); }
#else
inline void lib_sincosf(float x, float* s, float* c) { sincosf(x, s, c); }
inline float lib_atan2f(float y, float x) { return atan2f(y, x); }
inline float lib_hypotf(float x, float y) { return hypotf(x, y); }
inline float lib_tanhf(float x) { return tanhf(x); }
#endif
inline C from_phi(float phi)                                                                             // :79-85
{
    float s, c;
    lib_sincosf(phi, &s, &c);
    return {c, s};
}

const float kPiF = 3.14159265358979323846f;  // CUDART_PI_F

// msk_context.cuh:137-145
void fill_pp(float* pp)
{
    for(int i = 0; i < 12; i++)
    {
        float angle = i * kPiF / 12.0f;
        pp[i] = sinf(angle);
    }
}

// msk_context.cuh:176-196
void fill_cb42(C* cb)
{
    float pp[12];
    fill_pp(pp);
    const int* s8 = kSync8Pm;
    float cbi[42], cbq[42];
    for(int i = 0; i < 6; i++) cbq[0 + i] = pp[6 + i] * s8[0];
    for(int i = 0; i < 12; i++) cbq[6 + i] = pp[i] * s8[2];
    for(int i = 0; i < 12; i++) cbq[18 + i] = pp[i] * s8[4];
    for(int i = 0; i < 12; i++) cbq[30 + i] = pp[i] * s8[6];
    for(int i = 0; i < 12; i++) cbi[0 + i] = pp[i] * s8[1];
    for(int i = 0; i < 12; i++) cbi[12 + i] = pp[i] * s8[3];
    for(int i = 0; i < 12; i++) cbi[24 + i] = pp[i] * s8[5];
    for(int i = 0; i < 6; i++) cbi[36 + i] = pp[i] * s8[7];
    for(int i = 0; i < 42; i++) cb[i] = {cbi[i], cbq[i]};
}

// scan_kernel.cuh:45-69 / softbits_kernel.cuh:27-52: mix the whole window down by f.
void mix_window(const orc_ctx* ctx, int b, const orc_complex* cdat, C* cdat2, float* f0_out)
{
    const float f0 = -1 * orc_frequency(ctx, b);
    const float twopi = 2.0f * kPiF;
    for(int n = 0; n < kWindowSamples; n++)
    {
        const float phi = static_cast<float>(n) * twopi * f0 / kSampleRate;
        C w = from_phi(phi);
        cdat2[n] = cmul_dev(w, C{cdat[n].re, cdat[n].im});
    }
    *f0_out = f0;
}

// 32-lane __shfl_down emulation: lanes beyond the warp keep their own value.
template<typename T>
inline T shfl_down32(const T* v, int lane, int delta)
{
    return (lane + delta < 32) ? v[lane + delta] : v[lane];
}

// sum_reduction.cuh:14-44 - value of lane 0 of a 32-lane shuffle tree
template<typename T>
T warp_tree_sum(T* v /*[32], clobbered*/)
{
    for(int d = 1; d <= 16; d *= 2)
    {
        T nv[32];
        for(int l = 0; l < 32; l++) nv[l] = v[l] + shfl_down32(v, l, d);
        std::memcpy(v, nv, sizeof(nv));
    }
    return v[0];
}

// sum_reduction_two_cycles over `nthreads` threads (multiple of 32, <= 1024)
template<typename T>
T block_sum_two_cycles(const T* vals, int nthreads)
{
    T second[32];
    for(int i = 0; i < 32; i++) second[i] = 0;
    const int nwarps = nthreads / 32;
    for(int w = 0; w < nwarps; w++)
    {
        T lane[32];
        for(int l = 0; l < 32; l++) lane[l] = vals[w * 32 + l];
        second[w] = warp_tree_sum(lane);
    }
    return warp_tree_sum(second);
}

// ldpc_context.cuh:185-213
void gen_crc13_table(uint16_t* table)
{
    const int LengthCRC = 13;
    const uint16_t polynomial = kCrc13Poly;
    const uint16_t high_bit_mask = (1 << (LengthCRC - 1));
    for(int i = 0; i < 256; i++)
    {
        uint16_t dividend = i;
        uint16_t remainder = 0;
        for(int bit = 0; bit < 8; bit++)
        {
            if(dividend & 0x80) remainder ^= high_bit_mask;
            bool const quotient = remainder & high_bit_mask;
            remainder <<= 1;
            if(quotient) remainder ^= polynomial;
            dividend <<= 1;
        }
        table[i] = remainder;
    }
}

const uint16_t* crc_table()
{
    static uint16_t table[256];
    static bool init = false;
    if(!init)
    {
        gen_crc13_table(table);
        init = true;
    }
    return table;
}

// bit-major edge map rebuilt from the check-major table: mp[n][k] = (slot, check), k ascending in
// check - the layout of ldpc_context.cuh:10-139.
struct EdgeMap
{
    int8_t slot[128][3];
    int8_t check[128][3];
    bool full_row[38];
    EdgeMap()
    {
        int cnt[128] = {0};
        for(int c = 0; c < kChecks; c++)
        {
            full_row[c] = kCheckBits[c][10] >= 0;
            for(int j = 0; j < kMaxCheckDegree; j++)
            {
                int n = kCheckBits[c][j];
                if(n < 0) continue;
                slot[n][cnt[n]] = static_cast<int8_t>(j);
                check[n][cnt[n]] = static_cast<int8_t>(c);
                cnt[n]++;
            }
        }
    }
};
const EdgeMap& edge_map()
{
    static EdgeMap m;
    return m;
}

// ldpc_kernel.cuh:65-93
float platanh(float x)
{
    float isign = 1.0f;
    float z = x;
    if(x < 0.0)
    {
        isign = -1.0f;
        z = fabsf(x);
    }
    if(z <= 0.664f) return x / 0.83f;
    else if(z <= 0.9217f) return isign * (z - 0.4064f) / 0.322f;
    else if(z <= 0.9951f) return isign * (z - 0.8378f) / 0.0524f;
    else if(z <= 0.9998f) return isign * (z - 0.9914f) / 0.0012f;
    return isign * 7.0f;
}

// xb of one scanned position: scan_kernel.cuh:91-124
inline float scan_position(const C* cdat2, const C* cb42, const uint8_t* mask, unsigned base)
{
    C s = {0.0f, 0.0f};
    for(unsigned idx = 0; idx < (unsigned)kSyncTaps; idx++)
    {
        const unsigned long_idx = base + idx;
        C y = {0.0f, 0.0f};
        for(unsigned m = 0; m < (unsigned)kPatternBits; m++)
        {
            if(mask[m])
            {
                unsigned idx_a = long_idx + kFrameSamples * m;
                if(idx_a >= (unsigned)kWindowSamples) idx_a -= kWindowSamples;
                y = cadd(y, cdat2[idx_a]);
                unsigned idx_b = long_idx + kFrameSamples * m + kSecondSyncSample;
                if(idx_b >= (unsigned)kWindowSamples) idx_b -= kWindowSamples;
                y = cadd(y, cdat2[idx_b]);
            }
        }
        s = cmac_dev(s, cconj(y), cb42[idx]);
    }
    return lib_hypotf(s.re, s.im);
}

struct Slot
{
    unsigned pos;
    float xb;
};

// One slice (256 positions): arg-max with the reference's shuffle-tree tie rules
// (scan_kernel.cuh:140-268), then the 8-slot replacement rule (:276-353).
void slice_update(const float* xb256, unsigned slice_base, Slot* slots)
{
    // intra-warp (32 lanes) trees
    float wxb[8];
    unsigned wpos[8];
    for(int w = 0; w < 8; w++)
    {
        float xb[32];
        unsigned pos[32];
        for(int l = 0; l < 32; l++)
        {
            xb[l] = xb256[w * 32 + l];
            pos[l] = slice_base + w * 32 + l;
        }
        for(int d = 1; d <= 16; d *= 2)
        {
            float nxb[32];
            unsigned npos[32];
            for(int l = 0; l < 32; l++)
            {
                float xo = shfl_down32(xb, l, d);
                unsigned po = shfl_down32(pos, l, d);
                nxb[l] = xb[l];
                npos[l] = pos[l];
                if(xo > xb[l])
                {
                    nxb[l] = xo;
                    npos[l] = po;
                }
            }
            std::memcpy(xb, nxb, sizeof(xb));
            std::memcpy(pos, npos, sizeof(pos));
        }
        wxb[w] = xb[0];
        wpos[w] = pos[0];
    }
    // inter-warp: thread t loads entry t%8, 3 shuffle stages; thread 0 decides
    float xb[32];
    unsigned pos[32];
    for(int l = 0; l < 32; l++)
    {
        xb[l] = wxb[l % 8];
        pos[l] = wpos[l % 8];
    }
    for(int d = 1; d <= 4; d *= 2)
    {
        float nxb[32];
        unsigned npos[32];
        for(int l = 0; l < 32; l++)
        {
            float xo = shfl_down32(xb, l, d);
            unsigned po = shfl_down32(pos, l, d);
            nxb[l] = xb[l];
            npos[l] = pos[l];
            if(xo > xb[l])
            {
                nxb[l] = xo;
                npos[l] = po;
            }
        }
        std::memcpy(xb, nxb, sizeof(xb));
        std::memcpy(pos, npos, sizeof(pos));
    }
    const float xb_best = xb[0];
    const unsigned pos_best = pos[0];

    // arg-min over the stored slots, 4 stages (the 4th is a no-op for 8 slots)
    float sxb[32];
    unsigned sidx[32];
    for(int l = 0; l < 32; l++)
    {
        sidx[l] = l % kSlotsPerPattern;
        sxb[l] = slots[sidx[l]].xb;
    }
    for(int d = 1; d <= 8; d *= 2)
    {
        float nxb[32];
        unsigned nidx[32];
        for(int l = 0; l < 32; l++)
        {
            float xo = shfl_down32(sxb, l, d);
            unsigned io = shfl_down32(sidx, l, d);
            nxb[l] = sxb[l];
            nidx[l] = sidx[l];
            if(xo < sxb[l])
            {
                nxb[l] = xo;
                nidx[l] = io;
            }
        }
        std::memcpy(sxb, nxb, sizeof(sxb));
        std::memcpy(sidx, nidx, sizeof(sidx));
    }
    Slot& worst = slots[sidx[0]];
    if(xb_best > worst.xb)
    {
        worst.xb = xb_best;
        worst.pos = pos_best;
    }
}

orc_item& item_at(const orc_ctx* ctx, orc_item* items, unsigned b, unsigned p, unsigned c)
{
    return items[b * ctx->scan_depth * kSlotsPerPattern + p * kSlotsPerPattern + c];  // result_keeper.cuh:85-91
}

// softbits of one candidate given the mixed window: softbits_kernel.cuh:56-247
void softbits_core(const C* cdat2, const C* cb42, const float* pp, const uint8_t* mask, unsigned pos, float* soft144, float* llr128, int* nbadsync)
{
    // fold (:59-82).  Reads wrap with a true modulo (the reference reads out of bounds for
    // mask 111111 and pos >= 5185 - SURVEY.md A.9; documented deviation).
    C c3[kFrameSamples];
    for(unsigned n = 0; n < (unsigned)kFrameSamples; n++)
    {
        C s = {0.0f, 0.0f};
        for(unsigned m = 0; m < (unsigned)kPatternBits; m++)
        {
            if(mask[m])
            {
                unsigned i = pos + n + kFrameSamples * m;
                if(i >= (unsigned)kWindowSamples) i -= kWindowSamples;
                if(i >= (unsigned)kWindowSamples) i -= kWindowSamples;
                s = cadd(s, cdat2[i]);
            }
        }
        c3[n] = s;
    }

    // carrier phase from the 84 sync samples (:88-128)
    C r[42];
    for(unsigned t = 0; t < 42; t++) r[t] = cmul_dev(c3[t], cconj(cb42[t]));
    for(unsigned t = 42; t < 84; t++) r[t - 42] = cmac_dev(r[t - 42], c3[kSecondSyncSample + (t % 42)], cconj(cb42[t % 42]));
    for(unsigned t = 0; t < 10; t++) r[t] = cadd(r[t], r[32 + t]);
    for(int size = 16; size > 0; size /= 2)
        for(int t = 0; t < size; t++) r[t] = cadd(r[t], r[t + size]);

    const C s = r[0];
    const float phase0 = lib_atan2f(s.im, s.re);  // :137
    const C w = from_phi(phase0);
    const C cfac = cconj(w);
    for(unsigned n = 0; n < (unsigned)kFrameSamples; n++) c3[n] = cmul_dev(c3[n], cfac);  // :146-153

    // matched filter (:157-180)
    float softbits[kSoftBits];
    for(int t = 0; t < kSoftBits; t++)
    {
        const int pos_iq = t % 72;
        const int iq_selection = t / 72;
        const int base1 = iq_selection ? 0 : (kFrameSamples - 6);
        const int d = 12 * pos_iq;
        float sb = 0.0f;
        for(int idx = 0; idx < 12; idx++)
        {
            const int k = (base1 + d + idx) % kFrameSamples;
            const float v = iq_selection ? c3[k].re : c3[k].im;
            sb = mac_dev(sb, v, pp[idx]);
        }
        softbits[pos_iq * 2 + iq_selection] = sb;
    }

    // normalisation (:186-211); the block has 160 threads, threads >= 144 contribute 0
    float loc_sav[160], loc_s2av[160];
    for(int t = 0; t < 160; t++)
    {
        loc_sav[t] = 0.0f;
        loc_s2av[t] = 0.0f;
        if(t < 144)
        {
            loc_sav[t] = softbits[t];
            loc_s2av[t] = loc_sav[t] * loc_sav[t];
        }
    }
    const float sum_sav = block_sum_two_cycles(loc_sav, 160);
    const float sum_s2av = block_sum_two_cycles(loc_s2av, 160);
    const float sav = sum_sav / 144.0;    // double divide, stored to float (:196)
    const float s2av = sum_s2av / 144.0;  // (:197)
    const float ssig = sqrtf(s2av - sav * sav);
    const float sigma = 0.60f;
    const float scale = 2.0f / (ssig * sigma * sigma);

    for(int t = 0; t < 48; t++) llr128[t] = scale * softbits[8 + t];
    for(int t = 0; t < 80; t++) llr128[48 + t] = scale * softbits[8 + 48 + 8 + t];
    if(soft144) std::memcpy(soft144, softbits, sizeof(softbits));

    // sync-word disagreement count (:214-241); threads 0..7 -> second sync word, 8..15 -> first
    int mm[2];
    for(int g = 0; g < 2; g++)
    {
        const int base = g ? 0 : kSecondSyncBit;
        int v = 0;
        // shuffle tree over 8 lanes: ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)); integer, order irrelevant
        for(int bit = 0; bit < 8; bit++)
        {
            const float sb = softbits[base + bit];
            const int hardbit = (sb < 0.0f) ? -1 : 1;
            v += hardbit * kSync8Pm[bit];
        }
        mm[g] = (8 - v) / 2;
    }
    *nbadsync = mm[0] + mm[1];
}

// ldpc_kernel.cuh:100-249 for one LLR vector
bool ldpc_core(const float* softbits, char* message77, int* iters_out, int* nhard_out)
{
    const EdgeMap& mp = edge_map();
    float toc[11][38];
    float tov[3][128];
    float zn[128];
    char cw[128];
    char chk_cw_toc[11][38];
    std::memset(toc, 0, sizeof(toc));
    std::memset(chk_cw_toc, 0, sizeof(chk_cw_toc));
    for(int k = 0; k < 3; k++)
        for(int n = 0; n < 128; n++) tov[k][n] = 0.0f;

    for(unsigned iter = 0; iter < (unsigned)kLdpcIterations; iter++)
    {
        for(int n = 0; n < 128; n++)
        {
            float sum = 0.0f;
            for(int k = 0; k < 3; k++) sum += tov[k][n];
            zn[n] = softbits[n] + sum;
            cw[n] = (zn[n] > 0.0f) ? 1 : 0;
        }
        for(int n = 0; n < 128; n++)
            for(int k = 0; k < 3; k++) chk_cw_toc[mp.slot[n][k]][mp.check[n][k]] = cw[n];

        int ncheck = 0;
        for(int c = 0; c < 38; c++)
        {
            int sum = 0;
            for(int i = 0; i < 11; i++) sum += chk_cw_toc[i][c];
            ncheck += sum % 2;
        }

        bool is_crc_valid = false;
        if(ncheck == 0) is_crc_valid = orc_check_crc_bits(cw) != 0;

        int num_hard_errors = 0;
        for(int n = 0; n < 128; n++)
        {
            int is_bit_bad = ((cw[n] == 1 && softbits[n] > 0.0f) || (cw[n] == 0 && softbits[n] <= 0.0f)) ? 0 : 1;
            num_hard_errors += is_bit_bad;
        }
        const bool message_found = is_crc_valid && num_hard_errors < kMaxHardErrors;
        if(message_found)
        {
            for(int i = 0; i < kMessageBits; i++) message77[i] = cw[i];
            *iters_out = static_cast<int>(iter);
            *nhard_out = num_hard_errors;
            return true;
        }

        for(int n = 0; n < 128; n++)
            for(int k = 0; k < 3; k++) toc[mp.slot[n][k]][mp.check[n][k]] = zn[n] - tov[k][n];

        for(int n = 0; n < 128; n++)
        {
            for(int k = 0; k < 3; k++)
            {
                const int column = mp.check[n][k];
                const int row_to_exclude = mp.slot[n][k];
                float product = 1.0f;
                for(int j = 0; j < 11; j++)
                {
                    if((j < 10 || mp.full_row[column]) && j != row_to_exclude) product *= lib_tanhf(-0.5f * toc[j][column]);
                }
                tov[k][n] = 2.0f * platanh(-product);
            }
        }
    }
    return false;
}

// radix-2 in-place FFT, float data, twiddles rounded from double. sign=-1 forward, +1 inverse.
void fft_radix2(std::vector<C>& a, int sign)
{
    const int n = static_cast<int>(a.size());
    for(int i = 1, j = 0; i < n; i++)
    {
        int bit = n >> 1;
        for(; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if(i < j) std::swap(a[i], a[j]);
    }
    for(int len = 2; len <= n; len <<= 1)
    {
        const int half = len / 2;
        std::vector<C> tw(half);
        for(int k = 0; k < half; k++)
        {
            const double ang = sign * 2.0 * M_PI * k / len;
            tw[k] = {static_cast<float>(cos(ang)), static_cast<float>(sin(ang))};
        }
        for(int i = 0; i < n; i += len)
        {
            for(int k = 0; k < half; k++)
            {
                C u = a[i + k];
                C v = cmul(a[i + k + half], tw[k]);
                a[i + k] = cadd(u, v);
                a[i + k + half] = {u.re - v.re, u.im - v.im};
            }
        }
    }
}

}  // namespace

// =============================================================================================
extern "C" {

const char* orc_build_variant(void)
{
    return kBuildVariant;
}

int orc_sizeof_item(void)
{
    return static_cast<int>(sizeof(orc_item));
}

// msk_context.cuh:23-38, 95-113
void orc_ctx_init(orc_ctx* ctx, float center_freq, float search_width, float search_step, int scan_depth, int nbadsync_threshold)
{
    ctx->scan_depth = clamp_scan_depth(scan_depth);
    ctx->nbadsync_threshold = nbadsync_threshold;
    ctx->center_freq = center_freq;
    ctx->step = search_step;
    const int half_len = grid_half_len(search_width, search_step);
    ctx->num_blocks = half_len * 2 + 1;
    ctx->if1 = -1 * half_len * search_step;
    ctx->total_items = ctx->num_blocks * ctx->scan_depth * kSlotsPerPattern;
    ctx->num_threads = 1;
}

void orc_set_threads(orc_ctx* ctx, int n)
{
    ctx->num_threads = n < 1 ? 1 : n;
}

// msk_context.cuh:135
float orc_frequency(const orc_ctx* ctx, int block_idx)
{
    return ctx->center_freq + ctx->if1 + static_cast<int>(block_idx) * ctx->step;
}

void orc_get_cb42(orc_complex* cb42)
{
    C cb[42];
    fill_cb42(cb);
    for(int i = 0; i < 42; i++) cb42[i] = {cb[i].re, cb[i].im};
}

void orc_get_pp12(float* pp)
{
    fill_pp(pp);
}

// main.cu:301-307,323
void orc_normalize_audio(const int16_t* win, orc_complex* out)
{
    float acc = 0.0f;
    for(int i = 0; i < kWindowSamples; i++)
    {
        float b = static_cast<float>(win[i]);
        acc = static_cast<float>(static_cast<double>(acc + b * b));  // lambda returns double, accumulator is float
    }
    const float sum_rms2 = acc;
    const float rms = sqrtf(sum_rms2 / kWindowSamples);
    const float fac = 1.0f / rms;
    for(int i = 0; i < kWindowSamples; i++) out[i] = {fac * win[i], 0.0f};
}

// main.cu:365-371
void orc_convert_iq(const int8_t* win, orc_complex* out)
{
    for(int idx = 0; idx < kWindowSamples; idx++)
    {
        const float i = static_cast<float>(win[idx * 2 + 0]);
        const float q = static_cast<float>(win[idx * 2 + 1]);
        const float divider = 128.0f;
        out[idx] = {i / divider, q / divider};
    }
}

// analytic2.cuh:235-281.  The two FIR passes run "in place" in 32-sample slices with all reads of
// a slice before its writes; reads only reach samples the pass has not written yet (ahead of the
// ascending pass, behind the descending one), so each pass sees the previous stage's values.
void orc_analytic2(const orc_complex* in, orc_complex* out, int with_shift)
{
    static const C w_left[8] = {{kSin45, -kSin45}, {0.0f, -1.0f}, {-kSin45, -kSin45}, {-1.0f, 0.0f},
                                {-kSin45, kSin45}, {0.0f, 1.0f},  {kSin45, kSin45},   {1.0f, 0.0f}};  // :15-41
    static const C w_right[8] = {{1.0f, 0.0f},  {kSin45, kSin45},   {0.0f, 1.0f},  {-kSin45, kSin45},
                                 {-1.0f, 0.0f}, {-kSin45, -kSin45}, {0.0f, -1.0f}, {kSin45, -kSin45}};  // :56-82
    std::vector<C> c(kFirBuffer);
    for(int i = 0; i < kFirPad; i++) c[i] = {0.0f, 0.0f};
    for(int i = 0; i < kFirPad; i++) c[kFirBuffer - i - 1] = {0.0f, 0.0f};
    for(int n = 0; n < kWindowSamples; n++) c[kFirPad + n] = {in[n].re, in[n].im};  // :93-115

    if(with_shift)
        for(int i = 0; i < kFirBuffer; i++) c[i] = cmul_dev(c[i], w_left[i & 7]);  // :44-48

    const int n_filtered = kFirBuffer - 32;  // (NumSlices-1)*32 = 5216 outputs per pass
    // forward pass (:163-187)
    {
        std::vector<C> y(c);
        for(int i = 0; i < n_filtered; i++)
        {
            C s = {0.0f, 0.0f};
            for(int t = 0; t < kFirTaps; t++) s = smac_dev(s, kFirTapValue[t], c[i + (16 - kFirTapIndex[t])]);
            y[i] = s;
        }
        c.swap(y);
    }
    // reverse pass (:195-219)
    {
        std::vector<C> z(c);
        for(int i = kFirBuffer - 1; i >= kFirBuffer - n_filtered; i--)
        {
            C s = {0.0f, 0.0f};
            for(int t = 0; t < kFirTaps; t++) s = smac_dev(s, kFirTapValue[t], c[i - (16 - kFirTapIndex[t])]);
            z[i] = s;
        }
        c.swap(z);
    }
    if(with_shift)
        for(int i = 0; i < kFirBuffer; i++) c[i] = cmul_dev(c[i], w_right[i & 7]);  // :85-89

    for(int n = 0; n < kWindowSamples; n++) out[n] = {c[kFirPad + n].re, c[kFirPad + n].im};  // :224-233
}

// analytic_fft.cu:18-157 (cuFFT replaced by a plain radix-2 float FFT)
void orc_analytic_fft(const orc_complex* in, orc_complex* out)
{
    const int nfft = kFftSize;
    const int nh = nfft / 2;
    // filter table (:32-57)
    std::vector<float> h(nfft, 0.0f);
    {
        const float df = 12000.0f / nfft;
        const float pi = kPiF;
        const float t = 1.0f / 2000.0f;
        const float beta = 0.1f;
        for(int i = 0; i < nh; i++)
        {
            float ff = i * df;
            float f = ff - 1500.0f;
            h[i] = 1.0f;
            if(fabsf(f) > (1 - beta) / (2 * t) && fabsf(f) <= (1 + beta) / (2 * t))
            {
                h[i] = h[i] * 0.5f * (1.0f + static_cast<float>(cos((pi * t / beta) * (fabsf(f) - (1 - beta) / (2 * t)))));
            }
            else if(fabsf(f) > (1 + beta) / (2 * t))
            {
                h[i] = 0;
            }
        }
    }
    const float fac = 2.0f / nfft;  // :88
    std::vector<C> buf(nfft, C{0.0f, 0.0f});
    for(int i = 0; i < kWindowSamples; i++) buf[i] = smul(fac, C{in[i].re, in[i].im});
    fft_radix2(buf, -1);
    for(int i = 0; i < nh; i++) buf[i] = {buf[i].re * h[i], buf[i].im * h[i]};  // :118-121, operator* at :13-16
    buf[0] = {buf[0].re * 0.5f, buf[0].im * 0.5f};                               // :124
    for(int i = nh; i < nfft; i++) buf[i] = {0.0f, 0.0f};                        // :127
    fft_radix2(buf, +1);                                                         // :132 (unnormalised)
    for(int i = 0; i < kWindowSamples; i++) out[i] = {buf[i].re, buf[i].im};
}

void orc_frontend_audio(const int16_t* win, int analytic_method, orc_complex* out)
{
    std::vector<orc_complex> a(kWindowSamples);
    orc_normalize_audio(win, a.data());
    if(analytic_method == 1) orc_analytic_fft(a.data(), out);
    else orc_analytic2(a.data(), out, 1);
}

void orc_frontend_iq(const int8_t* win, orc_complex* out)
{
    std::vector<orc_complex> a(kWindowSamples);
    orc_convert_iq(win, a.data());
    orc_analytic2(a.data(), out, 0);
}

// result_keeper.cuh:61-73
void orc_clear_items(const orc_ctx* ctx, orc_item* items)
{
    std::memset(items, 0, sizeof(orc_item) * ctx->total_items);
}

// scan_kernel.cuh:27-366
void orc_scan(const orc_ctx* ctx, const orc_complex* cdat, orc_item* items)
{
    C cb42[42];
    fill_cb42(cb42);
#pragma omp parallel for schedule(dynamic, 1) num_threads(ctx->num_threads)
    for(int b = 0; b < ctx->num_blocks; b++)
    {
        std::vector<C> cdat2(kWindowSamples);
        float f0;
        mix_window(ctx, b, cdat, cdat2.data(), &f0);
        for(int p = 0; p < ctx->scan_depth; p++)
        {
            Slot slots[kSlotsPerPattern];
            for(int i = 0; i < kSlotsPerPattern; i++) slots[i] = {0u, 0.0f};
            for(int slice = 0; slice < kScanSlices; slice++)
            {
                float xb[kSlicePositions];
                for(int t = 0; t < kSlicePositions; t++) xb[t] = scan_position(cdat2.data(), cb42, kPatternMask[p], slice * kSlicePositions + t);
                slice_update(xb, slice * kSlicePositions, slots);
            }
            for(int c = 0; c < kSlotsPerPattern; c++)
            {
                orc_item& it = item_at(ctx, items, b, p, c);  // put_candidate, result_keeper.cuh:93-103
                it.block_idx = b;
                it.pattern_idx = p;
                it.pos = slots[c].pos;
                it.xb = slots[c].xb;
                it.f0 = -f0;
                it.num_avg = kPatternNumAvg[p];
            }
        }
    }
}

void orc_scan_xb(const orc_ctx* ctx, const orc_complex* cdat, int block_idx, int pattern_idx, float* xb)
{
    C cb42[42];
    fill_cb42(cb42);
    std::vector<C> cdat2(kWindowSamples);
    float f0;
    mix_window(ctx, block_idx, cdat, cdat2.data(), &f0);
    for(int pos = 0; pos < kScanPositions; pos++) xb[pos] = scan_position(cdat2.data(), cb42, kPatternMask[pattern_idx], pos);
}

// softbits_kernel.cuh:9-249
void orc_softbits(const orc_ctx* ctx, const orc_complex* cdat, orc_item* items)
{
    C cb42[42];
    fill_cb42(cb42);
    float pp[12];
    fill_pp(pp);
#pragma omp parallel for schedule(dynamic, 1) num_threads(ctx->num_threads)
    for(int b = 0; b < ctx->num_blocks; b++)
    {
        std::vector<C> cdat2(kWindowSamples);
        float f0;
        mix_window(ctx, b, cdat, cdat2.data(), &f0);
        for(int p = 0; p < ctx->scan_depth; p++)
        {
            for(int c = 0; c < kSlotsPerPattern; c++)
            {
                orc_item& it = item_at(ctx, items, b, p, c);
                int nbadsync;
                softbits_core(cdat2.data(), cb42, pp, kPatternMask[p], it.pos, nullptr, it.softbits_wo_sync, &nbadsync);
                it.nbadsync = nbadsync;  // put_softbits, result_keeper.cuh:105-115
            }
        }
    }
}

void orc_softbits_at(const orc_ctx* ctx, const orc_complex* cdat, int block_idx, int pattern_idx, uint32_t pos, float* soft144, float* llr128,
                     int32_t* nbadsync)
{
    C cb42[42];
    fill_cb42(cb42);
    float pp[12];
    fill_pp(pp);
    std::vector<C> cdat2(kWindowSamples);
    float f0;
    mix_window(ctx, block_idx, cdat, cdat2.data(), &f0);
    int nb;
    softbits_core(cdat2.data(), cb42, pp, kPatternMask[pattern_idx], pos, soft144, llr128, &nb);
    *nbadsync = nb;
}

// index_kernel.cuh:7-76
int orc_index(const orc_ctx* ctx, const orc_item* items, int32_t* indexes)
{
    int total = 0;
    for(int k = 0; k < ctx->total_items; k++)
    {
        if(items[k].nbadsync <= ctx->nbadsync_threshold) indexes[total++] = k;
    }
    return total;
}

// ldpc_kernel.cuh:100-249 over the index list
void orc_ldpc(const orc_ctx* ctx, orc_item* items, const int32_t* indexes, int n_indexed)
{
#pragma omp parallel for schedule(dynamic, 16) num_threads(ctx->num_threads)
    for(int i = 0; i < n_indexed; i++)
    {
        orc_item& it = items[indexes[i]];
        char msg[kMessageBits];
        int iters = 0, nhard = 0;
        if(ldpc_core(it.softbits_wo_sync, msg, &iters, &nhard))
        {
            it.is_message_present = 1;  // put_ldpc_decode_result, result_keeper.cuh:140-152
            it.ldpc_num_iterations = iters;
            it.ldpc_num_hard_errors = nhard;
            std::memcpy(it.message, msg, kMessageBits);
        }
    }
}

int orc_ldpc_one(const float* llr128, char* message77, int32_t* iters, int32_t* nhard)
{
    int it = 0, nh = 0;
    bool ok = ldpc_core(llr128, message77, &it, &nh);
    *iters = it;
    *nhard = nh;
    return ok ? 1 : 0;
}

// main.cu:461-468
int orc_decode_window(const orc_ctx* ctx, const orc_complex* cdat, orc_item* items, int32_t* indexes)
{
    orc_clear_items(ctx, items);
    orc_scan(ctx, cdat, items);
    orc_softbits(ctx, cdat, items);
    const int n = orc_index(ctx, items, indexes);
    if(n > 0) orc_ldpc(ctx, items, indexes, n);
    return n;
}

// ldpc_kernel.cuh:32-43
uint16_t orc_crc13(const uint8_t* buf, int length)
{
    const uint16_t* table = crc_table();
    uint16_t remainder = 0;
    for(int i = 0; i < length; i++)
    {
        const int index = (remainder >> (13 - 8)) & 0xff;
        remainder <<= 8;
        remainder |= buf[i];
        remainder ^= table[index];
    }
    return remainder & 0x1fff;
}

// convert_cw_to_bytes (ldpc_kernel.cuh:9-30) + check_crc (:45-63)
int orc_check_crc_bits(const char* cw)
{
    uint8_t byte_buf[16];
    for(int i = 0; i < 16; i++)
    {
        int v = 0;
        for(int b = 0; b < 8; b++)
        {
            const int idx = i * 8 + b;
            const int bit = (idx < 90) ? (cw[idx] & 1) : 0;  // bytes 12..15 are never read; bits 90..95 are masked off
            v = (v << 1) | bit;
        }
        byte_buf[i] = static_cast<uint8_t>(v);
    }
    const uint32_t unaligned = ((static_cast<uint32_t>(byte_buf[9]) & 0x7) << 16) | (static_cast<uint32_t>(byte_buf[10]) << 8) | (byte_buf[11] & 0xc0);
    const uint16_t crc_from_message = unaligned >> 6;
    byte_buf[9] &= 0xf8;
    byte_buf[10] = 0;
    byte_buf[11] = 0;
    const uint16_t calculated = orc_crc13(byte_buf, 12);
    return crc_from_message == calculated;
}

void orc_snr_init(orc_snr_tracker* t)
{
    t->noise_power = 0.0f;
    t->snr = 0.0f;
}

// snr_tracker.cu:21-37
void orc_segment_power(const orc_complex* data, unsigned length, float* seg8)
{
    const int num_elements = 8;
    for(int i = 0; i < num_elements; i++) seg8[i] = 0.0f;
    const int block_size = length / num_elements;
    for(int idx = 0; idx < block_size * num_elements; idx++)
    {
        C d = {data[idx].re, data[idx].im};
        C y = cmul(cconj(d), d);
        seg8[idx / block_size] += y.re;
    }
}

// snr_tracker.cu:21-69
void orc_snr_process(orc_snr_tracker* t, const orc_complex* data, unsigned length)
{
    float arr[8];
    orc_segment_power(data, length, arr);
    float summ = 0.0f;
    for(int i = 0; i < 8; i++) summ = summ + arr[i];
    const float avg = summ / 8;
    float peak = arr[0];
    for(int i = 1; i < 8; i++)
        if(peak < arr[i]) peak = arr[i];

    if(t->noise_power <= 0.0f) t->noise_power = avg;
    else if(avg > t->noise_power) t->noise_power = 0.9f * t->noise_power + 0.1f * avg;
    else t->noise_power = avg;

    if(t->noise_power > 0.0f) t->snr = 10.0f * log10f(peak / t->noise_power - 1.0f);
    else t->snr = 0.0f;
    if(t->snr > 24.0f) t->snr = 24.0f;
    if(t->snr < -8.0f) t->snr = -8.0f;
}

int orc_snr_int(const orc_snr_tracker* t)
{
    return static_cast<int>(t->snr);
}

// decode_softbits.cpp:25-30: 1 = passes the i3/n3 gate
int orc_message_gate(const char* m)
{
    const int n3 = (m[71] << 2) | (m[72] << 1) | m[73];
    const int i3 = (m[74] << 2) | (m[75] << 1) | m[76];
    if((i3 == 0 && (n3 == 1 || n3 == 3 || n3 == 4 || n3 > 5)) || i3 == 3 || i3 > 5) return 0;
    return 1;
}

}  // extern "C"
