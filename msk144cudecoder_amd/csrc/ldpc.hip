// ldpc_bp: log-domain belief propagation for the (128,90) code + CRC-13 + hard-error gate.
//
// Replaces ldpc_kernel (ldpc_kernel.cuh:9-249).  The reference runs one 128-thread
// block per gated candidate; here ONE 64-lane wave decodes one codeword (2 bits, 6 edges per lane; which bits a lane
// owns, which lane walks which check and in which round it meets which edge: ldpc_layout.h, generated for the LDS banks):
//   * hard decisions live in two 64-bit ballot masks (SGPRs); the 38 parity checks are
//     popcount(cw & H_row) on the 38 check lanes, the hard-error count is a popcount of a ballot -
//     no 11x38 byte scatter, no block reductions, no barriers;
//   * tanh(-toc/2) is evaluated once per edge (384 per iteration, the reference recomputes 3840) with
//     a 5-instruction exp2/rcp form (absolute error ~1.5e-7) and parked in a per-wave, bit-major LDS tile with M0-relative
//     add-TID stores; lanes 0..37 gather their check's eleven factors, one per round (within a 32-lane group the real reads of a
//     round hit 32 different banks and the constant-1.0 reads of the degree-10 checks a bank they leave free: ldpc_layout.h),
//     form the leave-one-out products with prefix/suffix products (27 multiplies instead of 110) and store them check-major
//     with add-TID stores at searched row residues; each edge reads its own product back.  34 LDS instructions and 2 bank-conflict
//     cycles per iteration (PMC: profiles/counters.json): the CU's LDS pipe charges an instruction 2.3-4.3 cycles whatever its
//     active lanes, and four SIMDs share it;
//   * the piecewise-linear atanh keeps the reference's breakpoints and offsets; (z-c)/d is evaluated
//     as z*(2/d) - c*(2/d) in one fma on the common piece, (z-c)*(2/d) on the rare upper pieces: within 1.6 ulp of the quotient;
//   * CRC-13 runs as a wave-uniform bit-serial division only when all 38 checks are satisfied;
//   * the 10th message update of the reference (whose result is never used) is skipped.
// Work distribution: grid = (blocks per channel, channels); waves stride over the channel's index list
// (idx/n_idx from index_kernel), so no host read of N_idx sits between the kernels
// (the reference dereferences a device pointer on the host here, result_keeper.cuh:162).
#include "ldpc_layout.h"
#include "msk144_kernels.h"
#include "wave64.h"

namespace msk144
{

namespace
{

constexpr int kLdpcThreads = 64;   // one wave per workgroup: waves never wait for a slower sibling to free the workgroup's slot (-1.5 % against 256)
constexpr int kLdpcWaves = kLdpcThreads / 64;
// Per-wave LDS (ldpc_layout.h): forward tile Tf[3h + i][lane] (bit-major: the tanh of the edge instruction i of half h handles in
// lane l), 32 cells holding the constant 1.0, one per bank (the missing eleventh factor of the degree-10 checks, read from a bank
// the real reads of the same access leave free), then the eleven backward rows Tb[round] at kRowBase[round] + lane of the check.
constexpr int kFwdCells = 2 * kEdgesPerBit * 64;   // 384
static_assert(kOnesBase == kFwdCells && kOnesBase % 32 == 0, "cell kOnesBase + b sits on bank b, right behind the forward tile");
constexpr int kTileFloats = kTileCells;

// Edge tables derived at compile time from the check-major graph and the layout of ldpc_layout.h (bit -> lane, check -> lane,
// first-edge order and the round in which a check lane meets each of its edges; tools/layout/make_layout.py).
struct EdgeTables
{
    uint16_t cell[2][64][kEdgesPerBit];  // backward cell (tile-relative) of the edge handled by instruction i of (half h, lane l)
    uint16_t fwd[64][kMaxCheckDegree];   // forward-tile cell check lane L reads in round r (a 1.0 cell where the check has no edge)
    uint64_t hlo[64];                    // parity-check row of the lane's check as masks over ballot(half 0), ballot(half 1)
    uint64_t hhi[64];
    uint8_t full[64];                    // the lane's check has 11 bits (ldpc_context.cuh:160-163)
    uint8_t pos_of_bit[kCodeBits];       // codeword bit n sits at ballot position h*64 + lane
};

constexpr EdgeTables make_edge_tables()
{
    EdgeTables t{};
    for(int h = 0; h < 2; h++)
        for(int l = 0; l < 64; l++) t.pos_of_bit[kBitOfLane[h][l]] = static_cast<uint8_t>(h * 64 + l);
    int cnt[kCodeBits] = {};
    int8_t e_check[kCodeBits][kEdgesPerBit] = {};
    int8_t e_slot[kCodeBits][kEdgesPerBit] = {};
    for(int l = 0; l < 64; l++)
    {
        t.hlo[l] = 0;
        t.hhi[l] = 0;
        t.full[l] = 0;
        for(int r = 0; r < kMaxCheckDegree; r++) t.fwd[l][r] = kOnesBase;  // lanes >= 38 never read
    }
    for(int c = 0; c < kChecks; c++)
    {
        const int cl = kLaneOfCheck[c];
        t.full[cl] = kCheckBits[c][kMaxCheckDegree - 1] >= 0 ? 1 : 0;
        if(!t.full[cl]) t.fwd[cl][kRoundOfSlot[c][kMaxCheckDegree - 1]] = kOneCellOfCheck[c];  // the empty round of a degree-10 check
        for(int j = 0; j < kMaxCheckDegree; j++)
        {
            const int n = kCheckBits[c][j];
            if(n < 0) continue;
            e_check[n][cnt[n]] = static_cast<int8_t>(c);  // edges of a bit in ascending check order = the reference's k
            e_slot[n][cnt[n]] = static_cast<int8_t>(kRoundOfSlot[c][j]);  // the round in which the check lane handles this edge
            cnt[n]++;
            const int p = t.pos_of_bit[n];
            if(p < 64) t.hlo[cl] |= (1ull << p);
            else t.hhi[cl] |= (1ull << (p - 64));
        }
    }
    for(int h = 0; h < 2; h++)
        for(int l = 0; l < 64; l++)
        {
            const int n = kBitOfLane[h][l];
            for(int i = 0; i < kEdgesPerBit; i++)
            {
                // instructions 0 and 1 may take the bit's first two edges in either order: (tov0 + tov1) + tov2 is commutative in them
                const int k = (i < 2 && kSwapFirstEdges[n]) ? 1 - i : i;
                t.cell[h][l][i] = static_cast<uint16_t>(kRowBase[e_slot[n][k]] + kLaneOfCheck[e_check[n][k]]);
                t.fwd[kLaneOfCheck[e_check[n][k]]][e_slot[n][k]] = static_cast<uint16_t>((h * kEdgesPerBit + i) * 64 + l);
            }
        }
    return t;
}

constexpr EdgeTables kEdges = make_edge_tables();

// BP runs in LOG2-SCALED units: every LLR-domain quantity (llr, zn, toc, tov) carries a factor
// log2(e), so that exp(-toc) is a bare v_exp_f32 of -toc' with no multiply in front of it.  The factor is
// applied once per codeword to the two LLRs of a lane and is folded into the constants below; hard
// decisions (signs) are unaffected.
constexpr float kLog2e = 1.4426950408889634f;

// log2(e) * 2 * platanh(x), platanh = ldpc_kernel.cuh:65-93: same breakpoints and offsets; the reference's
// (z - c) / d becomes (z - c) * (2 log2e / d).  z - c is exact (Sterbenz) and the product is within 1 ulp of
// the scaled quotient.  The first two pieces (x/0.83 and (z-0.4064)/0.322) meet exactly at the 0.664
// breakpoint and the slope increases there, so that branch is a max(); the upper breakpoints, where the
// reference's function jumps, stay explicit selects.
__device__ __forceinline__ float two_platanh_scaled_full(float x)
{
    const float z = __builtin_fabsf(x);
    float c = 0.4064f, r = kLog2e * 2.0f / 0.322f;
    if(z > 0.9217f)
    {
        c = 0.8378f;
        r = kLog2e * 2.0f / 0.0524f;
    }
    if(z > 0.9951f)
    {
        c = 0.9914f;
        r = kLog2e * 2.0f / 0.0012f;
    }
    float v = __builtin_fmaxf(z * (kLog2e * 2.0f / 0.83f), (z - c) * r);
    if(z > 0.9998f) v = kLog2e * 14.0f;
    return __builtin_copysignf(v, x);
}

// Same function, priced for what BP actually feeds it: a leave-one-out product of 9-10 tanh values exceeds
// the 0.9217 breakpoint on ~0.1 % of the edges of a noise codeword (~6 % of the 64-lane edge instructions have
// such a lane), so the two lower pieces (one max, no selects) run unconditionally and the three upper pieces
// sit behind a wave-uniform branch.
__device__ __forceinline__ float two_platanh_scaled(float x)
{
    const float z = __builtin_fabsf(x);
    // second piece as one fma, z r - c r: the rounded constant c r moves it by <= 1.6 ulp against (z - c) r - the same
    // order as multiplying by the rounded reciprocal instead of dividing - and saves an instruction on every edge
    constexpr float kR2 = kLog2e * 2.0f / 0.322f;
    float v = __builtin_fmaxf(z * (kLog2e * 2.0f / 0.83f), __builtin_fmaf(z, kR2, -0.4064f * kR2));
    if(__builtin_expect(__builtin_amdgcn_ballot_w64(z > 0.9217f) != 0ull, 0))
    {
        float c = 0.4064f, r = kLog2e * 2.0f / 0.322f;
        if(z > 0.9217f)
        {
            c = 0.8378f;
            r = kLog2e * 2.0f / 0.0524f;
        }
        if(z > 0.9951f)
        {
            c = 0.9914f;
            r = kLog2e * 2.0f / 0.0012f;
        }
        const float second = (z > 0.9217f) ? (z - c) * r : __builtin_fmaf(z, kR2, -0.4064f * kR2);  // lanes below the breakpoint keep the fast form
        v = __builtin_fmaxf(z * (kLog2e * 2.0f / 0.83f), second);
        if(z > 0.9998f) v = kLog2e * 14.0f;
    }
    return __builtin_copysignf(v, x);
}

// tanh(-x/2) = 1 - 2/(exp(-x)+1) for x given in log2-scaled units: exp(-x) = exp2(-x').  Hardware exp2 and
// rcp: absolute error <= ~1.5e-7 (the reference's tanhf: 6e-8 near +-1).  Relative accuracy for tiny |x| is
// deliberately not pursued: a small factor only ever produces a small check->bit message, and messages are
// added to LLRs of order 1.
__device__ __forceinline__ float tanh_neg_half_scaled(float xs)
{
    const float e = __builtin_amdgcn_exp2f(-xs);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// CRC-13 of the 96-bit block (77 message bits + zeros), bit-serial; equals the table walk of
// ldpc_kernel.cuh:32-43 with ldpc_context.cuh:185-213's table (polynomial x^13 + 0x15D7).
__device__ __forceinline__ uint32_t crc13_96(uint64_t m_hi64, uint32_t m_lo32)
{
    uint32_t rem = 0;
    for(int i = 63; i >= 0; i--)
    {
        rem = (rem << 1) | static_cast<uint32_t>((m_hi64 >> i) & 1ull);
        if(rem & 0x2000u) rem ^= (0x2000u | kCrc13Poly);
    }
    for(int i = 31; i >= 0; i--)
    {
        rem = (rem << 1) | ((m_lo32 >> i) & 1u);
        if(rem & 0x2000u) rem ^= (0x2000u | kCrc13Poly);
    }
    return rem & 0x1FFFu;
}

__global__ __launch_bounds__(kLdpcThreads) void ldpc_kernel(const DeviceStore st)
{
    __shared__ float s_t[kLdpcWaves][kTileFloats];

    const int ch = st.ch0 + blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_idx = st.n_idx[ch];
    const size_t off = static_cast<size_t>(ch) * st.K;
    // the gated-item list is read through the constant address space: uniform addresses become scalar loads (s_load_dword)
    // that return into SGPRs asynchronously - a VMEM load of a uniform value is followed at once by vmcnt(0) + readfirstlane
    typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;
    const const_i32_ptr idx = (const_i32_ptr)(st.idx + off);
    float* T = s_t[wave];

    // per-lane graph constants: bits lane and lane+64
    int e_addr[2][kEdgesPerBit];
#pragma unroll
    for(int h = 0; h < 2; h++)
#pragma unroll
        for(int k = 0; k < kEdgesPerBit; k++) e_addr[h][k] = kEdges.cell[h][lane][k];
    // forward gather: the cell this check lane reads in round r (lanes >= 38 read the constant)
    const float* f_addr[kMaxCheckDegree];
#pragma unroll
    for(int r = 0; r < kMaxCheckDegree; r++) f_addr[r] = T + kEdges.fwd[lane][r];
    const uint64_t hlo = kEdges.hlo[lane];  // zero for lanes >= 38
    const uint64_t hhi = kEdges.hhi[lane];
    const int bit_of[2] = {kBitOfLane[0][lane], kBitOfLane[1][lane]};

    // LDS byte address of this wave's tile, for the M0-relative column stores
    const uint32_t tile_m0 = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<size_t>((__attribute__((address_space(3))) float*)T)));

    // the eleventh factor of the degree-10 checks, once per bank
    if(lane < 32) T[kOnesBase + lane] = 1.0f;

    // Software pipeline over this wave's codewords: the index entry of codeword n+2 and the two LLRs of codeword n+1 are
    // fetched while codeword n iterates, so a codeword starts without the two dependent global-memory latencies
    // (idx -> LLR row) it would otherwise wait out - the kernel is latency-bound (7 waves per SIMD, three LDS round trips
    // per iteration), not issue-bound.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int stride = static_cast<int>(gridDim.x) * kLdpcWaves;
    const float* __restrict__ llr_rows = st.llr + static_cast<size_t>(blockIdx.y) * st.K * kCodeBits;
    int i = static_cast<int>(blockIdx.x) * kLdpcWaves + wave_u;
    int item_next = i < n_idx ? idx[i] : 0;
    int item_after = i + stride < n_idx ? idx[i + stride] : 0;
    float llr_next[2] = {0.0f, 0.0f};
    if(i < n_idx)
    {
        const float* __restrict__ L = llr_rows + static_cast<size_t>(item_next) * kCodeBits;
        llr_next[0] = L[bit_of[0]];
        llr_next[1] = L[bit_of[1]];
    }
    for(; i < n_idx; i += stride)
    {
        const int item = item_next;
        const float llr[2] = {llr_next[0], llr_next[1]};
        item_next = item_after;
        if(i + stride < n_idx)
        {
            const float* __restrict__ L = llr_rows + static_cast<size_t>(item_next) * kCodeBits;
            llr_next[0] = L[bit_of[0]];
            llr_next[1] = L[bit_of[1]];
        }
        if(i + 2 * stride < n_idx) item_after = idx[i + 2 * stride];
        const float llr_s[2] = {llr[0] * kLog2e, llr[1] * kLog2e};  // log2-scaled copy used by the message passing
        float tov[2][kEdgesPerBit] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};

        for(int iter = 0; iter < kLdpcIterations; iter++)
        {
            float zn[2];
            bool cw[2];
#pragma unroll
            for(int h = 0; h < 2; h++)
            {
                const float sum = f32_add(f32_add(tov[h][0], tov[h][1]), tov[h][2]);
                zn[h] = f32_add(llr_s[h], sum);
                cw[h] = zn[h] > 0.0f;
            }
            const uint64_t lo = __ballot(cw[0]);
            const uint64_t hi = __ballot(cw[1]);

            // 38 parity checks: lane c < 38 evaluates check c
            // parity of popcount(cw & H_row): fold the four masked dwords with XOR first and count once (v_bcnt is half rate)
            // v_bitop3_b32 with truth table 0x6a is (a & b) ^ c: one instruction per masked dword after the first
            uint32_t folded = static_cast<uint32_t>(lo) & static_cast<uint32_t>(hlo);
            // gfx940-family hazard: a VALU may read an SGPR written by a VALU (the ballots' v_cmp) only two wait states later;
            // the compiler pads its own instructions, an asm statement has to carry its own s_nop
            asm("s_nop 1\n\t"
                "v_bitop3_b32 %0, %1, %4, %0 bitop3:0x6a\n\t"
                "v_bitop3_b32 %0, %2, %5, %0 bitop3:0x6a\n\t"
                "v_bitop3_b32 %0, %3, %6, %0 bitop3:0x6a"
                : "+v"(folded)
                : "s"(static_cast<uint32_t>(lo >> 32)), "s"(static_cast<uint32_t>(hi)), "s"(static_cast<uint32_t>(hi >> 32)),
                  "v"(static_cast<uint32_t>(hlo >> 32)), "v"(static_cast<uint32_t>(hhi)), "v"(static_cast<uint32_t>(hhi >> 32)));
            const int par = __popc(folded) & 1;
            const uint64_t syndrome = __ballot(par != 0) & ((1ull << kChecks) - 1ull);

            if(syndrome == 0)
            {
                // hard-error count (ldpc_kernel.cuh:203-204); only an accepted codeword needs it
                const bool bad0 = cw[0] ? !(llr[0] > 0.0f) : !(llr[0] <= 0.0f);
                const bool bad1 = cw[1] ? !(llr[1] > 0.0f) : !(llr[1] <= 0.0f);
                const int nhard = __popcll(__ballot(bad0)) + __popcll(__ballot(bad1));
                // back to the codeword's own bit order: lane n fetches bits n and n + 64 from the layout-ordered ballots
                const int src_lo = kEdges.pos_of_bit[lane], src_hi = kEdges.pos_of_bit[lane + 64];  // ballot positions of codeword bits lane, lane + 64 (table read only here: rare path)
                const uint64_t w_lo = (src_lo & 64) ? hi : lo;
                const uint64_t w_hi = (src_hi & 64) ? hi : lo;
                const uint64_t c_lo = __ballot(((w_lo >> (src_lo & 63)) & 1ull) != 0);
                const uint64_t c_hi = __ballot(((w_hi >> (src_hi & 63)) & 1ull) != 0);
                // codeword MSB first: cw[0] -> bit 63 of m0, cw[64] -> bit 63 of m1
                const uint64_t m0 = __brevll(c_lo);
                const uint64_t m1 = __brevll(c_hi);
                const uint32_t crc_rx = static_cast<uint32_t>((m1 >> 38) & 0x1FFFull);        // cw[77..89]
                const uint32_t tail = static_cast<uint32_t>((m1 & 0xFFF8000000000000ull) >> 32);  // cw[64..76]
                const uint32_t crc = crc13_96(m0, tail);
                if(crc == crc_rx && nhard < kMaxHardErrors)
                {
                    if(lane == 0)
                    {
                        st.dec_flag[off + item] = 1;
                        st.dec_iter[off + item] = static_cast<uint8_t>(iter);
                        st.dec_nhard[off + item] = static_cast<uint8_t>(nhard);
                        uint32_t* w = st.dec_msg + (off + item) * 3;
                        w[0] = static_cast<uint32_t>(m0 >> 32);
                        w[1] = static_cast<uint32_t>(m0);
                        w[2] = tail;
                    }
                    break;
                }
            }
            if(iter == kLdpcIterations - 1) break;  // the reference's last update is never consumed

            // bit -> check messages, tanh once per edge (ldpc_kernel.cuh:225-241), into the bit-major forward tile: cell
            // (3h + k)*64 + lane is M0 + offset + 4*lane, so the six stores are ds_write_addtid_b32 (no address VGPR: 2.25
            // LDS cycles each against 4.25 for ds_write_b32, tools/ubench/lds_exec_groups.hip) and never conflict
            {
                float th[2][kEdgesPerBit];
                if(iter == 0)
                {
                    // first update: every check -> bit message is still 0, so the three edges of a bit all carry toc = zn = llr
                    // (ldpc_kernel.cuh:225-229 with tov = 0): one tanh per bit instead of three (zn - 0 is exact, same values)
#pragma unroll
                    for(int h = 0; h < 2; h++)
                    {
                        const float t0 = tanh_neg_half_scaled(zn[h]);
#pragma unroll
                        for(int k = 0; k < kEdgesPerBit; k++) th[h][k] = t0;
                    }
                }
                else
                {
#pragma unroll
                    for(int h = 0; h < 2; h++)
#pragma unroll
                        for(int k = 0; k < kEdgesPerBit; k++) th[h][k] = tanh_neg_half_scaled(zn[h] - tov[h][k]);
                }
                static_assert(kEdgesPerBit == 3, "six forward stores below");
                asm volatile("s_mov_b32 m0, %6\n\t"
                             "s_nop 0\n\t"  // SALU write of M0 -> LDS add-TID instruction: one wait state
                             "ds_write_addtid_b32 %0 offset:0\n\t"
                             "ds_write_addtid_b32 %1 offset:256\n\t"
                             "ds_write_addtid_b32 %2 offset:512\n\t"
                             "ds_write_addtid_b32 %3 offset:768\n\t"
                             "ds_write_addtid_b32 %4 offset:1024\n\t"
                             "ds_write_addtid_b32 %5 offset:1280"
                             :
                             : "v"(th[0][0]), "v"(th[0][1]), "v"(th[0][2]), "v"(th[1][0]), "v"(th[1][1]), "v"(th[1][2]), "s"(tile_m0)
                             : "memory");
            }
            __builtin_amdgcn_wave_barrier();

            // check node c (lane c): column T[0..10][c] -> leave-one-out products, in place
            if(lane < kChecks)
            {
                // round r: the tanh of this check's edge of round r, gathered from the bit-major tile; within a 32-lane group the
                // real reads of a round sit on 32 different banks and the 1.0 reads on a bank they leave free (ldpc_layout.h)
                float t[kMaxCheckDegree];
#pragma unroll
                for(int r = 0; r < kMaxCheckDegree; r++) t[r] = *f_addr[r];
                float pre[kMaxCheckDegree];  // pre[j] = t0*...*t(j-1)
                pre[0] = 1.0f;
#pragma unroll
                for(int j = 1; j < kMaxCheckDegree; j++) pre[j] = pre[j - 1] * t[j - 1];
                float suf = -1.0f;           // -(t(j+1)*...*t10): the column is stored NEGATED, ready for platanh(-product)
                float out[kMaxCheckDegree];
#pragma unroll
                for(int j = kMaxCheckDegree - 1; j >= 0; j--)
                {
                    out[j] = pre[j] * suf;
                    suf *= t[j];
                }
                // The column store is 4 kRowBase[r] + 4 lane: ds_write_addtid_b32 takes that address from M0 + offset + 4*lane
                // and moves no address VGPR to the LDS.  A degree-10 check also stores the product of its empty round: no edge reads it.
                static_assert(kMaxCheckDegree == 11, "eleven column stores below");
                asm volatile("s_mov_b32 m0, %11\n\t"
                             "s_nop 0\n\t"  // SALU write of M0 -> LDS add-TID instruction: one wait state
                             "ds_write_addtid_b32 %0 offset:%c12\n\t"
                             "ds_write_addtid_b32 %1 offset:%c13\n\t"
                             "ds_write_addtid_b32 %2 offset:%c14\n\t"
                             "ds_write_addtid_b32 %3 offset:%c15\n\t"
                             "ds_write_addtid_b32 %4 offset:%c16\n\t"
                             "ds_write_addtid_b32 %5 offset:%c17\n\t"
                             "ds_write_addtid_b32 %6 offset:%c18\n\t"
                             "ds_write_addtid_b32 %7 offset:%c19\n\t"
                             "ds_write_addtid_b32 %8 offset:%c20\n\t"
                             "ds_write_addtid_b32 %9 offset:%c21\n\t"
                             "ds_write_addtid_b32 %10 offset:%c22"
                             :
                             : "v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7]), "v"(out[8]), "v"(out[9]), "v"(out[10]),
                               "s"(tile_m0), "n"(kRowBase[0] * 4), "n"(kRowBase[1] * 4), "n"(kRowBase[2] * 4),
                               "n"(kRowBase[3] * 4), "n"(kRowBase[4] * 4), "n"(kRowBase[5] * 4), "n"(kRowBase[6] * 4),
                               "n"(kRowBase[7] * 4), "n"(kRowBase[8] * 4), "n"(kRowBase[9] * 4), "n"(kRowBase[10] * 4)
                             : "memory");
            }
            __builtin_amdgcn_wave_barrier();

            // check -> bit messages: all six products are read before the first (branchy) platanh so that the LDS
            // latencies overlap
            float prod[2][kEdgesPerBit];
#pragma unroll
            for(int h = 0; h < 2; h++)
#pragma unroll
                for(int k = 0; k < kEdgesPerBit; k++) prod[h][k] = T[e_addr[h][k]];
#pragma unroll
            for(int h = 0; h < 2; h++)
#pragma unroll
                for(int k = 0; k < kEdgesPerBit; k++) tov[h][k] = two_platanh_scaled(prod[h][k]);
        }
    }
}

}  // namespace

void launch_ldpc(const DeviceStore& st, hipStream_t stream)
{
    // Sixteen times the 8192 waves the chip holds at once: codewords need 1 to 10 iterations, and with a static stride the only
    // load balancing is the dispatcher handing out fresh workgroups - 131072 waves per launch (about eight codewords per wave
    // at the bench workload) measured 2.5 % faster than 32768 and 13 % faster than one resident set (8192).
    constexpr int kWavesPerLaunch = 131072;
    const int max_waves_per_channel = (st.K + kLdpcWaves - 1) / kLdpcWaves;
    int blocks = (kWavesPerLaunch / kLdpcWaves + st.nch - 1) / st.nch;
    if(blocks > max_waves_per_channel) blocks = max_waves_per_channel;
    if(blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ldpc_kernel, dim3(blocks, st.nch), dim3(kLdpcThreads), 0, stream, st);
}

}  // namespace msk144
