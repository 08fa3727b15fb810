// Developer microbenchmark: what one gfx950 SIMD sustains per wave64 VALU instruction, per opcode class and per
// number of resident waves.  Method (MI355X_MICROARCH.md, "DVFS give-back" item 6 / In-kernel stamps):
//   * the chip is first loaded for >= 2 s with back-to-back launches of the kernel under test (clock settles);
//   * every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its instruction loop; cycles
//     per instruction = median over waves of dt_memtime * waves_per_SIMD / instructions_per_wave - in SHADER cycles
//     at the clock the chip actually holds, which is reported next to it (dt_memtime / dt_memrealtime * 100 MHz);
//   * 8 independent accumulators (no dependent-issue stalls); operand forms read at most TWO distinct VGPRs unless
//     the row says otherwise (three distinct VGPR reads cost an extra register-file cycle);
//   * waves per SIMD are pinned with dynamic LDS: 256-thread blocks (one wave per SIMD), floor(160 KB / w) each.
// Output: one table row per (op, waves/SIMD).  Used to price the kernels' instruction mixes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define A8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)

enum Op
{
    FMA2,       // v_fma_f32 d, d, b, d           (2 distinct VGPRs)
    FMA3,       // v_fma_f32 d, d, b, c           (3 distinct VGPRs)
    FMAK,       // v_fma_f32 d, d, 0.5, d         (inline constant)
    MUL,        // v_mul_f32 d, d, b
    ADD,        // v_add_f32 d, d, b
    MAXF,       // v_max_f32 d, d, b
    PKFMA,      // v_pk_fma_f32 D, D, B, D
    PKMUL,      // v_pk_mul_f32 D, D, B
    PKADD,      // v_pk_add_f32 D, D, B
    EXP,        // v_exp_f32 d, d
    RCP,        // v_rcp_f32 d, d
    EXP_FMA,    // 1 v_exp_f32 : 3 v_fma_f32 interleaved
    CNDMASK,    // v_cndmask_b32 d, d, b, vcc
    CMP,        // v_cmp_gt_f32 s[..], d, b       (SGPR-pair result)
    BFI,        // v_bfi_b32 d, s, d, b
    ANDB,       // v_and_b32 d, d, b
    BCNT,       // v_bcnt_u32_b32 d, d, b
    ADD_DPP,    // v_add_f32_dpp d, d, d quad_perm
    MOV_DPP,    // v_mov_b32_dpp d, b row_shr:1
    CND_SGPR,   // v_cndmask_b32 d, d, b, s[20:21]  (VOP3, mask in an SGPR pair set before the loop)
    CND_VCC_SET,// v_cndmask_b32 d, d, b, vcc       (vcc written by a v_cmp right before each group)
    MINF,       // v_min_f32
    SUB,        // v_sub_f32
    MAX3,       // v_max3_f32 d, d, b, d
    MED3,       // v_med3_f32 d, d, b, c
    CMP_VCC,    // v_cmp_gt_f32 vcc, d, b
    MOV,        // v_mov_b32 d, b
    LSHL,       // v_lshlrev_b32 d, 1, d
    ADDU,       // v_add_u32 d, d, b
    READLANE,   // v_readlane_b32 s20, d, 5
    FMAC,       // v_fmac_f32 d, d, b   (VOP2)
    SQRT,       // v_sqrt_f32
    EXP_RCP,    // v_exp_f32 / v_rcp_f32 alternating
    MAX_ALT,    // v_max_f32 d, b, d  with NaN-free small values (operand order swapped)
    N_OPS
};

static const char* kNames[N_OPS] = {"v_fma_f32 (2 vgpr srcs)", "v_fma_f32 (3 vgpr srcs)", "v_fma_f32 (inline const)", "v_mul_f32", "v_add_f32", "v_max_f32",
                                    "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32", "1 v_exp : 3 v_fma", "v_cndmask_b32", "v_cmp_gt_f32 -> sgpr",
                                    "v_bfi_b32", "v_and_b32", "v_bcnt_u32_b32", "v_add_f32_dpp quad_perm", "v_mov_b32_dpp row_shr", "v_cndmask_b32 (sgpr mask)", "v_cmp + 8 v_cndmask (vcc)",
                                    "v_min_f32", "v_sub_f32", "v_max3_f32", "v_med3_f32", "v_cmp_gt_f32 -> vcc", "v_mov_b32", "v_lshlrev_b32", "v_add_u32", "v_readlane_b32",
                                    "v_fmac_f32", "v_sqrt_f32", "v_exp/v_rcp alternating", "v_max_f32 (swapped srcs)"};

template<int OP>
__global__ __launch_bounds__(256) void k(uint64_t* stamps, float* sink, int iters, float seed)
{
    extern __shared__ char pad[];
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float b = 0.999f + 1e-6f * threadIdx.x, c = 0.001f;
    v2f pb = {b, b};
    asm volatile("s_mov_b32 s22, 0x7fffffff\n s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x33333333" ::: "s22", "s20", "s21");
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for(int i = 0; i < iters; i++)
    {
#define F8(body) asm volatile(REP4(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21", "s22");
#define P8(body) asm volatile(REP4(body) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        if(OP == FMA2) F8("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n")
        if(OP == FMA3) F8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
        if(OP == FMAK) F8("v_fma_f32 %0, %0, 0.5, %0\n v_fma_f32 %1, %1, 0.5, %1\n v_fma_f32 %2, %2, 0.5, %2\n v_fma_f32 %3, %3, 0.5, %3\n v_fma_f32 %4, %4, 0.5, %4\n v_fma_f32 %5, %5, 0.5, %5\n v_fma_f32 %6, %6, 0.5, %6\n v_fma_f32 %7, %7, 0.5, %7\n")
        if(OP == MUL) F8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
        if(OP == ADD) F8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
        if(OP == MAXF) F8("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
        if(OP == PKFMA) P8("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n")
        if(OP == PKMUL) P8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
        if(OP == PKADD) P8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
        if(OP == EXP) F8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
        if(OP == RCP) F8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
        if(OP == EXP_FMA) F8("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n")
        if(OP == CNDMASK) F8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
        if(OP == CMP) F8("v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[20:21], %1, %8\n v_cmp_gt_f32 s[20:21], %2, %8\n v_cmp_gt_f32 s[20:21], %3, %8\n v_cmp_gt_f32 s[20:21], %4, %8\n v_cmp_gt_f32 s[20:21], %5, %8\n v_cmp_gt_f32 s[20:21], %6, %8\n v_cmp_gt_f32 s[20:21], %7, %8\n")
        if(OP == BFI) F8("v_bfi_b32 %0, s22, %0, %8\n v_bfi_b32 %1, s22, %1, %8\n v_bfi_b32 %2, s22, %2, %8\n v_bfi_b32 %3, s22, %3, %8\n v_bfi_b32 %4, s22, %4, %8\n v_bfi_b32 %5, s22, %5, %8\n v_bfi_b32 %6, s22, %6, %8\n v_bfi_b32 %7, s22, %7, %8\n")
        if(OP == ANDB) F8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
        if(OP == BCNT) F8("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n")
        if(OP == ADD_DPP) F8("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
        if(OP == MOV_DPP) F8("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n")
        if(OP == CND_SGPR) F8("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n ")
        if(OP == MINF) F8("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n ")
        if(OP == SUB) F8("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n ")
        if(OP == MAX3) F8("v_max3_f32 %0, %0, %8, %0\n v_max3_f32 %1, %1, %8, %1\n v_max3_f32 %2, %2, %8, %2\n v_max3_f32 %3, %3, %8, %3\n v_max3_f32 %4, %4, %8, %4\n v_max3_f32 %5, %5, %8, %5\n v_max3_f32 %6, %6, %8, %6\n v_max3_f32 %7, %7, %8, %7\n ")
        if(OP == MED3) F8("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n ")
        if(OP == CMP_VCC) F8("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n ")
        if(OP == MOV) F8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n ")
        if(OP == LSHL) F8("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n ")
        if(OP == ADDU) F8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n ")
        if(OP == READLANE) F8("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s20, %1, 5\n v_readlane_b32 s20, %2, 5\n v_readlane_b32 s20, %3, 5\n v_readlane_b32 s20, %4, 5\n v_readlane_b32 s20, %5, 5\n v_readlane_b32 s20, %6, 5\n v_readlane_b32 s20, %7, 5\n ")
        if(OP == FMAC) F8("v_fmac_f32 %0, %0, %8\n v_fmac_f32 %1, %1, %8\n v_fmac_f32 %2, %2, %8\n v_fmac_f32 %3, %3, %8\n v_fmac_f32 %4, %4, %8\n v_fmac_f32 %5, %5, %8\n v_fmac_f32 %6, %6, %8\n v_fmac_f32 %7, %7, %8\n ")
        if(OP == SQRT) F8("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n ")
        if(OP == MAX_ALT) F8("v_max_f32 %0, %8, %0\n v_max_f32 %1, %8, %1\n v_max_f32 %2, %8, %2\n v_max_f32 %3, %8, %3\n v_max_f32 %4, %8, %4\n v_max_f32 %5, %8, %5\n v_max_f32 %6, %8, %6\n v_max_f32 %7, %8, %7\n ")
        if(OP == CND_VCC_SET) F8("v_cmp_gt_f32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n ")
        if(OP == EXP_RCP) F8("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_exp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_exp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_exp_f32 %6, %6\n v_rcp_f32 %7, %7\n ")
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if((threadIdx.x & 63) == 0)
    {
        stamps[2 * wave] = t1 - t0;       // stamps live in a buffer of their own; no output value depends on them
        stamps[2 * wave + 1] = r1 - r0;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

static int g_warm_ms = 2000;

template<int OP>
void run(int w)
{
    const int iters = 4000, blocks = 256 * w;
    const size_t lds = (160 * 1024) / w - (w == 1 ? 0 : 64);  // exactly w blocks fit one CU
    uint64_t* d_st;
    float* d_sink;
    hipMalloc(&d_st, sizeof(uint64_t) * 2 * blocks * 4);
    hipMalloc(&d_sink, sizeof(float) * blocks * 256);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    const auto t_begin = std::chrono::steady_clock::now();
    int launches = 0;
    while(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_begin).count() < g_warm_ms || launches < 3)
    {
        for(int i = 0; i < 8; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d_st, d_sink, iters, 1.0f);
        hipDeviceSynchronize();
        launches += 8;
    }
    // kernel-level rate: a batch of back-to-back launches under HIP events, after the warm-up
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int batch = 32;
    hipEventRecord(e0);
    for(int i = 0; i < batch; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d_st, d_sink, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float batch_ms = 0.0f;
    hipEventElapsedTime(&batch_ms, e0, e1);
    std::vector<uint64_t> st(2 * blocks * 4);
    hipMemcpy(st.data(), d_st, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for(int i = 0; i < blocks * 4; i++)
    {
        cyc.push_back(static_cast<double>(st[2 * i]));
        if(st[2 * i + 1]) clk.push_back(static_cast<double>(st[2 * i]) / static_cast<double>(st[2 * i + 1]) * 100e6);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double n_instr = (OP == CND_VCC_SET ? 36.0 : 32.0) * iters;  // per wave
    const double clock = clk.empty() ? 2.4e9 : clk[clk.size() / 2];
    // per SIMD: w waves x n_instr instructions per launch; elapsed shader cycles per launch = batch_ms/batch * clock
    const double simd_cyc = batch_ms * 1e-3 / batch * clock / (w * n_instr);
    printf("%-28s waves/SIMD=%d  %6.2f cycles per wave-instr per SIMD (kernel-level)   one wave sees %6.2f cycles/instr (median; p10 %.2f p90 %.2f)   clock %.2f GHz\n", kNames[OP], w,
           simd_cyc, cyc[cyc.size() / 2] / n_instr, cyc[cyc.size() / 10] / n_instr, cyc[cyc.size() * 9 / 10] / n_instr, clock / 1e9);
    fflush(stdout);
    hipFree(d_st);
    hipFree(d_sink);
}

template<int OP>
void sweep()
{
    for(int w : {1, 2, 8}) run<OP>(w);
}

int main(int argc, char** argv)
{
    if(argc > 1) g_warm_ms = atoi(argv[1]);
    sweep<FMA2>();
    sweep<FMA3>();
    sweep<FMAK>();
    sweep<MUL>();
    sweep<ADD>();
    sweep<MAXF>();
    sweep<PKFMA>();
    sweep<PKMUL>();
    sweep<PKADD>();
    sweep<EXP>();
    sweep<RCP>();
    sweep<EXP_FMA>();
    sweep<CNDMASK>();
    sweep<CMP>();
    sweep<BFI>();
    sweep<ANDB>();
    sweep<BCNT>();
    sweep<ADD_DPP>();
    sweep<MOV_DPP>();
    sweep<CND_SGPR>();
    sweep<CND_VCC_SET>();
    sweep<MINF>();
    sweep<MAX_ALT>();
    sweep<SUB>();
    sweep<MAX3>();
    sweep<MED3>();
    sweep<CMP_VCC>();
    sweep<MOV>();
    sweep<LSHL>();
    sweep<ADDU>();
    sweep<READLANE>();
    sweep<FMAC>();
    sweep<SQRT>();
    sweep<EXP_RCP>();
    return 0;
}
