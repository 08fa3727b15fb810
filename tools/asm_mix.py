"""Developer tool: price a stretch of gfx950 assembly with the measured per-instruction issue costs
(profiles/r02_valu_issue_microbench.txt, 8 waves/SIMD): cycles one SIMD needs per wave for the VALU instructions between
two line numbers of a hipcc -S listing.  usage: asm_mix.py file.s first_line last_line"""
import re
import sys

# the VCC form of v_cndmask measured 15-23 cycles in the microbenchmark's 32-in-a-row pattern, but replacing such selects in the
# kernels (scan fold, softbits sqrt) changed nothing: priced as any other half-rate instruction
FULL, HALF, QUARTER, VCCSEL = 2.45, 4.3, 8.2, 4.3


def cost(op, line):
    if op.startswith("v_cndmask_b32_e32") or (op.startswith("v_cndmask") and line.rstrip().endswith("vcc")):
        return "cndmask(vcc)", VCCSEL
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "transcendental", QUARTER
    if "_dpp" in op or "_sdwa" in op:
        return "dpp", HALF
    if op.startswith("v_pk_"):
        return "packed", HALF
    if op.startswith(("v_max", "v_min", "v_med3", "v_cmp", "v_cmpx", "v_bfi", "v_bfe", "v_bcnt", "v_lshl", "v_lshr", "v_ashr", "v_cndmask", "v_readlane", "v_readfirstlane",
                      "v_writelane", "v_mbcnt", "v_alignbit", "v_perm", "v_cvt", "v_mad_u", "v_mul_lo", "v_mul_hi", "v_mad_i", "v_rndne", "v_trunc", "v_floor", "v_fract", "v_ldexp",
                      "v_frexp", "v_div", "v_lshl_add", "v_add3", "v_or3", "v_and_or", "v_xad", "v_add_lshl", "v_lshl_or", "v_mad_u64")):
        return "half-rate", HALF
    return "full-rate", FULL


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    tot = {}
    other = {}
    for i, line in enumerate(open(path), 1):
        if i < a or i > b:
            continue
        m = re.match(r"\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            cls, c = cost(op, line.split(";")[0])
            n, cyc = tot.get(cls, (0, 0.0))
            tot[cls] = (n + 1, cyc + c)
        else:
            k = "ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "salu/other"
            other[k] = other.get(k, 0) + 1
    n_all = sum(n for n, _ in tot.values())
    c_all = sum(c for _, c in tot.values())
    for cls, (n, c) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {cls:16s} {n:5d} instr  {c:8.0f} cycles")
    print(f"  VALU total       {n_all:5d} instr  {c_all:8.0f} cycles  ({c_all / max(n_all, 1):.2f} avg);  others: {other}")


if __name__ == "__main__":
    main()
