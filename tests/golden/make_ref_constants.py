"""Snapshot of the reference's constant tables -> tests/golden/ref_constants.json.

BUILD CONTAINER ONLY: reads /root/reference/src as TEXT (no import, no compile) and extracts the literal constants the
hot path is built on, each with the file:line it was found at.  The JSON is data (numbers), not source.  tests/
test_ref_constants.py then checks that msk144_protocol.h, the HIP kernels and the oracle use exactly these values, so
that the tables all our components share (Tanner graph, sync word, masks, FIR taps, platanh breakpoints, geometry) are
pinned to the reference mechanically instead of by a one-off reading.

    python tests/golden/make_ref_constants.py        # rewrites ref_constants.json
"""
import json
import os
import re

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_constants.json")


def read(name):
    return open(os.path.join(REF, name)).read()


def line_of(text, pos):
    return text.count("\n", 0, pos) + 1


def flt(s):
    return float(s.rstrip("fF"))


def main():
    out = {"_source": "alexander-sholohov/msk144cudecoder @ 2024_10_08, /root/reference/src read as text by tests/golden/make_ref_constants.py",
           "_cite": {}}
    cite = out["_cite"]

    # ---- ldpc_context.cuh: Tanner graph (bit-major: [bit][edge] = (slot row, check)), CRC polynomial, full rows ----
    t = read("ldpc_context.cuh")
    m = re.search(r"ldpc_reverse_map\[128\]\[3\]\[2\]\s*=\s*\{(.*?)\};", t, re.S)
    nums = [int(x) for x in re.findall(r"-?\d+", m.group(1))]
    assert len(nums) == 128 * 3 * 2
    out["ldpc_reverse_map"] = [[[nums[(n * 3 + k) * 2], nums[(n * 3 + k) * 2 + 1]] for k in range(3)] for n in range(128)]
    cite["ldpc_reverse_map"] = f"ldpc_context.cuh:{line_of(t, m.start())}-{line_of(t, m.end())}"
    m = re.search(r"#define\s+CRC13_POLY\s+(0x[0-9A-Fa-f]+)", t)
    out["crc13_poly"] = int(m.group(1), 16)
    cite["crc13_poly"] = f"ldpc_context.cuh:{line_of(t, m.start())}"
    full = [(int(x.group(1)), line_of(t, x.start())) for x in re.finditer(r"is_full_row\[(\d+)\]\s*=\s*true", t)]
    out["is_full_row"] = [c for c, _ in full]
    cite["is_full_row"] = f"ldpc_context.cuh:{full[0][1]}-{full[-1][1]}"
    m = re.search(r"is_full_row\((\d+)\)", t)
    out["num_checks"] = int(m.group(1))

    # ---- msk_context.cuh: sync word, half-sine length, template layout, averaging patterns, depth clamp ----
    t = read("msk_context.cuh")
    m = re.search(r"s8_org\[\]\s*=\s*\{([^}]*)\}", t)
    out["sync8"] = [int(x) for x in re.findall(r"\d+", m.group(1))]
    cite["sync8"] = f"msk_context.cuh:{line_of(t, m.start())}"
    m = re.search(r"for\(size_t i = 0; i < (\d+); i\+\+\)\s*\{\s*float angle = i \* pi / ([\d.]+)f;\s*pp\[i\] = sinf\(angle\);", t)
    out["pp_len"], out["pp_angle_div"] = int(m.group(1)), float(m.group(2))
    cite["pp"] = f"msk_context.cuh:{line_of(t, m.start())}"
    tpl = []
    for x in re.finditer(r"i < (\d+);\s*i\+\+\)\s*\{\s*(cb[iq])\[\s*(\d+) \+ i\] = pp\[(?:(\d+) \+ )?i\] \* s8\[(\d+)\];", t):
        tpl.append({"array": x.group(2), "base": int(x.group(3)), "count": int(x.group(1)), "pp_offset": int(x.group(4) or 0), "s8_index": int(x.group(5))})
    assert len(tpl) == 8
    out["template_segments"] = tpl
    first = re.search(r"cbq\[ 0 \+ i\]", t)
    cite["template_segments"] = f"msk_context.cuh:{line_of(t, first.start())}-{line_of(t, first.start()) + 8}"
    pats = [(tuple(int(v) for v in x.group(1).split(",")), line_of(t, x.start())) for x in re.finditer(r"PatternItem\(([01, ]+)\),", t)]
    out["patterns"] = [list(p) for p, _ in pats]
    cite["patterns"] = f"msk_context.cuh:{pats[0][1]}-{pats[-1][1]}"

    # ---- common.h: geometry ----
    t = read("common.h")
    geo = {}
    for x in re.finditer(r"constexpr\s+(?:unsigned|float)\s+(\w+)\s*=\s*([^;]+);", t):
        expr = x.group(2).strip().rstrip("f")
        if re.fullmatch(r"[\d\s+*().]+", expr):
            geo[x.group(1)] = eval(expr)  # noqa: S307 - digits and + * ( ) only
    out["common"] = geo
    cite["common"] = "common.h:14-47"

    # ---- analytic2.cuh: sc45, fs/8 rotation tables, FIR taps ----
    t = read("analytic2.cuh")
    m = re.search(r"constexpr float sc45 = ([\d.]+)f", t)
    out["sc45"] = flt(m.group(1))
    cite["sc45"] = f"analytic2.cuh:{line_of(t, m.start())}"

    def rot_table(fn):
        body = t[t.index(fn):]
        body = body[:body.index("// Apply frequency shifting")]
        rows = []
        for x in re.finditer(r"case (\d):\s*wttt = Complex\(([^,]+), ([^)]+)\);", body):
            def val(s):
                s = s.strip()
                sign = -1.0 if s.startswith("-") else 1.0
                s = s.lstrip("-")
                return sign * (out["sc45"] if s == "sc45" else flt(s))
            rows.append([val(x.group(2)), val(x.group(3))])
        assert len(rows) == 8
        return rows
    out["shift_left"] = rot_table("_frequency_shift_fs8_left")
    out["shift_right"] = rot_table("_frequency_shift_fs8_right")
    cite["shift_left"] = f"analytic2.cuh:{line_of(t, t.index('_frequency_shift_fs8_left'))}"
    cite["shift_right"] = f"analytic2.cuh:{line_of(t, t.index('_frequency_shift_fs8_right'))}"
    taps = {}
    lines = []
    conv = t[t.index("_lpf_convolution"):]
    off = t.index("_lpf_convolution")
    for x in re.finditer(r"const float h(\d+) = (-?[\d.]+)f;", conv):
        taps[int(x.group(1))] = flt(x.group(2))
        lines.append(line_of(t, off + x.start()))
    out["fir_taps"] = {str(k): taps[k] for k in sorted(taps)}
    cite["fir_taps"] = f"analytic2.cuh:{min(lines)}-{max(lines)}"

    # ---- ldpc_kernel.cuh: platanh pieces, accept rule ----
    t = read("ldpc_kernel.cuh")
    body = t[t.index("__device__ float platanh"):t.index("// The algorighm is taken")]
    bps = [flt(x) for x in re.findall(r"z <= ([\d.]+f)", body)]
    lin = [(flt(a), flt(b)) for a, b in re.findall(r"\(z - ([\d.]+f)\) / ([\d.]+f)", body)]
    first_div = flt(re.search(r"return x / ([\d.]+f)", body).group(1))
    sat = flt(re.search(r"return isign \* ([\d.]+f);", body).group(1))
    out["platanh"] = {"breakpoints": bps, "first_piece_divisor": first_div, "pieces_offset_divisor": [list(p) for p in lin], "saturation": sat}
    cite["platanh"] = f"ldpc_kernel.cuh:{line_of(t, t.index('__device__ float platanh'))}-{line_of(t, t.index('// The algorighm is taken')) - 2}"
    m = re.search(r"message_found = is_crc_valid && num_hard_errors < (\d+);", t)
    out["max_hard_errors_exclusive"] = int(m.group(1))
    cite["max_hard_errors_exclusive"] = f"ldpc_kernel.cuh:{line_of(t, m.start())}"

    # ---- softbits_kernel.cuh: normalisation constants ----
    t = read("softbits_kernel.cuh")
    m = re.search(r"const float sigma = ([\d.]+)f;", t)
    out["softbits_sigma"] = flt(m.group(1))
    cite["softbits_sigma"] = f"softbits_kernel.cuh:{line_of(t, m.start())}"
    m = re.search(r"const float sav = sum_sav / ([\d.]+);", t)
    out["softbits_mean_divisor"] = float(m.group(1))

    # ---- analytic_fft.cu: raised-cosine mask parameters ----
    t = read("analytic_fft.cu")
    out["fft_mask"] = {"t_inv": flt(re.search(r"float t = 1\.0f / ([\d.]+)f;", t).group(1)), "beta": flt(re.search(r"float beta = ([\d.]+)f;", t).group(1)),
                       "center_hz": flt(re.search(r"float f = ff - ([\d.]+)f;", t).group(1)), "sample_rate": flt(re.search(r"float df = ([\d.]+)f / nfft;", t).group(1))}
    cite["fft_mask"] = f"analytic_fft.cu:{line_of(t, t.index('float df ='))}-{line_of(t, t.index('float ac[5]')) - 2}"

    # ---- main.cu: code defaults (NOT the help text), FFT size, hop ----
    t = read("main.cu")
    d = {}
    for name in ("default_center_frequency_audio", "default_center_frequency_iq", "search_step_in_hz", "search_width_in_hz"):
        d[name] = flt(re.search(name + r" = ([\d.]+)f;", t).group(1))
    for name in ("scan_depth", "analytic_method", "nbadsync_threshold"):
        d[name] = int(re.search(r"int " + name + r" = (\d+);", t).group(1))
    out["main_defaults"] = d
    cite["main_defaults"] = f"main.cu:{line_of(t, t.index('default_center_frequency_audio'))}-{line_of(t, t.index('int nbadsync_threshold'))}"

    # ---- main.cu: help text (showHelp) and the labels of the stderr parameter block, as printed ----
    body = t[t.index("void showHelp"):t.index("// clang-format on", t.index("void showHelp"))]
    help_lines = []
    for ln in body.splitlines():
        m = re.search(r'std::cout << (.*) << std::endl;', ln)
        if not m:
            continue
        parts = re.findall(r'"((?:[^"\\]|\\.)*)"|(prog)', m.group(1))
        help_lines.append("".join("{prog}" if p[1] else p[0] for p in parts))
    out["help_lines"] = help_lines
    cite["help_lines"] = f"main.cu:{line_of(t, t.index('void showHelp')) + 3}-{line_of(t, t.index('void showHelp')) + 12}"
    blk = t[t.index('std::cerr << "Actual parameters:"'):t.index("sm1.stop();")]
    out["stderr_block_labels"] = re.findall(r'<< "([A-Z][^"]*?)(?:: |:)?"', blk)
    cite["stderr_block_labels"] = f"main.cu:{line_of(t, t.index('Actual parameters:'))}-{line_of(t, t.index('sm1.stop();')) - 2}"

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
