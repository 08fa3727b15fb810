"""Pins the tables every component shares (HIP kernels, host library, oracle, synthesiser, numpy model) to the REFERENCE's
own constants: tests/golden/ref_constants.json is a snapshot extracted from /root/reference/src as text
(tests/golden/make_ref_constants.py, file:line of every item under "_cite").  CPU only."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from msk144cudecoder_amd import protocol as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")))
CSRC = os.path.join(ROOT, "msk144cudecoder_amd", "csrc")


def _src(name):
    return open(os.path.join(CSRC, name)).read()


def _header_int(name):
    m = re.search(r"constexpr int " + name + r"\s*=\s*([^;]+);", _src("msk144_protocol.h"))
    expr = m.group(1)
    for ident in sorted(set(re.findall(r"k[A-Z]\w+", expr)), key=len, reverse=True):
        expr = expr.replace(ident, str(_header_int(ident)))
    assert re.fullmatch(r"[\d\s+*/()-]+", expr), expr
    return int(eval(expr))  # noqa: S307 - digits and arithmetic only


@pytest.fixture(scope="module")
def hostlib():
    L = C.CDLL(os.path.join(ROOT, "msk144cudecoder_amd", "libmsk144host.so"))
    fp = C.POINTER(C.c_float)
    L.msk144host_sync_template.argtypes = [fp, fp, fp]
    L.msk144host_sync_template.restype = None
    L.msk144host_frequency_grid.argtypes = [C.c_float, C.c_float, C.c_float, fp, C.c_int]
    L.msk144host_fft_band_mask.argtypes = [fp, C.c_int]
    return L


def test_tanner_graph_is_the_references():
    """Our check-major kCheckBits, turned back into the reference's bit-major [bit][edge] = (slot row, check) map, must equal
    ldpc_context.cuh:10-139 entry for entry - including slot numbering and edge order, which fix the order of every sum and
    product in the BP decoder."""
    ref = REF["ldpc_reverse_map"]
    rows = P.CHECK_BITS
    assert len(rows) == REF["num_checks"] == 38
    derived = [[] for _ in range(128)]
    for c, row in enumerate(rows):                       # ascending check order = edge order k
        for slot, n in enumerate(row):
            if n >= 0:
                derived[n].append([slot, c])
    assert derived == ref
    assert [c for c, row in enumerate(rows) if row[10] >= 0] == REF["is_full_row"]
    assert all(len([n for n in row if n >= 0]) in (10, 11) for row in rows)


def test_protocol_header_constants():
    c = REF["common"]
    assert P.SYNC8 == REF["sync8"]
    assert P.PATTERN_MASK == REF["patterns"]
    assert P.PATTERN_NUM_AVG == [sum(p) for p in REF["patterns"]]
    assert P.CRC13_POLY == REF["crc13_poly"] == 0x15D7
    assert _header_int("kFrameSamples") == c["Num864"] and _header_int("kWindowSamples") == c["Num6x864"]
    assert _header_int("kSyncTaps") == c["Num42"] and _header_int("kSecondSyncSample") == c["SecondSyncBase"]
    assert _header_int("kSecondSyncBit") == c["SecondHardbitsSyncBase"]
    assert _header_int("kSoftBits") == c["NumberOfSoftBits"] and _header_int("kCodeBits") == c["NumberOfSoftBitsWithoutSync"]
    assert _header_int("kMessageBits") == c["NumberOfMessageBits"] and _header_int("kLdpcIterations") == c["NumberOfLDPCIterations"]
    assert _header_int("kSlicePositions") == c["NumScanThreads"] and _header_int("kSlotsPerPattern") == c["NumCandidatesPerPattern"]
    assert _header_int("kScanDepthMax") == c["ScanDepthMax"] and _header_int("kPatternBits") == c["FixedNumBitsInPattern"]
    assert _header_int("kMaxHardErrors") == REF["max_hard_errors_exclusive"]
    assert float(re.search(r"kSampleRate = ([\d.]+)f", _src("msk144_protocol.h")).group(1)) == c["SampleRate"]
    # FIR: same 13 non-zero taps at the same positions (h3 and h13 are absent in the reference as well)
    assert {str(i): v for i, v in zip(P.FIR_TAP_INDEX, P.FIR_TAP_VALUE)} == REF["fir_taps"]
    assert float(re.search(r"kSin45 = ([\d.]+)f", _src("msk144_protocol.h")).group(1)) == REF["sc45"]
    # code defaults (main.cu:124-133), as msk144_default_params sets them
    api = _src("msk144_api.cpp")
    d = REF["main_defaults"]
    for field, key in (("center_hz", "default_center_frequency_audio"), ("width_hz", "search_width_in_hz"), ("step_hz", "search_step_in_hz")):
        assert float(re.search(r"p->" + field + r" = ([\d.]+)f;", api).group(1)) == d[key]
    for field, key in (("scan_depth", "scan_depth"), ("nbadsync_threshold", "nbadsync_threshold"), ("analytic_method", "analytic_method")):
        assert int(re.search(r"p->" + field + r" = (\d+);", api).group(1)) == d[key]


def test_frontend_rotation_tables_in_kernel_source():
    """frontend.hip's fs/8 rotation tables are the reference's 8 literal phasors, in its order (analytic2.cuh:15-41,56-82)."""
    src = _src("frontend.hip")

    def table(name):
        m = re.search(name + r"\s*=\s*\{(.*?)\};", src, re.S)
        assert m, name
        vals = []
        for a, b in re.findall(r"\{\s*([^,{}]+),\s*([^,{}]+)\}", m.group(1)):
            vals.append([_lit(a), _lit(b)])
        return vals

    def _lit(s):
        s = s.strip()
        sign = -1.0 if s.startswith("-") else 1.0
        s = s.lstrip("-").strip()
        return sign * (REF["sc45"] if s == "kSin45" else float(s.rstrip("f")))

    assert table(r"w_left\[8\]") == REF["shift_left"]
    assert table(r"w_right\[8\]") == REF["shift_right"]


def test_sync_template_grid_and_mask_tables(hostlib):
    """The tables libmsk144hip.so hands to its kernels (csrc/msk144_tables.h, exported through the host test library)."""
    re_, im_, pp = (np.zeros(n, dtype=np.float32) for n in (42, 42, 12))
    fp = C.POINTER(C.c_float)
    hostlib.msk144host_sync_template(re_.ctypes.data_as(fp), im_.ctypes.data_as(fp), pp.ctypes.data_as(fp))
    assert REF["pp_len"] == 12 and REF["pp_angle_div"] == 12.0
    angle = (np.arange(12, dtype=np.float32) * np.float32(np.pi)) / np.float32(REF["pp_angle_div"])   # float32, as the reference
    want_pp = np.sin(angle.astype(np.float64))
    assert np.abs(pp - want_pp).max() < 6e-8
    s8 = 2 * np.array(REF["sync8"]) - 1
    want = {"cbi": np.full(42, np.nan, dtype=np.float32), "cbq": np.full(42, np.nan, dtype=np.float32)}
    for seg in REF["template_segments"]:                 # the reference's eight fill loops, as data
        for i in range(seg["count"]):
            want[seg["array"]][seg["base"] + i] = pp[seg["pp_offset"] + i] * np.float32(s8[seg["s8_index"]])
    assert not np.isnan(want["cbi"]).any() and not np.isnan(want["cbq"]).any()
    assert np.array_equal(re_, want["cbi"]) and np.array_equal(im_, want["cbq"])          # bit for bit given pp

    # frequency grid: F = 2*int((w/2)/s)+1, f_b = center + if1 + b*step in float32 (msk_context.cuh:95-107,135)
    for center, width, step in ((1500.0, 500.0, 1.0), (1500.0, 200.0, 2.0), (0.0, 100.0, 0.25), (1500.0, 7.0, 3.0)):
        out = np.zeros(4096, dtype=np.float32)
        n = hostlib.msk144host_frequency_grid(center, width, step, out.ctypes.data_as(fp), len(out))
        half = int(np.float32(np.float32(np.float32(width) / np.float32(2)) / np.float32(step)))
        assert n == 2 * half + 1
        if1 = np.float32(-1 * half) * np.float32(step)
        want_f = np.array([np.float32(np.float32(np.float32(center) + if1) + np.float32(b) * np.float32(step)) for b in range(n)], dtype=np.float32)
        assert np.array_equal(out[:n], want_f)

    m = REF["fft_mask"]
    out = np.zeros(8192, dtype=np.float32)
    n = hostlib.msk144host_fft_band_mask(out.ctypes.data_as(fp), len(out))
    assert n == 4096
    f = np.abs(np.arange(4096) * (m["sample_rate"] / 8192.0) - m["center_hz"])
    lo, hi = (1 - m["beta"]) * m["t_inv"] / 2, (1 + m["beta"]) * m["t_inv"] / 2
    want_m = np.where(f <= lo, 1.0, np.where(f <= hi, 0.5 * (1 + np.cos(np.pi / m["t_inv"] / m["beta"] * (f - lo))), 0.0))
    assert np.abs(out[:n] - want_m).max() < 2e-6


def _ref_platanh(x):
    p = REF["platanh"]
    x = np.asarray(x, dtype=np.float32)
    z = np.abs(x)
    sgn = np.where(x < 0, np.float32(-1), np.float32(1))
    b = [np.float32(v) for v in p["breakpoints"]]
    (c2, d2), (c3, d3), (c4, d4) = [(np.float32(a), np.float32(d)) for a, d in p["pieces_offset_divisor"]]
    return np.where(z <= b[0], x / np.float32(p["first_piece_divisor"]),
                    np.where(z <= b[1], sgn * (z - c2) / d2,
                             np.where(z <= b[2], sgn * (z - c3) / d3,
                                      np.where(z <= b[3], sgn * (z - c4) / d4, sgn * np.float32(p["saturation"]))))).astype(np.float32)


def _hip_two_platanh_scaled(x, consts):
    """numpy float32 restatement of ldpc.hip's two_platanh_scaled (both the fast path and the full path compute this)."""
    f = np.float32
    L2E = f(1.4426950408889634)
    x = np.asarray(x, dtype=f)
    z = np.abs(x)
    c = np.where(z > f(consts["b1"]), f(consts["c3"]), f(consts["c2"]))
    r = np.where(z > f(consts["b1"]), f(L2E * f(2.0) / f(consts["d3"])), f(L2E * f(2.0) / f(consts["d2"])))
    c = np.where(z > f(consts["b2"]), f(consts["c4"]), c)
    r = np.where(z > f(consts["b2"]), f(L2E * f(2.0) / f(consts["d4"])), r)
    v = np.maximum(z * f(L2E * f(2.0) / f(consts["d1"])), (z - c).astype(f) * r).astype(f)
    v = np.where(z > f(consts["b3"]), f(L2E * f(consts["sat2"])), v)
    return np.copysign(v, x).astype(f)


def test_platanh_constants_and_branch_free_first_breakpoint():
    """ldpc.hip keeps the reference's breakpoints, offsets and divisors (ldpc_kernel.cuh:65-93); the 0.664 breakpoint is
    folded into a max() - verified here to select the reference's piece for EVERY float around it - and the result equals
    log2(e) * 2 * platanh(x) to float rounding."""
    src = _src("ldpc.hip")
    body = src[src.index("two_platanh_scaled_full(float x)"):src.index("// Same function, priced")]
    lits = [float(v) for v in re.findall(r"(?<![\w.])(\d+\.\d+)f", body)]
    p = REF["platanh"]
    (c2, d2), (c3, d3), (c4, d4) = p["pieces_offset_divisor"]
    for v in (c2, d2, c3, d3, c4, d4, p["first_piece_divisor"], p["breakpoints"][1], p["breakpoints"][2], p["breakpoints"][3], 2 * p["saturation"]):
        assert v in lits, (v, lits)
    consts = dict(b1=p["breakpoints"][1], b2=p["breakpoints"][2], b3=p["breakpoints"][3], c2=c2, d2=d2, c3=c3, d3=d3, c4=c4, d4=d4,
                  d1=p["first_piece_divisor"], sat2=2 * p["saturation"])
    # dense float neighbourhoods of all four breakpoints (every representable float within +-4096 ulp), plus a sweep
    xs = [np.linspace(0.0, 1.0, 200001, dtype=np.float32), np.float32([0.0, 1.0, 0.99999994])]
    for b in p["breakpoints"]:
        i0 = np.float32(b).view(np.uint32)
        xs.append((np.arange(int(i0) - 4096, int(i0) + 4097, dtype=np.uint32)).view(np.float32))
    x = np.concatenate(xs)
    x = np.concatenate([x, -x])
    want = np.float64(_ref_platanh(x)) * 2.0 * 1.4426950408889634
    got = np.float64(_hip_two_platanh_scaled(x, consts))
    assert np.all(np.abs(got - want) <= 3e-7 * np.maximum(np.abs(want), 1e-30)), np.abs(got / np.where(want == 0, 1, want) - 1).max()
    # the oracle's platanh is the reference's, literally
    orc = open(os.path.join(ROOT, "oracle", "msk144_oracle.cpp")).read()
    ob = orc[orc.index("float platanh(float x)"):]
    ob = ob[:ob.index("\n}\n")]
    assert [float(v) for v in re.findall(r"z <= ([\d.]+)f", ob)] == p["breakpoints"]
    assert [[float(a), float(b)] for a, b in re.findall(r"\(z - ([\d.]+)f\) / ([\d.]+)f", ob)] == p["pieces_offset_divisor"]
    assert float(re.search(r"return x / ([\d.]+)f", ob).group(1)) == p["first_piece_divisor"]


def test_softbits_normalisation_constants():
    sb = _src("softbits.hip")
    assert float(re.search(r"const float sigma = ([\d.]+)f;", sb).group(1)) == REF["softbits_sigma"]
    assert "(1.0f / 144.0f)" in sb and REF["softbits_mean_divisor"] == 144.0
