"""Python view of csrc/msk144_protocol.h (single source of truth: the header is parsed, not copied)."""
from __future__ import annotations

import os
import re

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "msk144_protocol.h")


def _parse():
    src = open(_HDR).read()

    def block(name):
        m = re.search(name + r"\s*(?:\[[^\]]*\])+\s*=\s*\{(.*?)\};", src, re.S)
        if not m:
            raise RuntimeError(f"{name} not found in {_HDR}")
        return m.group(1)

    def ints(name):
        return [int(x) for x in re.findall(r"-?\d+", block(name))]

    def floats(name):
        return [float(x.rstrip("f")) for x in re.findall(r"-?\d+\.\d+f?", block(name))]

    cb = ints("kCheckBits")
    assert len(cb) == 38 * 11
    pm = ints("kPatternMask")
    assert len(pm) == 8 * 6
    poly = int(re.search(r"kCrc13Poly\s*=\s*(0x[0-9A-Fa-f]+)", src).group(1), 16)
    return {
        "check_bits": [cb[i * 11:(i + 1) * 11] for i in range(38)],
        "pattern_mask": [pm[i * 6:(i + 1) * 6] for i in range(8)],
        "pattern_num_avg": ints("kPatternNumAvg"),
        "sync8": ints(r"kSync8"),
        "fir_idx": ints("kFirTapIndex"),
        "fir_val": floats("kFirTapValue"),
        "crc_poly": poly,
    }


_P = _parse()

CHECK_BITS = _P["check_bits"]
PATTERN_MASK = _P["pattern_mask"]
PATTERN_NUM_AVG = _P["pattern_num_avg"]
SYNC8 = _P["sync8"]
CRC13_POLY = _P["crc_poly"]
FIR_TAP_INDEX = _P["fir_idx"]
FIR_TAP_VALUE = _P["fir_val"]

FRAME_SAMPLES = 864
WINDOW_SAMPLES = 5184
HOP_SAMPLES = 2592
SAMPLE_RATE = 12000.0
SLOTS_PER_PATTERN = 8
SCAN_POSITIONS = 5376
SLICE_POSITIONS = 256
CODE_BITS = 128
MESSAGE_BITS = 77
ITEM_BYTES = 632


def grid(search_width: float, search_step: float):
    """(F, if1) exactly as msk_context.cuh:95-107 (float32 arithmetic)."""
    import numpy as np
    w = np.float32(search_width)
    s = np.float32(search_step)
    half = int(np.float32(np.float32(w / np.float32(2)) / s))
    return 2 * half + 1, float(np.float32(-1 * half) * s)


def clamp_scan_depth(d: int) -> int:
    return max(1, min(8, int(d)))
