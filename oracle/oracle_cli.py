#!/usr/bin/env python3
"""Oracle-driven stream decoder - TEST INFRASTRUCTURE (the CPU stand-in for BASELINE configs[0]).

Same stdin -> stdout contract as the reference program, computed entirely on the CPU: every window goes
through the oracle (oracle/msk144_oracle.cpp), the accepted candidates through the host library's
post-processing (the same C++ the HIP program links: msk144cudecoder_amd/host/).  The reference never bundled
a CPU MSK144 path (its WSJT-X submodule only supplied unpack77), so this is what "the CPU path" means here.

    python oracle/oracle_cli.py --search-width=100 --scan-depth=3 < samples.s16
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

HOST_SO = os.path.join(ROOT, "msk144cudecoder_amd", "libmsk144host.so")


class Accepted(C.Structure):
    _fields_ = [("f0", C.c_float), ("num_avg", C.c_int), ("nbadsync", C.c_int), ("pattern_idx", C.c_int), ("bits", C.c_ubyte * 77)]


def windows_of(stream: np.ndarray, read_mode: int):
    per = 1 if read_mode == 1 else 2
    win = 5184 * per
    hop = win // 2
    s = 0
    while s + win <= len(stream):
        yield stream[s:s + win]
        s += hop


def decode_stream(stream: np.ndarray, cfg: dict, read_mode: int = 1, analytic_method: int = 2, quirk: bool = True, threads: int = 8, mask_date: bool = True,
                  payloads: set = None):
    """All output lines (without the final 'Done') the reference program would print for `stream`.  `quirk` (default, as in the
    program): the reference's per-window text cache (main.cu:437-445, 497-504) - every accepted candidate of a window gets the text, or
    the unpack failure, of the window's first accepted candidate; False = msk144hipdecoder --strict-decode (each payload on its own).  `payloads`, when given, collects
    the 77-bit payload (as a '0'/'1' string) of every accepted candidate of every window - the text-independent artefact."""
    H = C.CDLL(HOST_SO)
    H.msk144host_table_new.restype = C.c_void_p
    H.msk144host_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    table = H.msk144host_table_new()
    o = orc.Oracle(threads=threads, **cfg)
    snr = orc.Snr()
    out = []
    for w in windows_of(stream, read_mode):
        cd = o.frontend_audio(w, analytic_method) if read_mode == 1 else o.frontend_iq(w)
        s = snr.process(cd)
        items, _ = o.decode_window(cd)
        acc = items[items["is_message_present"] == 1]
        if payloads is not None:
            payloads.update("".join(str(int(b)) for b in it["message"]) for it in acc)
        arr = (Accepted * max(len(acc), 1))()
        for a, it in zip(arr, acc):
            a.f0, a.num_avg, a.nbadsync, a.pattern_idx = float(it["f0"]), int(it["num_avg"]), int(it["nbadsync"]), int(it["pattern_idx"])
            a.bits[:] = [int(b) for b in it["message"]]
        buf = C.create_string_buffer(65536)
        n = H.msk144host_postprocess(table, arr, len(acc), s, 1 if quirk else 0, buf, len(buf))
        if n:
            out += buf.value.decode().split("\n")
    if mask_date:
        out = [re.sub(r"date=\d{14}", "date=X", l) for l in out]
    return out


def printed_payloads(stream: np.ndarray, cfg: dict, read_mode: int = 1, analytic_method: int = 2, threads: int = 8, quirk: bool = True) -> set:
    """The 77-bit payloads `msk144hipdecoder --print-bits` would print for `stream`: accepted payloads whose own text (decoded with
    a fresh hash table, as the program does for that option) is the text of an output line.  A payload the text layer rejects is
    accepted by the decoder but never printed; in the reference mode (`quirk`) neither is one whose window started with another text."""
    acc = set()
    lines = decode_stream(stream, cfg, read_mode, analytic_method, quirk=quirk, threads=threads, payloads=acc)
    texts = set(re.findall(r"msg='(.*)'; $", "\n".join(lines), flags=re.M))
    H = C.CDLL(HOST_SO)
    H.msk144host_table_new.restype = C.c_void_p
    H.msk144host_table_free.argtypes = [C.c_void_p]
    H.msk144host_decode_message.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    out = set()
    for bits in acc:
        table = H.msk144host_table_new()
        buf = C.create_string_buffer(64)
        ok = H.msk144host_decode_message(table, bytes(int(b) for b in bits), buf)
        H.msk144host_table_free(table)
        if ok and buf.value.decode() in texts:
            out.add(bits)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--center-frequency", type=float, default=None)
    ap.add_argument("--search-step", type=float, default=2.0)
    ap.add_argument("--search-width", type=float, default=200.0)
    ap.add_argument("--scan-depth", type=int, default=4)
    ap.add_argument("--read-mode", type=int, default=1)
    ap.add_argument("--analytic-method", type=int, default=2)
    ap.add_argument("--nbadsync-threshold", type=int, default=1)
    ap.add_argument("--reference-decode-cache", action="store_true", help="accepted for compatibility: the reference's cache behaviour is the default")
    ap.add_argument("--strict-decode", action="store_true", help="unpack every distinct payload of a window on its own")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    center = a.center_frequency if a.center_frequency is not None else (1500.0 if a.read_mode == 1 else 0.0)
    raw = sys.stdin.buffer.read()
    stream = np.frombuffer(raw, dtype=np.int16 if a.read_mode == 1 else np.int8)
    cfg = dict(center=center, width=a.search_width, step=a.search_step, depth=a.scan_depth, nbadsync_threshold=a.nbadsync_threshold)
    for line in decode_stream(stream, cfg, a.read_mode, a.analytic_method, quirk=not a.strict_decode, threads=a.threads, mask_date=False):
        print(line)
    print("Done")


if __name__ == "__main__":
    main()
