#!/usr/bin/env python3
"""EXPERIMENT (round 6): the 52 launches of a 1024-channel step replayed from a captured HIP graph against
launching them one by one - same handle, same inputs, interleaved rounds on one box, no profiling events in either mode.

    python tools/graph_ab.py [--rounds 5] [--steps 40]   ->  one JSON line
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import torch
    import bench
    be = bench.HipBackend(0, 0, 1024, 0)
    L = be.dec.L
    L.msk144_set_graph_replay.argtypes = [C.c_void_p, C.c_int32]
    want = None
    out = {"plain_ms": [], "graph_ms": []}
    for r in range(a.rounds + 1):
        for mode in ("plain", "graph"):
            assert L.msk144_set_graph_replay(be.dec.h, 1 if mode == "graph" else 0) == 0
            for i in range(3):
                be.step(i)
            be.fence()
            t0 = time.perf_counter()
            for i in range(a.steps):
                be.step(i)
            be.fence()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            got = be.results().tobytes()
            want = want or got
            assert got == want, mode                      # same last window, same list
            if r > 0:                                     # round 0 warms the clock up
                out[mode + "_ms"].append(round(ms, 4))
    p, g = sorted(out["plain_ms"]), sorted(out["graph_ms"])
    out["plain_median_ms"], out["graph_median_ms"] = p[len(p) // 2], g[len(g) // 2]
    out["graph_vs_plain"] = out["graph_median_ms"] / out["plain_median_ms"] - 1.0
    print(json.dumps(out))
    be.close()


if __name__ == "__main__":
    main()
