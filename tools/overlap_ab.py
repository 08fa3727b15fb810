#!/usr/bin/env python3
"""Same-box A/B of the block schedules of msk144_set_overlap on the bench workload (1024 channels, deep options): mode 0 (serial:
one kernel at a time) against modes 1..3 (LDPC of block b on a second stream beside scan/softbits/index of block b+1), interleaved
rounds, result lists compared byte for byte.  Prints one JSON object.

    python tools/overlap_ab.py [--rounds 4] [--steps 20] [--modes 0,1,2,3] [--llr-block 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--modes", default="0,1,2,3")
    ap.add_argument("--channels", type=int, default=bench.CHANNELS_PER_GPU)
    ap.add_argument("--llr-block", type=int, default=0)
    a = ap.parse_args()
    modes = [int(m) for m in a.modes.split(",")]
    be = bench.HipBackend(0, 0, a.channels, 0, a.llr_block)
    for i in range(40):             # clock and caches warm before the first measurement
        be.step(i)
    be.fence()
    ms = {m: [] for m in modes}
    lists = {}
    for r in range(a.rounds):
        for m in modes:
            be.dec.set_overlap(m)
            for i in range(3):
                be.step(i)
            be.fence()
            t0 = time.perf_counter()
            for i in range(a.steps):
                be.step(3 + i)
            be.fence()
            ms[m].append((time.perf_counter() - t0) / a.steps * 1e3)
            res = be.results()
            lists.setdefault(m, res.tobytes())
            assert lists[m] == res.tobytes(), f"mode {m}: result list changed between rounds"
    same = all(lists[m] == lists[modes[0]] for m in modes)
    out = {"workload": f"{a.channels} channels, width 500 / step 1 / depth 6 / threshold 3, llr block {be.llr_block}", "steps_per_measurement": a.steps,
           "ms_per_step": {str(m): [round(x, 3) for x in ms[m]] for m in modes},
           "median_ms": {str(m): round(float(np.median(ms[m])), 3) for m in modes},
           "result_lists_identical_across_modes": same, "records": len(be.results())}
    base = out["median_ms"][str(modes[0])]
    out["vs_first_mode"] = {str(m): round(out["median_ms"][str(m)] / base - 1.0, 4) for m in modes}
    print(json.dumps(out))
    be.close()
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
