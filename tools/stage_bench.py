"""Developer tool: time ONE stage of the staged pipeline in isolation on the bench workload (1024 channels, deep config).
The full pipeline runs once (so the stage's inputs are real), then only the chosen stage is re-launched.

    python tools/stage_bench.py --stage ldpc [--lib path/to/libmsk144hip.so] [--channels 1024] [--reps 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import bench  # noqa: E402
from msk144cudecoder_amd import hipdecoder as hd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stage", default="ldpc", choices=["scan", "softbits", "index", "ldpc", "collect"])
ap.add_argument("--lib", default=None)
ap.add_argument("--channels", type=int, default=1024)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--threshold", type=int, default=3)
args = ap.parse_args()
if args.lib:
    hd._lib = hd.load_library(os.path.abspath(args.lib))
bits = {"scan": hd.STAGE_SCAN, "softbits": hd.STAGE_SOFTBITS, "index": hd.STAGE_INDEX, "ldpc": hd.STAGE_LDPC, "collect": hd.STAGE_COLLECT}
wins, _ = bench.make_inputs(0, args.channels)
with hd.HipDecoder(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=args.threshold, channels=args.channels, max_results=1 << 20, llr_block_channels=args.channels) as d:
    d.submit_audio(wins[1])
    d.decode()
    n0 = d.result_count()
    for _ in range(2):
        d.decode(bits[args.stage])
    d.synchronize()
    d.set_profiling(True)
    d.stage_times(reset=True)
    for _ in range(args.reps):
        d.decode(bits[args.stage])
    d.synchronize()
    ms = d.stage_times()[args.stage][0]
    d.decode(hd.STAGE_COLLECT)
    print(f"{args.stage}: {ms:.3f} ms per launch ({args.channels} channels, lib={args.lib or 'default'}), decodes {n0} -> {d.result_count()}")
