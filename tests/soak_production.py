#!/usr/bin/env python3
"""One-off soak of the PRODUCTION path (default 128-channel blocks, gated softbits - what bench.py times) against the oracle at
BASELINE configs[2] size: for every staged window of the bench inputs, the 1024-channel batch is decoded once, and N sampled channels are
compared with the oracle exactly as tests/test_gpu_full.py does (tests/parity.py: records == accepted candidates of the channel's
dump, dump vs decode_window stage by stage with verified near-ties only).  TEST INFRASTRUCTURE: uses oracle/ and tests/parity.py.

    python tests/soak_production.py [--channels-per-window 48]   ->  one JSON line
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels-per-window", type=int, default=48)
    ap.add_argument("--rank", type=int, default=0, help="input set: bench.make_inputs(rank, 1024) (0 = the bench's own)")
    ap.add_argument("--seed", type=int, default=99, help="seed of the channel sample")
    a = ap.parse_args()
    import bench
    import parity
    from msk144cudecoder_amd import hipdecoder as hip
    from oracle import oracle as orc
    deep = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
    wins, truth = bench.make_inputs(a.rank, 1024)
    o = orc.Oracle(threads=bench.usable_cores(), **deep)
    rng = np.random.default_rng(a.seed)
    tot = dict(input_set=a.rank, sample_seed=a.seed, windows=0, channels=0, records=0, scan_near_ties=0, nbadsync_marginal=0, bp_marginal=0)
    t0 = time.time()
    with hip.HipDecoder(channels=1024, max_results=1 << 20, **deep) as prod, hip.HipDecoder(channels=1, **deep) as single:
        for t in range(wins.shape[0]):
            prod.submit_audio(wins[t])
            prod.decode()
            records = prod.results().copy()
            decoded = sorted(set(int(c) for c in records["channel"]))
            quiet = [c for c in range(1024) if c not in truth]
            n = a.channels_per_window
            sample = list(rng.choice(decoded, size=min(n // 2, len(decoded)), replace=False)) + list(rng.choice(quiet, size=n - min(n // 2, len(decoded)), replace=False))
            dumps, cds = {}, {}
            for ch in sample:
                ch = int(ch)
                single.submit_audio(wins[t, ch])
                single.decode()
                dumps[ch] = single.dump_candidates(0)
                cds[ch] = o.frontend_audio(wins[t, ch], 2)
            rep = parity.compare_result_list_with_oracle(o, orc, records, cds, dumps)
            tot["windows"] += 1
            tot["channels"] += rep["channels"]
            tot["records"] += rep["decodes"]
            for v in rep["per_channel"].values():
                tot["scan_near_ties"] += v["scan_near_ties"]
                tot["nbadsync_marginal"] += v["nbadsync_marginal"] or 0
                tot["bp_marginal"] += v["bp_marginal"]
            print(f"window {t}: {rep['channels']} channels, {rep['decodes']} records compared ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    tot["candidates_compared"] = tot["channels"] * 24048
    tot["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(tot), flush=True)


if __name__ == "__main__":
    main()
