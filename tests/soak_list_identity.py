#!/usr/bin/env python3
"""One-off soak: the production result list (default 128-channel blocks, gated softbits, periodic copies handed to the lower slot of their
group) against the retain-everything list (every candidate demodulated and decoded on its own, as the reference does) on many
1024-channel bench windows - byte for byte.  A handed-over copy reports its lower slot's nbadsync / iterations / hard errors /
payload; computed on its own (other float association of the same frame sums) those could differ only in a marginal case, which
would show here as a differing record.

    python tests/soak_list_identity.py [--ranks 6]   ->  one JSON line
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def iq_soak(n_seeds):
    """The LDPC-iteration-heavy stress workload (pings at -6..-2 dB: marginal softbits and BP decisions are likelier than in the audio bench
    windows): hand-over on against hand-over off on one blocked handle."""
    import parity
    from msk144cudecoder_amd import hipdecoder as hip
    from msk144cudecoder_amd import synth
    cfg = dict(center=0.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
    tot = dict(workload="configs[4]: 4096 int8 IQ channels per list", lists=0, records=0, copies_among_records=0, handed_over_slots=0, slots=0, differing_lists=0, differing_records=0)
    with hip.HipDecoder(read_mode=2, channels=4096, max_results=1 << 20, **cfg) as d:
        for seed in range(100, 100 + n_seeds):
            wins, _ = synth.iq_low_snr_batch(4096, seed)
            d.set_copy_handover(True)
            d.submit_iq(wins)
            d.decode()
            p = d.results().copy()
            handed = parity.handed_over_records(d, p)
            tot["handed_over_slots"] += d.copy_count()
            d.set_copy_handover(False)
            d.decode()
            f = d.results().copy()
            tot["lists"] += 1
            tot["records"] += len(f)
            tot["slots"] += 4096 * d.K
            tot["copies_among_records"] += int(handed.sum())
            if p.tobytes() != f.tobytes():
                tot["differing_lists"] += 1
                n = min(len(p), len(f))
                tot["differing_records"] += int(np.count_nonzero(p[:n] != f[:n])) + abs(len(p) - len(f))
            print(f"seed {seed}: {tot}", file=sys.stderr, flush=True)
    print(json.dumps(tot), flush=True)
    return 0 if tot["differing_lists"] == 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=6, help="input sets (bench.make_inputs(rank, 1024)), four windows each")
    ap.add_argument("--iq-seeds", type=int, default=0, help="instead: this many BASELINE configs[4] batches (4096 low-SNR IQ channels, synth.iq_low_snr_batch(4096, seed)), "
                                                              "production list against the SAME handle with the hand-over switched off (every slot computed on its own)")
    a = ap.parse_args()
    if a.iq_seeds > 0:
        return iq_soak(a.iq_seeds)
    import bench
    import parity
    from msk144cudecoder_amd import hipdecoder as hip
    deep = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
    tot = dict(lists=0, records=0, copies_among_records=0, handed_over_slots=0, slots=0, differing_lists=0, differing_records=0)
    with hip.HipDecoder(channels=1024, max_results=1 << 20, **deep) as prod, hip.HipDecoder(channels=1024, max_results=1 << 20, llr_block_channels=1024, **deep) as full:
        for rank in range(a.ranks):
            wins, _ = bench.make_inputs(rank, 1024)
            bench._INPUTS.clear()
            for t in range(wins.shape[0]):
                full.submit_audio(wins[t])
                full.decode()
                f = full.results().copy()
                prod.submit_audio(wins[t])
                prod.decode()
                p = prod.results().copy()
                tot["lists"] += 1
                tot["records"] += len(f)
                # records whose slot was handed over: not in the production handle's index list (never demodulated or decoded itself)
                handed = parity.handed_over_records(prod, p) if len(p) == len(f) else np.zeros(len(p), dtype=bool)
                tot["copies_among_records"] += int(handed.sum())
                tot["handed_over_slots"] += prod.copy_count()
                tot["slots"] += 1024 * prod.K
                if p.tobytes() != f.tobytes():
                    tot["differing_lists"] += 1
                    n = min(len(p), len(f))
                    tot["differing_records"] += int(np.count_nonzero(p[:n] != f[:n])) + abs(len(p) - len(f))
            print(f"rank {rank}: {tot}", file=sys.stderr, flush=True)
    print(json.dumps(tot), flush=True)
    return 0 if tot["differing_lists"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
