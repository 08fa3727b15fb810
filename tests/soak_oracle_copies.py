#!/usr/bin/env python3
"""CPU soak: how often does a COPY decide differently from the lower slot it would be handed to?  Measured on the oracle, which - like the
reference (softbits_kernel.cuh:56-83, ldpc_kernel.cuh:100-249) - demodulates and decodes EVERY slot on its own.

Two slots of one (frequency, pattern) group fold the same frames when their scan positions are congruent modulo the 5184-sample ring (the
scan walks 5376 positions: a "ring-wrap twin", the same computation bit for bit) or, for masks 111111 / 100100, modulo the period 864 / 2592
(a "periodic copy": the same frames added in another order, LLRs equal up to float association).  The HIP path in blocked staging computes
the LOWEST congruent slot of a group and lets the others report its nbadsync / iterations / hard errors / payload (csrc/softbits.hip,
index.hip).  That is a deviation from the reference exactly where a copy's own computation would have decided otherwise; this
script counts those cases over many windows of the workloads the benchmarks and tests use:

    audio   bench-style: AWGN sigma 1000 LSB, half the windows with one ping of 2..7 frames at -4 / 0 / +6 dB        (configs[2] style)
    iq      S3: complex AWGN sigma 20 LSB per rail, one ping of 3..6 frames at -6..-2 dB, centre 0 Hz               (configs[4] style)
    depth   6 (mask 111111 periodic) and 8 (mask 100100 too), nbadsync thresholds 0..5, search widths 40..160 Hz at step 1

    python tests/soak_oracle_copies.py [--windows 2000] [--threads 8] [--seed 1]   ->  one JSON line (profiles/r06_oracle_copy_soak.json)

tests/test_oracle_copies.py runs three windows of the same comparison in the CPU suite.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PERIOD = {5: 864, 6: 2592}   # pattern_idx -> period of the folded correlation (masks 111111, 100100)
RING = 5184

COUNT_KEYS = ("pairs", "nbadsync_differs", "gate_differs", "both_gated", "accept_differs", "both_accepted", "iterations_differ", "hard_errors_differ",
              "payload_differs", "reported_record_differs")


def copy_pairs(items):
    """(copy item, lower item, is_periodic) arrays per the rule of softbits_kernel<true, true>: a slot is a copy when a LOWER slot of its
    (frequency, pattern) group has the same residue - position modulo the ring and, for the periodic masks, modulo the period - and it is
    handed to the LOWEST such slot.  is_periodic: the two positions differ modulo the ring (same frames, other order of addition); otherwise
    the pair is a ring-wrap twin (the same computation)."""
    pos = items["pos"].astype(np.int64).reshape(-1, 8) % RING
    pat = items["pattern_idx"].astype(np.int64).reshape(-1, 8)[:, 0]
    per = np.full(len(pat), RING, dtype=np.int64)
    for p, n in PERIOD.items():
        per[pat == p] = n
    res = pos % per[:, None]
    copies, lowers, periodic = [], [], []
    for sl in range(1, 8):
        same = res[:, :sl] == res[:, sl:sl + 1]                      # [groups][sl]
        has = same.any(axis=1)
        first = same.argmax(axis=1)
        g = np.nonzero(has)[0]
        copies.append(g * 8 + sl)
        lowers.append(g * 8 + first[g])
        periodic.append(pos[g, sl] != pos[g, first[g]])
    return np.concatenate(copies), np.concatenate(lowers), np.concatenate(periodic)


def compare_pairs(items, copies, lowers, threshold):
    """Counts over the given pairs: what the copy's OWN computation gives against its lower slot's."""
    a, b = items[copies], items[lowers]
    out = dict.fromkeys(COUNT_KEYS, 0)
    out["pairs"] = int(len(a))
    nb_diff = a["nbadsync"] != b["nbadsync"]
    ga, gb = a["nbadsync"] <= threshold, b["nbadsync"] <= threshold
    out["nbadsync_differs"] = int(nb_diff.sum())
    out["gate_differs"] = int((ga != gb).sum())
    both = ga & gb
    out["both_gated"] = int(both.sum())
    acc_a = ga & (a["is_message_present"] == 1)
    acc_b = gb & (b["is_message_present"] == 1)
    out["accept_differs"] = int((both & (acc_a != acc_b)).sum())
    ba = acc_a & acc_b
    out["both_accepted"] = int(ba.sum())
    it = ba & (a["ldpc_num_iterations"] != b["ldpc_num_iterations"])
    nh = ba & (a["ldpc_num_hard_errors"] != b["ldpc_num_hard_errors"])
    pl = ba & (a["message"] != b["message"]).any(axis=1)
    out["iterations_differ"], out["hard_errors_differ"], out["payload_differs"] = int(it.sum()), int(nh.sum()), int(pl.sum())
    # what a user of the result list would see: the copy slot's record exists or not, and its nbadsync / iterations / hard errors / payload
    out["reported_record_differs"] = int(((acc_a != acc_b) | (ba & (nb_diff | it | nh | pl))).sum())
    d = np.abs(a["softbits_wo_sync"].astype(np.float64) - b["softbits_wo_sync"]) / np.maximum(1.0, np.abs(b["softbits_wo_sync"]))
    ok = ~nb_diff & np.isfinite(d).all(axis=1)
    out["llr_max_rel"] = float(d[ok].max()) if ok.any() else 0.0
    return out


def draw_marginal_window(rng, i):
    """--mode marginal: every window carries ONE ping that fills the window (6 frames from a start inside the first frame, so that the
    six-frame mask 111111 - the pattern whose slots are mostly periodic copies - is the one that decodes it) at an SNR around the
    threshold of the averaged decode (-10 .. -4 dB per frame): accepted copies with marginal softbits and BP runs that need many
    iterations are as likely as this workload family can make them."""
    from msk144cudecoder_amd import synth
    kind = "audio" if i % 2 == 0 else "iq"
    width = float((40.0, 80.0)[int(rng.integers(0, 2))])
    center = 1500.0 if kind == "audio" else 0.0
    cfg = dict(center=center, width=width, step=1.0, depth=6 if i % 4 < 3 else 8, nbadsync_threshold=int(rng.integers(3, 6)))
    ping = synth.Ping(synth.random_message(rng), int(rng.integers(0, 864)), 6, center + float(rng.uniform(-0.4, 0.4)) * width, float(rng.uniform(-10.0, -4.0)),
                      float(rng.uniform(0, 6.28)))
    raw = synth.synth_audio(5184, [ping], 1000.0, rng) if kind == "audio" else synth.synth_iq(5184, [ping], 20.0, rng)
    return kind, cfg, raw, 1


def draw_window(rng, i):
    """(kind, oracle config, analytic-signal maker) of window i: the workload mix of the module docstring."""
    from msk144cudecoder_amd import synth
    kind = ("audio", "iq", "audio", "iq", "audio")[i % 5]
    depth = 8 if i % 3 == 2 else 6
    thr = 3 if i % 2 == 0 else int(rng.integers(0, 6))
    width = float((40.0, 80.0, 120.0, 160.0)[int(rng.integers(0, 4))])
    pings = []
    if kind == "audio":
        center = 1500.0
        if rng.random() < 0.5:
            pings = [synth.Ping(synth.random_message(rng), int(rng.integers(0, 3000)), int(rng.integers(2, 8)), center + float(rng.uniform(-0.45, 0.45)) * width,
                                float((-4.0, 0.0, 6.0)[int(rng.integers(0, 3))]), float(rng.uniform(0, 6.28)))]
        raw = synth.synth_audio(5184, pings, 1000.0, rng)
    else:
        center = 0.0
        pings = [synth.Ping(synth.random_message(rng), int(rng.integers(0, 1500)), int(rng.integers(3, 7)), float(rng.uniform(-0.45, 0.45)) * width,
                            float(rng.uniform(-6, -2)), float(rng.uniform(0, 6.28)))]
        raw = synth.synth_iq(5184, pings, 20.0, rng)
    cfg = dict(center=center, width=width, step=1.0, depth=depth, nbadsync_threshold=thr)
    return kind, cfg, raw, len(pings)


def run(windows: int, threads: int, seed: int, progress=None, mode: str = "mix", variant: str = ""):
    from oracle import oracle as orc
    orc.build()
    library = orc.fma_lib(variant) if variant else None     # one of the builds that bracket the reference's own FMA contraction / math library (oracle/Makefile `fma`)
    rng = np.random.default_rng(770000 + seed)
    tot = {"ring_wrap_twins": dict.fromkeys(COUNT_KEYS, 0), "periodic_copies": dict.fromkeys(COUNT_KEYS, 0)}
    tot["ring_wrap_twins"]["llr_max_rel"] = tot["periodic_copies"]["llr_max_rel"] = 0.0
    mix = {}
    slots = 0
    t0 = time.perf_counter()
    oracles = {}
    for i in range(windows):
        kind, cfg, raw, n_pings = (draw_marginal_window if mode == "marginal" else draw_window)(rng, i)
        key = tuple(sorted(cfg.items()))
        if key not in oracles:
            oracles[key] = orc.Oracle(threads=threads, library=library, **cfg)
        o = oracles[key]
        cd = o.frontend_audio(raw, 2) if kind == "audio" else o.frontend_iq(raw)
        items, _ = o.decode_window(cd)
        slots += len(items)
        c, l, per = copy_pairs(items)
        for name, sel in (("ring_wrap_twins", ~per), ("periodic_copies", per)):
            r = compare_pairs(items, c[sel], l[sel], cfg["nbadsync_threshold"])
            for k in COUNT_KEYS:
                tot[name][k] += r[k]
            tot[name]["llr_max_rel"] = max(tot[name]["llr_max_rel"], r["llr_max_rel"])
        m = mix.setdefault(f"{kind}/depth{cfg['depth']}", dict(windows=0, with_ping=0))
        m["windows"] += 1
        m["with_ping"] += int(n_pings > 0)
        if progress and (i + 1) % progress == 0:
            print(f"{i + 1}/{windows} windows, {time.perf_counter() - t0:.0f} s: periodic {tot['periodic_copies']}", file=sys.stderr, flush=True)
    out = {"windows": windows, "seed": seed, "mode": mode, "oracle_build": variant or "parity (-ffp-contract=off)", "slots": slots, "seconds": round(time.perf_counter() - t0, 1), "mix": mix, **tot}
    for name in ("ring_wrap_twins", "periodic_copies"):
        t = tot[name]
        n = t["pairs"]
        # rule of three: an event seen 0 times in n trials has a rate below 3 / n at 95 % confidence
        t["reported_record_differs_rate"] = t["reported_record_differs"] / n if n else None
        t["reported_record_differs_rate_upper_95"] = (3.0 / n if t["reported_record_differs"] == 0 else
                                                      (t["reported_record_differs"] + 2.0 * np.sqrt(t["reported_record_differs"]) + 2.0) / n) if n else None
        ng = t["both_gated"]
        t["decision_differs_per_gated_pair"] = (t["accept_differs"] + t["iterations_differ"] + t["hard_errors_differ"] + t["payload_differs"]) / ng if ng else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=2000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", choices=("mix", "marginal"), default="mix", help="mix: the workload mix of the docstring; marginal: window-filling pings around the decode threshold")
    ap.add_argument("--variant", default="", choices=("", "contract-fast", "forced-fma", "cuda-libm", "cuda-like"),
                    help="run on one of the oracle builds that bracket the reference's FMA contraction and math library instead of the parity build")
    a = ap.parse_args()
    print(json.dumps(run(a.windows, a.threads, a.seed, progress=100, mode=a.mode, variant=a.variant)), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
