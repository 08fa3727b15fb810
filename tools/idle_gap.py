#!/usr/bin/env python3
"""Why do the product program's kernels run ~10 % slower than bench.py's (scan 11.35 vs 9.21 ms)?

The stream program decodes one 1024-stream hop every 216 ms: ~45 ms of kernels, then ~170 ms with nothing on the GPU.  bench.py runs
its steps back to back.  This tool runs the bench step (1024 channels, deep configuration, inputs resident in HBM) in both rhythms on
ONE box and records, per rhythm, the per-stage device times (HIP events) and the shader clock a one-wave probe reads BESIDE the
running step (msk144_clock_probe: s_memtime / s_memrealtime) at several offsets into the step.

    python tools/idle_gap.py [--steps 30] [--idle-ms 170] [--out gpurun_out/idle_gap.json]

Rhythms: "b2b" (back to back, after 2 s of warm-up), "idle" (synchronise, sleep idle-ms, one step, ...), "idle+warm" (the same, but a
front-end launch of the same windows is issued every warm-every-ms during the pause: the cheapest work the loop already has).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import bench  # noqa: E402
from msk144cudecoder_amd import hipdecoder as hd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--idle-ms", type=float, default=170.0)
    ap.add_argument("--warm-every-ms", type=float, default=10.0)
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--probe-us", type=int, default=1500)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    wins, _ = bench.make_inputs(0, a.channels)
    dev = torch.from_numpy(wins).cuda()
    out = {"channels": a.channels, "idle_ms": a.idle_ms, "steps": a.steps, "probe_us": a.probe_us, "rhythms": {}}
    with hd.HipDecoder(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3, channels=a.channels, max_results=1 << 20) as d:
        def step(i):
            d.submit_audio_device(dev[i % dev.shape[0]].data_ptr())
            d.decode()

        def probes(n):
            return [round(d.clock_probe(a.probe_us), 1) for _ in range(n)]

        # warm-up: >= 2 s of back-to-back steps (clock settles), unprofiled
        t0 = time.perf_counter()
        i = 0
        while time.perf_counter() - t0 < 2.0:
            step(i)
            i += 1
            if i % 8 == 0:
                d.synchronize()
        d.synchronize()
        d.set_profiling(True)

        def record(name, body):
            d.stage_times(reset=True)
            clocks, walls = [], []
            for k in range(a.steps):
                w0 = time.perf_counter()
                clocks.append(body(k))
                d.synchronize()
                walls.append((time.perf_counter() - w0) * 1e3)
            st = d.stage_times(reset=True)
            kernels = sum(st[n][0] for n in ("frontend", "scan", "softbits", "index", "ldpc", "collect"))
            rows = [c for c in clocks if c]
            out["rhythms"][name] = {
                "stage_ms": {n: round(st[n][0], 4) for n in hd.T_NAMES}, "kernels_ms": round(kernels, 3),
                "clock_mhz_by_probe_index": [round(sum(r[j] for r in rows) / len(rows), 1) for j in range(len(rows[0]))] if rows else None,
                "clock_mhz_first_step": rows[0] if rows else None, "clock_mhz_last_step": rows[-1] if rows else None,
                "wall_ms_mean": round(sum(walls) / len(walls), 3)}
            print(name, json.dumps(out["rhythms"][name]), flush=True)

        n_probe = 12   # 12 x 1.5 ms + launch gaps: the first ~25 ms of the step (front end, scan, first softbits/LDPC blocks)

        def b2b(k):
            step(k)
            return probes(n_probe) if k % 5 == 0 else None

        def idle(k):
            time.sleep(a.idle_ms * 1e-3)
            step(k)
            return probes(n_probe)

        def idle_warm(k):
            t_end = time.perf_counter() + a.idle_ms * 1e-3
            while time.perf_counter() < t_end:
                d.submit_audio_device(dev[0].data_ptr())          # front end only: 0.2 ms of GPU work
                time.sleep(a.warm_every_ms * 1e-3)
            step(k)
            return probes(n_probe)

        import threading

        def idle_sleeper(k):
            # one wave asleep on one CU for the whole pause (the clock probe's own kernel, s_sleep + a 100 MHz counter read per round):
            # does a BUSY queue hold the clock, or does the governor look at activity?
            th = threading.Thread(target=lambda: [d.clock_probe(85000) for _ in range(max(1, int(a.idle_ms // 85)))])
            th.start()
            time.sleep(a.idle_ms * 1e-3)
            th.join()
            step(k)
            return probes(n_probe)

        record("b2b", b2b)
        record("idle", idle)
        record("b2b_again", b2b)
        record("idle+warm", idle_warm)
        record("idle+sleeper", idle_sleeper)
        for ms in (1.0, 2.0, 5.0, 10.0, 20.0, 50.0, 100.0, 400.0):
            a.idle_ms = ms
            record(f"idle_{int(ms)}ms", idle)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
