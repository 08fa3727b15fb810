#!/usr/bin/env python3
"""The product PROGRAM at BASELINE scale: N real-time streams through `msk144hipdecoder --inputs-file=...` on one MI355X.

Every stream is a FIFO fed at the real-time rate of the reference's input (12 ksps: the first 5184 samples, then 2592 samples
every 216 ms, main.cu:269-294); the decoder runs the deep configuration (--search-width=500 --search-step=1 --scan-depth=6
--nbadsync-threshold=3) and reports, per hop, where the host's wall time goes (--timing: ingest, window assembly, submit, wait
for GPU + D2H, post-processing, print; device-side H2D / kernels / D2H from HIP events) and how many stream hops were answered
later than the reference's 210 ms soft limit (main.cu:398-403).

    python tools/host_scale.py --streams 1024 --hops 20            # paced, prints one JSON line
    python tools/host_scale.py --streams 1024 --hops 20 --pace-ms 0  # as fast as the pipes deliver (throughput of the host loop)
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.environ.get("MSK144_DECODER_EXE") or os.path.join(ROOT, "msk144cudecoder_amd", "msk144hipdecoder")   # the override: CPU tests of the loop against tests/stub_hip
DEEP = ["--search-width=500", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=3"]


def make_streams(n_streams: int, n_hops: int, seed: int = 7, meta: dict = None):
    """int16 [n_streams][5184 + n_hops*2592]: AWGN sigma = 1000 LSB; every 4th stream carries one 0 dB ping (bench.py's recipe).
    `meta`, when given, receives {stream: (first sample, frames) of its ping}."""
    from msk144cudecoder_amd import synth
    n = 5184 + n_hops * 2592
    rng = np.random.default_rng(seed)
    x = rng.normal(0.0, 1000.0, size=(n_streams, n)).astype(np.float32)
    sent = {}
    amp = np.sqrt(2.0 * 1000.0 ** 2 * (2500.0 / 6000.0))
    for c in range(0, n_streams, 4):
        msg = synth.random_message(rng)
        frames = int(rng.integers(3, 7))
        start = int(rng.integers(0, n - frames * 864))
        freq = 1500.0 + float(rng.uniform(-240, 240))
        bb = synth.modulate_frame(synth.frame_bits(synth.encode_message(msg)))
        t = np.arange(start, start + frames * 864)
        x[c, start:start + frames * 864] += (amp * np.real(np.tile(bb, frames) * np.exp(1j * (2 * np.pi * freq * t / 12000.0 + rng.uniform(0, 6.28))))).astype(np.float32)
        sent[c] = "".join(str(int(b)) for b in msg)
        if meta is not None:
            meta[c] = (start, frames)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16), sent


def parse_timing(err: str):
    """{row: {mean_ms, max_ms}} and the device-side split from one --timing block of the program's stderr."""
    rows = {}
    for name, mean, worst in re.findall(r"timing: (.+?)\s+mean\s+([\d.]+) ms\s+max\s+([\d.]+) ms", err):
        rows[name.strip()] = {"mean_ms": float(mean), "max_ms": float(worst)}
    dev = None
    m = re.search(r"device per batch \(HIP events\): (.*) ms", err)
    if m:
        dev = {k.strip(): float(v) for k, v in re.findall(r"([A-Za-z0-9 ]+?)\s+([\d.]+)(?:\s{2}|$)", m.group(1))}
    return rows, dev


def run(n_streams: int, n_hops: int, pace_ms: float = 216.0, extra_args=(), feeders: int = 8, hop_timeout_ms: int = None, timeout_s: float = 180.0, devices: str = None,
        phase_spread_ms: float = 0.0, keep_lines=(), phase_per_stream: bool = False, phase_seed: int = 11, loop_hops: int = 0):
    """devices: value for --devices (e.g. "0,0" or "0,1,2,3"): one ingest+post thread pair per entry, per-device rows in the result.
    phase_spread_ms: hop clocks offset by a fixed random phase in [0, phase_spread_ms) - streams that are NOT phase-aligned (real
    receivers are not), instead of all hops falling due together; one phase per feeder thread, or (phase_per_stream) one per STREAM,
    drawn with phase_seed.  loop_hops > 0: only that many hops of signal are synthesised per stream and fed round and round, so that a
    run of minutes (n_hops in the hundreds) does not need gigabytes of samples; the decode statistics of such a run are not meaningful,
    its timing is."""
    import resource
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    if soft < n_streams + 256:
        resource.setrlimit(resource.RLIMIT_NOFILE, (min(hard, n_streams + 256) if hard != resource.RLIM_INFINITY else n_streams + 256, hard))
    base_hops = min(n_hops, loop_hops) if loop_hops > 0 else n_hops
    streams, sent = make_streams(n_streams, base_hops)
    tmp = tempfile.mkdtemp(prefix="msk144_fifos_")
    paths = [os.path.join(tmp, f"s{c:05d}.fifo") for c in range(n_streams)]
    for p in paths:
        os.mkfifo(p)
    lst = os.path.join(tmp, "inputs.txt")
    with open(lst, "w") as f:
        f.write("\n".join(paths) + "\n")
    cmd = [EXE] + DEEP + ["--print-bits", "--timing", f"--inputs-file={lst}"] + ([f"--hop-timeout-ms={hop_timeout_ms}"] if hop_timeout_ms is not None else []) + \
        ([f"--devices={devices}"] if devices else []) + list(extra_args)
    out_path, err_path = os.path.join(tmp, "stdout.txt"), os.path.join(tmp, "stderr.txt")
    with open(out_path, "wb") as fo, open(err_path, "wb") as fe:
        proc = subprocess.Popen(cmd, stdout=fo, stderr=fe)
        fds = [None] * n_streams
        errors = []

        def open_range(lo, hi):
            try:
                for c in range(lo, hi):
                    fds[c] = os.open(paths[c], os.O_WRONLY)          # returns once the decoder has opened its end
            except OSError as e:
                errors.append(e)

        per = -(-n_streams // feeders)
        ths = [threading.Thread(target=open_range, args=(k * per, min(n_streams, (k + 1) * per))) for k in range(feeders)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        if errors or any(fd is None for fd in fds):
            proc.kill()
            raise RuntimeError(f"could not open the FIFOs for writing: {errors[:1]}")
        raw = streams.view(np.uint8).reshape(n_streams, -1)
        t0 = time.monotonic()
        start = threading.Barrier(feeders)

        prng = np.random.default_rng(phase_seed)
        phases = prng.uniform(0.0, phase_spread_ms * 1e-3, size=feeders) if phase_spread_ms > 0 else np.zeros(feeders)
        stream_phase = prng.uniform(0.0, phase_spread_ms * 1e-3, size=n_streams) if (phase_spread_ms > 0 and phase_per_stream) else None

        def feed(lo, hi):
            try:
                start.wait()
                if stream_phase is not None:
                    # every stream on its own hop clock: the feeder walks its streams in phase order, hop after hop
                    order = sorted(range(lo, hi), key=lambda c: stream_phase[c])
                    for h in range(-1, n_hops):
                        hh = h % base_hops
                        a, b = (0, 5184 * 2) if h < 0 else (5184 * 2 + hh * 5184, 5184 * 2 + (hh + 1) * 5184)
                        for c in order:
                            dt = t0 + 0.5 + float(stream_phase[c]) + pace_ms * 1e-3 * (h + 1) - time.monotonic()
                            if dt > 0 and pace_ms > 0:
                                time.sleep(dt)
                            os.write(fds[c], raw[c, a:b].tobytes())
                    return
                phase = float(phases[lo // per])
                if phase > 0:
                    time.sleep(phase)
                for h in range(-1, n_hops):
                    if h >= 0 and pace_ms > 0:
                        time.sleep(max(0.0, t0 + 0.5 + phase + pace_ms * 1e-3 * (h + 1) - time.monotonic()))
                    hh = h % base_hops
                    a, b = (0, 5184 * 2) if h < 0 else (5184 * 2 + hh * 5184, 5184 * 2 + (hh + 1) * 5184)
                    for c in range(lo, hi):
                        os.write(fds[c], raw[c, a:b].tobytes())
            except OSError as e:
                errors.append(e)
            finally:
                for c in range(lo, hi):
                    os.close(fds[c])

        ths = [threading.Thread(target=feed, args=(k * per, min(n_streams, (k + 1) * per))) for k in range(feeders)]
        for t in ths:
            t.start()
        deadline = time.monotonic() + timeout_s + n_hops * pace_ms * 1e-3
        for t in ths:
            t.join(timeout=max(0.1, deadline - time.monotonic()))
        if any(t.is_alive() for t in ths):
            proc.kill()                                   # a decoder that stopped reading: unblock the feeders and report it
            errors.append(TimeoutError("decoder stopped reading its inputs"))
            for t in ths:
                t.join(timeout=10)
        try:
            rc = proc.wait(timeout=max(1.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            proc.kill()
            rc = proc.wait()
            errors.append(TimeoutError("decoder did not exit after its inputs ended"))
    wall = time.monotonic() - t0
    out = open(out_path, errors="replace").read()
    err = open(err_path, errors="replace").read()
    for p in paths + [lst, out_path, err_path]:
        os.unlink(p)
    os.rmdir(tmp)
    res = {"streams": n_streams, "hops_per_stream": n_hops + 1, "pace_ms": pace_ms, "returncode": rc, "feeder_errors": len(errors), "wall_s": round(wall, 3),
           "options": " ".join(DEEP), "lines": out.count("\n") - 1}
    m = re.search(r"msk144hipdecoder: (\d+) batches, (\d+) stream hops, (\d+) late, worst latency (\d+) ms", err)   # the total line (per-device lines name their device first)
    if m:
        res.update(batches=int(m.group(1)), stream_hops=int(m.group(2)), late_hops=int(m.group(3)), worst_latency_ms=int(m.group(4)))
    sections = re.split(r"timing: ---- device ", err)
    if len(sections) > 1:
        # --devices: one block per device loop (its own ingest and post-processing threads)
        res["devices"] = devices
        res["per_device"] = []
        for sec in sections[1:]:
            m = re.match(r"(\d+): (\d+) streams \((\d+)\.\.(\d+)\)", sec)
            rows, dev = parse_timing(sec)
            mb = re.search(r"timing: \d+ streams, (\d+) batches in ([\d.]+) s wall", sec)
            res["per_device"].append({"device": int(m.group(1)), "streams": int(m.group(2)), "first_stream": int(m.group(3)), "batches": int(mb.group(1)) if mb else None,
                                      "host_ms_per_batch": rows, "device_ms_per_batch": dev})
        res["host_ms_per_batch"] = res["per_device"][0]["host_ms_per_batch"]
    else:
        rows, dev = parse_timing(err)
        res["host_ms_per_batch"] = rows
        if dev:
            res["device_ms_per_batch"] = dev
    res["phase_spread_ms"] = phase_spread_ms
    res["phases"] = ("one per stream" if phase_per_stream else "one per feeder thread") + f", seed {phase_seed}" if phase_spread_ms > 0 else "aligned"
    res["signal_seconds"] = round((n_hops + 1) * pace_ms * 1e-3, 1)
    if loop_hops > 0 and base_hops < n_hops:
        res["looped_signal"] = f"{base_hops} hops of synthesised signal per stream, fed cyclically (timing run: decode statistics not meaningful)"
    if "worst_latency_ms" in res:
        res["margin_ms_to_210"] = 210 - res["worst_latency_ms"]
    decoded = {}
    for ch, bits in re.findall(r"ch=(\d+); .*?bits='([01]{77})'", out):
        decoded.setdefault(int(ch), set()).add(bits)
    res["decoded_streams"] = sorted(c for c, b in sent.items() if b in decoded.get(c, set()))
    if keep_lines:
        keep = set(int(c) for c in keep_lines)
        res["lines_by_stream"] = {c: [] for c in keep}
        for line in out.split("\n"):
            m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", line)
            if m and int(m.group(1)) in keep:
                res["lines_by_stream"][int(m.group(1))].append("***  " + m.group(2))
    res["streams_with_ping"] = len(sent)
    res["pings_decoded"] = sum(1 for c, b in sent.items() if b in decoded.get(c, set()))
    res["warnings"] = err.count("Warning: Working loop takes too much time")
    res["stderr_tail"] = err[-600:] if (rc != 0 or errors) else ""
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--hops", type=int, default=20)
    ap.add_argument("--pace-ms", type=float, default=216.0)
    ap.add_argument("--hop-timeout-ms", type=int, default=None, help="default: the decoder's own (20 ms)")
    ap.add_argument("--devices", default=None, help="passed to the decoder as --devices=... (one loop per entry; an ordinal may repeat)")
    ap.add_argument("--phase-spread-ms", type=float, default=0.0, help="spread the feeders' hop clocks over this many ms instead of phase-aligning every stream")
    ap.add_argument("--feeders", type=int, default=8)
    ap.add_argument("--phase-per-stream", action="store_true", help="with --phase-spread-ms: one random phase per STREAM instead of one per feeder thread")
    ap.add_argument("--phase-seed", type=int, default=11)
    ap.add_argument("--loop-hops", type=int, default=0, help="synthesise only this many hops per stream and feed them cyclically (long timing runs)")
    a = ap.parse_args()
    print(json.dumps(run(a.streams, a.hops, a.pace_ms, hop_timeout_ms=a.hop_timeout_ms, devices=a.devices, phase_spread_ms=a.phase_spread_ms, feeders=a.feeders,
                         phase_per_stream=a.phase_per_stream, phase_seed=a.phase_seed, loop_hops=a.loop_hops, timeout_s=300.0)), flush=True)


if __name__ == "__main__":
    main()
