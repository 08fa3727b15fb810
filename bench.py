#!/usr/bin/env python3
"""Headline benchmark: candidate decodes/s (front end + scan + softbits + index + LDPC + collect) at
--search-width=500 --search-step=1 --scan-depth=6 --nbadsync-threshold=3.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one decode window (5184 samples) of every channel of the batch: BASELINE.json configs[2],
1024 synthetic 12 ksps audio channels per MI355X (F=501 x D=6 x 8 = 24 048 candidates per window,
24.6 M candidates per step and GPU).  Inputs are int16 windows already resident in HBM.

N > 1: `python bench.py --gpus N` starts the N ranks ITSELF - the parent process never imports torch.cuda
or touches HIP; it runs `python -m torch.distributed.run --nproc-per-node N bench.py ... --worker` as a
child, relays rank 0's JSON line and exits with the child's status.  Launched by the driver through
torch.distributed.run (WORLD_SIZE set) it is a worker directly.  Every rank decodes its own 1024 channels
(the path shards by channel, no data-path collective) and each step ends with ONE exchange: an RCCL
gather of the fixed-size decoded-record buffers (global channel ids, (n, total) trailer) to rank 0, which
validates total <= capacity for every rank and step ("scaling": "weak").

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel with SURVEY.md 8(d)'s algorithmic
634 B/candidate against the 8 TB/s HBM peak; the path is VALU/LDS-bound (10^3..10^6 flop per
compulsory HBM byte), so that fraction is small by construction and is reported, not engineered.
`cpu_baseline` times the CPU oracle (the reference's exhaustive algorithm restated in C++, OpenMP over
frequency hypotheses) on a bounded sample of the same windows - a reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, STEP, DEPTH, NBADSYNC = 500.0, 1.0, 6, 3
CHANNELS_PER_GPU = 1024
WINDOWS_PER_CHANNEL = 4          # distinct consecutive windows staged per channel, cycled by the steps
B_ALG_PER_CANDIDATE = 634.0      # one 632-byte result record written + window share read
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2.0   # 256 CUs x 4 SIMD-32, one wave64 VALU op per 2 cycles at 2.4 GHz
COUNTERS_FILE = os.path.join(ROOT, "profiles", "counters.json")


def kernel_source_sha() -> str:
    """Hash of the kernel sources' CODE - the .hip files and the headers they include (// comments and blank space dropped, so that
    rewording a comment does not orphan the counters; the host-side msk144_api.cpp holds no device code or launch geometry and is
    left out); profiles/counters.json records the one it was collected on, so stale static counters are never mixed with live
    timings."""
    csrc = os.path.join(ROOT, "msk144cudecoder_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            for line in open(os.path.join(csrc, name), "r", encoding="utf-8"):
                cut = line.find("//")
                while cut > 0 and line[:cut].count('"') % 2 == 1:      # "//" inside a string literal (asm text, URLs)
                    cut = line.find("//", cut + 2)
                code = (line if cut < 0 else line[:cut]).strip()
                if code:
                    h.update(" ".join(code.split()).encode())
                    h.update(b"\n")
    return h.hexdigest()[:16]


_INPUTS = {}


def make_inputs(rank: int, channels: int):
    """Cached per (rank, channels): the CPU baseline and the GPU backend of one process see the same arrays."""
    key = (rank, channels)
    if key not in _INPUTS:
        _INPUTS[key] = _make_inputs(rank, channels)
    return _INPUTS[key]


def _make_inputs(rank: int, channels: int):
    """[T][channels][5184] int16: AWGN sigma=1000 LSB, every 4th channel carries one 0 dB ping
    (S2-style) somewhere in its 1.08 s of signal.  Returns (windows, {channel: msg})."""
    from msk144cudecoder_amd import synth
    n = 5184 + (WINDOWS_PER_CHANNEL - 1) * 2592
    rng = np.random.default_rng(synth.SEED_BASE + 0x30000 + rank)
    streams = np.rint(rng.normal(0.0, 1000.0, size=(channels, n)))
    truth = {}
    for ch in range(0, channels, 4):
        msg = synth.random_message(rng)
        n_frames = int(rng.integers(2, 8))
        start = int(rng.integers(0, n - n_frames * 864))
        freq = 1500.0 + float(rng.uniform(-240, 240))
        phase = float(rng.uniform(0, 2 * np.pi))
        amp = np.sqrt(2.0 * 1000.0 ** 2 * (2500.0 / 6000.0))            # 0 dB in 2500 Hz at sigma=1000
        bb = synth.modulate_frame(synth.frame_bits(synth.encode_message(msg)))
        t = np.arange(start, start + n_frames * 864)
        carrier = np.exp(1j * (2 * np.pi * freq * t / 12000.0 + phase))
        streams[ch, start:start + n_frames * 864] += amp * np.real(np.tile(bb, n_frames) * carrier)
        truth[ch] = bytes(np.packbits(np.concatenate([msg, np.zeros(3, dtype=np.uint8)])))
    streams = np.clip(streams, -32768, 32767).astype(np.int16)
    wins = np.stack([streams[:, t * 2592:t * 2592 + 5184] for t in range(WINDOWS_PER_CHANNEL)])
    return np.ascontiguousarray(wins), truth


class _DevView:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class HipBackend:
    """The product path: libmsk144hip.so through its C ABI (ctypes), inputs resident in HBM."""
    name = "hip"
    dist_backend = "nccl"
    data = "synthetic"

    def __init__(self, rank: int, local_rank: int, channels: int, channel_base: int, llr_block: int = 0):
        import torch
        from msk144cudecoder_amd.hipdecoder import RESULT_DTYPE, T_NAMES, HipDecoder  # fails loudly without the HIP library
        self.torch = torch
        self.T_NAMES = T_NAMES
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        # ONE explicit stream carries everything of this rank - decode kernels, timing markers, the copy into the gather's send
        # buffer, and the collective joins it: torch's default stream is the null stream (handle 0, which the library reads as
        # "use your own stream"), and a non-blocking stream is not ordered against it.
        self.stream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.stream)
        self.wins_host, self.truth = make_inputs(rank, channels)
        self.wins_dev = torch.from_numpy(self.wins_host).cuda(local_rank)   # inputs resident in HBM before timing
        self.dec = HipDecoder(center=1500.0, width=WIDTH, step=STEP, depth=DEPTH, nbadsync_threshold=NBADSYNC, read_mode=1,
                              analytic_method=2, channels=channels, device=local_rank, max_results=1 << 20, llr_block_channels=llr_block)
        self.llr_block = self.dec.llr_block
        assert self.stream.cuda_stream != 0
        self.dec.set_stream(self.stream.cuda_stream)
        self.dec.set_channel_base(channel_base)
        self.channel_base = channel_base
        self.F, self.D, self.K = self.dec.F, self.dec.D, self.dec.K
        self.cand_per_step = channels * self.dec.K
        rec_ptr, cnt_ptr = self.dec.results_device()
        self.rec_view = torch.as_tensor(_DevView(rec_ptr, (1 << 20) * RESULT_DTYPE.itemsize), device=self.device)
        self.cnt_view = torch.as_tensor(_DevView(cnt_ptr, 4), device=self.device).view(torch.int32)

    def step(self, i: int):
        w = self.wins_dev[i % WINDOWS_PER_CHANNEL]
        self.dec.submit_audio_device(w.data_ptr())
        self.dec.decode()

    def fence(self):
        self.torch.cuda.synchronize()

    def start_profiling(self):
        self.dec.set_profiling(True)
        self.dec.stage_times(reset=True)

    def stage_times(self):
        return self.dec.stage_times()

    def results(self):
        return self.dec.results()

    def marker(self):
        """Timing event on the decode stream (torch's current stream: the handle runs on it, set_stream above)."""
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    @staticmethod
    def ms_between(a, b):
        return a.elapsed_time(b)

    def clock_probe(self):
        """Shader clock (MHz) read by a one-wave kernel BESIDE the queued steps (msk144_clock_probe)."""
        return self.dec.clock_probe(1000)

    def close(self):
        self.dec.close()


def usable_cores() -> int:
    """Every host core this process may run on: the scheduler affinity mask, cut to the cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            w = open(path).read().split()
            if path.endswith("cpu.max"):
                if w[0] != "max":
                    n = min(n, max(1, int(int(w[0]) / int(w[1]) + 0.5)))
            else:
                q = int(w[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(windows: np.ndarray, budget_s: float = 12.0):
    """Oracle ('port' of the reference algorithm) on ALL usable host cores, bounded sample of the same workload.  Timed on its
    own build of the oracle source (-O3 -march=native, compiled on this host: oracle.build_bench_library); the parity tests keep
    the -O2 build."""
    from oracle import oracle as orc  # test infrastructure: used here only as the timed CPU baseline
    cores = usable_cores()
    flags = " ".join(orc.BENCH_FLAGS)
    try:
        library = orc.bench_lib()
    except Exception as e:  # noqa: BLE001  (no compiler or a read-only tree on this host: time the parity build and say so)
        library = None
        flags = f"-O2 parity build of oracle/Makefile (the -O3 -march=native build failed here: {type(e).__name__})"
    o = orc.Oracle(center=1500.0, width=WIDTH, step=STEP, depth=DEPTH, nbadsync_threshold=NBADSYNC, threads=cores, library=library)
    done = 0
    t0 = time.perf_counter()
    while True:
        w = windows[0, done % windows.shape[1]]
        cd = o.frontend_audio(w, 2)
        o.decode_window(cd)
        done += 1
        el = time.perf_counter() - t0
        if el + el / done > budget_s or done >= 64:
            break
    return {"value": done * o.total_items / el, "unit": "candidates/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "flags": flags + "; OpenMP over frequency hypotheses (scan, softbits) and gated candidates (BP)",
            "sample": f"{done} window(s) of channel(s) 0..{done - 1}, step 0 (of {windows.shape[1]} channels), {el:.1f} s; exhaustive reference algorithm, not WSJT-X msk144spd"}


def roofline_valu(valu, lds, dom):
    """The roofline that binds the dominant kernel: VALU issue (wave64 instructions per second per SIMD against the guide's 2 cycles
    per instruction) and, for LDPC, the CU's single LDS pipe.  The nominal fraction understates the kernels: 30-45 % of their
    instructions are half- or quarter-rate by nature (compare/select/DPP/max: ~4.3 cycles, exp/rcp/sqrt: ~8.2, measured by
    tools/ubench/valu_rates.hip, profiles/r02_valu_issue_microbench.txt).  None when the static counters are stale for these sources."""
    if not valu or (dom + "_kernel") not in valu:
        return None
    k = dom + "_kernel"
    out = {"bound": "valu-issue", "kernel": k, "unit": "wave64 VALU instr/s", "achieved": valu[k]["achieved"], "peak": valu["peak"],
           "frac": valu[k]["frac"], "peak_note": valu.get("peak_note"), "static": valu.get("static")}
    if lds and k in lds:
        out["lds_pipe_busy"] = lds[k]["frac"]
        out["lds_conflict_share"] = lds[k]["conflict_share"]
    return out


def board_power_watts():
    """Best effort: every amdgpu hwmon power reading the host exposes (sysfs, microwatts), in watts; [] when none is readable.  Context
    for the sustained leg only - the test of a sustained clock is the in-kernel probe, not this (MI355X_MICROARCH.md, DVFS give-back)."""
    import glob
    out = []
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        for path in sorted(glob.glob(pat)):
            try:
                out.append(round(int(open(path).read().strip()) * 1e-6, 1))
            except (OSError, ValueError):
                pass
        if out:
            break
    return out


def sustained_leg(be, step, fence, n_steps: int, first_index: int, depth: int = 4, window: int = 50):
    """The same step, n_steps more times, never letting the GPU run dry: the host keeps `depth` steps queued and waits for step
    i - depth before it enqueues step i.  Every step ends with a timing marker on the decode stream; the step time of a stretch is
    the marker-to-marker time over it.  The shader clock is probed beside the running steps at the start, the middle and the end
    (HipBackend.clock_probe: a one-wave kernel on a side stream, s_memtime / s_memrealtime).  Every rank runs the same n_steps
    (the gather of a distributed run is a collective); rank 0 reports."""
    window = max(2, min(window, n_steps // 3))
    marks, clocks, power = [], {}, []
    probe_at = {depth + 2: "first", n_steps // 2: "mid", n_steps - 3: "last"} if hasattr(be, "clock_probe") else {}
    fence()
    t0 = time.perf_counter()
    marks.append(be.marker())
    for i in range(n_steps):
        if i >= depth:
            marks[i - depth + 1].synchronize()
        step(first_index + i)
        marks.append(be.marker())
        if i in probe_at:
            clocks[probe_at[i]] = round(be.clock_probe(), 1)
            if probe_at[i] == "mid":
                power = board_power_watts()
    fence()
    wall = time.perf_counter() - t0

    def stretch(a, b):
        return be.ms_between(marks[a], marks[b]) / (b - a)

    mid = (n_steps - window) // 2
    ms = {"first%d" % window: stretch(0, window), "mid%d" % window: stretch(mid, mid + window), "last%d" % window: stretch(n_steps - window, n_steps),
          "all": stretch(0, n_steps)}
    first, last = ms["first%d" % window], ms["last%d" % window]
    return {"steps": n_steps, "seconds": wall, "queue_depth": depth, "ms_per_step": {k: round(v, 4) for k, v in ms.items()},
            "drift_last_vs_first": (last / first - 1.0) if first > 0 else None, "wall_ms_per_step": wall / n_steps * 1e3,
            "clock_mhz": clocks or None,
            "board_power_w_mid_run": {"max": max(power), "every_gpu_of_the_host": power,
                                      "note": "amdgpu hwmon power1_average of every GPU the host exposes (sysfs), read in the middle of the leg; the job's own "
                                              "GPU is not identified - on an otherwise idle host it is the maximum"} if power else None,
            "note": "same step and inputs as the timed region, run on after it; markers = HIP events on the decode stream; clock_mhz = shader clock read by a "
                    "one-wave probe beside the running steps (s_memtime / s_memrealtime x 100 MHz); `value` is NOT taken from this leg"}


def every_slot_leg(be, n_steps: int, first_index: int):
    """The same step on the same inputs with the copy hand-over switched OFF (msk144_set_copy_handover(h, 0)): every one of the
    F*D*8 slots of every window is demodulated (up to the nbadsync gate) and, when gated, decoded on its own - what the reference's
    softbits_kernel / ldpc_kernel do (softbits_kernel.cuh:56-83, ldpc_kernel.cuh:100-249) and what SURVEY.md 8(d) calls a slot "fully
    evaluated".  Blocked staging and the early gate stay as in the timed region.  Reported beside `value`, never as `value`."""
    dec = be.dec
    dec.set_copy_handover(False)
    try:
        for i in range(2):
            be.step(first_index + i)
        be.fence()
        dec.stage_times(reset=True)
        t0 = time.perf_counter()
        for i in range(n_steps):
            be.step(first_index + 2 + i)
        be.fence()
        el = time.perf_counter() - t0
        st = dec.stage_times(reset=True)
        handed = dec.copy_count()
        records = len(dec.results())
    finally:
        dec.set_copy_handover(True)
    return {"value": be.cand_per_step * n_steps / el, "unit": "candidates/s", "steps": n_steps, "warmup": 2, "ms_per_step": el / n_steps * 1e3,
            "stage_ms": {n: round(st[n][0], 4) for n in be.T_NAMES}, "slots_handed_over": handed, "records_last_step": records,
            "note": "copy hand-over off: every slot demodulated / decoded on its own as in the reference; same inputs, same handle, blocked staging and early "
                    "nbadsync gate unchanged; run after the timed region"}


def pcie_leg(be, n_steps: int):
    """The same step with the windows coming from pinned HOST memory (the reference's hop includes its H2D copy, main.cu:325): each
    step is msk144_submit_slot (asynchronous H2D of the slot's 1024 windows + front end), msk144_decode, msk144_fetch_async
    (asynchronous D2H of count + records + segment powers), pipelined over the handle's two pinned staging slots exactly as
    msk144hipdecoder's loop drives them.  The slots are filled before the clock starts (they ARE the host buffer a reader fills).
    Reported beside `value`, never as `value`."""
    dec = be.dec
    for s in range(2):
        dec.input_slot(s)[:] = be.wins_host[s % WINDOWS_PER_CHANNEL]
    for i in range(2):                       # both slots once, untimed
        dec.submit_slot(i)
        dec.decode()
        dec.fetch_async(i)
    for i in range(2):
        dec.fetch_wait(i)
    be.fence()
    dec.stage_times(reset=True)
    t0 = time.perf_counter()
    records = 0
    for i in range(n_steps):
        s = i % 2
        if i >= 2:
            records = len(dec.fetch_wait(s)[0])
        dec.submit_slot(s)
        dec.decode()
        dec.fetch_async(s)
    for i in range(max(0, n_steps - 2), n_steps):
        records = len(dec.fetch_wait(i % 2)[0])
    be.fence()
    el = time.perf_counter() - t0
    st = dec.stage_times(reset=True)
    return {"value": be.cand_per_step * n_steps / el, "unit": "candidates/s", "steps": n_steps, "ms_per_step": el / n_steps * 1e3,
            "h2d_ms_per_step": round(st["h2d"][0], 4), "d2h_ms_per_step": round(st["d2h"][0], 4), "records_last_step": records,
            "h2d_bytes_per_step": int(be.wins_host[0].nbytes),
            "note": "windows in pinned host memory -> msk144_submit_slot / msk144_decode / msk144_fetch_async over the two staging slots, copies and kernels "
                    "on the one decode stream (no copy/compute overlap); timed from the first submit to the last msk144_fetch_wait"}


def frontend_leg(be, n_steps: int = 5):
    """Stage time of the front-end kernels alone on the bench's 1024 windows: analytic method 1 (frontend_fft_kernel, the reference's
    Analytic::execute, analytic_fft.cu:84-157) beside method 2 (frontend_fir_kernel, the default).  The front end does not depend on
    the search grid, so the two handles are created with a narrow one."""
    from msk144cudecoder_amd.hipdecoder import HipDecoder
    out = {}
    w = be.wins_dev[0]
    for method, name in ((1, "frontend_fft_kernel"), (2, "frontend_fir_kernel")):
        with HipDecoder(center=1500.0, width=12.0, step=2.0, depth=1, nbadsync_threshold=0, read_mode=1, analytic_method=method,
                        channels=w.shape[0], device=be.device.index, max_results=1024) as d:
            d.set_stream(be.stream.cuda_stream)
            for _ in range(3):
                d.submit_audio_device(w.data_ptr())
            d.synchronize()
            d.set_profiling(True)
            d.stage_times(reset=True)
            for _ in range(n_steps):
                d.submit_audio_device(w.data_ptr())
            ms, cnt = d.stage_times()["frontend"]
            out[name] = {"analytic_method": method, "ms_per_launch": round(ms, 4), "launches": cnt, "channels": int(w.shape[0]),
                         "windows_per_s": w.shape[0] / (ms * 1e-3) if ms > 0 else None}
    out["note"] = "HIP events around the one front-end launch of msk144_submit_audio_device, inputs resident in HBM; method 2 is what the timed region runs"
    return out


def configs4_leg(be, n_steps: int = 3, channels: int = 4096):
    """BASELINE configs[4]: IQ --read-mode=2, 4096 low-SNR synthetic channels (complex AWGN sigma 20 LSB per rail, every 4th channel one ping of
    3-6 frames at -6..-2 dB), width 500 / step 1 / depth 6 / threshold 3 - the LDPC-iteration-heavy stress - as a short leg on its own handle."""
    import torch
    from msk144cudecoder_amd import synth
    from msk144cudecoder_amd.hipdecoder import HipDecoder
    wins, truth = synth.iq_low_snr_batch(channels, 5)
    dev = torch.from_numpy(wins).cuda(be.device.index)
    with HipDecoder(center=0.0, width=WIDTH, step=STEP, depth=DEPTH, nbadsync_threshold=NBADSYNC, read_mode=2, channels=channels,
                    device=be.device.index, max_results=1 << 20) as d:
        d.set_stream(be.stream.cuda_stream)
        d.submit_iq_device(dev.data_ptr())
        d.decode()
        d.synchronize()
        d.set_profiling(True)
        d.stage_times(reset=True)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            d.submit_iq_device(dev.data_ptr())
            d.decode()
        d.synchronize()
        el = time.perf_counter() - t0
        st = d.stage_times()
        res = d.results()
        hit = {int(r["channel"]) for r in res if truth.get(int(r["channel"])) == bytes(r["message"])}
        return {"workload": f"BASELINE configs[4]: {channels} int8 I/Q channels (--read-mode=2, centre 0 Hz), width=500 step=1 depth=6 nbadsync-threshold=3 "
                            f"(F={d.F}, {d.K} candidates/window), one window per channel per step",
                "value": channels * d.K * n_steps / el, "unit": "candidates/s", "steps": n_steps, "warmup": 1, "ms_per_step": el / n_steps * 1e3,
                "stage_ms": {n: round(st[n][0], 4) for n in be.T_NAMES}, "llr_block_channels": d.llr_block,
                "decodes_last_step": int(len(res)), "pinged_channels": len(truth), "pinged_channels_decoded": len(hit)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=CHANNELS_PER_GPU, help="channels per GPU (default: the BASELINE config)")
    ap.add_argument("--llr-block", type=int, default=0, help="channels per softbits->index->LDPC block (0 = library default; = channels: retain every LLR row)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="time budget of the CPU baseline's bounded sample")
    ap.add_argument("--force-cpu-baseline", action="store_true",
                    help="TEST HOOK: time the CPU baseline although --backend-module replaces the GPU backend (rehearsal of the N-rank launch on the CPU)")
    ap.add_argument("--sustain-seconds", type=float, default=20.0,
                    help="after the K timed steps, keep running the same step for about this long (untimed inputs unchanged) and report the step time of its "
                         "first / middle / last 50 steps and the shader clock: `value` stays the K-step figure (0 = skip)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short legs appended after the timed region: value_every_slot_decoded (the same step with the copy hand-over off; every rank "
                         "runs it) and, single-GPU runs only, value_incl_h2d (pinned host windows), frontend_method1 (FFT front end beside the FIR one) and "
                         "configs4_iq (BASELINE configs[4])")
    ap.add_argument("--every-slot", action="store_true",
                    help="run the TIMED region with the copy hand-over off (msk144_set_copy_handover(h, 0)): `value` is then slots fully evaluated per second, as the "
                         "reference computes them; the value_every_slot_decoded leg is dropped (it would repeat the timed region)")
    ap.add_argument("--launcher", action="store_true", help="go through the N-rank launcher even for --gpus 1 (exercises the RCCL gather path)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend-module", default=None, help="TEST HOOK: module providing Backend (e.g. tests/stub_backend.py on gloo); the line is then labelled as such")
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """Parent of an N-rank run.  Touches no GPU API: it only starts torch.distributed.run as a CHILD process
    (never exec: a process that has initialised the GPU must not be replaced, and this one has not even
    imported torch) and relays rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cpu_file = None
    if not args.no_cpu_baseline and (not args.backend_module or args.force_cpu_baseline):
        # The CPU baseline is timed HERE, before any rank exists: nothing else of the job runs on the host cores meanwhile (timed on
        # rank 0 it shared them with the other ranks' start-up).  numpy + the oracle library only - still no GPU API in this process.
        import tempfile
        cpu = cpu_baseline(make_inputs(0, args.channels)[0], args.cpu_baseline_seconds)
        cpu["sample"] += "; timed in the launcher parent before the ranks were started"
        fd, cpu_file = tempfile.mkstemp(prefix="msk144_cpu_baseline_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(cpu, f)
        env["MSK144_BENCH_CPU_BASELINE_FILE"] = cpu_file
        _INPUTS.clear()
    passed = [a for a in argv if a != "--launcher"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + passed + ["--worker"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    if cpu_file:
        os.unlink(cpu_file)
    line = None
    for ln in proc.stdout.splitlines():
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode
    if line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def run_worker(args) -> int:
    import importlib

    import torch
    import torch.distributed as dist

    from msk144cudecoder_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "WORLD_SIZE" in os.environ       # under torch.distributed.run the gather path runs even with one rank
    if distributed and args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled run", file=sys.stderr)
        return 2

    Backend = importlib.import_module(args.backend_module).Backend if args.backend_module else HipBackend
    channels = args.channels

    # The CPU baseline runs FIRST, on rank 0, before anything touches the GPU: the GPU phase is then the tail of the process
    # (a sampler of GPU activity catches it), and the other ranks of a distributed run simply wait for rank 0 at the rendezvous.
    cpu = None
    handed = os.environ.get("MSK144_BENCH_CPU_BASELINE_FILE")
    if rank == 0 and not args.no_cpu_baseline and (Backend is HipBackend or args.force_cpu_baseline):
        if handed and os.path.exists(handed):
            cpu = json.load(open(handed))          # bench.py --gpus N: timed by the launcher parent before the ranks existed
        else:
            cpu = cpu_baseline(make_inputs(0, channels)[0], args.cpu_baseline_seconds)
            _INPUTS.clear()                        # the GPU phase stages its own copy; do not keep a second one alive
            if distributed and world > 1:
                cpu["sample"] += f"; timed while the other {world - 1} rank(s) of the job were starting up (torch import, rendezvous wait) on the same host"

    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if Backend.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(Backend.dist_backend)

    channel_base, _ = sharding.shard_channels(channels * world, rank, world)   # rank r owns global channels [r*C, (r+1)*C)
    be = Backend(rank, local_rank, channels, channel_base, args.llr_block)
    cand_per_step = be.cand_per_step
    if args.every_slot and Backend is HipBackend:
        be.dec.set_copy_handover(False)

    gather = None
    if distributed:
        cap = sharding.gather_capacity(channels)
        gather = sharding.RecordGather(cap, be.device, world, rank)

    timing = {"on": False}

    def step(i):
        be.step(i)
        if gather is not None:
            # the path's only exchange: fixed-size decoded-record buffers -> rank 0 (RCCL over xGMI)
            gather.step(be.rec_view, be.cnt_view, timed=timing["on"])

    def fence():
        if distributed:
            dist.barrier()
        be.fence()

    for i in range(args.warmup):
        step(i)
    fence()
    be.start_profiling()
    timing["on"] = True
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0

    rank_ms = [elapsed / args.steps * 1e3]
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=be.device)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        elapsed = max(float(x.item()) for x in every)            # MAX over ranks

    stage = be.stage_times()

    sustained = None
    if args.sustain_seconds > 0 and hasattr(be, "marker"):
        n_sus = int(min(100000, max(150, -(-args.sustain_seconds * args.steps // max(elapsed, 1e-9)))))   # same count on every rank: `elapsed` is the max over ranks
        sustained = sustained_leg(be, step, fence, n_sus, args.warmup + args.steps)
        st2 = be.stage_times()      # averages over the timed region AND the sustained leg (profiling stayed on)
        sustained["stage_ms_incl_timed_region"] = {n: round(st2[n][0], 4) for n in be.T_NAMES}
    last = be.results()         # of the last step run (the sustained leg's when there is one): what the last gather must have carried
    # What `value` counts: every slot is REPORTED each step; with the hand-over on (the default of blocked staging) some are not computed
    # again.  The share of the last step, and - same inputs, same box, right after the timed region - the rate with every slot computed.
    handover = None
    if Backend is HipBackend:
        on = be.dec.copy_handover()
        handed = be.dec.copy_count()
        handover = {"enabled": on, "slots_handed_over_last_step": handed, "share_of_slots": handed / cand_per_step}
        if on and not args.no_extra_legs:
            handover["value_every_slot_decoded"] = every_slot_leg(be, max(4, min(args.steps, 10)), args.warmup + args.steps)
    # Short extra legs, single-GPU runs of the product backend only (each on rank 0 would leave the other ranks waiting): after
    # everything the main line reports has been read, so they cannot disturb it.
    extra = {}
    if not args.no_extra_legs and world == 1 and not distributed and Backend is HipBackend:
        extra["value_incl_h2d"] = pcie_leg(be, max(4, min(args.steps, 10)))
        extra["frontend_method1"] = frontend_leg(be)
        extra["configs4_iq"] = configs4_leg(be)
    # Payloads that are not the channel's transmitted message.  The reference algorithm accepts on CRC-13
    # + < 18 hard errors, so at 1.6e7 BP attempts per step a few false positives are expected; they are
    # the oracle's too (tests/test_gpu_full.py), not decoder errors.
    wrong = sum(1 for r in last if be.truth.get(int(r["channel"]) - channel_base) != bytes(r["message"]))
    chans_decoded = len({int(r["channel"]) for r in last})

    rc = 0
    gathered_records = None
    if gather is not None:
        try:
            per_rank = gather.finish()                      # raises OverflowError if any rank exceeded the capacity
            if rank == 0:
                gathered_records = int(sum(len(r) for r in per_rank))
                for r, rec in enumerate(per_rank):
                    lo, cnt = sharding.shard_channels(channels * world, r, world)
                    if len(rec) and not ((rec["channel"] >= lo) & (rec["channel"] < lo + cnt)).all():
                        raise AssertionError(f"rank {r}: gathered channel ids outside its shard [{lo}, {lo + cnt})")
                if not np.array_equal(per_rank[0], last[:len(per_rank[0])]) or len(per_rank[0]) != len(last):
                    raise AssertionError("rank 0's gathered records differ from its own result list")
        except (OverflowError, AssertionError) as e:
            print(f"bench.py: gather validation failed: {e}", file=sys.stderr)
            rc = 3

    if rank == 0 and rc == 0:
        T_NAMES = be.T_NAMES
        ms_per_step = elapsed / args.steps * 1e3
        value = world * cand_per_step * args.steps / elapsed
        dom = max((n for n in T_NAMES), key=lambda n: stage[n][0])
        # stage times are per STEP; softbits/index/ldpc are launched once per channel block (blocked staging), so one launch
        # of the dominant kernel covers cand_per_step / launches candidates and lasts step_ms / launches on average
        llr_block = getattr(be, "llr_block", channels) or channels
        launches = -(-channels // llr_block) if dom in ("softbits", "index", "ldpc") else 1
        dom_ms = stage[dom][0] / launches
        cand_per_launch = cand_per_step / launches
        achieved = B_ALG_PER_CANDIDATE * cand_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # Static PMC counters (separate rocprofv3 --pmc passes, committed under profiles/): used only when they were
        # collected on exactly these kernel sources and this workload; labelled with their origin.
        traffic, valu, lds, static = None, None, None, None
        if os.path.exists(COUNTERS_FILE) and channels == CHANNELS_PER_GPU and Backend is HipBackend:
            try:
                cj = json.load(open(COUNTERS_FILE))
                shape_ok = abs(cj.get("kernels", {}).get(dom + "_kernel", {}).get("launches_per_step", launches) - launches) < 0.5    # per-launch figures: same block size
                if cj.get("kernel_source_sha") == kernel_source_sha() and shape_ok:
                    static = f"profiles/counters.json@{cj.get('commit', '?')} (kernel_source_sha {cj['kernel_source_sha']})"
                    traffic = cj.get("kernels", {}).get(dom + "_kernel", {}).get("hbm_bytes_per_launch")
                    valu = {"unit": "wave-instr/s", "peak": VALU_PEAK_WAVE_INSTR,
                            "peak_note": "256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 VALU op (MI355X_MICROARCH.md); transcendentals and "
                                         "v_pk_* cost more than one slot, see profiles/r02_valu_issue_microbench.txt",
                            "static": static}
                    lds = {"unit": "busy fraction of the LDS array (SQ_LDS_IDX_ACTIVE cycles / (256 CUs x 2.4 GHz x measured stage time))",
                           "static": static}
                    for k in ("scan", "softbits", "ldpc"):
                        n_valu = cj.get("kernels", {}).get(k + "_kernel", {}).get("SQ_INSTS_VALU")
                        if n_valu and stage[k][0] > 0:
                            rate = n_valu / (stage[k][0] * 1e-3)
                            valu[k + "_kernel"] = {"achieved": rate, "frac": rate / VALU_PEAK_WAVE_INSTR}
                        n_lds = cj.get("kernels", {}).get(k + "_kernel", {}).get("SQ_LDS_IDX_ACTIVE")
                        if n_lds and stage[k][0] > 0:
                            lds[k + "_kernel"] = {"frac": n_lds / (256 * 2.4e9 * stage[k][0] * 1e-3),
                                                  "conflict_share": cj["kernels"][k + "_kernel"].get("SQ_LDS_BANK_CONFLICT", 0.0) / n_lds}
                else:
                    static = "profiles/counters.json is stale for these kernel sources or this block size: traffic/valu_issue omitted"
            except Exception as e:  # noqa: BLE001
                static = f"profiles/counters.json unreadable: {e}"
        if llr_block >= channels:
            mode = "retained: every candidate demodulated in full, every LLR row kept (parity-dump mode)"
        else:
            mode = ("blocked staging: LLR rows are written only for candidates that pass the nbadsync gate and live for one "
                    f"{llr_block}-channel block; a gated-out candidate stops after its sync check (softbits_kernel<true, .>); a candidate that folds the same "
                    "frames as a lower slot of its (frequency, pattern) group - ring-wrap twins, the periodic copies of masks 111111 / 100100 - ")
            if handover and handover["enabled"]:
                mode += (f"is neither demodulated nor decoded again and reports that slot's result: {100.0 * handover['share_of_slots']:.1f} % of the slots of this "
                         "run's last step (DESIGN.md 3; every slot is reported, the list is byte-identical to computing each one; value_every_slot_decoded = "
                         "the same step with the hand-over off)")
            else:
                mode += "is computed again, as in the reference (copy hand-over off or not reported by this backend)"
        out = {
            "metric": "candidate decodes/sec (scan+softbits+LDPC), width=500 step=1 depth=6",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": Backend.data,
            "config": {"workload": f"BASELINE configs[2]: {channels} synthetic 12 ksps int16 audio channels per GPU, one 5184-sample window per "
                                   f"channel per step, width=500 step=1 depth=6 nbadsync-threshold=3 (F={be.F}, D={be.D}, {be.K} candidates/window)",
                       "channels_per_gpu": channels, "candidates_per_step_per_gpu": cand_per_step, "parallelism": f"channel-shard x{world}",
                       "analytic_method": 2, "llr_block_channels": getattr(be, "llr_block", None),
                       "llr_store": f"blocked/{llr_block}" if llr_block < channels else "retained",
                       "softbits_gate_early": bool(llr_block < channels),
                       "copies_computed_once": bool(handover["enabled"]) if handover else bool(llr_block < channels),
                       "value_counts": (f"ResultItem slots REPORTED per second (every slot of every window, each step); {100.0 * handover['share_of_slots']:.1f} % of them "
                                        "(measured on the last step of this run) fold the same frames as a lower slot of their group and were handed over - not "
                                        "demodulated / decoded again, they report that slot's result; `value_every_slot_decoded` is the same step with every slot "
                                        "computed on its own, as the reference does") if handover and handover["enabled"]
                                       else "ResultItem slots fully evaluated per second (every slot demodulated up to the nbadsync gate and, when gated, decoded)",
                       "backend": Backend.name, "launch": "torch.distributed.run" if distributed else "single process",
                       "real_time_channels": value / be.K / (12000.0 / 2592.0),
                       "real_time_channels_note": "hot-clock GPU-only arithmetic (windows/s / 4.63 at the back-to-back clock of ~2.35 GHz).  The stream decoder program itself, "
                                                  "measured over 60 s of signal per stream (tools/host_scale.py, profiles/r06_host_scale_*_60s.json): with every stream's hop "
                                                  "falling due together 4096 real-time streams - 0 late of 1.14 M hops, worst hop latency 170 ms of the 210 ms limit - and 4608 at the "
                                                  "edge (0 late, 190 ms); with every stream on its own hop phase 5376 / 5888 streams - 0 late, worst 53 / 69 ms.  A GPU that idles "
                                                  "between hops starts each batch at 1.8-2.0 GHz (profiles/r04_idle_gap.json), so a lightly loaded program runs its kernels ~8 % slower "
                                                  "than this line"},
            "roofline": {"bound": "hbm", "kernel": dom + "_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_static": static,
                         "algorithmic_bytes_per_launch": B_ALG_PER_CANDIDATE * cand_per_launch, "avg_launch_ms": dom_ms, "launches_per_step": launches,
                         "candidates_per_launch": cand_per_launch,
                         "mode": mode,
                         "note": "the contract's HBM figure; the path is VALU-issue/LDS-pipe bound (SURVEY.md 8d) - the binding roofline is roofline_valu"},
            "roofline_valu": roofline_valu(valu, lds, dom),
            "valu_issue": valu,
            "lds_array": lds,
            "stage_ms": {n: round(stage[n][0], 4) for n in T_NAMES},
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": [round(x, 4) for x in rank_ms]},
            "decodes_last_step": int(len(last)), "channels_decoded_last_step": chans_decoded, "crc13_false_positives_last_step": wrong,
        }
        if gather is not None:
            out["gather"] = {"records_last_step": gathered_records, "capacity_per_rank": gather.cap, "bytes_per_rank": int(gather.send.numel()),
                             "peak_records_per_rank": [int(x) for x in gather.max_total.cpu().numpy()], "backend": Backend.dist_backend,
                             "ms_per_step": gather.mean_ms(), "ms_note": "copy into the send buffer + gather, timed on rank 0 around RecordGather.step"}
        if handover is not None:
            every = handover.pop("value_every_slot_decoded", None)
            out["copy_handover"] = handover
            if every is not None:
                out["value_every_slot_decoded"] = every
        if sustained is not None:
            out["sustained"] = sustained
        out.update(extra)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    be.close()
    if distributed:
        flag = torch.tensor([rc], dtype=torch.int32, device=be.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        rc = int(flag.item())
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and not args.worker and (args.gpus > 1 or args.launcher):
        return launch_ranks(args, argv)
    return run_worker(args)


if __name__ == "__main__":
    sys.exit(main())
