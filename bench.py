#!/usr/bin/env python3
"""Headline benchmark: candidate decodes/s (front end + scan + softbits + index + LDPC + collect) at
--search-width=500 --search-step=1 --scan-depth=6 --nbadsync-threshold=3.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one decode window (5184 samples) of every channel of the batch: BASELINE.json configs[2],
1024 synthetic 12 ksps audio channels per MI355X (F=501 x D=6 x 8 = 24 048 candidates per window,
24.6 M candidates per step and GPU).  Inputs are int16 windows already resident in HBM.  For N > 1
(launched by torch.distributed.run, one rank per GPU) every rank decodes its own 1024 channels - the
path shards by channel with no data-path collective - and each step ends with one RCCL gather of the
fixed-size decoded-record buffers to rank 0 ("scaling": "weak").

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel with SURVEY.md 8(d)'s algorithmic
634 B/candidate against the 8 TB/s HBM peak; the path is VALU/LDS-bound (10^3..10^6 flop per
compulsory HBM byte), so that fraction is small by construction and is reported, not engineered.
`cpu_baseline` times the CPU oracle (the reference's exhaustive algorithm restated in C++, OpenMP over
frequency hypotheses) on a bounded sample of the same windows - a reported baseline, not a target.
"""
from __future__ import annotations

import argparse
import json
import os
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
GATHER_CAP = 8192                # decoded records gathered per rank and step (fixed-size buffer)


def make_inputs(rank: int, channels: int):
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


def cpu_baseline(windows: np.ndarray, budget_s: float = 12.0):
    """Oracle ('port' of the reference algorithm) on the host cores, bounded sample of the same workload."""
    from oracle import oracle as orc  # test infrastructure: used here only as the timed CPU baseline
    cores = min(os.cpu_count() or 1, 16)
    o = orc.Oracle(center=1500.0, width=WIDTH, step=STEP, depth=DEPTH, nbadsync_threshold=NBADSYNC, threads=cores)
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
    return {"value": done * o.total_items / el, "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": f"{done} window(s) of channel(s) 0..{done - 1}, step 0 (of {windows.shape[1]} channels), {el:.1f} s; exhaustive reference algorithm, not WSJT-X msk144spd"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=CHANNELS_PER_GPU, help="channels per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun the RCCL path runs even with one rank
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world:
        if rank == 0 and distributed:
            print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    if distributed:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from msk144cudecoder_amd.hipdecoder import RESULT_DTYPE, T_NAMES, HipDecoder  # fails loudly without the HIP library

    channels = args.channels
    wins_host, truth = make_inputs(rank, channels)
    wins_dev = torch.from_numpy(wins_host).cuda(local_rank)           # inputs resident in HBM before timing

    dec = HipDecoder(center=1500.0, width=WIDTH, step=STEP, depth=DEPTH, nbadsync_threshold=NBADSYNC, read_mode=1, analytic_method=2,
                     channels=channels, device=local_rank, max_results=1 << 20)
    dec.set_stream(torch.cuda.current_stream().cuda_stream)
    cand_per_step = channels * dec.K

    rec_ptr, cnt_ptr = dec.results_device()
    rec_view = torch.as_tensor(_DevView(rec_ptr, GATHER_CAP * RESULT_DTYPE.itemsize), device=f"cuda:{local_rank}")
    cnt_view = torch.as_tensor(_DevView(cnt_ptr, 4), device=f"cuda:{local_rank}")
    send = torch.empty(GATHER_CAP * RESULT_DTYPE.itemsize + 4, dtype=torch.uint8, device=f"cuda:{local_rank}")
    gathered = [torch.empty_like(send) for _ in range(world)] if (distributed and rank == 0) else None

    def step(i):
        w = wins_dev[i % WINDOWS_PER_CHANNEL]
        dec.submit_audio_device(w.data_ptr())
        dec.decode()
        if distributed:
            # the path's only exchange: fixed-size decoded-record buffers -> rank 0 (RCCL over xGMI)
            send[:-4].copy_(rec_view, non_blocking=True)
            send[-4:].copy_(cnt_view, non_blocking=True)
            dist.gather(send, gathered, dst=0)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    dec.set_profiling(True)
    dec.stage_times(reset=True)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage = dec.stage_times()
    last = dec.results()
    # Payloads that are not the channel's transmitted message.  The reference algorithm accepts on CRC-13
    # + < 18 hard errors, so at 1.6e7 BP attempts per step a few false positives are expected; they are
    # the oracle's too (tests/test_gpu_full.py), not decoder errors.
    wrong = sum(1 for r in last if truth.get(int(r["channel"])) != bytes(r["message"]))
    chans_decoded = len({int(r["channel"]) for r in last})

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * cand_per_step * args.steps / elapsed
        dom = max((n for n in T_NAMES), key=lambda n: stage[n][0])
        dom_ms = stage[dom][0]
        achieved = B_ALG_PER_CANDIDATE * cand_per_step / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(dom + "_kernel", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # The bound that actually applies: VALU instruction issue.  Instruction counts per launch come from the
        # committed SQ counter pass of this same command (profiles/r01_sq_counters.json); the rate uses the
        # HIP-event time of THIS run.  Ceiling ~8e11 wave-instr/s (tools/ubench/valu_rates.hip).
        valu = None
        cfile = os.path.join(ROOT, "profiles", "r01_sq_counters.json")
        if os.path.exists(cfile) and channels == CHANNELS_PER_GPU:
            try:
                sq = json.load(open(cfile))
                valu = {"unit": "wave-instr/s", "ceiling": 8.0e11, "source": "profiles/r01_sq_counters.json (SQ_INSTS_VALU) / stage_ms"}
                for k in ("scan", "softbits", "ldpc"):
                    if stage[k][0] > 0:
                        rate = sq[k + "_kernel"]["SQ_INSTS_VALU"] / (stage[k][0] * 1e-3)
                        valu[k + "_kernel"] = {"achieved": rate, "frac": rate / 8.0e11}
            except Exception:
                valu = None
        out = {
            "metric": "candidate decodes/sec (scan+softbits+LDPC), width=500 step=1 depth=6",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {channels} synthetic 12 ksps int16 audio channels per GPU, one 5184-sample window per "
                                   f"channel per step, width=500 step=1 depth=6 nbadsync-threshold=3 (F={dec.F}, D={dec.D}, {dec.K} candidates/window)",
                       "channels_per_gpu": channels, "candidates_per_step_per_gpu": cand_per_step, "parallelism": f"channel-shard x{world}",
                       "analytic_method": 2, "real_time_channels": value / dec.K / (12000.0 / 2592.0)},
            "roofline": {"bound": "hbm", "kernel": dom + "_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": B_ALG_PER_CANDIDATE * cand_per_step, "avg_launch_ms": dom_ms,
                         "note": "path is VALU/LDS-bound (SURVEY.md 8d); HBM fraction reported as measured"},
            "valu_issue": valu,
            "stage_ms": {n: round(stage[n][0], 4) for n in T_NAMES},
            "decodes_last_step": int(len(last)), "channels_decoded_last_step": chans_decoded, "crc13_false_positives_last_step": wrong,
        }
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wins_host)
        print(json.dumps(out), flush=True)

    dec.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
