"""bench.py --gpus N must start N ranks by itself (the driver runs `python bench.py --gpus N`), run the per-step record
gather through msk144cudecoder_amd.sharding and report n_gpus = N.  Exercised here on the CPU: gloo + tests/stub_backend."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300, channels=8):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--channels", str(channels), "--sustain-seconds", "0.2",
                           "--backend-module", "stub_backend"] + extra, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_gpus2_self_launch_reports_two_ranks():
    p = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["backend"] == "stub" and out["data"].startswith("stub")      # never mistaken for a measurement
    assert out["config"]["launch"] == "torch.distributed.run"
    g = out["gather"]
    assert g["backend"] == "gloo" and g["records_last_step"] == 5 + 6                   # rank 0: 5 records, rank 1: 6
    assert g["peak_records_per_rank"] == [5, 6] and g["capacity_per_rank"] >= 1024
    assert out["value"] > 0 and out["steps"] == 3
    # what a driver's --gpus N record must carry besides value: per-rank step times (value uses the MAX) and the gather alone
    rk = out["rank_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["min"] <= rk["max"] and abs(rk["max"] - out["ms_per_step"]) < 1e-6
    assert g["ms_per_step"] is not None and g["ms_per_step"] >= 0.0
    assert "roofline" in out and out["roofline"]["mode"]
    # the sustained leg: the same step run on after the timed region, the same number of steps on every rank (the gather is a collective)
    su = out["sustained"]
    assert su["steps"] >= 150 and set(su["ms_per_step"]) == {"first50", "mid50", "last50", "all"} and su["drift_last_vs_first"] is not None
    assert su["clock_mhz"] is None                    # the stub has no GPU to probe


def test_gpus8_self_launch_on_gloo():
    """The shape of the driver's 8-GPU run (BASELINE configs[3]: channels sharded over eight ranks, one gather per step), rehearsed
    with eight gloo ranks on the CPU and the stub backend."""
    p = _run(["--gpus", "8"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "channel-shard x8"
    g = out["gather"]
    assert g["peak_records_per_rank"] == [5, 6, 7, 8, 8, 8, 8, 8] and g["records_last_step"] == 58      # min(channels, 5 + rank) records per rank
    assert len(out["rank_ms_per_step"]["per_rank"]) == 8


def test_gpus8_rehearsal_full_size_inputs_and_cpu_baseline():
    """First-time-right check for the driver's `python bench.py --gpus 8` (no 8-GPU node has run it yet): the
    logistics of the real launch on the CPU - the launcher parent times the CPU baseline before any rank exists, eight ranks start
    through torch.distributed.run, EVERY rank synthesises its full 1024-channel input set (bench.make_inputs, what HipBackend stages),
    the per-step gather runs on gloo - and the one line carries everything the driver's record needs.  The wall clock of this
    rehearsal (8 host cores here, 16 on a GPU box's quota) is the host-side floor of the real run's driver_run_s; DESIGN 6 states it."""
    import time
    t0 = time.monotonic()
    p = _run(["--gpus", "8", "--force-cpu-baseline", "--cpu-baseline-seconds", "3"], env_extra={"MSK144_STUB_REAL_INPUTS": "1"}, timeout=900, channels=1024)
    wall = time.monotonic() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["channels_per_gpu"] == 1024 and out["config"]["parallelism"] == "channel-shard x8"
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["kernel"] and out["roofline"]["launches_per_step"] >= 1
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and "launcher parent" in cpu["sample"]
    assert len(out["rank_ms_per_step"]["per_rank"]) == 8 and len(out["gather"]["peak_records_per_rank"]) == 8
    assert out["gather"]["capacity_per_rank"] == 32 * 1024 and out["gather"]["records_last_step"] == sum(5 + r for r in range(8))
    assert "sustained" in out and out["sustained"]["steps"] >= 150
    assert wall < 600, wall
    print(f"8-rank rehearsal wall clock: {wall:.1f} s")


def test_gpus1_through_launcher_matches_contract():
    p = _run(["--gpus", "1", "--launcher"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["gather"]["records_last_step"] == 5


def test_world_size_mismatch_is_refused():
    """Under torch.distributed.run with WORLD_SIZE != --gpus the worker must refuse rather than print a mislabelled line."""
    p = _run(["--gpus", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_gather_overflow_fails_the_run():
    """A rank that decodes more records than the gather capacity must fail the run, not truncate silently."""
    p = _run(["--gpus", "2"], env_extra={"MSK144_STUB_OVERFLOW": "1"})
    assert p.returncode != 0
    assert "gather" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
