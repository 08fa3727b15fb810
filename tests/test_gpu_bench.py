"""-m gpu: bench.py itself - the line the driver records - through the N-rank launcher with the RCCL record gather (one rank on the
one-GPU box) against the plain single-process run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2"] + extra, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_launcher_line_matches_the_plain_run():
    plain = _bench(["--no-cpu-baseline", "--sustain-seconds", "0"])       # without the sustained leg both runs end on the same window
    dist = _bench(["--gpus", "1", "--launcher", "--no-cpu-baseline", "--sustain-seconds", "0"])
    assert plain["n_gpus"] == dist["n_gpus"] == 1
    assert plain["config"]["launch"] == "single process" and dist["config"]["launch"] == "torch.distributed.run"
    g = dist["gather"]
    assert g["backend"] == "nccl" and g["records_last_step"] == dist["decodes_last_step"] == plain["decodes_last_step"]
    assert g["peak_records_per_rank"][0] <= g["capacity_per_rank"] and g["ms_per_step"] is not None
    assert abs(dist["value"] / plain["value"] - 1.0) < 0.05, (dist["value"], plain["value"])
    for line in (plain, dist):
        assert line["config"]["softbits_gate_early"] is True and line["config"]["llr_store"] == "blocked/128"
        assert line["roofline"]["kernel"] == "ldpc_kernel" and 0.0 < line["roofline"]["frac"] < 1.0
        assert line["rank_ms_per_step"]["max"] == pytest.approx(line["ms_per_step"])
        # `value` counts reported slots; the line says how many were handed over and carries the every-slot figure beside it
        ho, every = line["copy_handover"], line["value_every_slot_decoded"]
        assert ho["enabled"] is True and 0.10 < ho["share_of_slots"] < 0.20 and line["config"]["copies_computed_once"] is True
        assert every["slots_handed_over"] == 0 and every["records_last_step"] > 0
        assert 0.80 < every["value"] / line["value"] < 0.97, (every["value"], line["value"])
        assert f"{100.0 * ho['share_of_slots']:.1f} %" in line["config"]["value_counts"]


def test_distributed_line_carries_the_cpu_baseline():
    """What a driver's `bench.py --gpus N` prints must be gradeable: roofline, cpu_baseline and gather in one line."""
    dist = _bench(["--gpus", "1", "--launcher"])
    cb = dist["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "-march=native" in cb["flags"]
    assert "gather" in dist and "roofline" in dist
    assert "timed in the launcher parent" in cb["sample"]
    # the sustained leg: the same step held for ~10 s; markers on the decode stream must SEE the kernels (a marker on another
    # stream reads a fraction of the step time), the probe must read a shader clock, and the gather of the leg's last step must
    # have carried that step's records (bench.py validates it and would have exited non-zero)
    su = dist["sustained"]
    assert su["seconds"] >= 16.0 and su["steps"] >= 300
    for k in ("first50", "mid50", "last50"):
        assert abs(su["ms_per_step"][k] / dist["ms_per_step"] - 1.0) < 0.10, (k, su["ms_per_step"], dist["ms_per_step"])
    assert abs(su["drift_last_vs_first"]) < 0.05
    assert all(1200.0 < su["clock_mhz"][k] < 2700.0 for k in ("first", "mid", "last")), su["clock_mhz"]
    assert dist["gather"]["records_last_step"] == dist["decodes_last_step"]


def test_every_slot_flag_times_the_region_with_the_hand_over_off():
    line = _bench(["--every-slot", "--no-cpu-baseline", "--sustain-seconds", "0", "--no-extra-legs"])
    assert line["copy_handover"] == {"enabled": False, "slots_handed_over_last_step": 0, "share_of_slots": 0.0}
    assert line["config"]["copies_computed_once"] is False and "fully evaluated" in line["config"]["value_counts"]
    assert "value_every_slot_decoded" not in line and "computed again, as in the reference" in line["roofline"]["mode"]
