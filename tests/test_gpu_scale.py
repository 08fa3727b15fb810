"""-m gpu: the product program at BASELINE scale (main.cu:261-422 with its 210 ms watchdog :398-403): 1024
deep-configuration streams, each a FIFO fed in real time, through msk144hipdecoder's pipelined multi-stream loop on one GPU."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_1024_realtime_streams_deep_config_no_late_hops():
    import host_scale
    res = host_scale.run(1024, 20, pace_ms=216.0)
    assert res["returncode"] == 0 and res["feeder_errors"] == 0, res
    assert res["stream_hops"] == 1024 * 21                      # nothing dropped
    assert res["late_hops"] == 0 and res["worst_latency_ms"] <= 210, res
    assert res["pings_decoded"] >= 0.7 * res["streams_with_ping"], res     # 0 dB pings of 3-6 frames: most, not all, decode
    rows = res["host_ms_per_batch"]
    # the host side of a hop (everything but waiting for the GPU) must leave the 216 ms period to the GPU
    busy = sum(rows[k]["mean_ms"] for k in rows if not k.startswith(("wait", "batch released")))
    assert busy < 100.0, rows
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    import json
    with open(os.path.join(ROOT, "gpurun_out", "host_scale_1024.json"), "w") as f:
        json.dump(res, f, indent=1)
