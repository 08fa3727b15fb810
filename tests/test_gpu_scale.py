"""-m gpu: the product program at BASELINE scale (main.cu:261-422 with its 210 ms watchdog :398-403): 1024
deep-configuration streams, each a FIFO fed in real time, through msk144hipdecoder's pipelined multi-stream loop on one GPU."""
import json
import os
import re
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


DEEP_CFG = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
LINE_SAMPLE = (0, 1, 2, 4, 8, 12)     # four streams with a ping (0, 4, 8, 12), two without (1, 2): every line against the oracle-driven decoder
RATE_SAMPLE = tuple(range(0, 96, 4))  # 24 streams with a ping: decoded or not, stream by stream, against the oracle


def _oracle_prints_ping(oracle_cli, stream, start, frames, bits):
    """Would the oracle-driven decoder print the stream's transmitted payload?  Only the windows that overlap the ping can hold it."""
    n_win = (len(stream) - 5184) // 2592 + 1
    hit = [w for w in range(n_win) if not (w * 2592 + 5184 <= start or w * 2592 >= start + frames * 864)]
    sub = stream[hit[0] * 2592: hit[-1] * 2592 + 5184]
    return bits in oracle_cli.printed_payloads(sub, DEEP_CFG, 1, 2, threads=16)


def test_1024_realtime_streams_deep_config_no_late_hops(orc, parity_report):
    import host_scale
    from oracle import oracle_cli
    res = host_scale.run(1024, 20, pace_ms=216.0, keep_lines=LINE_SAMPLE)
    assert LINE_SAMPLE and RATE_SAMPLE
    assert res["returncode"] == 0 and res["feeder_errors"] == 0, res
    assert res["stream_hops"] == 1024 * 21                      # nothing dropped
    assert res["late_hops"] == 0 and res["worst_latency_ms"] <= 210, res
    # ---- what the 1024 streams printed, against the oracle ----
    meta = {}
    streams, sent = host_scale.make_streams(1024, 20, meta=meta)            # the harness's own seeded streams, regenerated
    # 1. six streams, all 21 windows each: every output line (snr, f0, num_avg, nbadsync, pattern, text) is the oracle-driven
    #    CPU decoder's, in order; the 77 printed bits of the pinged streams are the transmitted payload
    for c in LINE_SAMPLE:
        got = [re.sub(r"' bits='[01]{77}", "", re.sub(r"date=\d{14}", "date=X", l)) for l in res["lines_by_stream"][c]]
        want = oracle_cli.decode_stream(streams[c], DEEP_CFG, 1, 2, threads=16)
        assert got == want, (c, got[:3], want[:3])
        # the printed bits are payloads the oracle accepted in that stream; a stream is "decoded" iff its transmitted payload is among them
        accepted = set()
        if got:
            oracle_cli.decode_stream(streams[c], DEEP_CFG, 1, 2, threads=16, payloads=accepted)
        printed = set(re.findall(r"bits='([01]{77})'", "\n".join(res["lines_by_stream"][c])))
        assert printed <= accepted and ((c in res["decoded_streams"]) == (c in sent and sent[c] in printed))
    # 2. decoded-or-not for 24 pinged streams, stream by stream: the program's verdict is the oracle's
    oracle_says = {c: _oracle_prints_ping(oracle_cli, streams[c], meta[c][0], meta[c][1], sent[c]) for c in RATE_SAMPLE}
    program_says = {c: c in res["decoded_streams"] for c in RATE_SAMPLE}
    assert program_says == oracle_says
    # 3. the overall decode rate is bounded by the oracle's rate on that sample (not by a hand-picked constant): 0 dB pings of 3-6
    #    frames decode mostly, not always - in the oracle too
    rate_o = float(np.mean(list(oracle_says.values())))
    sigma = float(np.sqrt(max(rate_o * (1.0 - rate_o), 0.01) / len(RATE_SAMPLE)))
    rate_p = res["pings_decoded"] / res["streams_with_ping"]
    assert rate_p >= rate_o - 3.0 * sigma - 0.02, (rate_p, rate_o, sigma)
    parity_report("host_scale_1024_vs_oracle", dict(line_sample=list(LINE_SAMPLE), lines_compared=sum(len(res["lines_by_stream"][c]) for c in LINE_SAMPLE),
                                                     rate_sample=len(RATE_SAMPLE), oracle_rate_on_sample=rate_o, program_rate_all_256=rate_p, verdicts_equal=True))
    res.pop("lines_by_stream", None)
    rows = res["host_ms_per_batch"]
    # the host side of a hop (everything but waiting for the GPU) must leave the 216 ms period to the GPU
    busy = sum(rows[k]["mean_ms"] for k in rows if not k.startswith(("wait", "batch released")))
    assert busy < 100.0, rows
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "host_scale_1024.json"), "w") as f:
        json.dump(res, f, indent=1)
