"""Generate tests/golden/*.npz: seeded inputs and the CPU oracle's outputs for them.

The reference has no golden vectors of its own and cannot run here (CUDA), so these fixtures are
produced by the oracle restatement (oracle/msk144_oracle.cpp) - "parity unpinned", see DESIGN.md.
They freeze the oracle's behaviour: tests/test_golden.py checks that the oracle still reproduces
them (CPU) and that the HIP path matches them (GPU).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from msk144cudecoder_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    # name: (read_mode, analytic_method, decoder config, ping spec)
    "audio_fir": dict(read_mode=1, method=2, cfg=dict(center=1500.0, width=20.0, step=2.0, depth=6, nbadsync_threshold=2),
                      ping=dict(start=700, n_frames=7, freq=1506.0, snr=3.0, phase=1.0), sigma=1000.0, seed=3),
    "audio_fft": dict(read_mode=1, method=1, cfg=dict(center=1500.0, width=12.0, step=1.0, depth=4, nbadsync_threshold=1),
                      ping=dict(start=123, n_frames=4, freq=1497.0, snr=6.0, phase=0.3), sigma=1000.0, seed=11),
    "iq_fir": dict(read_mode=2, method=2, cfg=dict(center=0.0, width=16.0, step=2.0, depth=8, nbadsync_threshold=3),
                   ping=dict(start=2000, n_frames=5, freq=-4.0, snr=2.0, phase=2.0), sigma=20.0, seed=29),
    "audio_noise": dict(read_mode=1, method=2, cfg=dict(center=1500.0, width=8.0, step=2.0, depth=6, nbadsync_threshold=3),
                        ping=None, sigma=1000.0, seed=5),
}


def make_case(name, spec):
    rng = np.random.default_rng(spec["seed"])
    pings = []
    msg = np.zeros(77, dtype=np.uint8)
    if spec["ping"]:
        msg = synth.random_message(rng)
        p = spec["ping"]
        pings = [synth.Ping(msg, p["start"], p["n_frames"], p["freq"], p["snr"], p["phase"])]
    if spec["read_mode"] == 1:
        x = synth.synth_audio(5184, pings, spec["sigma"], rng)
    else:
        x = synth.synth_iq(5184, pings, spec["sigma"], rng)
    o = orc.Oracle(threads=8, **spec["cfg"])
    cd = o.frontend_audio(x, spec["method"]) if spec["read_mode"] == 1 else o.frontend_iq(x)
    items, idx = o.decode_window(cd)
    cfg = spec["cfg"]
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        input=x, tx_message=msg, read_mode=spec["read_mode"], analytic_method=spec["method"],
        center=cfg["center"], width=cfg["width"], step=cfg["step"], depth=cfg["depth"], nbadsync_threshold=cfg["nbadsync_threshold"],
        analytic=cd, seg_power=orc.segment_power(cd),
        pos=items["pos"], xb=items["xb"], f0=items["f0"], num_avg=items["num_avg"], nbadsync=items["nbadsync"],
        llr=items["softbits_wo_sync"], present=items["is_message_present"], iters=items["ldpc_num_iterations"],
        nhard=items["ldpc_num_hard_errors"], message=items["message"], indexes=idx)
    print(name, "items", len(items), "gated", len(idx), "decodes", int(items["is_message_present"].sum()))


if __name__ == "__main__":
    orc.build()
    for n, s in CASES.items():
        make_case(n, s)
