"""The oracle must keep reproducing the committed fixtures (tests/golden/*.npz, made by
tests/golden/make_golden.py).  The reference has no fixtures of its own: parity unpinned."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


def test_fixtures_present():
    assert {"audio_fir", "audio_fft", "iq_fir", "audio_noise"} <= set(CASES)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(orc, name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    o = orc.Oracle(center=float(g["center"]), width=float(g["width"]), step=float(g["step"]), depth=int(g["depth"]),
                   nbadsync_threshold=int(g["nbadsync_threshold"]), threads=4)
    if int(g["read_mode"]) == 1:
        cd = o.frontend_audio(g["input"], int(g["analytic_method"]))
    else:
        cd = o.frontend_iq(g["input"])
    assert np.array_equal(cd.view(np.uint32), g["analytic"].view(np.uint32))
    items, idx = o.decode_window(cd)
    assert np.array_equal(items["pos"], g["pos"])
    assert np.array_equal(items["xb"].view(np.uint32), g["xb"].view(np.uint32))
    assert np.array_equal(items["nbadsync"], g["nbadsync"])
    assert np.array_equal(items["softbits_wo_sync"].view(np.uint32), g["llr"].view(np.uint32))
    assert np.array_equal(idx, g["indexes"])
    assert np.array_equal(items["is_message_present"], g["present"])
    pres = g["present"] == 1
    assert np.array_equal(items["message"][pres], g["message"][pres])
    assert np.array_equal(items["ldpc_num_iterations"][pres], g["iters"][pres])
    if pres.any():
        assert all(np.array_equal(m, g["tx_message"].astype(np.int8)) for m in items["message"][pres])
