"""-m gpu: HIP path vs the committed golden fixtures (tests/golden/*.npz, made by the oracle)."""
import glob
import os

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _as_items(g, dtype):
    n = len(g["pos"])
    it = np.zeros(n, dtype=dtype)
    D = int(g["depth"])
    k = np.arange(n)
    it["block_idx"] = k // (D * 8)
    it["pattern_idx"] = (k % (D * 8)) // 8
    for src, dst in (("pos", "pos"), ("xb", "xb"), ("f0", "f0"), ("num_avg", "num_avg"), ("nbadsync", "nbadsync"), ("llr", "softbits_wo_sync"),
                     ("present", "is_message_present"), ("iters", "ldpc_num_iterations"), ("nhard", "ldpc_num_hard_errors"), ("message", "message")):
        it[dst] = g[src]
    return it


@pytest.mark.parametrize("name", CASES)
def test_golden(orc, hip, name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    cfg = dict(center=float(g["center"]), width=float(g["width"]), step=float(g["step"]), depth=int(g["depth"]),
               nbadsync_threshold=int(g["nbadsync_threshold"]))
    rm, am = int(g["read_mode"]), int(g["analytic_method"])
    with hip.HipDecoder(read_mode=rm, analytic_method=am, channels=1, **cfg) as d:
        (d.submit_audio if rm == 1 else d.submit_iq)(g["input"])
        d.decode()
        analytic = d.dump_analytic(0)
        seg = d.segment_power()[0]
        items_g = d.dump_candidates(0)
        idx_g = d.dump_indexes(0)
    if am == 2 or rm == 2:
        assert np.array_equal(analytic.view(np.uint32), g["analytic"].view(np.uint32))
        assert np.array_equal(seg.view(np.uint32), g["seg_power"].view(np.uint32))
    else:
        rms = np.sqrt(np.mean(np.abs(g["analytic"]) ** 2))
        assert np.abs(analytic - g["analytic"]).max() <= 1e-5 * rms
    items_o = _as_items(g, orc.ITEM_DTYPE)
    o = orc.Oracle(threads=4, **cfg)
    cd = g["analytic"]
    parity.compare_scan(o, cd, items_o, items_g)
    parity.compare_softbits(o, cd, items_o, items_g)
    if np.array_equal(items_g["nbadsync"], items_o["nbadsync"]):
        assert np.array_equal(idx_g, g["indexes"])
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o)
    if g["present"].any():
        assert parity.decoded_messages(items_g) == {bytes(g["tx_message"].astype(np.uint8))}
    else:
        assert items_g["is_message_present"].sum() == 0
