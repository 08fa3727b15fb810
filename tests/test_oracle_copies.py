"""CPU: the property of the REFERENCE ALGORITHM (as the oracle restates it) that the HIP path's copy hand-over rests on.

Two slots of one (frequency, pattern) group whose positions are congruent modulo the ring (5184: the scan walks 5376 positions) or, for masks
111111 / 100100, modulo 864 / 2592 fold the same frames (softbits_kernel.cuh:56-83).  The reference demodulates and decodes both; the HIP
path, in blocked staging, computes the lower slot and lets the copy report its result (msk144cudecoder_amd/csrc/softbits.hip, index.hip;
DESIGN.md 3; msk144_set_copy_handover switches it off).  That is exact for ring-wrap twins (the same computation) and exact up to float
association for periodic copies.  Here the oracle computes EVERY slot on its own, as the reference does, and the test counts how often a
copy's own result differs from its lower slot's: never, on these windows.  tests/soak_oracle_copies.py is the same comparison over
thousands of windows (profiles/r06_oracle_copy_soak.json: the measured rate and its bound); tests/soak_list_identity.py measures it on the kernels."""
import numpy as np
import pytest

from msk144cudecoder_amd import synth

import soak_oracle_copies as soak


def test_pairing_rule_on_a_handmade_group():
    """copy_pairs = the rule of softbits_kernel<true, true>: lowest congruent slot of the group; twin vs periodic."""
    items = np.zeros(16, dtype=[("pos", "<u4"), ("pattern_idx", "<u4")])
    items["pattern_idx"][:8] = 2
    items["pos"][:8] = [10, 5194, 11, 10, 900, 5184 + 900, 7, 8]          # slot 1 and 3 are twins of 0, slot 5 of 4 (mod 5184 only)
    items["pattern_idx"][8:] = 5
    items["pos"][8:] = [100, 964, 100 + 5184, 101, 1828, 965, 99, 963]      # mod 864: 100 100 100 101 100 101 99 99
    c, l, per = soak.copy_pairs(items)
    got = sorted(zip(c.tolist(), l.tolist(), per.tolist()))
    assert got == [(1, 0, False), (3, 0, False), (5, 4, False), (9, 8, True), (10, 8, False), (12, 8, True), (13, 11, True), (15, 14, True)]


@pytest.mark.parametrize("depth,width,seed", [(6, 160.0, 1), (6, 160.0, 2), (8, 60.0, 3)])
def test_a_copy_decodes_like_its_lower_slot(orc, depth, width, seed):
    cfg = dict(center=1500.0, width=width, step=1.0, depth=depth, nbadsync_threshold=3)
    rng = np.random.default_rng(9000 + seed)
    msg = synth.random_message(rng)
    pings = [synth.Ping(msg, int(rng.integers(0, 600)), 6, 1500.0 + float(rng.uniform(-20, 20)), 2.0, float(rng.uniform(0, 6.28)))] if seed != 2 else []
    x = synth.synth_audio(5184, pings, 1000.0, rng)
    o = orc.Oracle(threads=8, **cfg)
    items, _ = o.decode_window(o.frontend_audio(x, 2))
    c, l, per = soak.copy_pairs(items)
    five = items["pattern_idx"] == 5
    assert int((items["pattern_idx"][c] == 5).sum()) > 0.5 * int(five.sum())     # most slots of mask 111111 are copies
    assert (items["pattern_idx"][c] < 5).any()                                  # and a few ring-wrap twins elsewhere
    twins = soak.compare_pairs(items, c[~per], l[~per], 3)
    copies = soak.compare_pairs(items, c[per], l[per], 3)
    # ring-wrap twins are the same computation; periodic copies add the same frames in another order: LLRs within a few ulp
    assert twins["llr_max_rel"] == 0.0 and copies["llr_max_rel"] < 1e-5, (twins, copies)
    for r in (twins, copies):
        assert r["pairs"] > 0 and r["both_gated"] > 0
        assert all(r[k] == 0 for k in ("nbadsync_differs", "gate_differs", "accept_differs", "iterations_differ", "hard_errors_differ", "payload_differs",
                                       "reported_record_differs")), r
    if pings:
        assert twins["both_accepted"] + copies["both_accepted"] >= 1             # the ping's copies decode, like their lower slots
