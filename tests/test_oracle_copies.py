"""CPU: the property of the REFERENCE ALGORITHM (as the oracle restates it) that the HIP path's copy hand-over rests on.

Two slots of one (frequency, pattern) group whose positions are congruent modulo the ring (5184: the scan walks 5376 positions) or, for masks
111111 / 100100, modulo 864 / 2592 fold the same frames (softbits_kernel.cuh:56-83).  The reference demodulates and decodes both; the HIP
path, in blocked staging, computes the lower slot and lets the copy report its result (msk144cudecoder_amd/csrc/softbits.hip, index.hip;
DESIGN.md 3).  That is exact for ring-wrap twins (the same computation) and exact up to float association for periodic copies.  Here the
oracle computes EVERY slot on its own, as the reference does, and the test counts how often a copy's own result differs from its lower
slot's: never, on these windows - the quantity the GPU soak (tests/soak_list_identity.py) measures on the kernels."""
import numpy as np
import pytest

from msk144cudecoder_amd import synth

PERIOD = {5: 864, 6: 2592}


def _copies(items):
    """[(copy item, lower item)] per the rule of softbits_kernel<true>: lowest slot of the group with the same residue."""
    pos = items["pos"].astype(np.int64) % 5184
    out = []
    for g0 in range(0, len(items), 8):
        r = pos[g0:g0 + 8] % PERIOD.get(int(items["pattern_idx"][g0]), 5184)
        for sl in range(1, 8):
            same = np.nonzero(r[:sl] == r[sl])[0]
            if len(same):
                out.append((g0 + sl, g0 + int(same[0])))
    return out


@pytest.mark.parametrize("depth,width,seed", [(6, 160.0, 1), (6, 160.0, 2), (8, 60.0, 3)])
def test_a_copy_decodes_like_its_lower_slot(orc, depth, width, seed):
    cfg = dict(center=1500.0, width=width, step=1.0, depth=depth, nbadsync_threshold=3)
    rng = np.random.default_rng(9000 + seed)
    msg = synth.random_message(rng)
    pings = [synth.Ping(msg, int(rng.integers(0, 600)), 6, 1500.0 + float(rng.uniform(-20, 20)), 2.0, float(rng.uniform(0, 6.28)))] if seed != 2 else []
    x = synth.synth_audio(5184, pings, 1000.0, rng)
    o = orc.Oracle(threads=8, **cfg)
    items, _ = o.decode_window(o.frontend_audio(x, 2))
    pairs = _copies(items)
    five = items["pattern_idx"] == 5
    n_five_copies = sum(1 for c, _ in pairs if items["pattern_idx"][c] == 5)
    assert n_five_copies > 0.5 * int(five.sum())                       # most slots of mask 111111 are copies
    assert any(items["pattern_idx"][c] < 5 for c, _ in pairs)            # and a few ring-wrap twins elsewhere
    differ = dict(nbadsync=0, accept=0, iterations=0, hard_errors=0, payload=0, llr_max_rel=0.0)
    accepted_copies = 0
    for c, l in pairs:
        a, b = items[c], items[l]
        differ["nbadsync"] += int(a["nbadsync"] != b["nbadsync"])
        if a["nbadsync"] != b["nbadsync"]:
            continue
        d = np.abs(a["softbits_wo_sync"].astype(np.float64) - b["softbits_wo_sync"]) / np.maximum(1.0, np.abs(b["softbits_wo_sync"]))
        if np.isfinite(d).all():
            differ["llr_max_rel"] = max(differ["llr_max_rel"], float(d.max()))
        if a["nbadsync"] > cfg["nbadsync_threshold"]:
            continue
        differ["accept"] += int(a["is_message_present"] != b["is_message_present"])
        if a["is_message_present"] and b["is_message_present"]:
            accepted_copies += 1
            differ["iterations"] += int(a["ldpc_num_iterations"] != b["ldpc_num_iterations"])
            differ["hard_errors"] += int(a["ldpc_num_hard_errors"] != b["ldpc_num_hard_errors"])
            differ["payload"] += int(not np.array_equal(a["message"], b["message"]))
    # ring-wrap twins are the same computation; periodic copies add the same frames in another order: LLRs within a few ulp
    assert differ["llr_max_rel"] < 1e-5, differ
    assert differ["nbadsync"] == 0 and differ["accept"] == 0 and differ["iterations"] == 0 and differ["hard_errors"] == 0 and differ["payload"] == 0, differ
    if pings:
        assert accepted_copies >= 1                                      # the ping's copies decode, like their lower slots
