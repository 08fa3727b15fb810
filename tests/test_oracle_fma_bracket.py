"""Do the decoder's DECISIONS survive the reference's own floating-point build?

The parity build of the oracle contracts nothing (-ffp-contract=off); the reference's CUDA binary contracts a*b+c into FMAs in
device code (nvcc default, CMakeLists.txt:130-132), and cannot be built here.  Two further builds of the SAME oracle
source bracket what that binary may compute (oracle/Makefile target `fma`, notes at the top of oracle/msk144_oracle.cpp):
"contract-fast" (gcc -ffp-contract=fast -mfma, everything fusable fused), "forced-fma" (complex products and the three
accumulate hot spots as fully fused chains), "cuda-libm" (glibc's sincosf / atan2f / hypotf / tanhf moved by deterministic
pseudo-random ulps within CUDA's documented error bounds: 2 / 3 / 3 / 2) and "cuda-like" (forced FMAs and moved math library
together).  Over the golden corpus, >= 200 fuzz configurations and one deep window this test
asserts, for each contracting build against the parity build on the same raw input:

  * payload sets identical, index lists identical wherever nbadsync is, accept / iteration identical,
  * every scan arg-max, nbadsync or BP difference is a VERIFIED near-tie by the rules the GPU parity tests use (tests/parity.py),
  * reals within the stated tolerances (xb 1e-4 relative, LLR 1e-3).

and counts how many decisions move (`python tests/test_oracle_fma_bracket.py --seeds N --report FILE` for a longer soak).  CPU only.
"""
import glob
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import parity  # noqa: E402
from test_gpu_fuzz import _case  # noqa: E402  (the seeded configuration / window generator of the GPU fuzz sweep)

VARIANTS = ("contract-fast", "forced-fma", "cuda-libm", "cuda-like")


def _frontend(o, x, read_mode, method):
    return o.frontend_audio(x, method) if read_mode == 1 else o.frontend_iq(x)


def compare_builds(orc, base, var, x, read_mode, method, tally):
    """One raw window through the parity build (`base`) and a contracting build (`var`); returns nothing, raises on a real
    mismatch, adds to `tally`."""
    cd0 = _frontend(base, x, read_mode, method)
    cd1 = _frontend(var, x, read_mode, method)
    rms = float(np.sqrt(np.mean(np.abs(cd0.astype(np.complex128)) ** 2)))
    if np.isfinite(rms) and rms > 0:
        d = float(np.max(np.abs(cd1.astype(np.complex128) - cd0.astype(np.complex128)))) / rms
        assert d <= 1e-5, ("front end moved by more than 1e-5 of the window rms under contraction", d)
        tally["frontend_max_rel"] = max(tally["frontend_max_rel"], d)
        tally["frontend_samples_changed"] += int(np.count_nonzero(cd0.view(np.uint32) != cd1.view(np.uint32)))
        tally["frontend_samples"] += 2 * cd0.size
    items0, idx0 = base.decode_window(cd0)
    items1, idx1 = var.decode_window(cd1)
    if not np.isfinite(items0["xb"]).all():
        # the all-zero window: NaNs everywhere in both builds; nothing decodes
        assert not items0["is_message_present"].any() and not items1["is_message_present"].any()
        return
    scan = parity.compare_scan(base, cd0, items0, items1, enforce_limits=False)      # limits: the HIP kernels' measured rates, not this comparison's
    sb = parity.compare_softbits(base, cd0, items0, items1, enforce_limits=False)
    same = (items0["pos"] == items1["pos"]) & (items0["nbadsync"] == items1["nbadsync"])
    ld = parity.compare_ldpc_items(orc, items0, items1, same, enforce_limits=False)
    # index list: exactly the items whose nbadsync passes, in item order - identical wherever nbadsync is
    thr = base.ctx.nbadsync_threshold
    assert np.array_equal(idx1, np.nonzero(items1["nbadsync"] <= thr)[0])
    if np.array_equal(items0["nbadsync"], items1["nbadsync"]):
        assert np.array_equal(idx0, idx1)
    else:
        # nbadsync differs slot by slot: either the 8 slots of a (frequency, pattern) group hold the same candidates in another
        # order (exact-tie positions of the periodic patterns, near-ties - all verified by compare_scan above), or a sync softbit
        # sat on zero (verified by compare_softbits).  What the gate DECIDES is the number of candidates it passes per group:
        g0 = (items0["nbadsync"] <= thr).reshape(-1, 8).sum(axis=1)
        g1 = (items1["nbadsync"] <= thr).reshape(-1, 8).sum(axis=1)
        tally["index_lists_reordered"] += 1
        tally["gate_counts_changed_groups"] += int(np.count_nonzero(g0 != g1))
    p0, p1 = parity.decoded_messages(items0), parity.decoded_messages(items1)
    if p0 != p1:
        tally["payload_sets_changed"] += 1
        tally["payload_set_cases"].append(dict(only_parity=len(p0 - p1), only_contracted=len(p1 - p0)))
    tally["windows"] += 1
    tally["slots"] += scan["total"]
    tally["scan_near_ties"] += scan["near_ties"]
    tally["periodic_groups_moved"] += scan["periodic_groups"]
    tally["nbadsync_marginal"] += sb["nbadsync_marginal"]
    tally["llr_max_abs_diff"] = max(tally["llr_max_abs_diff"], sb["llr_max_abs_diff"])
    tally["bp_compared"] += ld["compared"]
    tally["bp_marginal_flips"] += ld["marginal_flips"]
    tally["decodes"] += int(items0["is_message_present"].sum())


def new_tally():
    return dict(windows=0, slots=0, scan_near_ties=0, periodic_groups_moved=0, nbadsync_marginal=0, llr_max_abs_diff=0.0, bp_compared=0,
                bp_marginal_flips=0, decodes=0, index_lists_reordered=0, gate_counts_changed_groups=0, payload_sets_changed=0, payload_set_cases=[], frontend_max_rel=0.0,
                frontend_samples_changed=0, frontend_samples=0)


def run_bracket(orc, variant, seeds, deep=True, threads=8):
    L = orc.fma_lib(variant)
    tally = new_tally()
    # 1. the golden corpus (tests/golden/*.npz: the inputs; the frozen outputs are the parity build's)
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))):
        g = np.load(path)
        cfg = dict(center=float(g["center"]), width=float(g["width"]), step=float(g["step"]), depth=int(g["depth"]), nbadsync_threshold=int(g["nbadsync_threshold"]))
        compare_builds(orc, orc.Oracle(threads=threads, **cfg), orc.Oracle(threads=threads, library=L, **cfg), g["input"], int(g["read_mode"]), int(g["analytic_method"]), tally)
    # 2. the fuzz sweep's configurations and windows
    for seed in seeds:
        cfg, read_mode, method, x, _ = _case(seed)
        compare_builds(orc, orc.Oracle(threads=threads, **cfg), orc.Oracle(threads=threads, library=L, **cfg), x, read_mode, method, tally)
    # 3. one deep window (BASELINE width 500 / step 1 / depth 6 / threshold 3: 24 048 candidates) with two pings
    if deep:
        from msk144cudecoder_amd import synth
        rng = np.random.default_rng(4242)
        pings = [synth.Ping(synth.random_message(rng), 700, 5, 1411.3, 2.0, 0.9), synth.Ping(synth.random_message(rng), 2900, 3, 1642.8, 0.0, 2.1)]
        x = synth.synth_audio(5184, pings, 1000.0, rng)
        cfg = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
        compare_builds(orc, orc.Oracle(threads=threads, **cfg), orc.Oracle(threads=threads, library=L, **cfg), x, 1, 2, tally)
    return tally


@pytest.mark.parametrize("variant", VARIANTS)
def test_decisions_survive_fma_contraction(orc, variant):
    if not orc.host_has_fma():
        pytest.skip("host CPU without FMA: the contracting builds cannot run here")
    n = int(os.environ.get("MSK144_BRACKET_SEEDS", "200"))
    tally = run_bracket(orc, variant, range(n))
    print(variant, json.dumps({k: v for k, v in tally.items() if k != "payload_set_cases"}))
    assert tally["windows"] >= n                                   # (the all-zero golden window returns early)
    assert tally["payload_sets_changed"] == 0, tally["payload_set_cases"]
    # the bracket is not vacuous: contraction moves the front end's samples, the math-library builds (no transcendental in the FIR
    # front end) move the LLRs
    assert tally["frontend_samples_changed"] > 0 or tally["llr_max_abs_diff"] > 0.0
    # every moved decision was verified as a near-tie inside compare_*; their share must stay what "rounding-level" means
    assert tally["scan_near_ties"] <= 1e-3 * tally["slots"] and tally["bp_marginal_flips"] <= 1e-3 * max(tally["bp_compared"], 1)
    # a group's gate count may move only with a verified near-tie or a verified marginal sync softbit
    assert tally["gate_counts_changed_groups"] <= tally["scan_near_ties"] + tally["nbadsync_marginal"] + tally["periodic_groups_moved"]


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=1000)
    ap.add_argument("--report", default=None)
    a = ap.parse_args()
    from oracle import oracle as orc_mod
    orc_mod.build()
    rep = {}
    for v in VARIANTS:
        rep[v] = run_bracket(orc_mod, v, range(a.seeds))
        print(v, json.dumps(rep[v]), flush=True)
    if a.report:
        json.dump(dict(seeds=a.seeds, golden=4, deep_windows=1, variants=rep), open(a.report, "w"), indent=1)
