"""The checker's own logic (CPU): a BP difference may only be waved through when the oracle's decision is demonstrably
unstable at that input (tests/parity.py: verify_marginal_bp)."""
import numpy as np
import pytest

from msk144cudecoder_amd import synth

import parity


def _clean_llr(seed, sigma):
    rng = np.random.default_rng(seed)
    msg = synth.random_message(rng)
    cw = synth.encode_message(msg)
    return (((2.0 * cw - 1.0) + rng.normal(0, sigma, 128)) * (2 / sigma ** 2)).astype(np.float32), msg


def test_stable_difference_is_rejected(orc):
    llr, msg = _clean_llr(3, 0.3)                                  # decodes at iteration 0, far from any threshold
    out = parity.bp_outcome(orc, llr)
    assert out[0] and out[1] == 0 and out[3] == bytes(msg)
    with pytest.raises(AssertionError, match="STABLE"):
        parity.verify_marginal_bp(orc, llr, (False, -1), seed=1)     # a kernel that rejected this codeword is simply wrong
    with pytest.raises(AssertionError, match="STABLE"):
        parity.verify_marginal_bp(orc, llr, (True, 3), seed=1)       # ... or that needed 3 iterations
    noise = np.random.default_rng(9).normal(0, 5.5, 128).astype(np.float32)
    assert not parity.bp_outcome(orc, noise)[0]
    with pytest.raises(AssertionError, match="STABLE"):
        parity.verify_marginal_bp(orc, noise, (True, 4), seed=1)     # a kernel that "decoded" noise


def test_identical_items_report_no_flips(orc):
    items = np.zeros(4, dtype=[("nbadsync", "<i4"), ("softbits_wo_sync", "<f4", (128,)), ("is_message_present", "u1"),
                               ("ldpc_num_iterations", "<i4"), ("ldpc_num_hard_errors", "<i4"), ("message", "i1", (77,))])
    for k in range(4):
        llr, msg = _clean_llr(10 + k, 0.45)
        ok, it, nh, m = parity.bp_outcome(orc, llr)
        items["softbits_wo_sync"][k] = llr
        items["is_message_present"][k] = ok
        items["ldpc_num_iterations"][k] = max(it, 0)
        items["ldpc_num_hard_errors"][k] = max(nh, 0)
        if ok:
            items["message"][k] = np.frombuffer(m, dtype=np.uint8)
    rep = parity.compare_ldpc_against_oracle_on_gpu_llrs(orc, items, threshold=3)
    assert rep["checked"] == 4 and rep["marginal_flips"] == 0
    items["ldpc_num_iterations"][0] += 2                              # a wrong iteration count on a stable case
    with pytest.raises(AssertionError, match="STABLE"):
        parity.compare_ldpc_against_oracle_on_gpu_llrs(orc, items, threshold=3)


def test_count_limits_follow_the_measured_rates():
    """The regression limits the GPU comparators enforce (tests/parity.py): Poisson quantiles at three times the measured rates."""
    assert parity.near_tie_limit(24048) == 3 and parity.nbadsync_marginal_limit(24048) == 1 and parity.bp_marginal_limit(24048) == 2    # the deep window
    assert parity.near_tie_limit(40) == 0 and parity.near_tie_limit(528) == 1 and parity.near_tie_limit(128064) == 8
    assert parity.count_limit(1e-6, 0) == 0
    # monotone in the number of slots, and the false-alarm probability at the limit is below 1e-3
    import math
    last = 0
    for n in (10, 100, 1000, 10 ** 4, 10 ** 5, 10 ** 6):
        k = parity.near_tie_limit(n)
        assert k >= last
        last = k
        lam = 3.0 * parity.NEAR_TIE_RATE * n
        assert 1.0 - sum(math.exp(-lam) * lam ** i / math.factorial(i) for i in range(k + 1)) < 1e-3
    assert parity.near_tie_limit(10 ** 9) > parity.near_tie_limit(10 ** 8) > parity.near_tie_limit(10 ** 7) >= last      # normal-tail branch
    assert parity.TOL_LLR_REGRESSION * 10 == pytest.approx(parity.TOL_LLR)
