"""Tolerance-aware comparison of HIP results with the oracle (used by the -m gpu tests).

Bit-equality of real numbers between a CPU restatement and a GPU is not attainable (different
sincos/tanh/sqrt implementations, FMA, wave64 reduction trees), so the contract is:

  * front end (FIR audio / IQ)   bit-exact (same operation order, no FMA contraction)
  * front end (FFT)              |diff| <= 1e-5 * rms(window)
  * xb                           relative 1e-4
  * LLR                          |diff| <= 1e-3 * max(1, |llr|)
  * integers (pos, nbadsync, index list, accept/iter/nhard, 77 bits) exact, except where the deciding
    real value lies within tolerance of its threshold; every such exception is verified to BE a
    near-tie against the oracle and counted, never waved through: scan arg-max swaps against the oracle's
    own xb values, nbadsync against the oracle's sync softbits, BP accept/iteration differences by showing
    that the oracle's own decision flips under LLR perturbations of 1e-6 .. 1e-4 (verify_marginal_bp).

Masks 111111 and 100100 make the folded correlation exactly periodic in the position (864 and 2592
samples: the same frames are summed), so their arg-max has mathematically exact ties that only
rounding decides; for those two patterns positions are compared as multisets modulo the period.
"""
from __future__ import annotations

import numpy as np

TOL_XB_REL = 1e-4
TOL_LLR = 1e-3
PERIODIC = {5: 864, 6: 2592}  # pattern_idx -> period of xb(pos)

# Regression guards, tighter than the contractual tolerances above and derived from what the soaks MEASURED on the final kernels
# (profiles/r02_soak_fuzz.txt, r03_soak_fuzz.txt, r03_production_soak.json, r04_parity_report.json), so that a kernel change that
# stays inside the contract but moves the numbers is seen:
#   * max |dLLR| 1.1e-5 over 8000 fuzz cases and the full-size windows  ->  guard 1e-4 * max(1, |llr|), a tenth of the contract
#   * verified scan near-ties: 82 in 15.39 M slots (production soak), 11 in 2.48 M, 1 in 0.81 M  ->  5.3e-6 per slot
#   * verified nbadsync marginals: 8 in 15.39 M candidates                                       ->  5.2e-7 per candidate
#   * verified marginal BP decisions: 0 in 2.27 M decodes (rule of three: < 1.3e-6)              ->  1.3e-6 per decode
#   * periodic-pattern groups that do not even agree modulo the period (fall back to the near-tie rule): 0 ever seen; counted with
#     the near-ties
# A test's limit is the count a Poisson variable with THREE times the measured rate exceeds with probability < 1e-3 (count_limit):
# the deep window (24 048 slots) may show 3 near-ties, 1 nbadsync marginal, 2 marginal BP decisions - observed 0 / 0 / 0 - where
# the round-4 limits were 24 / 8 / 4.  The rates are those of HIP kernels against the oracle; comparisons of another kind (the oracle's
# FMA-contracting builds against its parity build, tests/test_oracle_fma_bracket.py) pass enforce_limits=False and keep their own asserts.
TOL_LLR_REGRESSION = 1e-4
NEAR_TIE_RATE = 5.3e-6
NBADSYNC_MARGINAL_RATE = 5.2e-7
BP_MARGINAL_RATE = 1.3e-6


def count_limit(rate: float, n: int, factor: float = 3.0, p_false_alarm: float = 1e-3) -> int:
    """Smallest k with P(X > k) < p_false_alarm for X ~ Poisson(factor * rate * n)."""
    import math
    lam = factor * rate * max(int(n), 0)
    if lam > 30.0:  # normal tail (exp(-lam) underflows long before the sum would matter): P(Z > 3.3) < 5e-4
        return int(lam + 3.3 * math.sqrt(lam)) + 1
    term = math.exp(-lam)
    cdf = term
    k = 0
    while 1.0 - cdf >= p_false_alarm:
        k += 1
        term *= lam / k
        cdf += term
    return k


def near_tie_limit(slots: int) -> int:
    return count_limit(NEAR_TIE_RATE, slots)


def nbadsync_marginal_limit(candidates: int) -> int:
    return count_limit(NBADSYNC_MARGINAL_RATE, candidates)


def bp_marginal_limit(decodes: int) -> int:
    return count_limit(BP_MARGINAL_RATE, decodes)


def handed_over_records(dec, records):
    """Mask over `records` (the result list `dec` has just decoded with copies handed over): True where the record's slot was NOT in its
    channel's index list, i.e. it was never demodulated or decoded itself and reports the result of a lower slot of its group
    (csrc/softbits.hip, index.hip).  Reads the index lists of the channels that have records; call before the next decode."""
    mask = np.zeros(len(records), dtype=bool)
    for ch in np.unique(records["channel"]):          # channel base 0: record channels are the handle's channels
        sel = np.nonzero(records["channel"] == ch)[0]
        mask[sel] = ~np.isin(records["item"][sel], dec.dump_indexes(int(ch)))
    return mask


def copy_class(items, k):
    """Slots of one (frequency, pattern) group that fold the same frames - ring-wrap twins, the periodic copies of masks 111111 / 100100 -
    are one event when something marginal happens to them: (block, pattern, position modulo the ring and the pattern's period)."""
    p_idx = int(items["pattern_idx"][k])
    return (int(items["block_idx"][k]), p_idx, (int(items["pos"][k]) % 5184) % PERIODIC.get(p_idx, 5184))


def llr_close(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) <= TOL_LLR * np.maximum(1.0, np.abs(a))


def compare_scan(o, cd, items_o, items_g, near_tie_factor: float = 1.0, enforce_limits: bool = True):
    """Returns dict(exact, near_ties, periodic_groups, periodic_fallbacks, near_tie_limit).  Raises AssertionError on a real mismatch,
    and when the verified near-ties (plus periodic groups that needed the near-tie rule) exceed the measured-rate limit for this many
    slots (near_tie_factor widens it for inputs with fewer significant bits, e.g. denormals)."""
    D = o.D
    n = len(items_o)
    assert len(items_g) == n
    exact = 0
    near = 0
    groups = 0
    fallbacks = 0
    xb_cache = {}

    def xb_all(b, p):
        key = (b, p)
        if key not in xb_cache:
            xb_cache[key] = o.scan_xb(cd, b, p)
        return xb_cache[key]

    for g0 in range(0, n, 8):
        b = int(items_o["block_idx"][g0])
        p = int(items_o["pattern_idx"][g0])
        po = items_o["pos"][g0:g0 + 8].astype(np.int64)
        pg = items_g["pos"][g0:g0 + 8].astype(np.int64)
        xo = items_o["xb"][g0:g0 + 8].astype(np.float64)
        xg = items_g["xb"][g0:g0 + 8].astype(np.float64)
        scale = max(xo.max(), 1e-30)
        # the 8 stored values must agree as a sorted list whatever the slot order
        assert np.all(np.abs(np.sort(xo) - np.sort(xg)) <= TOL_XB_REL * scale), (b, p, xo, xg)
        if np.array_equal(po, pg):
            assert np.all(np.abs(xo - xg) <= TOL_XB_REL * scale), (b, p, xo, xg)
            exact += 8
            continue
        ref = xb_all(b, p).astype(np.float64)
        # every GPU slot must carry the true correlation value of its own position
        assert np.all(np.abs(ref[pg] - xg) <= TOL_XB_REL * scale), (b, p, pg, xg, ref[pg])
        if p in PERIODIC:
            per = PERIODIC[p]
            if sorted((po % per).tolist()) != sorted((pg % per).tolist()):
                # not even the same positions modulo the period: only acceptable as an ordinary verified near-tie, and counted as one
                assert _near_tie_sets(ref, po, pg, scale), (b, p, po, pg)
                fallbacks += 1
            groups += 1
        else:
            # a differing slot is acceptable only when the oracle itself sees a near-tie between the
            # two positions (arg-max decided inside the tolerance)
            assert _near_tie_sets(ref, po, pg, scale), (b, p, po, pg, xo, xg)
            near += int((po != pg).sum())
            exact += int((po == pg).sum())
    limit = count_limit(NEAR_TIE_RATE * near_tie_factor, n)
    rep = dict(exact=exact, near_ties=near, periodic_groups=groups, periodic_fallbacks=fallbacks, near_tie_limit=limit, total=n)
    assert not enforce_limits or near + fallbacks <= limit, ("more verified near-ties than three times the measured rate allows", rep)
    return rep


def _near_tie_sets(ref, po, pg, scale):
    """True when the sorted oracle values at the oracle's and at the GPU's positions agree within
    tolerance, i.e. swapping one set for the other changes no stored value by more than the tolerance."""
    a = np.sort(ref[po])
    b = np.sort(ref[pg])
    return bool(np.all(np.abs(a - b) <= 4 * TOL_XB_REL * scale))


def expected_softbits(o, cd, items_o, items_g):
    """Oracle LLRs / nbadsync evaluated at the GPU's positions (so a tolerated scan difference does not
    leak into the softbits comparison)."""
    exp_llr = items_o["softbits_wo_sync"].copy()
    exp_nb = items_o["nbadsync"].copy()
    diff = np.nonzero(items_o["pos"] != items_g["pos"])[0]
    for k in diff:
        soft, llr, nb = o.softbits_at(cd, int(items_o["block_idx"][k]), int(items_o["pattern_idx"][k]), int(items_g["pos"][k]))
        exp_llr[k] = llr
        exp_nb[k] = nb
    return exp_llr, exp_nb


def compare_softbits(o, cd, items_o, items_g, enforce_limits: bool = True):
    exp_llr, exp_nb = expected_softbits(o, cd, items_o, items_g)
    got = items_g["softbits_wo_sync"]
    finite = np.isfinite(exp_llr).all(axis=1)
    ok = llr_close(exp_llr[finite], got[finite])
    bad_rows = np.nonzero(~ok.all(axis=1))[0]
    assert len(bad_rows) == 0, ("LLR out of tolerance", bad_rows[:5], np.abs(exp_llr[finite] - got[finite]).max())
    # regression guard (not the contract): ten times what the soaks measured, a tenth of the contract
    e64, g64 = exp_llr[finite].astype(np.float64), got[finite].astype(np.float64)
    worst_rel = float((np.abs(e64 - g64) / np.maximum(1.0, np.abs(e64))).max()) if finite.any() else 0.0
    assert not enforce_limits or worst_rel <= TOL_LLR_REGRESSION, ("LLRs inside the 1e-3 contract but beyond the 1e-4 regression guard (measured: 1.1e-5)", worst_rel)
    # nbadsync: exact, unless a sync softbit of that candidate is within tolerance of zero
    nb_diff = np.nonzero(exp_nb != items_g["nbadsync"])[0]
    marginal = 0
    classes = set()
    for k in nb_diff:
        soft, _, _ = o.softbits_at(cd, int(items_o["block_idx"][k]), int(items_o["pattern_idx"][k]), int(items_g["pos"][k]))
        sync = np.concatenate([soft[0:8], soft[56:64]])
        rms = float(np.sqrt(np.mean(np.square(soft))))
        assert np.min(np.abs(sync)) <= 1e-3 * rms, ("nbadsync differs without a marginal sync softbit", k, exp_nb[k], items_g["nbadsync"][k])
        marginal += 1
        # slots that fold the same frames (ring-wrap twins, the periodic copies of masks 111111 / 100100) share their sync softbits: a
        # softbit that sits on zero shows up once per copy (round 5: two of them in one 24 048-candidate window of the 640-channel soak).
        # The limit counts such a class once; the rate it is derived from was counted per slot, which only errs on the strict side.
        classes.add(copy_class(items_g, k))
    limit = nbadsync_marginal_limit(len(items_g))
    rep = dict(llr_max_abs_diff=float(np.abs(exp_llr[finite] - got[finite]).max()) if finite.any() else 0.0, llr_max_rel_diff=worst_rel,
               nbadsync_marginal=marginal, nbadsync_marginal_classes=len(classes), nbadsync_marginal_limit=limit)
    assert not enforce_limits or len(classes) <= limit, ("more verified nbadsync marginals than three times the measured rate allows", rep)
    return rep


BP_PERTURBATION_SCALES = (1e-6, 1e-5, 1e-4)   # relative LLR perturbations, all far inside the stated LLR tolerance (1e-3)


def bp_outcome(orc_mod, llr):
    ok, msg, it, nh = orc_mod.ldpc_one(np.ascontiguousarray(llr, dtype=np.float32))
    return (bool(ok), int(it) if ok else -1, int(nh) if ok else -1, bytes(np.asarray(msg, dtype=np.uint8)) if ok else b"")


def verify_marginal_bp(orc_mod, llr, gpu_outcome, seed, trials=48):
    """A BP accept/iteration difference between the HIP kernel and the oracle is tolerated ONLY if the oracle's own decision
    is unstable at that input: re-running the oracle's BP on the same LLRs perturbed by a relative 1e-6 .. 1e-4 (the kernel's
    tanh/atanh differ from libm by ~1e-7) must reproduce the kernel's (accept, iteration) outcome in some trial AND the
    oracle's in another.  Returns the smallest scale at which that happens; raises AssertionError for a stable difference."""
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    rng = np.random.default_rng(seed)
    base = bp_outcome(orc_mod, llr)[:2]
    for scale in BP_PERTURBATION_SCALES:
        seen = {base}
        for _ in range(trials):
            pert = (llr * (1.0 + scale * rng.uniform(-1.0, 1.0, size=llr.shape))).astype(np.float32)
            seen.add(bp_outcome(orc_mod, pert)[:2])
            if tuple(gpu_outcome[:2]) in seen and len(seen) > 1:
                return scale
    raise AssertionError(("BP decision differs from the oracle and is STABLE under LLR perturbations up to 1e-4: not a marginal case",
                          dict(oracle=base, gpu=tuple(gpu_outcome[:2]))))


def compare_ldpc_against_oracle_on_gpu_llrs(orc_mod, items_g, threshold):
    """Run the oracle's BP on the LLRs the GPU produced; accept/iter/nhard/message must be identical.  A difference is
    accepted only after verify_marginal_bp has shown the decision to be unstable at that input; such cases are counted
    with the perturbation scale that exposed them."""
    flips = []
    checked = 0
    for k in range(len(items_g)):
        if items_g["nbadsync"][k] > threshold:
            assert items_g["is_message_present"][k] == 0
            continue
        llr = items_g["softbits_wo_sync"][k]
        if not np.isfinite(llr).all():
            continue
        ok, it, nh, msg = bp_outcome(orc_mod, llr)
        checked += 1
        g_ok = bool(items_g["is_message_present"][k])
        g_it = int(items_g["ldpc_num_iterations"][k]) if g_ok else -1
        if g_ok == ok and g_it == it:
            if ok:
                assert msg == bytes(np.asarray(items_g["message"][k], dtype=np.uint8)), k
                assert nh == items_g["ldpc_num_hard_errors"][k], k
            continue
        scale = verify_marginal_bp(orc_mod, llr, (g_ok, g_it), seed=1000 + k)
        flips.append(dict(item=int(k), oracle=[ok, it], gpu=[g_ok, g_it], unstable_at_relative_perturbation=scale))
    classes = {copy_class(items_g, f["item"]) for f in flips}      # a marginal codeword shows up once per copy of its slot
    rep = dict(checked=checked, marginal_flips=len(flips), marginal_classes=len(classes), marginal_limit=bp_marginal_limit(checked), flips=flips)
    assert len(classes) <= rep["marginal_limit"], ("more verified marginal BP decisions than three times the measured bound allows", rep)
    return rep


def compare_ldpc_items(orc_mod, items_o, items_g, same, enforce_limits: bool = True):
    """Full-window form: items_o are the oracle's results on the oracle's own LLRs; for every candidate in `same` (identical
    position and nbadsync) accept/iteration/hard errors/payload must be identical, except verified-marginal cases (the LLRs
    themselves differ by <= 1e-3 there, so the perturbation test is run around the GPU's LLRs with the oracle's outcome as the
    second required outcome)."""
    flips = []
    idx = np.nonzero(same & ((items_o["is_message_present"] != items_g["is_message_present"]) |
                             ((items_o["is_message_present"] == 1) & (items_o["ldpc_num_iterations"] != items_g["ldpc_num_iterations"]))))[0]
    for k in idx:
        g_ok = bool(items_g["is_message_present"][k])
        g_it = int(items_g["ldpc_num_iterations"][k]) if g_ok else -1
        o_ok = bool(items_o["is_message_present"][k])
        o_it = int(items_o["ldpc_num_iterations"][k]) if o_ok else -1
        llr = items_g["softbits_wo_sync"][k]
        on_gpu_llr = bp_outcome(orc_mod, llr)[:2]
        if on_gpu_llr == (g_ok, g_it):
            # the kernel agrees with the oracle's BP on its own LLRs; the difference comes from the (toleranced) LLRs:
            # it must then be reachable from the oracle's LLRs by a perturbation inside the LLR tolerance
            scale = verify_marginal_bp(orc_mod, items_o["softbits_wo_sync"][k], (g_ok, g_it), seed=2000 + int(k))
        else:
            scale = verify_marginal_bp(orc_mod, llr, (g_ok, g_it), seed=2000 + int(k))
        flips.append(dict(item=int(k), oracle=[o_ok, o_it], gpu=[g_ok, g_it], unstable_at_relative_perturbation=scale))
    both = same & (items_o["is_message_present"] == 1) & (items_g["is_message_present"] == 1)
    assert np.array_equal(items_o["message"][both], items_g["message"][both])
    assert np.array_equal(items_o["ldpc_num_hard_errors"][both], items_g["ldpc_num_hard_errors"][both])
    classes = {copy_class(items_g, f["item"]) for f in flips}      # a marginal codeword shows up once per copy of its slot
    rep = dict(compared=int(same.sum()), both_accepted=int(both.sum()), marginal_flips=len(flips), marginal_classes=len(classes),
               marginal_limit=bp_marginal_limit(int(same.sum())), flips=flips)
    assert not enforce_limits or len(classes) <= rep["marginal_limit"], ("more verified marginal BP decisions than three times the measured bound allows", rep)
    return rep


def decoded_set(items):
    """{(block_idx, pattern_idx, message bytes)} of accepted candidates."""
    out = set()
    for k in np.nonzero(items["is_message_present"])[0]:
        out.add((int(items["block_idx"][k]), int(items["pattern_idx"][k]), bytes(np.asarray(items["message"][k], dtype=np.uint8))))
    return out


def decoded_messages(items):
    return {bytes(np.asarray(items["message"][k], dtype=np.uint8)) for k in np.nonzero(items["is_message_present"])[0]}


def compare_result_list_with_oracle(o, orc_mod, records, analytic_by_channel, gpu_items_by_channel):
    """The compact result list of a decoder that does NOT retain candidates (the production path: blocked staging, gated softbits)
    against the oracle, for the listed channels, in two links that are both checked here:
      1. the channel's records are EXACTLY the accepted candidates of `gpu_items_by_channel[ch]` - the candidate dump of the same
         window decoded by a retaining handle (bit-identical to the batch, which the caller asserts) - item for item, field for
         field, bit for bit;
      2. that dump agrees with the oracle's decode_window stage by stage under the usual rules (compare_scan, compare_softbits,
         compare_ldpc_items: integers exact, every exception a verified near-tie or a verified marginal BP decision), and the
         sets of decoded payloads are equal."""
    decodes = 0
    per_channel = {}
    for ch, cd in analytic_by_channel.items():
        items_g = gpu_items_by_channel[ch]
        rec = records[records["channel"] == ch]
        acc = np.nonzero(items_g["is_message_present"])[0]
        assert np.array_equal(rec["item"], acc), ("records are not the accepted candidates of the dump", ch)
        for r, k in zip(rec, acc):
            assert bytes(r["message"]) == bytes(np.packbits(np.concatenate([items_g["message"][k].astype(np.uint8), np.zeros(3, np.uint8)]))), (ch, k)
            assert (int(r["pos"]), int(r["nbadsync"]), int(r["ldpc_iterations"]), int(r["ldpc_hard_errors"]), int(r["pattern_idx"]), int(r["num_avg"])) == \
                   (int(items_g["pos"][k]), int(items_g["nbadsync"][k]), int(items_g["ldpc_num_iterations"][k]), int(items_g["ldpc_num_hard_errors"][k]),
                    int(items_g["pattern_idx"][k]), int(items_g["num_avg"][k])), (ch, k)
            assert np.float32(r["f0"]).view(np.uint32) == np.float32(items_g["f0"][k]).view(np.uint32), (ch, k)
            assert np.float32(r["xb"]).view(np.uint32) == np.float32(items_g["xb"][k]).view(np.uint32), (ch, k)
        items_o, _ = o.decode_window(cd)
        scan = compare_scan(o, cd, items_o, items_g)
        sb = compare_softbits(o, cd, items_o, items_g)
        same = (items_o["pos"] == items_g["pos"]) & (items_o["nbadsync"] == items_g["nbadsync"])
        ld = compare_ldpc_items(orc_mod, items_o, items_g, same)
        assert decoded_messages(items_g) == decoded_messages(items_o), ("payload sets differ", ch)
        decodes += len(rec)
        per_channel[int(ch)] = dict(records=int(len(rec)), scan_near_ties=scan["near_ties"], nbadsync_marginal=sb.get("nbadsync_marginal"), bp_marginal=ld["marginal_flips"])
    return dict(channels=len(analytic_by_channel), decodes=decodes, per_channel=per_channel)
