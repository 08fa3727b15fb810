"""-m gpu: BASELINE-size checks (width 500, step 1, depth 6, threshold 3: F=501, 24 048 candidates/window)."""
import numpy as np
import pytest

from msk144cudecoder_amd import synth

import parity

pytestmark = pytest.mark.gpu
DEEP = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)


def test_deep_window_against_oracle(orc, hip, parity_report):
    """One full-size window, every stage compared with the oracle (about 1 s of 16 host cores)."""
    rng = np.random.default_rng(2025)
    msg = synth.random_message(rng)
    x = synth.synth_audio(5184, [synth.Ping(msg, 1234, 5, 1500.0 + 137.3, 0.0, 0.9)], 1000.0, rng)
    o = orc.Oracle(threads=16, **DEEP)
    cd = o.frontend_audio(x, 2)
    items_o, idx_o = o.decode_window(cd)
    with hip.HipDecoder(channels=1, **DEEP) as d:
        assert (d.F, d.D, d.K) == (501, 6, 24048)
        d.submit_audio(x)
        d.decode()
        assert np.array_equal(d.dump_analytic(0).view(np.uint32), cd.view(np.uint32))
        items_g = d.dump_candidates(0)
        idx_g = d.dump_indexes(0)
    rep = parity.compare_scan(o, cd, items_o, items_g)
    assert rep["near_ties"] + rep["periodic_fallbacks"] <= parity.near_tie_limit(24048) == 3, rep   # measured-rate limit (parity.py); observed 0
    sb = parity.compare_softbits(o, cd, items_o, items_g)
    assert sb["nbadsync_marginal_classes"] <= parity.nbadsync_marginal_limit(24048) == 1, sb
    assert sb["llr_max_rel_diff"] <= parity.TOL_LLR_REGRESSION
    assert np.array_equal(idx_g, np.nonzero(items_g["nbadsync"] <= 3)[0])
    same = (items_o["pos"] == items_g["pos"]) & (items_o["nbadsync"] == items_g["nbadsync"])
    # accept / iterations / hard errors / payload identical wherever the candidate itself is identical; any difference
    # must be a VERIFIED marginal case (oracle decision unstable under 1e-6..1e-4 LLR perturbations)
    ld = parity.compare_ldpc_items(orc, items_o, items_g, same)
    assert ld["marginal_classes"] <= parity.bp_marginal_limit(24048) == 2, ld
    assert ld["both_accepted"] > 10
    parity_report("deep_window_F501_D6", dict(scan=rep, softbits=sb, ldpc=ld))
    assert bytes(msg) in parity.decoded_messages(items_g)
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o)


def test_1024_channel_batch_properties(orc, hip, parity_report):
    """BASELINE configs[2] size: round trip of the injected pings, determinism, channel independence and
    agreement with the oracle on every payload that is not a transmitted one (CRC-13 false positives of the
    algorithm itself are expected at 1.6e7 BP attempts per step and must be the oracle's too)."""
    import bench
    wins, truth = bench.make_inputs(0, 1024)
    with hip.HipDecoder(channels=1024, max_results=1 << 20, llr_block_channels=1024, **DEEP) as d:   # parity-dump mode: every LLR row retained
        d.submit_audio(wins[2])
        d.decode()
        res1 = d.results().copy()
        ch5 = d.dump_candidates(5).tobytes()
        ch1000 = d.dump_candidates(1000).tobytes()
        d.submit_audio(wins[2])
        d.decode()
        res2 = d.results().copy()
    assert res1.tobytes() == res2.tobytes()                             # deterministic
    # the PRODUCTION path - default blocked staging (128-channel blocks, softbits_kernel<true, true>: gated-out candidates stop after
    # their sync check), what bench.py times - must give the retained-mode list byte for byte at full size
    with hip.HipDecoder(channels=1024, max_results=1 << 20, **DEEP) as dp:
        dp.submit_audio(wins[2])
        dp.decode()
        prod = dp.results().copy()
        handed = parity.handed_over_records(dp, prod)
        handed_slots = dp.copy_count()
    assert prod.tobytes() == res1.tobytes()
    # records whose slot was never computed itself (its lower slot's result, its own pos / xb): each one equals the record the
    # retained handle computed for that very slot
    assert handed.sum() > 1000 and prod[handed].tobytes() == res1[handed].tobytes()
    assert 0.10 < handed_slots / (1024 * 24048) < 0.20
    key = res1["channel"].astype(np.int64) * 100000 + res1["item"]
    assert np.all(np.diff(key) > 0)                                     # ordered by (channel, item)
    decoded_channels = set(int(c) for c in np.unique(res1["channel"]))
    pinged = set(truth)
    assert len(decoded_channels & pinged) >= 0.5 * len(pinged)          # most pings overlap this window
    unexpected = [r for r in res1 if truth.get(int(r["channel"])) != bytes(r["message"])]
    assert len(unexpected) <= 5
    o = orc.Oracle(threads=16, **DEEP)
    for r in unexpected:
        ch = int(r["channel"])
        items, _ = o.decode_window(o.frontend_audio(wins[2, ch], 2))
        k = int(r["item"])
        assert items["is_message_present"][k] == 1
        assert bytes(np.packbits(np.concatenate([items["message"][k].astype(np.uint8), np.zeros(3, np.uint8)]))) == bytes(r["message"])
    with hip.HipDecoder(channels=1, **DEEP) as d1:
        for ch, blob in ((5, ch5), (1000, ch1000)):
            d1.submit_audio(wins[2, ch])
            d1.decode()
            assert d1.dump_candidates(0).tobytes() == blob              # batch == single, bit for bit
    # production list against the oracle DIRECTLY: 16 sampled channels spread over the 128-channel blocks, eight with a decoded
    # ping and eight noise-only - the records are the accepted candidates of the channel's dump, and the dump agrees with the oracle
    with_ping = sorted(decoded_channels & pinged)
    noise = [c for c in range(1024) if c not in pinged]
    sample = [with_ping[i * len(with_ping) // 8] for i in range(8)] + [noise[i * len(noise) // 8] for i in range(8)]
    dumps = {}
    with hip.HipDecoder(channels=1, **DEEP) as d1:
        for ch in sample:
            d1.submit_audio(wins[2, ch])
            d1.decode()
            dumps[ch] = d1.dump_candidates(0)                           # single == batch bit for bit (asserted above)
    report = parity.compare_result_list_with_oracle(o, orc, prod, {ch: o.frontend_audio(wins[2, ch], 2) for ch in sample}, dumps)
    assert report["channels"] == 16 and report["decodes"] >= 8, report
    report["handed_over_records"] = int(handed.sum())
    report["handed_over_records_differing_from_their_own_decode"] = 0
    report["records"] = int(len(prod))
    report["handed_over_slots_share"] = handed_slots / (1024 * 24048)
    parity_report("production_path_1024ch_vs_oracle", report)


def test_fine_step_depth8_all_gated(orc, hip, parity_report):
    """Quarter-Hz grid, all 8 patterns, threshold 16 (every candidate is BP-decoded): F=401, 25 664 candidates."""
    cfg = dict(center=1500.0, width=100.0, step=0.25, depth=8, nbadsync_threshold=16)
    rng = np.random.default_rng(77)
    msg = synth.random_message(rng)
    x = synth.synth_audio(5184, [synth.Ping(msg, 2500, 3, 1500.0 + 17.37, 1.0, 2.2)], 1000.0, rng)
    o = orc.Oracle(threads=16, **cfg)
    cd = o.frontend_audio(x, 2)
    items_o, idx_o = o.decode_window(cd)
    with hip.HipDecoder(channels=1, **cfg) as d:
        assert (d.F, d.D, d.K) == (401, 8, 25664)
        d.submit_audio(x)
        d.decode()
        items_g = d.dump_candidates(0)
        idx_g = d.dump_indexes(0)
    assert np.array_equal(idx_g, np.arange(25664)) and np.array_equal(idx_o, idx_g)
    assert np.array_equal(items_o["f0"].view(np.uint32), items_g["f0"].view(np.uint32))
    rep = parity.compare_scan(o, cd, items_o, items_g)
    assert rep["near_ties"] + rep["periodic_fallbacks"] <= parity.near_tie_limit(25664) == 3, rep
    parity.compare_softbits(o, cd, items_o, items_g)
    same = (items_o["pos"] == items_g["pos"]) & (items_o["nbadsync"] == items_g["nbadsync"])
    ld = parity.compare_ldpc_items(orc, items_o, items_g, same)
    assert ld["marginal_classes"] <= parity.bp_marginal_limit(25664), ld
    parity_report("fine_step_depth8_all_gated", dict(scan=rep, ldpc=ld))
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o)
    assert bytes(msg) in parity.decoded_messages(items_g)


def test_config4_iq_4096_low_snr_channels(orc, hip, parity_report):
    """BASELINE configs[4] at full size: IQ --read-mode=2, 4096 low-SNR channels, width 500 / step 1 (BASELINE leaves the step
    open; 1 Hz = the deep setting, F=501) / depth 6 / nbadsync-threshold 3 - the LDPC-iteration-heavy stress (main.cu:334-380).
    Determinism, batch == single channel bit for bit, every payload that was not transmitted reproduced by the oracle at the same
    item, and most pinged channels decoded."""
    cfg = dict(center=0.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
    nch = 4096
    wins, truth = synth.iq_low_snr_batch(nch, 5)
    with hip.HipDecoder(read_mode=2, channels=nch, max_results=1 << 20, llr_block_channels=nch, **cfg) as d:   # parity-dump mode (50 GB of LLRs)
        assert (d.F, d.D, d.K) == (501, 6, 24048)
        d.submit_iq(wins)
        d.decode()
        res1 = d.results().copy()
        blobs = {ch: d.dump_candidates(ch).tobytes() for ch in (0, 1, 2047, 4095)}
        d.submit_iq(wins)
        d.decode()
        res2 = d.results().copy()
    assert res1.tobytes() == res2.tobytes()                             # deterministic
    # production path (default 128-channel blocks, gated softbits, copies handed over) at full size: same list, byte for byte
    with hip.HipDecoder(read_mode=2, channels=nch, max_results=1 << 20, **cfg) as dp:
        dp.submit_iq(wins)
        dp.decode()
        prod = dp.results().copy()
        handed = parity.handed_over_records(dp, prod)
        handed_slots = dp.copy_count()
    assert prod.tobytes() == res1.tobytes()
    assert handed.sum() > 5000 and prod[handed].tobytes() == res1[handed].tobytes()      # slots never computed themselves = their own decode (6541 on this workload)
    key = res1["channel"].astype(np.int64) * 100000 + res1["item"]
    assert np.all(np.diff(key) > 0)                                     # ordered by (channel, item)
    good = {int(r["channel"]) for r in res1 if truth.get(int(r["channel"])) == bytes(r["message"])}
    assert len(good) >= 0.75 * len(truth), (len(good), len(truth))      # -6..-2 dB pings: most decode
    unexpected = [r for r in res1 if truth.get(int(r["channel"])) != bytes(r["message"])]
    assert len(unexpected) <= 16, len(unexpected)                       # CRC-13 false positives of the algorithm itself
    o = orc.Oracle(threads=16, **cfg)
    for r in unexpected[:8]:
        ch, k = int(r["channel"]), int(r["item"])
        items, _ = o.decode_window(o.frontend_iq(wins[ch]))
        assert items["is_message_present"][k] == 1
        assert bytes(np.packbits(np.concatenate([items["message"][k].astype(np.uint8), np.zeros(3, np.uint8)]))) == bytes(r["message"])
    with hip.HipDecoder(read_mode=2, channels=1, **cfg) as d1:
        for ch, blob in blobs.items():
            d1.submit_iq(wins[ch])
            d1.decode()
            assert d1.dump_candidates(0).tobytes() == blob              # batch == single, bit for bit
            if ch == 0:
                # one pinged channel end to end against the oracle, stage by stage
                cd = o.frontend_iq(wins[ch])
                assert np.array_equal(d1.dump_analytic(0).view(np.uint32), cd.view(np.uint32))
                items_o, _ = o.decode_window(cd)
                items_g = d1.dump_candidates(0)
                rep = parity.compare_scan(o, cd, items_o, items_g)
                sb = parity.compare_softbits(o, cd, items_o, items_g)
                same = (items_o["pos"] == items_g["pos"]) & (items_o["nbadsync"] == items_g["nbadsync"])
                ld = parity.compare_ldpc_items(orc, items_o, items_g, same)
                assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o)
    # and against the oracle directly: 16 sampled channels (eight decoded pings, eight without a ping)
    no_ping = [c for c in range(nch) if c not in truth]
    gl = sorted(good)
    sample = [gl[i * len(gl) // 8] for i in range(8)] + [no_ping[i * len(no_ping) // 8] for i in range(8)]
    dumps = {}
    with hip.HipDecoder(read_mode=2, channels=1, **cfg) as d1:
        for ch in sample:
            d1.submit_iq(wins[ch])
            d1.decode()
            dumps[ch] = d1.dump_candidates(0)
    direct = parity.compare_result_list_with_oracle(o, orc, prod, {ch: o.frontend_iq(wins[ch]) for ch in sample}, dumps)
    assert direct["channels"] == 16 and direct["decodes"] >= 8, direct
    direct["handed_over_records"] = int(handed.sum())
    direct["handed_over_records_differing_from_their_own_decode"] = 0
    direct["records"] = int(len(prod))
    direct["handed_over_slots_share"] = handed_slots / (nch * 24048)
    parity_report("production_path_config4_vs_oracle", direct)
    parity_report("config4_iq_4096", dict(channels=nch, decodes=int(len(res1)), pinged=len(truth), pinged_decoded=len(good),
                                          not_transmitted=len(unexpected), channel0=dict(scan=rep, softbits=sb, ldpc=ld)))


def test_blocked_staging_equals_retained(hip):
    """Blocked staging (LLR rows of 64 channels at a time, the production default for big batches) must give exactly the result
    list of the retain-everything mode; candidate dumps and partial stage runs are refused rather than served stale."""
    import bench
    wins, _ = bench.make_inputs(0, 200)
    with hip.HipDecoder(channels=200, max_results=1 << 20, llr_block_channels=200, **DEEP) as d:
        d.submit_audio(wins[1])
        d.decode()
        want = d.results().copy()
    for blk in (0, 16, 7, 64):          # 0 = automatic (128 for more than 128 channels: 128 + 72 here); 7: last block is short
        with hip.HipDecoder(channels=200, max_results=1 << 20, llr_block_channels=blk, **DEEP) as d:
            assert d.llr_block == (blk or 128)      # msk144_llr_block_channels: the library's automatic choice is 128 channels per block
            d.submit_audio(wins[1])
            d.decode()
            got = d.results().copy()
            assert got.tobytes() == want.tobytes(), blk
            with pytest.raises(hip.Msk144Error) as e:
                d.dump_candidates(3)
            assert e.value.code == -6
            with pytest.raises(hip.Msk144Error) as e:
                d.decode(hip.STAGE_LDPC)
            assert e.value.code == -6
            d.decode(hip.STAGE_SCAN)        # stages outside the block loop stay individually runnable
    assert len(want) > 100


@pytest.mark.parametrize("depth,width", [(6, 500.0), (8, 120.0)])
def test_copies_are_handed_to_the_lower_slot(hip, parity_report, depth, width):
    """Slots of one (frequency, pattern) group that fold the SAME frames: the scan walks 5376 positions of a 5184-sample ring, so pos
    and pos + 5184 are one place, and masks 111111 and 100100 sum the same frames at pos and pos + 864 (+ 2592), so the eight slots of
    those patterns are mostly copies of two or three peaks (exact ties in exact arithmetic).  The reference demodulates and decodes
    each copy (softbits_kernel.cuh:56-83, ldpc_kernel.cuh:100-249).  In blocked staging a slot whose position is congruent to a LOWER
    slot's of its group is not computed: the index list leaves it out and the collect stage reports it with the nbadsync and the
    decode of that slot.  Checked here on two channels (one with a strong ping, so that accepted copies exist): (1) the index list of
    the blocked handle is exactly the retained handle's list minus the slots that have a congruent lower slot; (2) most slots of
    pattern 5 are such copies, a few per cent elsewhere (ring wrap only); (3) the result lists of the two handles are byte-identical -
    the copies are reported, with their own position and xb."""
    cfg = dict(center=1500.0, width=width, step=1.0, depth=depth, nbadsync_threshold=3)
    rng = np.random.default_rng(808)
    msg = synth.random_message(rng)
    wins = np.stack([synth.synth_audio(5184, [synth.Ping(msg, 200, 6, 1500.0 + 33.3, 3.0, 1.1)], 1000.0, rng),
                     np.rint(rng.normal(0.0, 1000.0, 5184)).astype(np.int16)])
    with hip.HipDecoder(channels=2, llr_block_channels=2, max_results=1 << 18, **cfg) as d:
        d.submit_audio(wins)
        d.decode()
        full = d.results().copy()
        items = [d.dump_candidates(c) for c in range(2)]
        idx_full = [d.dump_indexes(c) for c in range(2)]
        assert d.copy_handover() is False and d.copy_count() == 0
        with pytest.raises(hip.Msk144Error) as e:        # a handle that retains every row never hands a slot over
            d.set_copy_handover(True)
        assert e.value.code == -6
        d.set_copy_handover(False)
    with hip.HipDecoder(channels=2, llr_block_channels=1, max_results=1 << 18, **cfg) as d:
        d.submit_audio(wins)
        d.decode()
        blocked = d.results().copy()
        idx_blocked = [d.dump_indexes(c) for c in range(2)]
        assert d.copy_handover() is True
        copies_counted = d.copy_count()
        # (4) the switch: blocked staging WITHOUT the hand-over computes every slot, as the reference does
        d.set_copy_handover(False)
        d.submit_audio(wins)
        d.decode()
        every_slot = d.results().copy()
        idx_every_slot = [d.dump_indexes(c) for c in range(2)]
        assert d.copy_count() == 0 and d.copy_handover() is False
        d.set_copy_handover(True)
        d.decode()
        assert d.results().tobytes() == blocked.tobytes() and d.copy_count() == copies_counted
    assert blocked.tobytes() == full.tobytes() and len(full) > 20                                    # (3)
    assert every_slot.tobytes() == full.tobytes()
    period = {5: 864, 6: 2592}
    handed = kept = 0
    accepted_copies = 0
    for c in range(2):
        it = items[c]
        pos = it["pos"].astype(np.int64) % 5184
        drop = np.zeros(len(it), dtype=bool)
        for g0 in range(0, len(it), 8):
            r = pos[g0:g0 + 8] % period.get(int(it["pattern_idx"][g0]), 5184)
            for sl in range(1, 8):
                drop[g0 + sl] = bool((r[:sl] == r[sl]).any())
        want = np.array([k for k in idx_full[c] if not drop[k]], dtype=np.int32)
        assert np.array_equal(idx_blocked[c], want), c                                                # (1)
        assert np.array_equal(idx_every_slot[c], idx_full[c]), c                                      # (4): the retained index list exactly
        copies_counted -= int(drop.sum())
        handed += int(drop[idx_full[c]].sum())
        kept += len(want)
        five = it["pattern_idx"] == 5
        wrap_only = ~np.isin(it["pattern_idx"], list(period))
        assert drop[five].mean() > 0.5 and 0.0 < drop[wrap_only].mean() < 0.08                        # (2)
        accepted_copies += int((drop & (it["is_message_present"] == 1)).sum())
    assert accepted_copies >= 1            # the list identity above covered records that were never decoded themselves
    assert copies_counted == 0             # msk144_copy_count = the slots with a congruent lower slot, gated or not
    parity_report(f"copies_handed_over_depth{depth}", dict(gated_slots_handed_over=handed, gated_slots_decoded=kept, accepted_copies_in_the_result_list=accepted_copies,
                                                           result_lists_identical=True))


def test_single_stream_handle_without_llr_retention(hip):
    """msk144_set_llr_retention(h, 0) - what msk144hipdecoder asks of every handle, its single stream included: a one-block handle then
    runs the kernels of a blocked batch (early nbadsync gate, copies handed over).  Same result list as the retaining handle, dumps and
    partial stage runs refused until retention is switched back on."""
    cfg = dict(center=1500.0, width=200.0, step=1.0, depth=6, nbadsync_threshold=3)
    rng = np.random.default_rng(515)
    x = synth.synth_audio(5184, [synth.Ping(synth.random_message(rng), 100, 6, 1500.0 - 61.7, 3.0, 2.1)], 1000.0, rng)
    with hip.HipDecoder(channels=1, max_results=1 << 16, **cfg) as d:
        d.submit_audio(x)
        d.decode()
        want = d.results().copy()
        idx_all = d.dump_indexes(0)
        d.set_llr_retention(False)
        assert d.copy_handover() is True
        d.decode()
        assert d.results().tobytes() == want.tobytes() and len(want) > 20
        assert d.copy_count() > 0 and len(d.dump_indexes(0)) < len(idx_all)
        for call in (lambda: d.dump_candidates(0), lambda: d.decode(hip.STAGE_LDPC), lambda: d.load_candidates(np.zeros(d.K, dtype=hip.CANDIDATE_DTYPE))):
            with pytest.raises(hip.Msk144Error) as e:
                call()
            assert e.value.code == -6
        d.set_copy_handover(False)                       # every slot on its own, still gated early
        d.decode()
        assert d.results().tobytes() == want.tobytes() and d.copy_count() == 0
        assert np.array_equal(d.dump_indexes(0), idx_all)
        d.set_llr_retention(True)
        with pytest.raises(hip.Msk144Error) as e:        # retention is back on, but the rows in the store are those of a decode without it
            d.dump_candidates(0)
        assert e.value.code == -6
        d.decode()
        assert d.results().tobytes() == want.tobytes() and d.copy_handover() is False
        assert int((d.dump_candidates(0)["is_message_present"] == 1).sum()) == len(want)
    with hip.HipDecoder(channels=4, llr_block_channels=2, **cfg) as d:
        with pytest.raises(hip.Msk144Error) as e:        # a blocked handle cannot retain
            d.set_llr_retention(True)
        assert e.value.code == -6
        d.set_llr_retention(False)


def test_handed_over_records_equal_their_own_decode(hip, parity_report):
    """The four 1024-channel windows bench.py cycles through, production path (128-channel blocks, copies handed over) against the same
    handle with the hand-over switched off (every slot demodulated and decoded on its own, as the reference does): lists byte-identical,
    and the records whose slot was handed over - never computed themselves - are counted and compared on their own."""
    import bench
    wins, _ = bench.make_inputs(0, 1024)
    tot = dict(windows=0, records=0, handed_over_records=0, handed_over_records_differing_from_their_own_decode=0, handed_over_slots=0, slots=0)
    with hip.HipDecoder(channels=1024, max_results=1 << 20, **DEEP) as d:
        for t in range(wins.shape[0]):
            d.set_copy_handover(True)
            d.submit_audio(wins[t])
            d.decode()
            prod = d.results().copy()
            handed = parity.handed_over_records(d, prod)
            tot["handed_over_slots"] += d.copy_count()
            d.set_copy_handover(False)
            d.decode()
            own = d.results().copy()
            assert d.copy_count() == 0
            assert len(own) == len(prod)
            tot["windows"] += 1
            tot["records"] += len(prod)
            tot["slots"] += 1024 * d.K
            tot["handed_over_records"] += int(handed.sum())
            tot["handed_over_records_differing_from_their_own_decode"] += int(np.count_nonzero(prod[handed] != own[handed]))
            assert prod.tobytes() == own.tobytes(), t
    assert tot["handed_over_records"] > 5000 and tot["handed_over_records_differing_from_their_own_decode"] == 0, tot
    assert 0.10 < tot["handed_over_slots"] / tot["slots"] < 0.20, tot
    parity_report("handed_over_records_4_bench_windows", tot)


def test_silent_and_saturated_channels_in_a_blocked_batch(hip):
    """An all-zero window (1/0 normalisation: NaN samples, NaN correlations, arbitrary scan positions - possibly eight equal ones, i.e.
    seven copies per group) and a full-scale square wave inside a batch that is decoded in blocks with copies handed over: the list equals
    the retain-everything list byte for byte, the silent channel reports nothing and disturbs no one."""
    cfg = dict(center=1500.0, width=60.0, step=1.0, depth=6, nbadsync_threshold=3)
    rng = np.random.default_rng(31)
    n_ch = 70
    wins = np.rint(rng.normal(0.0, 1000.0, size=(n_ch, 5184))).astype(np.int16)
    msg = synth.random_message(rng)
    for c in (0, 33, 69):
        wins[c] = synth.synth_audio(5184, [synth.Ping(msg, 300 + c, 6, 1500.0 + 0.3 * c, 4.0, 0.1 * c)], 1000.0, rng)
    wins[5] = 0
    wins[66] = np.where(np.arange(5184) % 8 < 4, 32767, -32768).astype(np.int16)
    lists = []
    for blk in (n_ch, 64, 9):
        with hip.HipDecoder(channels=n_ch, llr_block_channels=blk, max_results=1 << 18, **cfg) as d:
            d.submit_audio(wins)
            d.decode()
            lists.append(d.results().copy())
    assert lists[0].tobytes() == lists[1].tobytes() == lists[2].tobytes()
    got = set(int(c) for c in lists[0]["channel"])
    assert {0, 33, 69} <= got and 5 not in got


def test_maximum_grid_single_channel(orc, hip, parity_report):
    """Largest search grid the option surface allows in practice: width 500 at a quarter-Hz step, all 8 patterns (F = 2001,
    128 064 candidates per window, threshold 4) - every stage against the oracle on one window with two pings."""
    cfg = dict(center=1500.0, width=500.0, step=0.25, depth=8, nbadsync_threshold=4)
    rng = np.random.default_rng(4242)
    m1, m2 = synth.random_message(rng), synth.random_message(rng)
    x = synth.synth_audio(5184, [synth.Ping(m1, 300, 4, 1500.0 - 201.3, 3.0, 0.4), synth.Ping(m2, 3000, 2, 1500.0 + 77.77, 6.0, 2.0)], 1000.0, rng)
    o = orc.Oracle(threads=16, **cfg)
    cd = o.frontend_audio(x, 2)
    items_o, idx_o = o.decode_window(cd)
    with hip.HipDecoder(channels=1, **cfg) as d:
        assert (d.F, d.D, d.K) == (2001, 8, 128064)
        d.submit_audio(x)
        d.decode()
        items_g = d.dump_candidates(0)
        idx_g = d.dump_indexes(0)
        res = d.results()
    assert np.array_equal(items_o["f0"].view(np.uint32), items_g["f0"].view(np.uint32))
    rep = parity.compare_scan(o, cd, items_o, items_g)
    assert rep["near_ties"] + rep["periodic_fallbacks"] <= parity.near_tie_limit(128064) == 8, rep       # 0.68 expected at the measured rate; observed 0
    sb = parity.compare_softbits(o, cd, items_o, items_g)
    assert np.array_equal(idx_g, np.nonzero(items_g["nbadsync"] <= 4)[0])
    same = (items_o["pos"] == items_g["pos"]) & (items_o["nbadsync"] == items_g["nbadsync"])
    ld = parity.compare_ldpc_items(orc, items_o, items_g, same)
    assert ld["marginal_classes"] <= parity.bp_marginal_limit(128064), ld
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o)
    assert {bytes(m1), bytes(m2)} <= parity.decoded_messages(items_g)
    assert np.array_equal(res["item"], np.nonzero(items_g["is_message_present"])[0])
    parity_report("maximum_grid_F2001_D8", dict(scan=rep, softbits=sb, ldpc=ld, decodes=int(len(res))))


def test_channel_sharded_decode_equals_unsharded(hip):
    """the path shards by channel.  2048 deep-configuration channels decoded by ONE handle, and the same channels as
    four shards of 512 on four handles with msk144_set_channel_base = first channel of the shard (what rank r of a 4-GPU run, or the
    r-th device loop, does): the concatenated shard records are the unsharded records, byte for byte, global channel ids included."""
    import bench
    from msk144cudecoder_amd import sharding
    wins, truth = bench.make_inputs(3, 2048)
    with hip.HipDecoder(channels=2048, max_results=1 << 20, **DEEP) as d:
        d.submit_audio(wins[1])
        d.decode()
        whole = d.results().copy()
    assert len(whole) > 1000 and len({int(c) for c in whole["channel"]}) > 100
    parts = []
    for r in range(4):
        lo, cnt = sharding.shard_channels(2048, r, 4)
        with hip.HipDecoder(channels=cnt, max_results=1 << 20, **DEEP) as d:
            d.set_channel_base(lo)
            d.submit_audio(wins[1, lo:lo + cnt])
            d.decode()
            rec = d.results().copy()
        assert len(rec) == 0 or (rec["channel"].min() >= lo and rec["channel"].max() < lo + cnt)
        parts.append(rec)
    assert np.concatenate(parts).tobytes() == whole.tobytes()
