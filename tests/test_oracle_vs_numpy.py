"""The C++ oracle against an independent float64 numpy reading of the same reference semantics
(tests/numpy_model.py).  Catches coding slips in the restatement; it cannot pin the oracle to the reference
(nothing can here: parity unpinned)."""
import numpy as np
import pytest

from msk144cudecoder_amd import protocol as P
from msk144cudecoder_amd import synth

import numpy_model as M


def _window(seed, snr=2.0, n_frames=5, freq=1503.0, start=1111):
    rng = np.random.default_rng(seed)
    msg = synth.random_message(rng)
    return synth.synth_audio(5184, [synth.Ping(msg, start, n_frames, freq, snr, 0.4)], 1000.0, rng), msg


def test_template(orc):
    assert np.allclose(orc.cb42(), M.cb42(), atol=1e-7)


@pytest.mark.parametrize("p", [0, 2, 5, 6, 7])
def test_scan_xb_and_slots(orc, p):
    x, _ = _window(1)
    o = orc.Oracle(center=1500.0, width=8.0, step=2.0, depth=8)
    cd = o.frontend_audio(x, 2)
    b = 3
    got = o.scan_xb(cd, b, p).astype(np.float64)
    want = M.scan_xb(M.mix(cd, o.frequency(b)), P.PATTERN_MASK[p])
    assert np.abs(got - want).max() <= 2e-5 * want.max()
    items = o.scan(cd)
    sel = (items["block_idx"] == b) & (items["pattern_idx"] == p)
    pos_m, xb_m = M.slots_from_xb(got)     # the slot rule applied to the oracle's own float32 values
    assert np.array_equal(items["pos"][sel], pos_m)
    assert np.allclose(items["xb"][sel], xb_m, rtol=0, atol=0)


@pytest.mark.parametrize("p,pos", [(0, 1111), (3, 247), (5, 5000), (7, 3000), (5, 5300)])
def test_softbits(orc, p, pos):
    x, _ = _window(2)
    o = orc.Oracle(center=1500.0, width=8.0, step=2.0, depth=8)
    cd = o.frontend_audio(x, 2)
    b = 2
    soft_o, llr_o, nb_o = o.softbits_at(cd, b, p, pos)
    soft_m, llr_m, nb_m = M.softbits(M.mix(cd, o.frequency(b)), P.PATTERN_MASK[p], pos % 5184)
    assert np.abs(soft_o - soft_m).max() <= 2e-4 * np.abs(soft_m).max()
    assert np.abs(llr_o - llr_m).max() <= 2e-4 * np.abs(llr_m).max()
    margin = np.abs(np.concatenate([soft_m[:8], soft_m[56:64]])).min()
    assert nb_o == nb_m or margin < 1e-3 * np.abs(soft_m).max()


def test_bp_decoder(orc):
    rng = np.random.default_rng(3)
    agree = 0
    for trial in range(12):
        m = synth.random_message(rng)
        cw = synth.encode_message(m)
        sigma = [1.0, 1.6, 2.2][trial % 3]
        llr = ((2.0 * cw - 1.0) * 3.0 + rng.normal(0, sigma, 128)).astype(np.float32)
        ok_o, msg_o, it_o, nh_o = orc.ldpc_one(llr)
        ok_m, msg_m, it_m, nh_m = M.bp_decode(llr.astype(np.float64))
        assert ok_o == ok_m
        if ok_o:
            assert np.array_equal(msg_o, msg_m) and it_o == it_m and nh_o == nh_m
            agree += 1
    assert agree >= 6
    # pure noise: both reject
    for _ in range(3):
        llr = rng.normal(0, 5.0, 128).astype(np.float32)
        assert orc.ldpc_one(llr)[0] == M.bp_decode(llr.astype(np.float64))[0]


def test_end_to_end_numpy_decode_matches_oracle(orc):
    """Full chain in numpy for the strongest candidate of one frequency: same position, same payload."""
    x, msg = _window(4, snr=6.0, n_frames=6, freq=1500.0, start=700)
    o = orc.Oracle(center=1500.0, width=0.0, step=2.0, depth=6, nbadsync_threshold=2)
    cd = o.frontend_audio(x, 2)
    items, _ = o.decode_window(cd)
    cd2 = M.mix(cd, 1500.0)
    p = 3
    xb = M.scan_xb(cd2, P.PATTERN_MASK[p])
    pos_m, xb_m = M.slots_from_xb(xb)
    sel = items[items["pattern_idx"] == p]
    assert sorted((sel["pos"] % 864).tolist()) == sorted((pos_m % 864).tolist()) or np.array_equal(sel["pos"], pos_m)
    best = int(pos_m[np.argmax(xb_m)])
    _, llr, nbad = M.softbits(cd2, P.PATTERN_MASK[p], best % 5184)
    ok, m_dec, it, nh = M.bp_decode(llr)
    assert ok and np.array_equal(m_dec, msg) and nbad <= 1
    k = np.nonzero((items["pattern_idx"] == p) & (items["pos"] == best))[0]
    assert len(k) == 1 and items["is_message_present"][k[0]] == 1
    assert np.array_equal(items["message"][k[0]], msg) and items["ldpc_num_iterations"][k[0]] == it
