"""The oracle itself: properties that hold for the reference algorithm, in lieu of
reference golden vectors (the reference has none - parity unpinned)."""
import numpy as np
import pytest

from msk144cudecoder_amd import synth


def _ping_window(seed, snr, n_frames, freq, start=700, sigma=1000.0):
    rng = np.random.default_rng(seed)
    msg = synth.random_message(rng)
    p = synth.Ping(msg, start, n_frames, freq, snr, float(rng.uniform(0, 6.28)))
    return synth.synth_audio(5184, [p], sigma, rng), msg


def test_crc_table_walk_equals_bit_serial_division(orc):
    rng = np.random.default_rng(1)
    for _ in range(200):
        buf = rng.integers(0, 256, size=12, dtype=np.uint8)
        bits = np.unpackbits(buf)
        rem = 0
        for b in bits:
            rem = (rem << 1) | int(b)
            if rem & 0x2000:
                rem ^= 0x2000 | 0x15D7
        assert orc.crc13(buf.tobytes()) == (rem & 0x1FFF)


def test_crc_accepts_codewords_and_rejects_single_flips(orc):
    rng = np.random.default_rng(2)
    for _ in range(20):
        cw = synth.encode_message(synth.random_message(rng)).astype(np.int8)
        assert orc.check_crc_bits(cw)
        for pos in rng.choice(90, size=8, replace=False):
            bad = cw.copy()
            bad[pos] ^= 1
            assert not orc.check_crc_bits(bad)


def test_bp_accepts_noiseless_codeword_at_iteration_zero(orc):
    rng = np.random.default_rng(3)
    for _ in range(10):
        m = synth.random_message(rng)
        cw = synth.encode_message(m)
        ok, msg, it, nh = orc.ldpc_one((2.0 * cw - 1.0) * 3.0)
        assert ok and it == 0 and nh == 0 and np.array_equal(msg, m)


def test_bp_corrects_errors_and_rejects_noise(orc):
    rng = np.random.default_rng(4)
    m = synth.random_message(rng)
    cw = synth.encode_message(m)
    llr = (2.0 * cw - 1.0) * 2.5 + rng.normal(0, 1.3, 128)
    ok, msg, it, nh = orc.ldpc_one(llr)
    assert ok and np.array_equal(msg, m) and nh == int(((llr > 0) != (cw > 0)).sum())
    accepted = sum(orc.ldpc_one(rng.normal(0, 2.0, 128))[0] for _ in range(300))
    assert accepted == 0                                            # noise-only LLRs are essentially never accepted


def test_template_matches_modulator(orc):
    """cb42 is the first 42 samples of a frame whose sync word is s8 (msk_context.cuh:188-196)."""
    cw = np.zeros(128, dtype=np.uint8)
    x = synth.modulate_frame(synth.frame_bits(cw))
    cb = orc.cb42()
    # samples 0..5 of Q belong to bit 0 and its predecessor (last data bit) overlaps only on Q's first half
    assert np.allclose(cb.real[:36], x.real[:36], atol=1e-6)
    assert np.allclose(cb.imag[6:42], x.imag[6:42], atol=1e-6)
    assert np.allclose(cb.imag[:6], (2 * 0 - 1) * np.sin(np.arange(6, 12) * np.pi / 12), atol=1e-6)


@pytest.mark.parametrize("method", [1, 2])
def test_audio_round_trip(orc, method):
    x, msg = _ping_window(5, snr=3.0, n_frames=7, freq=1506.0)
    o = orc.Oracle(center=1500.0, width=20.0, step=2.0, depth=3, nbadsync_threshold=2, threads=4)
    items, idx = o.decode_window(o.frontend_audio(x, method))
    ok = items[items["is_message_present"] == 1]
    assert len(ok) > 0
    assert all(np.array_equal(it["message"], msg) for it in ok)
    best = ok[np.argmax(ok["xb"])]
    assert abs(best["f0"] - 1506.0) <= 2.0
    assert best["pos"] % 864 in (699, 700, 701)                    # pos == ping offset (mod 864)
    assert np.array_equal(idx, np.nonzero(items["nbadsync"] <= 2)[0])


def test_iq_round_trip(orc):
    rng = np.random.default_rng(6)
    msg = synth.random_message(rng)
    x = synth.synth_iq(5184, [synth.Ping(msg, 1000, 6, -6.0, 3.0, 0.5)], 20.0, rng)
    o = orc.Oracle(center=0.0, width=20.0, step=2.0, depth=4, nbadsync_threshold=2, threads=4)
    items, _ = o.decode_window(o.frontend_iq(x))
    ok = items[items["is_message_present"] == 1]
    assert len(ok) > 0 and all(np.array_equal(it["message"], msg) for it in ok)


def test_high_snr_sync_is_clean_and_hard_bits_equal_codeword(orc):
    rng = np.random.default_rng(7)
    msg = synth.random_message(rng)
    cw = synth.encode_message(msg)
    x = synth.synth_audio(5184, [synth.Ping(msg, 0, 6, 1500.0, 30.0, 0.0)], 100.0, rng)
    o = orc.Oracle(center=1500.0, width=0.0, step=2.0, depth=6, nbadsync_threshold=0)
    cd = o.frontend_audio(x, 2)
    items, _ = o.decode_window(cd)
    top = items[(items["pattern_idx"] == 5)]
    top = top[np.argmax(top["xb"])]
    assert top["nbadsync"] == 0
    assert np.array_equal((top["softbits_wo_sync"] > 0).astype(np.uint8), cw)
    assert top["is_message_present"] == 1 and top["ldpc_num_iterations"] == 0


def test_scan_duplicate_positions_and_periodic_patterns(orc):
    x, _ = _ping_window(8, snr=0.0, n_frames=5, freq=1498.0)
    o = orc.Oracle(center=1500.0, width=4.0, step=2.0, depth=8)
    cd = o.frontend_audio(x, 2)
    for p in range(8):
        xb = o.scan_xb(cd, 1, p)
        assert np.array_equal(xb[5184:5376].view(np.uint32), xb[0:192].view(np.uint32))   # same samples, same order: bit-exact
    xb5 = o.scan_xb(cd, 1, 5)
    assert np.allclose(xb5[:4320], xb5[864:5184], rtol=2e-5, atol=1e-4)                   # mask 111111: period 864
    xb6 = o.scan_xb(cd, 1, 6)
    assert np.allclose(xb6[:2592], xb6[2592:5184], rtol=2e-5, atol=1e-4)                  # mask 100100: period 2592


def test_averaging_gain(orc):
    """For repeated frames xb of the best candidate grows ~linearly with num_avg."""
    rng = np.random.default_rng(9)
    msg = synth.random_message(rng)
    x = synth.synth_audio(5184, [synth.Ping(msg, 0, 6, 1500.0, 10.0, 0.3)], 300.0, rng)
    o = orc.Oracle(center=1500.0, width=0.0, step=2.0, depth=6)
    items = o.scan(o.frontend_audio(x, 2))
    best = [items[items["pattern_idx"] == p]["xb"].max() for p in range(6)]
    ratios = np.array(best) / best[0]
    assert np.all(np.abs(ratios - np.arange(1, 7)) < 0.25 * np.arange(1, 7))


def test_slot_rule_keeps_top8_of_slice_maxima(orc):
    x, _ = _ping_window(10, snr=-10.0, n_frames=1, freq=1500.0)
    o = orc.Oracle(center=1500.0, width=0.0, step=2.0, depth=1)
    cd = o.frontend_audio(x, 2)
    items = o.scan(cd)
    xb = o.scan_xb(cd, 0, 0)
    slice_max = xb.reshape(21, 256).max(axis=1)
    assert np.allclose(np.sort(items["xb"]), np.sort(slice_max)[-8:], rtol=0, atol=0)
    for it in items:
        assert xb[it["pos"]] == it["xb"]


def test_all_zero_window_gives_nan_and_no_decode(orc):
    o = orc.Oracle(center=1500.0, width=4.0, step=2.0, depth=2, nbadsync_threshold=16)
    cd = o.frontend_audio(np.zeros(5184, dtype=np.int16), 2)
    assert np.isnan(cd).all()                                       # fac = 1/0 (main.cu:306-307 is unguarded)
    items, idx = o.decode_window(cd)
    assert (items["is_message_present"] == 0).all()


def test_frontend_properties(orc):
    o = orc.Oracle()
    n = np.arange(5184)
    # a tone at 1500+300 Hz passes, its image at -(1800) Hz is suppressed: output ~ complex exponential
    x = np.round(8000 * np.cos(2 * np.pi * 1800.0 * n / 12000.0)).astype(np.int16)
    for method in (1, 2):
        y = o.frontend_audio(x, method)[200:-200]
        ref = np.exp(2j * np.pi * 1800.0 * n[200:-200] / 12000.0)
        c = np.vdot(ref, y) / np.linalg.norm(ref) / np.linalg.norm(y)
        assert abs(c) > 0.995
    # FFT front end against numpy's float64 FFT of the same recipe
    xw = np.random.default_rng(11).integers(-3000, 3000, 5184).astype(np.int16)
    a = o.normalize_audio(xw)
    got = o.analytic_fft(a)
    buf = np.zeros(8192, dtype=np.complex128)
    buf[:5184] = a.astype(np.complex128) * (2.0 / 8192)
    spec = np.fft.fft(buf)
    f = np.arange(4096) * (12000.0 / 8192) - 1500.0
    h = np.where(np.abs(f) <= 900, 1.0, np.where(np.abs(f) <= 1100, 0.5 * (1 + np.cos(np.pi / 200 * (np.abs(f) - 900))), 0.0))
    spec[:4096] *= h
    spec[0] *= 0.5
    spec[4096:] = 0
    ref = np.fft.ifft(spec) * 8192
    assert np.abs(got - ref[:5184]).max() < 2e-5 * np.sqrt(np.mean(np.abs(ref) ** 2))
    # rms normalisation
    assert abs(np.sqrt(np.mean(a.real ** 2)) - 1.0) < 1e-5


def test_snr_tracker_and_gate(orc):
    t = orc.Snr()
    base = (np.ones(5184) * (1 + 0j)).astype(np.complex64)
    assert t.process(base) == -8                                    # peak == noise -> log10(0) -> clamp
    loud = base.copy()
    loud[:648] *= 4.0
    assert t.process(loud) == int(10 * np.log10(16.0 * 648 / (0.9 * 648 + 0.1 * (648 * (16 + 7) / 8)) - 1))
    m = np.zeros(77, dtype=np.int8)
    m[74:77] = [0, 0, 1]
    assert orc.message_gate(m)
    m[74:77] = [0, 1, 1]
    assert not orc.message_gate(m)                                  # i3 == 3
    m[74:77] = [0, 0, 0]
    m[71:74] = [0, 0, 1]
    assert not orc.message_gate(m)                                  # i3 == 0, n3 == 1
