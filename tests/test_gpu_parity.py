"""-m gpu: HIP path vs the CPU oracle, stage by stage and end to end, through the C ABI."""
import numpy as np
import pytest

from msk144cudecoder_amd import synth
from msk144cudecoder_amd.hipdecoder import STAGE_COLLECT, STAGE_INDEX, STAGE_LDPC, STAGE_SCAN, STAGE_SOFTBITS, unpack_message

import parity

pytestmark = pytest.mark.gpu


def _audio_window(seed, snr=3.0, n_frames=6, freq=1504.0, start=500, sigma=1000.0):
    rng = np.random.default_rng(seed)
    msg = synth.random_message(rng)
    pings = [synth.Ping(msg, start, n_frames, freq, snr, float(rng.uniform(0, 6.28)))] if n_frames else []
    return synth.synth_audio(5184, pings, sigma, rng), msg


def _iq_window(seed, snr=3.0, n_frames=6, freq=3.0, start=900, sigma=20.0):
    rng = np.random.default_rng(seed)
    msg = synth.random_message(rng)
    pings = [synth.Ping(msg, start, n_frames, freq, snr, float(rng.uniform(0, 6.28)))] if n_frames else []
    return synth.synth_iq(5184, pings, sigma, rng), msg


# ------------------------------------------------------------------------------------------------
# front ends
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_frontend_audio_fir_bit_exact(orc, hip, seed):
    x, _ = _audio_window(seed)
    o = orc.Oracle(width=8.0, depth=1)
    with hip.HipDecoder(width=8.0, depth=1, channels=1) as d:
        d.submit_audio(x)
        got = d.dump_analytic(0)
        seg = d.segment_power()[0]
    exp = o.frontend_audio(x, 2)
    assert np.array_equal(exp.view(np.uint32), got.view(np.uint32))          # bit-exact
    assert np.array_equal(orc.segment_power(exp).view(np.uint32), seg.view(np.uint32))


@pytest.mark.parametrize("seed", [4, 5])
def test_frontend_iq_bit_exact(orc, hip, seed):
    x, _ = _iq_window(seed)
    o = orc.Oracle(center=0.0, width=8.0, depth=1)
    with hip.HipDecoder(center=0.0, width=8.0, depth=1, read_mode=2, channels=1) as d:
        d.submit_iq(x)
        got = d.dump_analytic(0)
        seg = d.segment_power()[0]
    exp = o.frontend_iq(x)
    assert np.array_equal(exp.view(np.uint32), got.view(np.uint32))
    assert np.array_equal(orc.segment_power(exp).view(np.uint32), seg.view(np.uint32))


@pytest.mark.parametrize("seed", [6, 7])
def test_frontend_fft_tolerance(orc, hip, seed):
    x, _ = _audio_window(seed)
    o = orc.Oracle(width=8.0, depth=1)
    with hip.HipDecoder(width=8.0, depth=1, analytic_method=1, channels=1) as d:
        d.submit_audio(x)
        got = d.dump_analytic(0)
    exp = o.frontend_audio(x, 1)
    rms = np.sqrt(np.mean(np.abs(exp) ** 2))
    assert np.abs(got - exp).max() <= 1e-5 * rms                             # stated tolerance: 1e-5 of window rms


def test_frontend_extremes(orc, hip):
    """Full-scale int16 square wave and a single-sample impulse."""
    o = orc.Oracle(width=8.0, depth=1)
    a = np.where(np.arange(5184) % 8 < 4, 32767, -32768).astype(np.int16)
    b = np.zeros(5184, dtype=np.int16)
    b[2600] = 1
    with hip.HipDecoder(width=8.0, depth=1, channels=2) as d:
        d.submit_audio(np.stack([a, b]))
        for ch, x in enumerate((a, b)):
            assert np.array_equal(o.frontend_audio(x, 2).view(np.uint32), d.dump_analytic(ch).view(np.uint32))


# ------------------------------------------------------------------------------------------------
# scan / softbits / index / ldpc on seeded windows
# ------------------------------------------------------------------------------------------------
CONFIGS = [
    dict(center=1500.0, width=20.0, step=2.0, depth=6, nbadsync_threshold=2),
    dict(center=1500.0, width=10.0, step=1.0, depth=8, nbadsync_threshold=3),
    dict(center=1500.0, width=12.0, step=3.0, depth=1, nbadsync_threshold=0),
    dict(center=1500.0, width=6.0, step=0.7, depth=4, nbadsync_threshold=16),   # inexact step, every candidate gated
]


@pytest.mark.parametrize("ci", range(len(CONFIGS)))
@pytest.mark.parametrize("seed", [21, 22])
def test_stages_against_oracle(orc, hip, parity_report, ci, seed):
    cfg = CONFIGS[ci]
    x, msg = _audio_window(seed, snr=2.0 + seed % 3, n_frames=3 + seed % 5, freq=1500.0 + (seed % 7) - 3)
    o = orc.Oracle(threads=8, **cfg)
    cd = o.frontend_audio(x, 2)
    items_o, idx_o = o.decode_window(cd)
    with hip.HipDecoder(channels=1, **cfg) as d:
        assert (d.F, d.D, d.K) == (o.F, o.D, o.total_items)
        for b in range(d.F):
            assert np.float32(d.frequency(b)) == np.float32(o.frequency(b))
        d.submit_audio(x)
        d.decode()
        items_g = d.dump_candidates(0)
        idx_g = d.dump_indexes(0)
        res = d.results()

    # metadata
    for f in ("block_idx", "pattern_idx", "num_avg"):
        assert np.array_equal(items_o[f], items_g[f])
    assert np.array_equal(items_o["f0"].view(np.uint32), items_g["f0"].view(np.uint32))

    rep = parity.compare_scan(o, cd, items_o, items_g)
    # outside the two periodic patterns positions must match exactly up to verified near-ties
    assert rep["near_ties"] + rep["periodic_fallbacks"] <= rep["near_tie_limit"] <= 1, rep     # a few hundred slots: at most one, at the measured rate
    sb = parity.compare_softbits(o, cd, items_o, items_g)
    assert sb["nbadsync_marginal"] <= sb["nbadsync_marginal_limit"] <= 1, sb

    # index list: exactly the ascending list of items with nbadsync <= threshold (of the GPU's own nbadsync)
    exp_idx = np.nonzero(items_g["nbadsync"] <= cfg["nbadsync_threshold"])[0].astype(np.int32)
    assert np.array_equal(idx_g, exp_idx)
    if np.array_equal(items_g["nbadsync"], items_o["nbadsync"]):
        assert np.array_equal(idx_g, idx_o)

    ld = parity.compare_ldpc_against_oracle_on_gpu_llrs(orc, items_g, cfg["nbadsync_threshold"])
    assert ld["marginal_classes"] <= ld["marginal_limit"] <= 1, ld          # each one verified unstable by parity.verify_marginal_bp
    parity_report(f"stages_cfg{ci}_seed{seed}", dict(scan=rep, softbits=sb, ldpc=ld))

    # decoded payloads: same set of messages as the oracle, and it is the transmitted one
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o)
    if items_o["is_message_present"].any():
        assert parity.decoded_messages(items_g) == {bytes(msg)}

    # compact result list == accepted items in item order
    acc = np.nonzero(items_g["is_message_present"])[0]
    assert np.array_equal(res["item"], acc)
    for r in res:
        k = r["item"]
        assert np.array_equal(unpack_message(r["message"]), items_g["message"][k].astype(np.uint8))
        assert r["pos"] == items_g["pos"][k] and r["nbadsync"] == items_g["nbadsync"][k]
        assert r["ldpc_iterations"] == items_g["ldpc_num_iterations"][k] and r["ldpc_hard_errors"] == items_g["ldpc_num_hard_errors"][k]
        assert r["pattern_idx"] == items_g["pattern_idx"][k] and r["num_avg"] == items_g["num_avg"][k]
        assert np.float32(r["f0"]) == items_g["f0"][k]


@pytest.mark.parametrize("seed", [31, 32])
def test_iq_end_to_end(orc, hip, seed):
    cfg = dict(center=0.0, width=16.0, step=2.0, depth=6, nbadsync_threshold=2)
    x, msg = _iq_window(seed, snr=2.0, n_frames=5, freq=-4.0 + seed % 5)
    o = orc.Oracle(threads=8, **cfg)
    cd = o.frontend_iq(x)
    items_o, _ = o.decode_window(cd)
    with hip.HipDecoder(read_mode=2, channels=1, **cfg) as d:
        d.submit_iq(x)
        d.decode()
        items_g = d.dump_candidates(0)
    parity.compare_scan(o, cd, items_o, items_g)
    parity.compare_softbits(o, cd, items_o, items_g)
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o) == {bytes(msg)}


def test_ldpc_stage_isolated_exact(orc, hip):
    """Feed the ORACLE's candidates (pos, nbadsync, LLRs) to the GPU and run index+ldpc+collect only:
    integer outputs must then match the oracle exactly."""
    cfg = dict(center=1500.0, width=20.0, step=2.0, depth=6, nbadsync_threshold=2)
    x, msg = _audio_window(41, snr=1.0, n_frames=6)
    o = orc.Oracle(threads=8, **cfg)
    cd = o.frontend_audio(x, 2)
    items_o, idx_o = o.decode_window(cd)
    with hip.HipDecoder(channels=1, **cfg) as d:
        d.submit_audio(x)
        d.load_candidates(items_o.view(hip.CANDIDATE_DTYPE), 0)
        d.decode(STAGE_INDEX | STAGE_LDPC | STAGE_COLLECT)
        items_g = d.dump_candidates(0)
        idx_g = d.dump_indexes(0)
    assert np.array_equal(idx_g, idx_o)
    assert np.array_equal(items_g["is_message_present"], items_o["is_message_present"])
    pres = items_o["is_message_present"] == 1
    assert pres.sum() > 0
    assert np.array_equal(items_g["message"][pres], items_o["message"][pres])
    assert np.array_equal(items_g["ldpc_num_iterations"][pres], items_o["ldpc_num_iterations"][pres])
    assert np.array_equal(items_g["ldpc_num_hard_errors"][pres], items_o["ldpc_num_hard_errors"][pres])


def test_softbits_stage_isolated(orc, hip):
    """Scan positions taken from the oracle, softbits stage alone."""
    cfg = dict(center=1500.0, width=20.0, step=2.0, depth=8, nbadsync_threshold=2)
    x, _ = _audio_window(43, snr=0.0, n_frames=8)
    o = orc.Oracle(threads=8, **cfg)
    cd = o.frontend_audio(x, 2)
    items_o, _ = o.decode_window(cd)
    seed_items = items_o.copy()
    seed_items["softbits_wo_sync"] = 0
    seed_items["nbadsync"] = 99
    with hip.HipDecoder(channels=1, **cfg) as d:
        d.submit_audio(x)
        d.load_candidates(seed_items.view(hip.CANDIDATE_DTYPE), 0)
        d.decode(STAGE_SOFTBITS)
        items_g = d.dump_candidates(0)
    assert np.array_equal(items_g["pos"], items_o["pos"])
    rep = parity.compare_softbits(o, cd, items_o, items_g)
    assert rep["nbadsync_marginal"] <= 1


# ------------------------------------------------------------------------------------------------
# edge cases
# ------------------------------------------------------------------------------------------------
def test_all_zero_window_does_not_hang(hip):
    """fac = 1/0 = inf -> NaN everywhere (main.cu:306-307 is unguarded); kernels must terminate and
    report no decode."""
    with hip.HipDecoder(width=10.0, depth=6, nbadsync_threshold=16, channels=2) as d:
        d.submit_audio(np.zeros((2, 5184), dtype=np.int16))
        d.decode()
        assert d.result_count() == 0
        items = d.dump_candidates(0)
        assert (items["pos"] == 0).all() and (items["xb"] == 0).all()   # no slice maximum ever beats the empty slots


def test_state_errors(hip):
    with hip.HipDecoder(width=4.0, depth=1, channels=1) as d:
        with pytest.raises(hip.Msk144Error):
            d.decode()                                   # nothing submitted yet
        with pytest.raises(hip.Msk144Error):
            d.submit_iq(np.zeros(2 * 5184, dtype=np.int8))  # audio handle
    with pytest.raises(hip.Msk144Error):
        hip.HipDecoder(step=0.0)
    with pytest.raises(hip.Msk144Error):
        hip.HipDecoder(read_mode=3)
    with pytest.raises(hip.Msk144Error):
        hip.HipDecoder(channels=0)


def test_scan_depth_clamp(hip):
    with hip.HipDecoder(width=4.0, depth=0, channels=1) as d:
        assert d.D == 1
    with hip.HipDecoder(width=4.0, depth=99, channels=1) as d:
        assert d.D == 8


def test_batch_equals_single(hip):
    """Channels are independent: a window decoded inside a batch gives the same candidates, bit for
    bit, as the same window decoded alone."""
    cfg = dict(center=1500.0, width=12.0, step=2.0, depth=6, nbadsync_threshold=2)
    wins = np.stack([_audio_window(50 + i, snr=2.0, n_frames=i % 7, freq=1500.0 + i % 5)[0] for i in range(12)])
    with hip.HipDecoder(channels=12, **cfg) as d:
        d.submit_audio(wins)
        d.decode()
        batch = [d.dump_candidates(c) for c in range(12)]
        res = d.results()
    with hip.HipDecoder(channels=1, **cfg) as d1:
        for c in (0, 5, 11):
            d1.submit_audio(wins[c])
            d1.decode()
            single = d1.dump_candidates(0)
            assert batch[c].tobytes() == single.tobytes()
    # result list is ordered by (channel, item)
    key = res["channel"].astype(np.int64) * 10 ** 6 + res["item"]
    assert np.all(np.diff(key) > 0)
    for c in range(12):
        assert (res["channel"] == c).sum() == batch[c]["is_message_present"].sum()


def test_device_pointer_submit_and_determinism(hip):
    torch = pytest.importorskip("torch")
    cfg = dict(center=1500.0, width=12.0, step=2.0, depth=6, nbadsync_threshold=2)
    wins = np.stack([_audio_window(70 + i, n_frames=4)[0] for i in range(4)])
    t = torch.from_numpy(wins).cuda()
    with hip.HipDecoder(channels=4, **cfg) as d:
        d.set_stream(torch.cuda.current_stream().cuda_stream)
        d.submit_audio_device(t.data_ptr())
        d.decode()
        a = [d.dump_candidates(c).tobytes() for c in range(4)]
        d.submit_audio(wins)
        d.decode()
        b = [d.dump_candidates(c).tobytes() for c in range(4)]
    assert a == b


def test_fft_frontend_end_to_end(orc, hip):
    """--analytic-method=1: decodes equal the oracle's (front end within tolerance, everything after compared
    with the oracle fed by the GPU's own analytic window)."""
    cfg = dict(center=1500.0, width=12.0, step=2.0, depth=4, nbadsync_threshold=1)
    x, msg = _audio_window(61, snr=4.0, n_frames=5, freq=1502.0)
    o = orc.Oracle(threads=8, **cfg)
    with hip.HipDecoder(analytic_method=1, channels=1, **cfg) as d:
        d.submit_audio(x)
        d.decode()
        cd_g = d.dump_analytic(0)
        items_g = d.dump_candidates(0)
    cd_o = o.frontend_audio(x, 1)
    assert np.abs(cd_g - cd_o).max() <= 1e-5 * np.sqrt(np.mean(np.abs(cd_o) ** 2))
    items_o, _ = o.decode_window(cd_g)            # same analytic input for both
    parity.compare_scan(o, cd_g, items_o, items_g)
    parity.compare_softbits(o, cd_g, items_o, items_g)
    assert parity.decoded_messages(items_g) == parity.decoded_messages(items_o) == {bytes(msg)}


def test_fft_frontend_1024_channel_batch(orc, hip, parity_report):
    """--analytic-method=1 at the bench's batch size (analytic_fft.cu:84-157, one workgroup per channel): 1024 windows in one launch.
    (1) batch == single: a channel's analytic window and segment powers are bit-for-bit what a one-channel handle computes for
    the same samples; (2) every sampled channel within 1e-5 of the window rms of the oracle's FFT front end; (3) end to end: the
    payload set of every pinged sample channel equals the oracle's, stage by stage on the GPU's own analytic window."""
    n_ch = 1024
    cfg = dict(center=1500.0, width=12.0, step=2.0, depth=4, nbadsync_threshold=1)
    rng = np.random.default_rng(4242)
    wins = np.rint(rng.normal(0.0, 1000.0, size=(n_ch, 5184))).astype(np.int16)
    pinged = list(range(0, n_ch, 64))
    msgs = {}
    for c in pinged:
        w, m = _audio_window(7000 + c, snr=5.0, n_frames=5, freq=1498.0 + (c // 64) % 5, start=100 + 7 * (c // 64))
        wins[c] = w
        msgs[c] = m
    wins[5] = 0                                   # a muted channel in the batch: 1/0 normalisation, NaNs stay in their own channel
    sample = pinged + [1, 2, 3, 4, 6, 511, 1023]
    o = orc.Oracle(threads=8, **cfg)
    worst = 0.0
    with hip.HipDecoder(analytic_method=1, channels=n_ch, llr_block_channels=n_ch, **cfg) as d:
        d.submit_audio(wins)
        d.decode()
        seg = d.segment_power()
        batch = {c: d.dump_analytic(c) for c in sample + [5]}
        items = {c: d.dump_candidates(c) for c in pinged}
    assert not np.isfinite(batch[5]).any()
    with hip.HipDecoder(analytic_method=1, channels=1, **cfg) as d1:
        for c in sample:
            d1.submit_audio(wins[c])
            assert np.array_equal(d1.dump_analytic(0).view(np.uint32), batch[c].view(np.uint32)), c      # (1)
            assert np.array_equal(d1.segment_power()[0].view(np.uint32), seg[c].view(np.uint32)), c
    for c in sample:
        exp = o.frontend_audio(wins[c], 1)
        rms = np.sqrt(np.mean(np.abs(exp) ** 2))
        err = float(np.abs(batch[c] - exp).max() / rms)
        worst = max(worst, err)
        assert err <= 1e-5, (c, err)                                                                      # (2)
    for c in pinged:
        items_o, _ = o.decode_window(batch[c])
        parity.compare_scan(o, batch[c], items_o, items[c])
        parity.compare_softbits(o, batch[c], items_o, items[c])
        assert parity.decoded_messages(items[c]) == parity.decoded_messages(items_o) == {bytes(msgs[c])}, c   # (3)
    parity_report("fft_frontend_1024_channels", dict(channels=n_ch, sampled=len(sample), max_err_over_rms=worst, tolerance=1e-5, batch_equals_single=True,
                                                     pinged_channels_decoded=len(pinged)))


def test_iq_batch_low_snr(orc, hip):
    """configs[4] shape at reduced size: IQ, low SNR, threshold 3; every channel equals its oracle decode set."""
    cfg = dict(center=0.0, width=24.0, step=1.0, depth=6, nbadsync_threshold=3)
    wins, msgs = [], []
    for i in range(6):
        w, m = _iq_window(90 + i, snr=-4.0 + i % 3, n_frames=4 + i % 3, freq=-6.0 + 2 * i, start=300 * i)
        wins.append(w)
        msgs.append(m)
    o = orc.Oracle(threads=8, **cfg)
    with hip.HipDecoder(read_mode=2, channels=6, **cfg) as d:
        d.submit_iq(np.stack(wins))
        d.decode()
        got = [d.dump_candidates(c) for c in range(6)]
    for c in range(6):
        items_o, _ = o.decode_window(o.frontend_iq(wins[c]))
        assert parity.decoded_messages(got[c]) == parity.decoded_messages(items_o)
    assert sum(bytes(msgs[c]) in parity.decoded_messages(got[c]) for c in range(6)) >= 4


def test_result_overflow_is_reported(hip):
    """max_results smaller than the number of decodes: count is exact, list truncated, status EOVERFLOW."""
    import ctypes as C
    cfg = dict(center=1500.0, width=20.0, step=1.0, depth=6, nbadsync_threshold=2)
    x, _ = _audio_window(3, snr=6.0, n_frames=6)
    with hip.HipDecoder(channels=1, max_results=4, **cfg) as d:
        d.submit_audio(x)
        d.decode()
        n = d.result_count()
        assert n > 4
        out = np.zeros(n, dtype=hip.RESULT_DTYPE)
        got = C.c_int32()
        rc = d.L.msk144_results(d.h, out.ctypes.data_as(C.c_void_p), n, C.byref(got))
        assert rc == -5 and got.value == n                      # MSK144_EOVERFLOW, exact count
        assert (out["item"][:4] > 0).all() and (out["item"][4:] == 0).all()
        assert np.all(np.diff(out["item"][:4]) > 0)


def test_scan_tiny_amplitude_window(orc, hip, parity_report):
    """ADVICE r3 (scan.hip octet merge): an analytic window scaled until |S|^2 is a DENORMAL float for the noise positions.  The
    kernels are built with denormals preserved (hipcc default, no -fgpu-flush-denormals-to-zero): the octet maxima survive the DPP
    max bit for bit, every slot is written, and the scan agrees with the oracle as on a full-scale window.  (A flushing build would
    have left s_oct cells unwritten with the former == test; the >= test stores in that case too.)"""
    cfg = dict(center=1500.0, width=8.0, step=1.0, depth=6, nbadsync_threshold=2)
    x, _ = _audio_window(41, snr=3.0, n_frames=4, freq=1501.0)
    o = orc.Oracle(threads=8, **cfg)
    cd = (o.frontend_audio(x, 2) * np.float32(2e-21)).astype(np.complex64)      # rms 2e-21: |S|^2 of noise positions ~1e-40
    items_o = o.scan(cd)
    assert float(np.min(items_o["xb"].astype(np.float64) ** 2)) < 1.17e-38     # denormal |S|^2 among the stored maxima
    with hip.HipDecoder(channels=1, **cfg) as d:
        d.submit_analytic(cd)
        d.decode(hip.STAGE_SCAN)
        items_g = d.dump_candidates(0)
    assert np.all(items_g["xb"] > 0) and np.all(np.isfinite(items_g["xb"]))
    rep = parity.compare_scan(o, cd, items_o, items_g, near_tie_factor=100.0)    # denormals carry fewer bits: the near-tie rate of full-precision windows does not apply
    assert rep["near_ties"] + rep["periodic_fallbacks"] <= rep["near_tie_limit"] <= 4, rep
    parity_report("scan_tiny_amplitude", rep)
