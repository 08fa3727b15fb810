"""-m gpu: the pinned staging slots of the C ABI (msk144_input_slot .. msk144_fetch_wait, include/msk144hip.h) - the pipelined form
of the reference's per-hop sequence fread -> H2D -> kernels -> D2H -> host loop (main.cu:261-422, 474-525).  A hop decoded through
a slot must return exactly what the serial calls return, whatever is in flight on the other slot."""
import threading

import numpy as np
import pytest

from msk144cudecoder_amd import synth

pytestmark = pytest.mark.gpu
CFG = dict(center=1500.0, width=60.0, step=1.0, depth=6, nbadsync_threshold=3)


def _windows(n_hops, channels, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_hops):
        w = np.empty((channels, 5184), dtype=np.int16)
        for c in range(channels):
            pings = [synth.Ping(synth.random_message(rng), int(rng.integers(0, 2000)), int(rng.integers(2, 6)), 1500.0 + float(rng.uniform(-25, 25)), 3.0,
                                float(rng.uniform(0, 6)))] if c % 2 == 0 else []
            w[c] = synth.synth_audio(5184, pings, 1000.0, rng)
        out.append(w)
    return out


def _serial(hip, wins, channels, **kw):
    ref = []
    with hip.HipDecoder(channels=channels, max_results=1 << 16, **CFG, **kw) as d:
        for w in wins:
            d.submit_audio(w)
            d.decode()
            ref.append((d.results().copy(), d.segment_power().copy()))
    return ref


def test_pipelined_slots_equal_serial_calls(hip):
    channels, hops = 12, 6
    wins = _windows(hops, channels, 11)
    ref = _serial(hip, wins, channels)
    assert sum(len(r) for r, _ in ref) > 20
    with hip.HipDecoder(channels=channels, max_results=1 << 16, **CFG) as d:
        got = []
        # software pipeline: hop n is submitted before hop n-1 is fetched, so both slots are in flight at once
        for n, w in enumerate(wins):
            s = n % 2
            d.input_slot(s)[:] = w
            d.submit_slot(s)
            d.decode()
            d.fetch_async(s)
            if n >= 1:
                got.append(d.fetch_wait(1 - s))
        got.append(d.fetch_wait((hops - 1) % 2))
    for (r0, p0), (r1, p1) in zip(ref, got):
        assert r0.tobytes() == r1.tobytes()
        assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))


def test_fetch_wait_on_a_second_thread(hip):
    """The post-processing thread of msk144hipdecoder waits for slot s while the ingest thread submits slot 1 - s."""
    channels, hops = 8, 8
    wins = _windows(hops, channels, 12)
    ref = _serial(hip, wins, channels)
    got, errors = [None] * hops, []
    with hip.HipDecoder(channels=channels, max_results=1 << 16, **CFG) as d:
        free = [threading.Semaphore(1), threading.Semaphore(1)]
        queue, cond = [], threading.Condition()

        def consumer():
            try:
                for n in range(hops):
                    with cond:
                        cond.wait_for(lambda: queue)
                        k = queue.pop(0)
                    got[k] = d.fetch_wait(k % 2)
                    free[k % 2].release()
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                for f in free:
                    f.release()

        t = threading.Thread(target=consumer)
        t.start()
        for n, w in enumerate(wins):
            s = n % 2
            free[s].acquire()
            if errors:
                break
            d.input_slot(s)[:] = w
            d.submit_slot(s)
            d.decode()
            d.fetch_async(s)
            with cond:
                queue.append(n)
                cond.notify()
        t.join(timeout=120)
        assert not t.is_alive() and not errors, errors
    for (r0, p0), (r1, p1) in zip(ref, got):
        assert r0.tobytes() == r1.tobytes() and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))


def test_short_estimate_takes_the_remainder_copy(hip):
    """The asynchronous copy covers max(1024, 2 x last count + 256) records; a hop with more decodes than that is completed by the
    synchronous remainder copy from the slot's own device list - also when the other slot has decoded in between."""
    cfg = dict(center=1500.0, width=40.0, step=0.25, depth=8, nbadsync_threshold=6)
    channels = 48
    rng = np.random.default_rng(5)
    quiet = np.stack([synth.synth_audio(5184, [], 1000.0, rng) for _ in range(channels)])
    msg = synth.random_message(rng)
    loud = np.stack([synth.synth_audio(5184, [synth.Ping(msg, 200, 5, 1500.0 + 2.0, 12.0, 0.3)], 1000.0, rng) for _ in range(channels)])
    with hip.HipDecoder(channels=channels, max_results=1 << 18, **cfg) as d:
        d.submit_audio(loud)
        d.decode()
        want_loud = d.results().copy()
        d.submit_audio(quiet)
        d.decode()
        want_quiet = d.results().copy()
    assert len(want_loud) > 1024 + 256 and len(want_quiet) < 256          # the estimate after a quiet hop is 1024: short for the loud one
    with hip.HipDecoder(channels=channels, max_results=1 << 18, **cfg) as d:
        d.input_slot(0)[:] = quiet
        d.submit_slot(0)
        d.decode()
        d.fetch_async(0)
        r, _ = d.fetch_wait(0)
        assert r.tobytes() == want_quiet.tobytes()
        d.input_slot(1)[:] = loud
        d.submit_slot(1)
        d.decode()
        d.fetch_async(1)
        d.input_slot(0)[:] = quiet                                          # slot 0 decodes while slot 1 still waits to be fetched
        d.submit_slot(0)
        d.decode()
        d.fetch_async(0)
        r1, _ = d.fetch_wait(1)
        r0, _ = d.fetch_wait(0)
        assert r1.tobytes() == want_loud.tobytes() and r0.tobytes() == want_quiet.tobytes()


def test_slot_state_errors(hip):
    with hip.HipDecoder(channels=2, **CFG) as d:
        with pytest.raises(hip.Msk144Error) as e:
            d.fetch_wait(0)                                                 # nothing in flight
        assert e.value.code == -4
        with pytest.raises(hip.Msk144Error):
            d.input_slot(2)
        d.input_slot(0)[:] = 0
        d.submit_slot(0)
        d.decode()
        with pytest.raises(hip.Msk144Error) as e:
            d.fetch_async(1)                                                # not the slot that was decoded
        assert e.value.code == -4
        d.fetch_async(0)
        with pytest.raises(hip.Msk144Error) as e:
            d.submit_slot(0)                                                # results not fetched yet
        assert e.value.code == -4
        d.fetch_wait(0)
        d.submit_slot(0)
        d.decode()
        assert len(d.results()) == 0                                        # the plain calls read the list of the slot just decoded


def test_partial_hop_covers_only_its_streams(hip):
    """msk144_submit_slot_n: a hop over the first n windows of the slot gives exactly the records of an n-channel handle (every
    kernel, copy and result sized for n), whatever the handle's capacity and whatever a previous, larger hop left behind."""
    capacity = 100                                   # 64-channel blocks: blocked staging with gated softbits, two blocks for n = 70
    wins = _windows(1, capacity, 21)[0]
    with hip.HipDecoder(channels=capacity, max_results=1 << 16, llr_block_channels=64, **CFG) as d:
        d.input_slot(0)[:] = wins
        d.submit_slot(0)
        d.decode()
        d.fetch_async(0)
        full, seg_full = d.fetch_wait(0)
        for n, slot in ((5, 1), (70, 0), (1, 1)):
            d.input_slot(slot)[:n] = wins[:n]
            d.submit_slot(slot, n)
            d.decode()
            d.fetch_async(slot)
            got, seg = d.fetch_wait(slot)
            with hip.HipDecoder(channels=n, max_results=1 << 16, **CFG) as small:
                small.submit_audio(wins[:n])
                small.decode()
                want = small.results().copy()
                want_seg = small.segment_power().copy()
            assert got.tobytes() == want.tobytes(), n
            assert got.tobytes() == full[full["channel"] < n].tobytes(), n
            assert np.array_equal(seg[:n].view(np.uint32), want_seg.view(np.uint32))
        with pytest.raises(hip.Msk144Error):
            d.submit_slot(0, capacity + 1)
    assert len(full) > 50


@pytest.mark.parametrize("read_mode", [1, 2])
def test_device_side_hop_ring_equals_host_windows(hip, read_mode):
    """msk144_hop_slot / msk144_push_hops: the device keeps every stream's 50 %-overlap window (main.cu:271-294, 337-359); shipping
    2592 new samples per stream and hop must give exactly the records of shipping the windows a host-side ring would hold - for
    streams that skip hops, for first hops in the middle of a run, on both slots, in both read modes."""
    channels, n_hops = 6, 5
    rng = np.random.default_rng(31 + read_mode)
    per = 1 if read_mode == 1 else 2
    cfg = dict(CFG, center=1500.0 if read_mode == 1 else 0.0)
    total = 5184 + n_hops * 2592
    streams = []
    for c in range(channels):
        pings = [synth.Ping(synth.random_message(rng), int(rng.integers(0, total - 6 * 864)), 6, cfg["center"] + float(rng.uniform(-20, 20)), 4.0, float(rng.uniform(0, 6)))]
        streams.append(synth.synth_audio(total, pings, 1000.0, rng) if read_mode == 1 else synth.synth_iq(total, pings, 20.0, rng))
    # which streams have a hop in which batch; stream 4 joins late (its first hop is batch 2)
    schedule = [[0, 1, 2, 3, 5], [0, 2, 3, 5], [1, 2, 4], [0, 1, 2, 3, 4, 5], [2, 4, 5], [0, 1, 3]]
    consumed = [0] * channels                      # samples of each stream handed over so far
    window = [None] * channels                     # the host-side ring, for the expected records
    H = 2592 * per
    got, want = [], []
    with hip.HipDecoder(channels=channels, read_mode=read_mode, max_results=1 << 16, **cfg) as d, \
            hip.HipDecoder(channels=channels, read_mode=read_mode, max_results=1 << 16, **cfg) as ref:
        for b, members in enumerate(schedule):
            s = b % 2
            hops, first_halves, ids, is_first = d.hop_slot(s)
            for j, c in enumerate(members):
                x = streams[c]
                if consumed[c] == 0:
                    first_halves[j, :] = x[:H]
                    hops[j, :] = x[H:2 * H]
                    window[c] = x[:2 * H].copy()
                    consumed[c] = 2 * H
                    is_first[j] = 1
                else:
                    hops[j, :] = x[consumed[c]:consumed[c] + H]
                    window[c] = np.concatenate([window[c][H:], x[consumed[c]:consumed[c] + H]])
                    consumed[c] += H
                    is_first[j] = 0
                ids[j] = c
            d.push_hops(s, len(members))
            d.decode()
            d.fetch_async(s)
            if b >= 1:
                got.append(d.fetch_wait(1 - s))
            ref.input_slot(0)[:len(members)] = np.stack([window[c] for c in members])
            ref.submit_slot(0, len(members))
            ref.decode()
            ref.fetch_async(0)
            want.append(ref.fetch_wait(0))
        got.append(d.fetch_wait((len(schedule) - 1) % 2))
        with pytest.raises(hip.Msk144Error):
            d.hop_slot(0)[2][:2] = [3, 1]                                   # not ascending
            d.push_hops(0, 2)
    assert sum(len(r) for r, _ in want) > 10
    for (r0, p0), (r1, p1), members in zip(want, got, schedule):
        assert r0.tobytes() == r1.tobytes()
        assert np.array_equal(p0[:len(members)].view(np.uint32), p1[:len(members)].view(np.uint32))


def test_two_live_handles_decode_concurrently_from_two_threads(hip):
    """Handles are independent (include/msk144hip.h; the multi-device stream program keeps one per GPU, each driven by its own
    threads): two handles alive at once, each decoding its own six hops through its slots from its own thread at the same time, with
    different channel counts and channel bases, must return exactly what each returns when it runs alone.  Also the new ABI entries:
    msk144_device_count >= 1 and a plausible shader clock from msk144_clock_probe while a decode is in flight."""
    assert hip.device_count() >= 1
    sets = [(_windows(6, 10, 31), 10, 0), (_windows(6, 7, 32), 7, 1000)]
    refs = []
    for wins, channels, base in sets:
        ref = []
        with hip.HipDecoder(channels=channels, max_results=1 << 16, **CFG) as d:
            d.set_channel_base(base)
            for w in wins:
                d.submit_audio(w)
                d.decode()
                ref.append(d.results().copy())
        refs.append(ref)
    assert all(sum(len(r) for r in ref) > 10 for ref in refs)
    got = [[], []]
    errors = []
    start = threading.Barrier(2)
    decs = [hip.HipDecoder(channels=channels, max_results=1 << 16, **CFG) for _, channels, _ in sets]
    clocks = []

    def work(k):
        try:
            d = decs[k]
            wins, _, base = sets[k]
            d.set_channel_base(base)
            start.wait()
            for n, w in enumerate(wins):
                s = n % 2
                d.input_slot(s)[:] = w
                d.submit_slot(s)
                d.decode()
                d.fetch_async(s)
                if k == 0 and n == 2:
                    clocks.append(d.clock_probe(500))
                if n >= 1:
                    got[k].append(d.fetch_wait(1 - s)[0].copy())
            got[k].append(d.fetch_wait((len(wins) - 1) % 2)[0].copy())
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for d in decs:
        d.close()
    assert not errors, errors
    for k in range(2):
        assert len(got[k]) == len(refs[k])
        for a, b in zip(refs[k], got[k]):
            assert a.tobytes() == b.tobytes()
        base = sets[k][2]
        assert all(((r["channel"] >= base) & (r["channel"] < base + sets[k][1])).all() for r in got[k] if len(r))
    assert clocks and 500.0 < clocks[0] < 3000.0, clocks
