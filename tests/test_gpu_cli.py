"""-m gpu: the stdin -> stdout program (msk144hipdecoder) against lines predicted from the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from msk144cudecoder_amd import synth

import pack77
from test_host import Accepted

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "msk144cudecoder_amd", "msk144hipdecoder")
HOST_SO = os.path.join(ROOT, "msk144cudecoder_amd", "libmsk144host.so")


def _expected_lines(orc, stream, cfg, read_mode, quirk):
    """Oracle decode of every window + the host library's post-processing (the same C++ the CLI links)."""
    from oracle import oracle_cli
    return oracle_cli.decode_stream(stream, cfg, read_mode, 2, quirk=quirk, threads=8)


def _run(args, data):
    p = subprocess.run([EXE] + args, input=data, capture_output=True, timeout=300)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _lines(out):
    lines = out.strip().split("\n")
    assert lines[-1] == "Done"
    return [re.sub(r"date=\d{14}", "date=X", l) for l in lines[:-1]]


def _both_modes(orc, args, stream, cfg, read_mode=1):
    """Identical arguments => identical stdout (date= aside): the DEFAULT run against the oracle-driven decoder with the reference's
    per-window text cache (main.cu:437-445, 497-504), --strict-decode against the per-payload decode, and --reference-decode-cache
    (accepted for compatibility) changes nothing.  Returns (default lines, strict lines, stderr of the default run)."""
    rc, out, err = _run(args, stream.tobytes())
    assert rc == 0, err
    got = _lines(out)
    assert got == _expected_lines(orc, stream, cfg, read_mode, quirk=True)
    rc, out_s, err_s = _run(args + ["--strict-decode"], stream.tobytes())
    assert rc == 0, err_s
    strict = _lines(out_s)
    assert strict == _expected_lines(orc, stream, cfg, read_mode, quirk=False)
    rc, out_c, _ = _run(args + ["--reference-decode-cache"], stream.tobytes())
    assert rc == 0 and _lines(out_c) == got
    # the default run hands copies to the lower slot of their group and gates before the full demodulation (the kernels bench.py times);
    # --every-slot computes each slot on its own as the reference does: the same stdout
    rc, out_e, err_e = _run(args + ["--every-slot"], stream.tobytes())
    assert rc == 0 and _lines(out_e) == got, err_e
    return got, strict, err


def test_cli_audio_stream(orc):
    rng = np.random.default_rng(77)
    n = 5184 + 4 * 2592
    msgs = [synth.random_message(rng) for _ in range(2)]
    pings = [synth.Ping(msgs[0], 1500, 6, 1503.0, 4.0, 0.4), synth.Ping(msgs[1], 9000, 5, 1495.0, 5.0, 1.4)]
    stream = synth.synth_audio(n, pings, 1000.0, rng)
    cfg = dict(center=1500.0, width=20.0, step=1.0, depth=6, nbadsync_threshold=2)
    args = ["--search-width=20", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=2"]
    got, strict, err = _both_modes(orc, args, stream, cfg)
    assert "Actual parameters:" in err and "Incomplete read error. rc=0" in err
    assert len(got) >= 2
    # every line has the reference's field layout (main.cu:409-417)
    pat = re.compile(r"^\*\*\*  snr=[ -]?\d+; f0=\s*[\d.]+; num_avg=\d; nbadsync=\d+; pattern_idx=\d; date=X; msg='.*'; $")
    assert all(pat.match(l) for l in got + strict)


def test_cli_two_payloads_in_one_window(orc):
    """Two stations in the SAME window (main.cu:480-525 walks the items in index order, lowest frequency bin first): the reference's cache
    hands the second station the first one's text, so the default run prints ONE text for the window (the lower bin's) where
    --strict-decode prints both - each mode line for line what the oracle-driven decoder predicts."""
    rng = np.random.default_rng(79)
    lo, hi = pack77.pack_standard("CQ", "K1ABC", "FN42"), pack77.pack_standard("W9XYZ", "G4ABC", "-07")
    pings = [synth.Ping(hi, 700, 5, 1506.0, 6.0, 0.3), synth.Ping(lo, 900, 5, 1494.0, 6.0, 1.1)]
    stream = synth.synth_audio(5184, pings, 1000.0, rng)       # exactly one window
    cfg = dict(center=1500.0, width=20.0, step=1.0, depth=6, nbadsync_threshold=2)
    args = ["--search-width=20", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=2"]
    got, strict, _ = _both_modes(orc, args, stream, cfg)
    text = lambda ls: sorted({re.search(r"msg='(.*)'; $", l).group(1) for l in ls})
    assert text(strict) == ["CQ K1ABC FN42", "W9XYZ G4ABC -07"]
    assert text(got) == ["CQ K1ABC FN42"]                      # the window's first accepted candidate sits in the lowest decoded bin


def test_cli_first_candidate_fails_the_type_gate(orc):
    """The window's first accepted candidate carries i3 = 3, which decode_message refuses before unpack77 (decode_softbits.cpp:25-30):
    the reference caches that failure for the whole window (main.cu:497-504), so the default run prints nothing for it although a
    valid message sits a few bins higher; --strict-decode prints the valid one.  The second window of the stream, which holds only the
    valid station, prints in both modes."""
    rng = np.random.default_rng(83)
    bad = synth.random_message(rng).copy()
    bad[74:77] = (0, 1, 1)                                      # i3 = 3
    good = pack77.pack_standard("CQ", "DL1ABC", "JO62")
    pings = [synth.Ping(bad, 200, 2, 1493.0, 7.0, 0.2), synth.Ping(good, 2200, 3, 1507.0, 7.0, 0.9), synth.Ping(good, 5184 + 600, 5, 1507.0, 7.0, 0.5)]
    stream = synth.synth_audio(5184 + 2 * 2592, pings, 1000.0, rng)
    cfg = dict(center=1500.0, width=20.0, step=1.0, depth=6, nbadsync_threshold=2)
    args = ["--search-width=20", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=2"]
    got, strict, _ = _both_modes(orc, args, stream, cfg)
    assert strict and all("msg='CQ DL1ABC JO62'" in l for l in strict)
    assert len(strict) == 3 and got == strict[1:]               # the first window is silenced by the cached failure, the later two are not


def test_cli_iq_and_option_quirks(orc):
    rng = np.random.default_rng(78)
    msg = synth.random_message(rng)
    stream = synth.synth_iq(5184 + 2592, [synth.Ping(msg, 800, 6, 2.0, 4.0, 0.1)], 20.0, rng)
    cfg = dict(center=0.0, width=12.0, step=2.0, depth=4, nbadsync_threshold=1)
    got, _, _ = _both_modes(orc, ["--read-mode=2", "--search-width=12"], stream, cfg, read_mode=2)
    assert len(got) >= 1
    # short input: error on stderr, Done on stdout, exit 0 (main.cu:274-278,424)
    rc, out, err = _run([], b"\x00" * 100)
    assert rc == 0 and out.strip() == "Done" and "Incomplete read error. rc=50" in err
    # bad read mode without a centre frequency -> exit 2 (main.cu:193-207)
    rc, out, err = _run(["--read-mode=5"], b"")
    assert rc == 2 and "Wrong read mode 5" in err
    rc, out, _ = _run(["--help"], b"")
    assert rc == 0 and "--nbadsync-threshold" in out


def test_cli_multi_stream_equals_single_streams(tmp_path):
    """--inputs=a,b,c decodes the streams as one GPU batch per hop; each channel's lines must be exactly what
    the single-stream program prints for that file (streams of different length end independently)."""
    rng = np.random.default_rng(80)
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2"]
    files, singles = [], []
    for i, n_hops in enumerate((4, 2, 5)):
        n = 5184 + n_hops * 2592
        text = [("CQ", "K1ABC", "FN42"), None, ("K1ABC", "W9XYZ", "-11")][i]
        pings = [] if text is None else [synth.Ping(pack77.pack_standard(*text), 1000 + 3000 * i, 6, 1500.0 + 2 * i, 5.0, 0.3 * i)]
        stream = synth.synth_audio(n, pings, 1000.0, rng)
        path = tmp_path / f"s{i}.s16"
        path.write_bytes(stream.tobytes())
        files.append(str(path))
        rc, out, _ = _run(args, stream.tobytes())
        assert rc == 0
        singles.append([re.sub(r"date=\d{14}", "date=X", l) for l in out.strip().split("\n")[:-1]])
    assert len(singles[0]) >= 1 and len(singles[2]) >= 1 and singles[1] == []
    assert all("msg='CQ K1ABC FN42'" in l for l in singles[0]) and all("msg='K1ABC W9XYZ -11'" in l for l in singles[2])
    rc, out, err = _run(args + ["--inputs=" + ",".join(files)], b"")
    assert rc == 0, err
    lines = out.strip().split("\n")
    assert lines[-1] == "Done"
    per_ch = {0: [], 1: [], 2: []}
    for l in lines[:-1]:
        m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
        assert m, l
        per_ch[int(m.group(1))].append("***  " + re.sub(r"date=\d{14}", "date=X", m.group(2)))
    for c in range(3):
        assert per_ch[c] == singles[c]
    assert err.count("Incomplete read error") == 3 and "3 input streams per GPU batch" in err


def test_cli_multi_stream_iq_equals_single_streams(tmp_path):
    """The same for --read-mode=2 (int8 I/Q, main.cu:334-380): four streams of different length as one batch per hop, partial
    batches once the short ones have ended."""
    rng = np.random.default_rng(81)
    args = ["--read-mode=2", "--search-width=12", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2"]
    files, singles = [], []
    for i, n_hops in enumerate((3, 1, 0, 4)):
        n = 5184 + n_hops * 2592
        msg = pack77.pack_standard(("CQ", "K1ABC", "W9XYZ", "G4ABC")[i], ("K1ABC", "W9XYZ", "G4ABC", "PA9XYZ")[i], ("FN42", "EN37", "-07", "RR73")[i])
        stream = synth.synth_iq(n, [synth.Ping(msg, 300 + 2000 * (i % 2), 6, 2.0 * (i - 1), 5.0, 0.2 * i)], 20.0, rng)
        path = tmp_path / f"q{i}.s8"
        path.write_bytes(stream.tobytes())
        files.append(str(path))
        rc, out, _ = _run(args, stream.tobytes())
        assert rc == 0
        singles.append([re.sub(r"date=\d{14}", "date=X", l) for l in out.strip().split("\n")[:-1]])
    assert sum(len(s) for s in singles) >= 4
    rc, out, err = _run(args + ["--inputs=" + ",".join(files)], b"")
    assert rc == 0, err
    per_ch = {c: [] for c in range(4)}
    for l in out.strip().split("\n")[:-1]:
        m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
        assert m, l
        per_ch[int(m.group(1))].append("***  " + re.sub(r"date=\d{14}", "date=X", m.group(2)))
    for c in range(4):
        assert per_ch[c] == singles[c], c


def test_cli_interleaved_equals_single_streams():
    """--interleaved=N (N streams as one interleaved stream on stdin, block per hop): every channel's lines are what the single-stream
    program prints for that stream."""
    rng = np.random.default_rng(82)
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2"]
    n_hops, streams, singles = 4, [], []
    for i in range(3):
        msg = pack77.pack_standard(("CQ", "K1ABC", "W9XYZ")[i], ("K1ABC", "W9XYZ", "G4ABC")[i], ("FN42", "-11", "RR73")[i])
        x = synth.synth_audio(5184 + n_hops * 2592, [synth.Ping(msg, 1000 + 2500 * i, 6, 1500.0 + 2 * i, 5.0, 0.3 * i)], 1000.0, rng)
        streams.append(x)
        rc, out, _ = _run(args, x.tobytes())
        assert rc == 0
        singles.append([re.sub(r"date=\d{14}", "date=X", l) for l in out.strip().split("\n")[:-1]])
    assert all(len(s) >= 1 for s in singles)
    blocks = [np.stack([s[:5184] for s in streams]).tobytes()]
    for h in range(n_hops):
        blocks.append(np.stack([s[5184 + h * 2592:5184 + (h + 1) * 2592] for s in streams]).tobytes())
    rc, out, err = _run(args + ["--interleaved=3"], b"".join(blocks))
    assert rc == 0, err
    per_ch = {c: [] for c in range(3)}
    for l in out.strip().split("\n")[:-1]:
        m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
        assert m, l
        per_ch[int(m.group(1))].append("***  " + re.sub(r"date=\d{14}", "date=X", m.group(2)))
    for c in range(3):
        assert per_ch[c] == singles[c], c
    assert "(interleaved on stdin)" in err
    # the same stdin split over two device loops (streams 0..1 | 2), one reader handing every loop its slice of each block
    rc, out, err = _run(args + ["--interleaved=3", "--devices=0,0"], b"".join(blocks))
    assert rc == 0, err
    split = {c: [] for c in range(3)}
    for l in out.strip().split("\n")[:-1]:
        m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
        assert m, l
        split[int(m.group(1))].append("***  " + re.sub(r"date=\d{14}", "date=X", m.group(2)))
    assert split == per_ch and "device 0 decodes streams 0..1" in err and "device 0 decodes streams 2..2" in err


def test_cli_silent_stream_among_live_ones(tmp_path):
    """A muted channel (all-zero samples: rms 0 -> the reference's unguarded 1/0 normalisation) in a batch must not
    disturb the other streams or hang the batch: it prints nothing, the others print what they print alone."""
    rng = np.random.default_rng(83)
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2"]
    n = 5184 + 3 * 2592
    live = synth.synth_audio(n, [synth.Ping(pack77.pack_standard("CQ", "K1ABC", "FN42"), 2000, 6, 1502.0, 6.0, 0.2)], 1000.0, rng)
    files = []
    for i, x in enumerate((live, np.zeros(n, dtype=np.int16), live)):
        path = tmp_path / f"z{i}.s16"
        path.write_bytes(x.tobytes())
        files.append(str(path))
    rc, alone, _ = _run(args, live.tobytes())
    want = [re.sub(r"date=\d{14}", "date=X", l) for l in alone.strip().split("\n")[:-1]]
    assert rc == 0 and len(want) >= 1
    rc, out, err = _run(args + ["--inputs=" + ",".join(files)], b"")
    assert rc == 0, err
    per_ch = {0: [], 1: [], 2: []}
    for l in out.strip().split("\n")[:-1]:
        m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
        assert m, l
        per_ch[int(m.group(1))].append("***  " + re.sub(r"date=\d{14}", "date=X", m.group(2)))
    assert per_ch[0] == want and per_ch[2] == want and per_ch[1] == []


def test_cli_s1_stream_light_config(orc):
    """BASELINE configs[0]/[1] stand-in (demo/0001.wav is absent): the S1 functional stream at the README's
    'optimal scan' options, HIP program vs the oracle-driven CPU decoder, line for line - the default run in the reference's mode,
    --strict-decode in the per-payload mode."""
    stream, pings = synth.stream_s1(0, seconds=8.0, n_pings=4, span=40.0)
    cfg = dict(center=1500.0, width=100.0, step=2.0, depth=3, nbadsync_threshold=1)
    got, strict, err = _both_modes(orc, ["--search-width=100", "--scan-depth=3"], stream, cfg)
    assert len(got) >= 2
    assert "Left Boundary: 1450Hz" in err and "Right Boundary: 1550Hz" in err


def test_cli_s1_stream_deep_config(orc):
    """BASELINE configs[1] stand-in (demo/0001.wav is absent): an S1 stream through stdin -> stdout at the deep options of
    README.md:65-67 (--search-width=500 --search-step=1 --scan-depth=6 --nbadsync-threshold=3: F=501, 24 048 candidates per window)
    - window ring, GPU path, SNR tracker, text layer and per-window filter - line for line against the oracle-driven CPU decoder
    (main.cu:261-422) in BOTH modes (default = the reference's text cache, --strict-decode = per payload), and, independent of the text
    layer, the printed 77-bit payloads against the oracle's accepted payloads."""
    from oracle import oracle_cli
    stream, pings = synth.stream_s1(1, seconds=3.5, n_pings=3, span=240.0)
    cfg = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
    args = ["--search-width=500", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=3"]
    rc, out, err = _run(args, stream.tobytes())
    assert rc == 0, err
    assert "Left Boundary: 1250Hz" in err and "Right Boundary: 1750Hz" in err and "Softbit-kernel CUDA blocks: 501*48=24048" in err
    accepted = set()
    want = oracle_cli.decode_stream(stream, cfg, 1, 2, quirk=True, threads=16, payloads=accepted)
    assert _lines(out) == want and len(want) >= 2
    rc, out, err = _run(args + ["--strict-decode"], stream.tobytes())
    assert rc == 0, err
    want_strict = oracle_cli.decode_stream(stream, cfg, 1, 2, quirk=False, threads=16)
    assert _lines(out) == want_strict and len(want_strict) >= len(want)
    rc, out, err = _run(args + ["--strict-decode", "--print-bits"], stream.tobytes())
    assert rc == 0, err
    printed = set(re.findall(r"bits='([01]{77})'", out))
    sent = {"".join(str(int(b)) for b in p.msg77) for p in pings}
    assert printed and printed <= accepted            # every printed payload is one the oracle accepted
    assert printed & sent == accepted & sent and printed & sent   # and every transmitted payload the oracle decodes is printed


def test_stderr_parameter_block_is_the_references():
    """main.cu:233-252: same labels, same order, values the reference would print for these options (F = 11, depth 3)."""
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")))
    rc, out, err = _run(["--search-width=20", "--search-step=2", "--scan-depth=3"], b"")
    assert rc == 0 and out.strip() == "Done"
    block = err.split("\n\n")[0].split("\n")
    labels = [l for l in ref["stderr_block_labels"] if l != "Hz"]
    assert len(block) == len(labels)
    for line, label in zip(block, labels):
        assert line.startswith(label.rstrip("(").rstrip(": ").rstrip(":")), (line, label)
    assert block[:8] == ["Actual parameters:", "Center Frequency: 1500Hz", "Search Step: 2Hz", "Search Width: 20Hz", "Scan Depth: 3", "Left Boundary: 1490Hz",
                         "Right Boundary: 1510Hz", "Read Mode: (Audio. 16 bits signed.)"]
    assert block[8:] == ["Analytic Method: 2", "Badsync Threshold: 1", "Scan-kernel CUDA blocks: 11", "Scan-kernel CUDA threads: 256", "Softbit-kernel CUDA blocks: 11*24=264",
                         "Softbit-kernel CUDA threads: 160"]


def test_skip_wav_header_flag(orc):
    """Off by default (the reference decodes a RIFF header as 22 samples, README.md:72); on: the first 44 bytes are dropped."""
    rng = np.random.default_rng(91)
    msg = synth.random_message(rng)
    stream = synth.synth_audio(5184 + 2 * 2592, [synth.Ping(msg, 2000, 5, 1502.0, 5.0, 0.3)], 1000.0, rng)
    args = ["--search-width=12", "--search-step=1", "--scan-depth=6"]
    rc, plain, _ = _run(args, stream.tobytes())
    rc, skipped, _ = _run(args + ["--skip-wav-header"], b"RIFF" + bytes(40) + stream.tobytes())
    rc, not_skipped, _ = _run(args, b"RIFF" + bytes(40) + stream.tobytes())
    strip = lambda o: [re.sub(r"date=\d{14}", "date=X", l) for l in o.strip().split("\n")]
    assert strip(skipped) == strip(plain) and len(strip(plain)) >= 2
    assert strip(not_skipped) != strip(plain)          # 22 extra samples shift the window alignment: different positions/lines


def test_paced_fifos_one_stalled_stream_does_not_hold_the_batch(tmp_path):
    """f-4: 64 FIFOs fed in real time (2592 samples per 216 ms, main.cu:284-294, 398-403).  Stream 5 stalls for three hops in the
    middle: the other 63 keep their cadence (their hops are answered within the hop period), the stalled one is decoded when its
    data arrives, and every stream's ping is found."""
    import threading
    import time
    n_ch, n_hops = 64, 7
    rng = np.random.default_rng(2718)
    n = 5184 + n_hops * 2592
    streams, msgs = [], []
    calls = ["K1ABC", "W9XYZ", "G4ABC", "DL1XYZ", "JA1ABC", "VK2DEF", "OH8XYZ", "PA3GHI"]
    for c in range(n_ch):
        msg = pack77.pack_standard(calls[c % 8], calls[(c // 8) % 8 if (c // 8) % 8 != c % 8 else (c + 1) % 8], ("FN42", "EN37", "IO91", "JO62")[c % 4])   # always unpacks to text
        msgs.append(msg)
        streams.append(synth.synth_audio(n, [synth.Ping(msg, 6000 + 40 * c, 5, 1500.0 + (c % 9) - 4, 6.0, 0.1 * c)], 1000.0, rng).tobytes())
    paths = []
    for c in range(n_ch):
        p = str(tmp_path / f"ch{c}.fifo")
        os.mkfifo(p)
        paths.append(p)
    proc = subprocess.Popen([EXE, "--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2", "--print-bits", "--hop-timeout-ms=100",
                             "--inputs=" + ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    hop = 2592 * 2

    def feed(c):
        with open(paths[c], "wb", buffering=0) as f:
            t0 = time.monotonic()
            f.write(streams[c][:5184 * 2])
            for h in range(n_hops):
                due = t0 + 0.216 * (h + 1) + (0.75 if (c == 5 and h >= 2) else 0.0)      # stream 5: its hops 2.. arrive 0.75 s late
                time.sleep(max(0.0, due - time.monotonic()))
                f.write(streams[c][5184 * 2 + h * hop:5184 * 2 + (h + 1) * hop])

    threads = [threading.Thread(target=feed, args=(c,)) for c in range(n_ch)]
    t_start = time.monotonic()
    for t in threads:
        t.start()
    out, err = proc.communicate(timeout=120)
    wall = time.monotonic() - t_start
    for t in threads:
        t.join()
    out, err = out.decode(), err.decode()
    assert proc.returncode == 0, err[-2000:]
    assert out.strip().split("\n")[-1] == "Done"
    decoded = {}
    for l in out.strip().split("\n")[:-1]:
        m = re.match(r"^\*\*\*  ch=(\d+); .*bits='([01]{77})'; $", l)
        assert m, l
        decoded.setdefault(int(m.group(1)), set()).add(m.group(2))
    for c in range(n_ch):
        assert "".join(str(int(b)) for b in msgs[c]) in decoded.get(c, set()), c       # every stream's ping decoded, the stalled one too
    # real-time behaviour: the run lasts about as long as the slowest stream's data (7 hops + the stall), not longer
    assert wall < 0.216 * n_hops + 0.75 + 3.0, wall
    m = re.search(r"(\d+) batches, (\d+) stream hops, (\d+) late, worst latency (\d+) ms", err)
    assert m, err[-1500:]
    batches, hops, late, worst = map(int, m.groups())
    assert hops == n_ch * (n_hops + 1)                      # nothing dropped
    assert batches > n_hops + 1                             # the stalled stream was served in batches of its own
    assert late == 0 and worst <= 210, err[-1500:]          # no stream waited on the stalled one beyond the soft limit


def test_cli_two_device_loops_on_one_gpu_equal_separate_runs(tmp_path):
    """--devices=0,0: two device loops - two live library handles, two ingest and two post-processing threads,
    one printer - over six streams on the one GPU of the box.  Streams 0..2 belong to the first loop, 3..5 to the second, ch= is the
    global stream number, and every stream's lines are byte-identical (date= aside) to two separate single-device runs over the two
    halves.  On an 8-GPU node the same code path runs with eight different ordinals (main.cu:115 binds the reference to one)."""
    rng = np.random.default_rng(90)
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2", "--print-bits"]
    texts = [("CQ", "K1ABC", "FN42"), None, ("K1ABC", "W9XYZ", "-11"), ("CQ", "DL1ABC", "JO62"), ("W9XYZ", "K1ABC", "R-09"), None]
    files = []
    for i, n_hops in enumerate((4, 2, 5, 3, 6, 1)):
        pings = [] if texts[i] is None else [synth.Ping(pack77.pack_standard(*texts[i]), 900 + 1500 * (i % 4), 6, 1496.0 + 2 * i, 5.0, 0.3 * i)]
        stream = synth.synth_audio(5184 + n_hops * 2592, pings, 1000.0, rng)
        path = tmp_path / f"s{i}.s16"
        path.write_bytes(stream.tobytes())
        files.append(str(path))

    def by_channel(out, offset=0):
        per = {}
        lines = out.strip().split("\n")
        assert lines[-1] == "Done" and out.count("Done") == 1
        for l in lines[:-1]:
            m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
            assert m, l
            per.setdefault(offset + int(m.group(1)), []).append(re.sub(r"date=\d{14}", "date=X", m.group(2)))
        return per

    rc, out, err = _run(args + ["--devices=0,0", "--timing", "--inputs=" + ",".join(files)], b"")
    assert rc == 0, err
    both = by_channel(out)
    assert "device 0 decodes streams 0..2" in err and "device 0 decodes streams 3..5" in err and err.count("---- device 0: 3 streams") == 2
    assert err.count("Incomplete read error") == 6
    want = {}
    for half in (0, 1):
        rc, out1, err1 = _run(args + ["--inputs=" + ",".join(files[3 * half:3 * half + 3])], b"")
        assert rc == 0, err1
        want.update(by_channel(out1, 3 * half))
    assert both == want and set(both) == {0, 2, 3, 4}                     # the four streams that carry a ping, nothing on the noise streams
    for c, t in ((0, "CQ K1ABC FN42"), (2, "K1ABC W9XYZ -11"), (3, "CQ DL1ABC JO62"), (4, "W9XYZ K1ABC R-09")):
        assert all(f"msg='{t}'" in l for l in both[c])
    m = re.search(r"msk144hipdecoder: (\d+) batches, (\d+) stream hops", err)
    assert int(m.group(2)) == sum(h + 1 for h in (4, 2, 5, 3, 6, 1))
    # an ordinal the box does not have is refused before anything runs
    rc, out, err = _run(args + ["--devices=0,99", "--inputs=" + ",".join(files)], b"")
    assert rc == 2 and "device 99" in err


def _demo_wav():
    """The reference's demo recording (README.md:72 `cat ../demo/0001.wav | ./msk144cudecoder`) is a missing blob in the reference
    tree this project was built from.  If it ever appears - MSK144_DEMO_WAV, or demo/0001.wav / tests/golden/0001.wav in this
    repository - the test below picks it up by itself."""
    for p in (os.environ.get("MSK144_DEMO_WAV"), os.path.join(ROOT, "demo", "0001.wav"), os.path.join(ROOT, "tests", "golden", "0001.wav")):
        if p and os.path.exists(p):
            return p
    return None


@pytest.mark.skipif(_demo_wav() is None, reason="demo/0001.wav (the reference's recording, BASELINE configs[0]/[1]) is not available: missing blob in the reference tree")
@pytest.mark.parametrize("name,args,cfg", [
    ("configs0_light", ["--search-width=100", "--scan-depth=3"], dict(center=1500.0, width=100.0, step=2.0, depth=3, nbadsync_threshold=1)),
    ("configs1_deep", ["--search-width=500", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=3"],
     dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)),
])
def test_cli_demo_wav_if_present(orc, name, args, cfg):
    """BASELINE configs[0]/[1] on the reference's own recording, fed exactly as its README does (the RIFF header goes in as samples,
    main.cu:273): msk144hipdecoder's stdout, line for line, against the oracle-driven CPU decoder in the reference's mode, and the
    decode is written to gpurun_out/demo_0001_<config>.txt (+ printed payloads) so that the first run with the file leaves the golden
    behind; a committed tests/golden/demo_0001_<config>.txt is compared as well."""
    from oracle import oracle_cli
    raw = open(_demo_wav(), "rb").read()
    rc, out, err = _run(args, raw)
    assert rc == 0, err
    got = _lines(out)
    stream = np.frombuffer(raw[:len(raw) // 2 * 2], dtype=np.int16)
    want = oracle_cli.decode_stream(stream, cfg, 1, 2, quirk=True, threads=16)
    rc, out_b, _ = _run(args + ["--print-bits"], raw)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"demo_0001_{name}.txt"), "w") as f:
        f.write("\n".join(_lines(out_b)) + "\n")
    assert got == want and len(got) >= 1
    golden = os.path.join(ROOT, "tests", "golden", f"demo_0001_{name}.txt")
    if os.path.exists(golden):
        assert [re.sub(r"' bits='[01]{77}", "", l) for l in open(golden).read().strip().split("\n")] == got


def test_cli_result_list_overflow_is_survived(tmp_path):
    """--max-results smaller than a busy hop's decode list: the library cuts the list (MSK144_EOVERFLOW), the program
    reports the hop, processes the records that fitted and keeps decoding - exit code 0, the later hops print normally.  Which
    decodes of the overflowed hop survive is unspecified (the compact list is filled in (stream, item) order: the first streams win);
    INTEGRATION.md says so."""
    rng = np.random.default_rng(92)
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2"]
    files = []
    texts = [("CQ", "K1ABC", "FN42"), ("K1ABC", "W9XYZ", "-11"), ("CQ", "DL1ABC", "JO62")]
    for i in range(3):
        # a strong ping in the first window of every stream (dozens of accepted candidates each), a second one three hops later
        pings = [synth.Ping(pack77.pack_standard(*texts[i]), 300, 5, 1498.0 + 2 * i, 8.0, 0.3 * i), synth.Ping(pack77.pack_standard(*texts[i]), 5184 + 2 * 2592 + 200, 5, 1498.0 + 2 * i, 8.0, 0.2)]
        path = tmp_path / f"o{i}.s16"
        path.write_bytes(synth.synth_audio(5184 + 5 * 2592, pings, 1000.0, rng).tobytes())
        files.append(str(path))
    rc, full, err = _run(args + ["--inputs=" + ",".join(files)], b"")
    assert rc == 0 and "list was cut" not in err
    rc, out, err = _run(args + ["--max-results=4", "--inputs=" + ",".join(files)], b"")
    assert rc == 0, err
    assert "held more decodes than the result list" in err and "the list was cut, decoding goes on" in err and "hops overflowed the result list" in err
    assert out.strip().endswith("Done")
    key = lambda ls: [re.search(r"(ch=\d+); .*(msg='.*'); $", l).groups() for l in ls]      # (stream, text): the filter's winner among the SURVIVING
    cut, whole = key(_lines(out)), key(_lines(full))                                        # records of a cut hop may be another candidate (f0, num_avg differ)
    assert 1 <= len(cut) < len(whole) and set(cut) <= set(whole)          # nothing invented, fewer lines
    assert ("ch=0", "msg='CQ K1ABC FN42'") in cut                         # stream 0 comes first in the compact list: it still prints


def test_cli_two_different_devices_equal_two_single_device_runs(hip, tmp_path):
    """--devices=0,1 with DIFFERENT ordinals (never run on this pool's one-GPU boxes): the two halves of six streams on two
    GPUs print, stream by stream, what two single-device runs over the halves print.  Runs by itself on the first multi-GPU box."""
    if hip.device_count() < 2:
        pytest.skip("needs two HIP devices (the one-GPU boxes of this pool run --devices=0,0 instead)")
    rng = np.random.default_rng(93)
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2"]
    files = []
    for i in range(6):
        pings = [synth.Ping(pack77.pack_standard("CQ", ("K1ABC", "W9XYZ", "G4ABC")[i % 3], "FN42"), 700 + 900 * i, 5, 1496.0 + i, 6.0, 0.1 * i)]
        path = tmp_path / f"d{i}.s16"
        path.write_bytes(synth.synth_audio(5184 + 3 * 2592, pings, 1000.0, rng).tobytes())
        files.append(str(path))

    def by_channel(out, offset=0):
        per = {}
        for l in _lines(out):
            m = re.match(r"^\*\*\*  ch=(\d+); (.*)$", l)
            per.setdefault(offset + int(m.group(1)), []).append(m.group(2))
        return per

    rc, out, err = _run(args + ["--devices=0,1", "--inputs=" + ",".join(files)], b"")
    assert rc == 0, err
    want = {}
    for half in (0, 1):
        rc, o1, e1 = _run(args + [f"--device={half}", "--inputs=" + ",".join(files[3 * half:3 * half + 3])], b"")
        assert rc == 0, e1
        want.update(by_channel(o1, 3 * half))
    assert by_channel(out) == want and len(want) >= 4
    assert "device 0 decodes streams 0..2" in err and "device 1 decodes streams 3..5" in err


@pytest.mark.parametrize("name,args,cfg,read_mode,method", [
    ("fft_front_end", ["--analytic-method=1", "--search-width=24", "--search-step=1", "--scan-depth=5", "--nbadsync-threshold=2"],
     dict(center=1500.0, width=24.0, step=1.0, depth=5, nbadsync_threshold=2), 1, 1),
    ("depth8_fractional_step_offset_centre", ["--center-frequency=1512.5", "--search-width=9", "--search-step=0.75", "--scan-depth=8", "--nbadsync-threshold=4"],
     dict(center=1512.5, width=9.0, step=0.75, depth=8, nbadsync_threshold=4), 1, 2),
    ("iq_depth_clamped", ["--read-mode=2", "--center-frequency=-3", "--search-width=14", "--search-step=1", "--scan-depth=11", "--nbadsync-threshold=3"],
     dict(center=-3.0, width=14.0, step=1.0, depth=8, nbadsync_threshold=3), 2, 2),          # scan depth is clamped to 8 (msk_context.cuh:29-33)
])
def test_cli_option_surface_against_oracle(orc, name, args, cfg, read_mode, method):
    """Corners of the reference's option surface (main.cu:136-190) through the program, line for line against the oracle-driven decoder in
    both text modes: the FFT front end (--analytic-method=1, analytic_fft.cu), all eight averaging patterns with a fractional search
    step around an offset centre frequency (msk_context.cuh:95-113), and IQ input with a scan depth beyond the clamp."""
    from oracle import oracle_cli
    rng = np.random.default_rng({"fft_front_end": 521, "depth8_fractional_step_offset_centre": 806, "iq_depth_clamped": 430}[name])
    n = 5184 + 3 * 2592
    f0 = cfg["center"] + 1.5
    msgs = [pack77.pack_standard("CQ", "K1ABC", "FN42"), pack77.pack_standard("K1ABC", "W9XYZ", "R-05")]
    pings = [synth.Ping(msgs[0], 700, 6, f0, 5.0, 0.3), synth.Ping(msgs[1], 5184 + 900, 4, f0 - 2.0, 5.0, 1.3)]
    stream = synth.synth_audio(n, pings, 1000.0, rng) if read_mode == 1 else synth.synth_iq(n, pings, 20.0, rng)
    for extra, quirk in (([], True), (["--strict-decode"], False)):
        rc, out, err = _run(args + extra, stream.tobytes())
        assert rc == 0, err
        want = oracle_cli.decode_stream(stream, cfg, read_mode, method, quirk=quirk, threads=8)
        assert _lines(out) == want, (name, quirk)
        assert len(want) >= 2
    assert f"Scan Depth: {cfg['depth']}" in err
