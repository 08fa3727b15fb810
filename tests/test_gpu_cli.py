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


def test_cli_audio_stream(orc):
    rng = np.random.default_rng(77)
    n = 5184 + 4 * 2592
    msgs = [synth.random_message(rng) for _ in range(2)]
    pings = [synth.Ping(msgs[0], 1500, 6, 1503.0, 4.0, 0.4), synth.Ping(msgs[1], 9000, 5, 1495.0, 5.0, 1.4)]
    stream = synth.synth_audio(n, pings, 1000.0, rng)
    cfg = dict(center=1500.0, width=20.0, step=1.0, depth=6, nbadsync_threshold=2)
    args = ["--search-width=20", "--search-step=1", "--scan-depth=6", "--nbadsync-threshold=2"]
    rc, out, err = _run(args + ["--strict-decode"], stream.tobytes())
    assert rc == 0, err
    lines = out.strip().split("\n")
    assert lines[-1] == "Done"
    got = [re.sub(r"date=\d{14}", "date=X", l) for l in lines[:-1]]
    assert "Actual parameters:" in err and "Incomplete read error. rc=0" in err
    want = _expected_lines(orc, stream, cfg, 1, quirk=False)
    assert len(want) >= 2
    assert got == want
    # every line has the reference's field layout (main.cu:409-417)
    pat = re.compile(r"^\*\*\*  snr=[ -]?\d+; f0=\s*[\d.]+; num_avg=\d; nbadsync=\d+; pattern_idx=\d; date=X; msg='.*'; $")
    assert all(pat.match(l) for l in got)
    # default mode reproduces the reference's first-candidate cache behaviour
    rc, out2, _ = _run(args, stream.tobytes())
    got2 = [re.sub(r"date=\d{14}", "date=X", l) for l in out2.strip().split("\n")[:-1]]
    assert got2 == _expected_lines(orc, stream, cfg, 1, quirk=True)


def test_cli_iq_and_option_quirks(orc):
    rng = np.random.default_rng(78)
    msg = synth.random_message(rng)
    stream = synth.synth_iq(5184 + 2592, [synth.Ping(msg, 800, 6, 2.0, 4.0, 0.1)], 20.0, rng)
    cfg = dict(center=0.0, width=12.0, step=2.0, depth=4, nbadsync_threshold=1)
    rc, out, err = _run(["--read-mode=2", "--search-width=12", "--strict-decode"], stream.tobytes())
    assert rc == 0 and out.strip().endswith("Done")
    got = [re.sub(r"date=\d{14}", "date=X", l) for l in out.strip().split("\n")[:-1]]
    assert got == _expected_lines(orc, stream, cfg, 2, quirk=False) and len(got) >= 1
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
    args = ["--search-width=16", "--search-step=2", "--scan-depth=6", "--nbadsync-threshold=2", "--strict-decode"]
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
    assert err.count("Incomplete read error") == 3 and "Input streams: 3" in err


def test_cli_s1_stream_light_config(orc):
    """BASELINE configs[0]/[1] stand-in (demo/0001.wav is absent): the S1 functional stream at the README's
    'optimal scan' options, HIP program vs the oracle-driven CPU decoder, line for line."""
    stream, pings = synth.stream_s1(0, seconds=8.0, n_pings=4, span=40.0)
    cfg = dict(center=1500.0, width=100.0, step=2.0, depth=3, nbadsync_threshold=1)
    args = ["--search-width=100", "--scan-depth=3"]
    rc, out, err = _run(args + ["--strict-decode"], stream.tobytes())
    assert rc == 0, err
    got = [re.sub(r"date=\d{14}", "date=X", l) for l in out.strip().split("\n")[:-1]]
    want = _expected_lines(orc, stream, cfg, 1, quirk=False)
    assert got == want
    assert "Left Boundary: 1450Hz" in err and "Right Boundary: 1550Hz" in err
