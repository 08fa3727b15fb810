"""msk144hipdecoder's multi-stream loop on the CPU: the real host sources (main.cpp, window_decoder.cpp, text layer, filter) linked
against tests/stub_hip (a stand-in for libmsk144hip.so that decodes nothing; every hop it reports which window it was handed).
Checks the loop's own logic - non-blocking ingest, FIFOs whose writer connects late, batch policy with a stalled stream, the
two-slot pipeline with its post-processing thread, window overlap across slots, end of stream - which the GPU tests then repeat
with the real library."""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "msk144cudecoder_amd", "host")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("stubhip"))
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", os.path.join(d, "libmsk144hip.so"),
                    os.path.join(ROOT, "tests", "stub_hip", "msk144hip_stub.cpp")], check=True)
    srcs = [os.path.join(HOST, f) for f in ("snr_tracker.cpp", "result_filter.cpp", "unpack77.cpp", "postprocess.cpp", "window_decoder.cpp", "main.cpp")]
    out = os.path.join(d, "msk144hipdecoder_stub")
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", out] + srcs + ["-L" + d, "-lmsk144hip", "-Wl,-rpath," + d], check=True)
    return out


def marked_stream(n_hops, tag):
    """Stream of n_hops + 1 windows; half-window k (2592 samples) starts with 0x7777, tag + k: the stub reports, per window, the
    second sample of both halves."""
    x = np.zeros(5184 + n_hops * 2592, dtype=np.int16)
    for k in range(n_hops + 2):
        x[k * 2592] = 0x7777
        x[k * 2592 + 1] = tag + k
    return x


def windows_seen(stdout, n_streams):
    """{channel: [(first half id, second half id), ...]} in output order, from the telemetry text of the stub's records."""
    seen = {c: [] for c in range(n_streams)}
    for line in stdout.strip().split("\n"):
        if line == "Done":
            continue
        m = re.match(r"^\*\*\*  (?:ch=(\d+); )?.*msg='([0-9A-F]+)'; $", line)
        assert m, line
        v = int(m.group(2), 16)
        ch = int(m.group(1) or 0)     # the stream the host attributes the record to; v >> 32 is only its position in the compact batch
        seen[ch].append(((v >> 16) & 0xFFFF, v & 0xFFFF))
    return seen


def test_files_of_different_length(exe, tmp_path):
    hops = [4, 1, 7, 0, 3]
    paths = []
    for c, h in enumerate(hops):
        p = tmp_path / f"s{c}.s16"
        p.write_bytes(marked_stream(h, 100 * c).tobytes())
        paths.append(str(p))
    r = subprocess.run([exe, "--inputs=" + ",".join(paths)], capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DECODE_MS="3"))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    seen = windows_seen(r.stdout.decode(), len(hops))
    for c, h in enumerate(hops):
        assert seen[c] == [(100 * c + k, 100 * c + k + 1) for k in range(h + 1)], c      # every window once, in order, halves overlapping by one
    err = r.stderr.decode()
    assert err.count("Incomplete read error") == len(hops)
    m = re.search(r"(\d+) batches, (\d+) stream hops, (\d+) late", err)
    assert m and int(m.group(1)) == max(hops) + 1 and int(m.group(2)) == sum(h + 1 for h in hops)


def test_inputs_file_and_stdin_single_stream(exe, tmp_path):
    x = marked_stream(3, 7)
    r = subprocess.run([exe], input=x.tobytes(), capture_output=True, timeout=60)
    assert r.returncode == 0 and windows_seen(r.stdout.decode(), 1)[0] == [(7 + k, 8 + k) for k in range(4)]
    p = tmp_path / "a.s16"
    p.write_bytes(x.tobytes())
    lst = tmp_path / "inputs.txt"
    lst.write_text(f"{p}\n{p}\n")
    r = subprocess.run([exe, f"--inputs-file={lst}", "--timing"], capture_output=True, timeout=60)
    assert r.returncode == 0
    seen = windows_seen(r.stdout.decode(), 2)
    assert seen[0] == seen[1] == [(7 + k, 8 + k) for k in range(4)]
    assert "timing: wait for GPU + D2H (post thread)" in r.stderr.decode()


def test_interleaved_stdin(exe):
    """--interleaved=N: one block of N x 5184 samples, then N x 2592 per hop, stream after stream, on stdin."""
    n, hops = 5, 6
    streams = [marked_stream(hops, 300 * c) for c in range(n)]
    blocks = [np.stack([s[:5184] for s in streams]).tobytes()]
    for h in range(hops):
        blocks.append(np.stack([s[5184 + h * 2592:5184 + (h + 1) * 2592] for s in streams]).tobytes())
    r = subprocess.run([exe, f"--interleaved={n}"], input=b"".join(blocks) + b"\0" * 64, capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DECODE_MS="5"))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    seen = windows_seen(r.stdout.decode(), n)
    for c in range(n):
        assert seen[c] == [(300 * c + k, 300 * c + k + 1) for k in range(hops + 1)], c
    err = r.stderr.decode()
    assert "Incomplete read error. rc=32" in err and f"{hops + 1} batches, {n * (hops + 1)} stream hops" in err
    r = subprocess.run([exe, "--interleaved=2", "--inputs=a,b"], input=b"", capture_output=True, timeout=60)
    assert r.returncode == 2 and "excludes --inputs" in r.stderr.decode()


def test_fifos_late_writer_and_stalled_stream(exe, tmp_path):
    """Stream 1's writer connects 0.4 s after the decoder has opened the FIFO (read() returns 0 until then: NOT end of stream);
    stream 2 stalls for 0.5 s in the middle: the others keep going in batches of their own and nothing is lost or reordered."""
    n, hops = 4, 8
    paths = [str(tmp_path / f"f{c}.fifo") for c in range(n)]
    for p in paths:
        os.mkfifo(p)
    data = [marked_stream(hops, 1000 * c).tobytes() for c in range(n)]
    proc = subprocess.Popen([exe, "--hop-timeout-ms=60", "--inputs=" + ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, MSK144_STUB_DECODE_MS="10"))

    def feed(c):
        if c == 1:
            time.sleep(0.4)
        with open(paths[c], "wb", buffering=0) as f:
            f.write(data[c][:5184 * 2])
            for h in range(hops):
                time.sleep(0.04 + (0.5 if (c == 2 and h == 3) else 0.0))
                f.write(data[c][5184 * 2 + h * 5184:5184 * 2 + (h + 1) * 5184])

    ths = [threading.Thread(target=feed, args=(c,)) for c in range(n)]
    for t in ths:
        t.start()
    out, err = proc.communicate(timeout=60)
    for t in ths:
        t.join()
    assert proc.returncode == 0, err.decode()[-1500:]
    seen = windows_seen(out.decode(), n)
    for c in range(n):
        assert seen[c] == [(1000 * c + k, 1000 * c + k + 1) for k in range(hops + 1)], c
    m = re.search(r"(\d+) batches, (\d+) stream hops", err.decode())
    assert int(m.group(2)) == n * (hops + 1) and int(m.group(1)) > hops + 1       # the late and the stalled stream were served in extra batches


def test_many_streams_unpaced_through_the_scale_harness(exe):
    """tools/host_scale.py (the BASELINE-scale harness of the GPU suite) against the stub: 256 FIFOs written as fast as the pipes
    take them, a 20 ms 'GPU': both slots in flight, back-pressure into the pipes, every hop accounted for."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import host_scale
    old = host_scale.EXE
    host_scale.EXE = exe
    os.environ["MSK144_STUB_DECODE_MS"] = "20"
    try:
        res = host_scale.run(256, 12, pace_ms=0.0, timeout_s=60.0)
    finally:
        host_scale.EXE = old
        del os.environ["MSK144_STUB_DECODE_MS"]
    assert res["returncode"] == 0 and res["feeder_errors"] == 0, res
    assert res["stream_hops"] == 256 * 13
    assert res["host_ms_per_batch"]["wait for GPU + D2H (post thread)"]["mean_ms"] >= 15.0
