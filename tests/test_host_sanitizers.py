"""The pipelined multi-stream loop of msk144hipdecoder (ingest thread + post-processing thread over two staging slots) under
ThreadSanitizer and AddressSanitizer/UBSan - CPU build against tests/stub_hip (GPU sanitizers are not available on the pool)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_host_loop import HOST, PROGRAM_SOURCES, ROOT, marked_stream, windows_seen


def _build(tmp, flag):
    d = str(tmp)
    common = ["g++", "-O1", "-g", f"-fsanitize={flag}", "-std=c++17", "-pthread"]
    r = subprocess.run(common + ["-fPIC", "-shared", "-o", os.path.join(d, "libmsk144hip.so"), os.path.join(ROOT, "tests", "stub_hip", "msk144hip_stub.cpp")],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip(f"-fsanitize={flag} not usable with this toolchain: {r.stderr[-200:]}")
    srcs = [os.path.join(HOST, f) for f in PROGRAM_SOURCES]
    exe = os.path.join(d, "msk144hipdecoder_san")
    subprocess.run(common + ["-ffp-contract=off", "-o", exe] + srcs + ["-L" + d, "-lmsk144hip", "-Wl,-rpath," + d], check=True)
    return exe


@pytest.mark.parametrize("flag", ["thread", "address,undefined"])
def test_multi_stream_loop_under_sanitizers(tmp_path, flag):
    exe = _build(tmp_path, flag)
    env = dict(os.environ, MSK144_STUB_DECODE_MS="4", TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1")
    hops = [5, 1, 8, 0, 3, 8]
    paths = []
    for c, h in enumerate(hops):
        p = tmp_path / f"s{c}.s16"
        p.write_bytes(marked_stream(h, 50 * c).tobytes())
        paths.append(str(p))
    r = subprocess.run([exe, "--timing", "--print-bits", "--inputs=" + ",".join(paths)], capture_output=True, timeout=300, env=env)
    err = r.stderr.decode()
    assert r.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert r.stdout.decode().strip().endswith("Done")
    # interleaved stdin through the same loop
    n, h = 4, 5
    streams = [marked_stream(h, 10 * c) for c in range(n)]
    blocks = [np.stack([s[:5184] for s in streams]).tobytes()] + [np.stack([s[5184 + k * 2592:5184 + (k + 1) * 2592] for s in streams]).tobytes() for k in range(h)]
    r = subprocess.run([exe, f"--interleaved={n}"], input=b"".join(blocks), capture_output=True, timeout=300, env=env)
    err = r.stderr.decode()
    assert r.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert windows_seen(r.stdout.decode(), n)[3] == [(30 + k, 31 + k) for k in range(h + 1)]
    # three device loops side by side (six threads, one printer), files of different length; then interleaved stdin split over two
    env3 = dict(env, MSK144_STUB_DEVICES="3")
    r = subprocess.run([exe, "--timing", "--devices=0,1,2", "--inputs=" + ",".join(paths)], capture_output=True, timeout=300, env=env3)
    err = r.stderr.decode()
    assert r.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    devs = {}
    seen = windows_seen(r.stdout.decode(), len(hops), devs)
    for c, hh in enumerate(hops):
        assert seen[c] == [(50 * c + k, 50 * c + k + 1) for k in range(hh + 1)] and devs[c] == {c // 2}, c
    r = subprocess.run([exe, f"--interleaved={n}", "--devices=0,1"], input=b"".join(blocks), capture_output=True, timeout=300, env=env3)
    err = r.stderr.decode()
    assert r.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert windows_seen(r.stdout.decode(), n)[3] == [(30 + k, 31 + k) for k in range(h + 1)]
