#!/usr/bin/env python3
"""Randomised stress of msk144hipdecoder's multi-stream / multi-device loop against tests/stub_hip (CPU, no GPU): random stream
counts (1-40), device lists (1-4 entries over four stub devices, ordinals may repeat), hop counts, input kinds (files, FIFOs written
in random chunk sizes with random pauses, interleaved stdin), stub decode times and hop timeouts; every run must exit 0 and print every
window of every stream once, in order.  Optionally under a sanitizer.

    ITERS=40 python tools/host_loop_stress.py <seed> [thread | address,undefined]
"""
import os, sys, random, subprocess, threading, time, tempfile, shutil
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_host_loop import marked_stream, windows_seen, PROGRAM_SOURCES, HOST, ROOT

d = tempfile.mkdtemp(prefix="stress_")
san = sys.argv[2] if len(sys.argv) > 2 else ""
flags = ["-fsanitize=" + san, "-g"] if san else []
subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread"] + flags + ["-o", os.path.join(d, "libmsk144hip.so"), os.path.join(ROOT, "tests", "stub_hip", "msk144hip_stub.cpp")], check=True)
exe = os.path.join(d, "dec")
subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-pthread"] + flags + ["-o", exe] + [os.path.join(HOST, f) for f in PROGRAM_SOURCES] + ["-L" + d, "-lmsk144hip", "-Wl,-rpath," + d], check=True)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(int(os.environ.get("ITERS", "40"))):
    n = rng.randint(1, 40)
    ndev = rng.randint(1, 4)
    hops = [rng.randint(0, 7) for _ in range(n)]
    mode = rng.choice(["fifo", "file", "interleaved"])
    env = dict(os.environ, MSK144_STUB_DECODE_MS=str(rng.choice([0, 1, 3, 8])), MSK144_STUB_DEVICES="4", TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=0")
    devs = ",".join(str(rng.randint(0, 3)) for _ in range(ndev))
    args = [exe, "--devices=" + devs, "--hop-timeout-ms=" + str(rng.choice([1, 5, 20, 60]))]
    tmp = tempfile.mkdtemp(prefix="it_", dir=d)
    data = [marked_stream(h, 50 * c) for c, h in enumerate(hops)]
    if mode == "interleaved":
        h = min(hops)
        hops = [h] * n
        data = [marked_stream(h, 50 * c) for c in range(n)]
        blocks = [np.stack([s[:5184] for s in data]).tobytes()] + [np.stack([s[5184 + k * 2592:5184 + (k + 1) * 2592] for s in data]).tobytes() for k in range(h)]
        r = subprocess.run(args + [f"--interleaved={n}"], input=b"".join(blocks), capture_output=True, timeout=120, env=env)
        out, err, rc = r.stdout.decode(), r.stderr.decode(), r.returncode
    else:
        paths = [os.path.join(tmp, f"s{c}") for c in range(n)]
        if mode == "file":
            for p, x in zip(paths, data):
                open(p, "wb").write(x.tobytes())
            r = subprocess.run(args + ["--inputs=" + ",".join(paths)], capture_output=True, timeout=120, env=env)
            out, err, rc = r.stdout.decode(), r.stderr.decode(), r.returncode
        else:
            for p in paths:
                os.mkfifo(p)
            proc = subprocess.Popen(args + ["--inputs=" + ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            def feed(c):
                time.sleep(rng.random() * 0.05)
                with open(paths[c], "wb", buffering=0) as f:
                    b = data[c].tobytes()
                    off = 0
                    while off < len(b):
                        k = rng.choice([700, 5184, 10368, 20000])
                        f.write(b[off:off + k]); off += k
                        if rng.random() < 0.2: time.sleep(rng.random() * 0.03)
            ths = [threading.Thread(target=feed, args=(c,)) for c in range(n)]
            [t.start() for t in ths]
            try:
                o, e = proc.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                proc.kill(); o, e = proc.communicate(); print("TIMEOUT", it, mode, n, devs)
            [t.join() for t in ths]
            out, err, rc = o.decode(), e.decode(), proc.returncode
    ok = rc == 0 and "Sanitizer" not in err
    if ok:
        seen = windows_seen(out, n)
        for c in range(n):
            if seen[c] != [((50 * c + k) & 0xFFFF, (50 * c + k + 1) & 0xFFFF) for k in range(hops[c] + 1)]:
                ok = False
    if not ok:
        bad += 1
        print("FAIL", it, mode, n, devs, hops, rc, err[-800:])
    shutil.rmtree(tmp)
print("done, failures:", bad)
shutil.rmtree(d)
