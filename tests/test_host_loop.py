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
PROGRAM_SOURCES = ("snr_tracker.cpp", "result_filter.cpp", "unpack77.cpp", "postprocess.cpp", "window_decoder.cpp", "stream_loop.cpp", "main.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("stubhip"))
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", os.path.join(d, "libmsk144hip.so"),
                    os.path.join(ROOT, "tests", "stub_hip", "msk144hip_stub.cpp")], check=True)
    srcs = [os.path.join(HOST, f) for f in PROGRAM_SOURCES]
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


def windows_seen(stdout, n_streams, devices=None):
    """{channel: [(first half id, second half id), ...]} in output order, from the telemetry text of the stub's records.
    `devices` (a dict) receives {channel: set of device ordinals whose handle decoded it}."""
    seen = {c: [] for c in range(n_streams)}
    for line in stdout.strip().split("\n"):
        if line == "Done":
            continue
        m = re.match(r"^\*\*\*  (?:ch=(\d+); )?.*msg='([0-9A-F]+)'; $", line)
        assert m, line
        v = int(m.group(2), 16)
        ch = int(m.group(1) or 0)     # the stream the host attributes the record to; v >> 32 is only its position in the compact batch
        seen[ch].append(((v >> 16) & 0xFFFF, v & 0xFFFF))
        if devices is not None:
            devices.setdefault(ch, set()).add(v >> 56)
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


def test_program_asks_for_the_batch_kernels_and_every_slot_switches_the_hand_over_off(exe, tmp_path):
    """Every handle the program creates - its single stdin stream and the per-device handles of --inputs alike - is told that no LLR row
    has to outlive its decode (msk144_set_llr_retention(h, 0): early gate, copies handed over); --every-slot then switches the hand-over
    off again (msk144_set_copy_handover(h, 0))."""
    env = dict(os.environ, MSK144_STUB_DECODE_MS="1", MSK144_STUB_LOG_MODES="1", MSK144_STUB_DEVICES="2")
    data = marked_stream(2, 100).tobytes()
    r = subprocess.run([exe], input=data, capture_output=True, timeout=60, env=env)
    assert r.returncode == 0 and r.stderr.decode().count("stub: msk144_set_llr_retention(0)") == 1 and b"msk144_set_copy_handover" not in r.stderr
    r = subprocess.run([exe, "--every-slot"], input=data, capture_output=True, timeout=60, env=env)
    assert r.returncode == 0 and b"stub: msk144_set_llr_retention(0)" in r.stderr and r.stderr.decode().count("stub: msk144_set_copy_handover(0)") == 1
    p = tmp_path / "s.s16"
    p.write_bytes(data)
    r = subprocess.run([exe, f"--inputs={p},{p},{p}", "--devices=0,1", "--every-slot"], capture_output=True, timeout=60, env=env)
    assert r.returncode == 0 and r.stderr.decode().count("stub: msk144_set_llr_retention(0)") == 2 and r.stderr.decode().count("stub: msk144_set_copy_handover(0)") == 2


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
    os.environ["MSK144_STUB_DEVICES"] = "2"
    try:
        res = host_scale.run(256, 12, pace_ms=0.0, timeout_s=60.0)
        two = host_scale.run(128, 6, pace_ms=20.0, timeout_s=60.0, devices="0,1", phase_spread_ms=15.0)
        # every stream on its own hop clock, 40 hops fed out of 5 hops of synthesised signal (the long-run mode of the harness)
        own = host_scale.run(64, 40, pace_ms=20.0, timeout_s=60.0, phase_spread_ms=20.0, phase_per_stream=True, phase_seed=3, loop_hops=5)
    finally:
        del os.environ["MSK144_STUB_DEVICES"]
        host_scale.EXE = old
        del os.environ["MSK144_STUB_DECODE_MS"]
    assert res["returncode"] == 0 and res["feeder_errors"] == 0, res
    assert res["stream_hops"] == 256 * 13
    assert res["host_ms_per_batch"]["wait for GPU + D2H (post thread)"]["mean_ms"] >= 15.0
    # --devices: one timing block per device loop
    assert two["returncode"] == 0 and two["stream_hops"] == 128 * 7 and len(two["per_device"]) == 2, two
    assert [d["streams"] for d in two["per_device"]] == [64, 64] and two["per_device"][1]["first_stream"] == 64
    assert all("ingest (read syscalls, all streams)" in d["host_ms_per_batch"] for d in two["per_device"])
    assert own["returncode"] == 0 and own["feeder_errors"] == 0 and own["stream_hops"] == 64 * 41, own
    assert own["phases"].startswith("one per stream") and "looped_signal" in own and own["margin_ms_to_210"] == 210 - own["worst_latency_ms"]


def _feed_fifos(paths, data, chunk=5184):
    """Writer threads: every FIFO gets its stream in hop-sized writes, as fast as the pipe takes them."""
    def feed(lo, hi):
        fds = [os.open(paths[c], os.O_WRONLY) for c in range(lo, hi)]
        off = 0
        n = max(len(data[c]) for c in range(lo, hi))
        while off < n:
            for k, c in enumerate(range(lo, hi)):
                if off < len(data[c]):
                    os.write(fds[k], data[c][off:off + chunk])
            off += chunk
        for fd in fds:
            os.close(fd)
    per = -(-len(paths) // 8)
    ths = [threading.Thread(target=feed, args=(k * per, min(len(paths), (k + 1) * per))) for k in range(8) if k * per < len(paths)]
    for t in ths:
        t.start()
    return ths


def _run_fifos(exe, tmp_path, tag, data, args, env):
    paths = [str(tmp_path / f"{tag}_{c}.fifo") for c in range(len(data))]
    for p in paths:
        os.mkfifo(p)
    lst = tmp_path / f"{tag}.txt"
    lst.write_text("\n".join(paths) + "\n")
    proc = subprocess.Popen([exe, f"--inputs-file={lst}"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    ths = _feed_fifos(paths, data)
    out, err = proc.communicate(timeout=120)
    for t in ths:
        t.join()
    return proc.returncode, out.decode(), err.decode()


def _lines_by_channel(stdout, offset=0):
    per = {}
    for line in stdout.strip().split("\n")[:-1]:
        ch = int(re.match(r"^\*\*\*  ch=(\d+); ", line).group(1))
        per.setdefault(offset + ch, []).append(re.sub(r"date=\d+; ", "", re.sub(r"ch=\d+; ", "", line, count=1)))   # date= is wall-clock time
    return per


def test_four_devices_equal_four_single_device_runs(exe, tmp_path):
    """--devices=0,1,2,3 over 4 x 64 FIFOs (the stub reports four devices): every stream is served by the loop of its own device,
    ch= is the global stream number, and each stream's lines are exactly those of a single-device run over that device's 64 streams."""
    n, hops = 256, 6
    data = [marked_stream(hops + (c % 3), 7 * c).tobytes() for c in range(n)]
    env = dict(os.environ, MSK144_STUB_DECODE_MS="6", MSK144_STUB_DEVICES="4")
    rc, out, err = _run_fifos(exe, tmp_path, "all", data, ["--devices=0,1,2,3", "--timing"], env)
    assert rc == 0, err[-2000:]
    devs = {}
    seen = windows_seen(out, n, devs)
    for c in range(n):
        assert seen[c] == [(7 * c + k, 7 * c + k + 1) for k in range(hops + (c % 3) + 1)], c
        assert devs[c] == {c // 64}, (c, devs[c])                       # contiguous shares: streams 64d .. 64d+63 on device d
    assert out.strip().endswith("Done") and out.count("Done") == 1
    for d in range(4):
        assert f"device {d} decodes streams {64 * d}..{64 * d + 63}" in err
        assert f"---- device {d}: 64 streams ({64 * d}..{64 * d + 63})" in err
    m = re.search(r"msk144hipdecoder: (\d+) batches, (\d+) stream hops, (\d+) late", err)
    assert m and int(m.group(2)) == sum(hops + (c % 3) + 1 for c in range(n))
    per_channel = _lines_by_channel(out)
    for d in range(4):
        rc1, out1, err1 = _run_fifos(exe, tmp_path, f"dev{d}", data[64 * d:64 * (d + 1)], [f"--device={d}"], env)
        assert rc1 == 0, err1[-2000:]
        single = _lines_by_channel(out1, 64 * d)
        for c in range(64 * d, 64 * (d + 1)):
            assert per_channel[c] == single[c], c


def test_devices_option_errors_and_interleaved_split(exe, tmp_path):
    r = subprocess.run([exe, "--devices=0,1"], input=b"", capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DEVICES="2"))
    assert r.returncode == 2 and "--devices splits the streams" in r.stderr.decode()
    p = tmp_path / "a.s16"
    p.write_bytes(marked_stream(2, 1).tobytes())
    r = subprocess.run([exe, f"--inputs={p},{p}", "--devices=0,gpu1"], capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DEVICES="2"))
    assert r.returncode == 2 and "not 'gpu1'" in r.stderr.decode()
    r = subprocess.run([exe, f"--inputs={p},{p}", "--devices=0,5"], capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DEVICES="2"))
    assert r.returncode == 2 and "device 5: device ordinal out of range" in r.stderr.decode()
    # more devices than streams: the surplus devices get no loop; --devices=all asks the library
    r = subprocess.run([exe, f"--inputs={p},{p}", "--devices=all"], capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DEVICES="3"))
    assert r.returncode == 0 and "device 2 decodes" not in r.stderr.decode() and "device 1 decodes streams 1..1" in r.stderr.decode()
    # --interleaved=5 over two devices: streams 0..2 on device 0, 3..4 on device 1, one reader splitting every block
    n, hops = 5, 6
    streams = [marked_stream(hops, 300 * c) for c in range(n)]
    blocks = [np.stack([s[:5184] for s in streams]).tobytes()] + [np.stack([s[5184 + h * 2592:5184 + (h + 1) * 2592] for s in streams]).tobytes() for h in range(hops)]
    r = subprocess.run([exe, f"--interleaved={n}", "--devices=0,1"], input=b"".join(blocks) + b"\0" * 64, capture_output=True, timeout=60,
                       env=dict(os.environ, MSK144_STUB_DECODE_MS="5", MSK144_STUB_DEVICES="2"))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    devs = {}
    seen = windows_seen(r.stdout.decode(), n, devs)
    for c in range(n):
        assert seen[c] == [(300 * c + k, 300 * c + k + 1) for k in range(hops + 1)], c
        assert devs[c] == {0 if c < 3 else 1}
    assert f"{n * (hops + 1)} stream hops" in r.stderr.decode() and "Incomplete read error. rc=32" in r.stderr.decode()


def test_result_list_overflow_is_reported_not_fatal(exe, tmp_path):
    """one busy hop must not kill every stream.  --max-results=2 with five streams that all 'decode' every hop: each hop
    is cut to two records and reported, the loop keeps running to the end of every stream and exits 0."""
    paths = []
    for c in range(5):
        p = tmp_path / f"s{c}.s16"
        p.write_bytes(marked_stream(4, 100 * c).tobytes())
        paths.append(str(p))
    r = subprocess.run([exe, "--max-results=2", "--inputs=" + ",".join(paths)], capture_output=True, timeout=60, env=dict(os.environ, MSK144_STUB_DECODE_MS="2"))
    err = r.stderr.decode()
    assert r.returncode == 0, err[-1500:]
    seen = windows_seen(r.stdout.decode(), 5)
    assert [len(seen[c]) for c in range(5)] == [5, 5, 0, 0, 0]              # the records that fitted were processed
    assert err.count("the list was cut, decoding goes on") == 5 and "5 hops overflowed the result list" in err
    assert "5 batches, 25 stream hops" in err


def test_fifo_writer_that_leaves_without_writing_ends_its_stream_at_once(exe, tmp_path):
    """a writer that opens and closes without a byte used to count as 'not connected yet' for the whole connect timeout
    (10 s), holding every batch for the linger.  poll() reports its hang-up; the stream ends, the other one is decoded in full."""
    paths = [str(tmp_path / "empty.fifo"), str(tmp_path / "full.fifo")]
    for p in paths:
        os.mkfifo(p)
    proc = subprocess.Popen([exe, "--connect-timeout-ms=30000", "--inputs=" + ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, MSK144_STUB_DECODE_MS="3"))
    t0 = time.monotonic()
    os.close(os.open(paths[0], os.O_WRONLY))
    with open(paths[1], "wb", buffering=0) as f:
        f.write(marked_stream(5, 40).tobytes())
    out, err = proc.communicate(timeout=25)                                  # well inside the 30 s connect timeout
    assert proc.returncode == 0 and time.monotonic() - t0 < 20, err.decode()[-1000:]
    seen = windows_seen(out.decode(), 2)
    assert seen[0] == [] and seen[1] == [(40 + k, 41 + k) for k in range(6)]
    assert "ch=0: Incomplete read error. rc=0" in err.decode()


def test_sigterm_finishes_hops_in_flight_and_exits_in_order(exe, tmp_path):
    """Service mode: SIGTERM (or Ctrl-C) while live FIFOs are being decoded - the loops stop reading, the hops already submitted are
    collected and printed, the summary and "Done" follow, exit code 0 (the reference has no handler: it dies mid-hop)."""
    import signal
    n = 6
    paths = [str(tmp_path / f"live{c}.fifo") for c in range(n)]
    for p in paths:
        os.mkfifo(p)
    proc = subprocess.Popen([exe, "--devices=0,1", "--inputs=" + ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, MSK144_STUB_DECODE_MS="5", MSK144_STUB_DEVICES="2"))
    stop = threading.Event()

    def feed(c):
        x = marked_stream(400, 100 * c).tobytes()
        try:
            with open(paths[c], "wb", buffering=0) as f:
                f.write(x[:5184 * 2])
                off = 5184 * 2
                while not stop.is_set() and off < len(x):
                    time.sleep(0.02)
                    f.write(x[off:off + 5184])
                    off += 5184
        except BrokenPipeError:
            pass

    ths = [threading.Thread(target=feed, args=(c,)) for c in range(n)]
    for t in ths:
        t.start()
    time.sleep(0.6)
    proc.send_signal(signal.SIGTERM)
    out, err = proc.communicate(timeout=30)
    stop.set()
    for t in ths:
        t.join()
    assert proc.returncode == 0, err.decode()[-1500:]
    assert out.decode().strip().endswith("Done") and "stopped by signal" in err.decode()
    seen = windows_seen(out.decode(), n)
    for c in range(n):
        assert len(seen[c]) >= 5 and seen[c] == [(100 * c + k, 100 * c + k + 1) for k in range(len(seen[c]))], c     # a prefix of the stream, in order
    m = re.search(r"msk144hipdecoder: (\d+) batches, (\d+) stream hops", err.decode())
    assert m and int(m.group(2)) == sum(len(seen[c]) for c in range(n))


def _interleaved_blocks(n, hops):
    streams = [marked_stream(hops, 300 * c) for c in range(n)]
    blocks = [np.stack([s[:5184] for s in streams]).tobytes()]
    for h in range(hops):
        blocks.append(np.stack([s[5184 + h * 2592:5184 + (h + 1) * 2592] for s in streams]).tobytes())
    return blocks


@pytest.mark.parametrize("decode_ms,source", [("5", "file"), ("400", "file"), ("5", "slow_pipe")])
def test_sigterm_stops_an_interleaved_run_wherever_the_reader_is(exe, tmp_path, decode_ms, source):
    """the handler only sets a flag, so the stop must not depend on the signal landing inside a blocking read.  Three places
    the --interleaved reader can be in: stdin is a file that never blocks (`cat big.raw | ...`: the reader lives in feed()'s
    back-pressure wait or between reads), the same with a slow "GPU" (always in the back-pressure wait), and a pipe that delivers
    a block every 100 ms (the reader sleeps in poll).  Each time: exit 0 within two seconds, "stopped by signal", "Done", the windows
    printed are a prefix of every stream, in order."""
    import signal
    n, hops = 4, 4000
    blocks = _interleaved_blocks(n, 40)
    if source == "file":
        path = tmp_path / "big.raw"
        with open(path, "wb") as f:
            f.write(blocks[0])
            for k in range(hops):
                f.write(blocks[1 + k % 40])               # ~80 MB: minutes of decoding at 5 ms per hop, nowhere near done when the signal comes
        stdin = open(path, "rb")
    else:
        stdin = subprocess.PIPE
    proc = subprocess.Popen([exe, f"--interleaved={n}", "--devices=0,1"], stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, MSK144_STUB_DECODE_MS=decode_ms, MSK144_STUB_DEVICES="2"))
    stop = threading.Event()

    def slow_writer():
        try:
            proc.stdin.write(blocks[0])
            k = 0
            while not stop.is_set():
                time.sleep(0.1)
                proc.stdin.write(blocks[1 + k % 40])
                proc.stdin.flush()
                k += 1
        except (BrokenPipeError, ValueError):
            pass

    th = None
    if source == "slow_pipe":
        th = threading.Thread(target=slow_writer)
        th.start()
    time.sleep(1.0)
    t0 = time.monotonic()
    proc.send_signal(signal.SIGTERM)
    try:
        out, err = proc.communicate(timeout=20) if source == "file" else (None, None)
        if source != "file":
            # communicate() would close stdin, which alone would end the reader: wait for the exit first
            proc.wait(timeout=20)
            stop.set()
            th.join()
            out, err = proc.stdout.read(), proc.stderr.read()
    finally:
        stop.set()
        if proc.poll() is None:
            proc.kill()
    took = time.monotonic() - t0
    if source == "file":
        stdin.close()
    assert proc.returncode == 0, err.decode()[-1500:]
    assert took < 2.0 + 4 * int(decode_ms) / 1000.0, took     # two queued blocks + two slots in flight may still finish
    assert out.decode().strip().endswith("Done") and "stopped by signal" in err.decode()
    assert "Incomplete read error" not in err.decode()       # a stop is not an end of input
    seen = windows_seen(out.decode(), n)
    for c in range(n):
        got = len(seen[c])
        assert got >= 1 and got < hops
        if source == "slow_pipe":
            assert seen[c] == [(300 * c + k, 300 * c + k + 1) for k in range(got)], c


def test_a_failing_device_loop_stops_its_siblings(exe, tmp_path):
    """in a multi-device run over live FIFOs a loop whose device fails used to leave the other loops decoding until their
    writers went away - the process never exited.  Now the failure is a stop request for every loop: hops in flight are finished and
    the program exits 2 while the writers are still there."""
    n = 4
    paths = [str(tmp_path / f"live{c}.fifo") for c in range(n)]
    for p in paths:
        os.mkfifo(p)
    proc = subprocess.Popen([exe, "--devices=0,1", "--inputs=" + ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, MSK144_STUB_DECODE_MS="5", MSK144_STUB_DEVICES="2", MSK144_STUB_FAIL_DEVICE="1", MSK144_STUB_FAIL_AFTER="3"))
    stop = threading.Event()

    def feed(c):
        x = marked_stream(2000, 100 * c).tobytes()
        try:
            with open(paths[c], "wb", buffering=0) as f:
                f.write(x[:5184 * 2])
                off = 5184 * 2
                while not stop.is_set() and off < len(x):
                    time.sleep(0.02)
                    f.write(x[off:off + 5184])
                    off += 5184
        except BrokenPipeError:
            pass

    ths = [threading.Thread(target=feed, args=(c,)) for c in range(n)]
    for t in ths:
        t.start()
    try:
        rc = proc.wait(timeout=15)                           # exits on its own, writers still connected
    finally:
        stop.set()
        if proc.poll() is None:
            proc.kill()
    out, err = proc.stdout.read().decode(), proc.stderr.read().decode()
    for t in ths:
        t.join()
    assert rc == 2 and "injected device failure" in err, err[-1500:]
    assert "Done" not in out
    seen = windows_seen(out + "Done\n", n)                   # (the helper strips the text: keep the last line's trailing blank)
    assert len(seen[0]) >= 3 and len(seen[2]) == 3           # device 0's streams ran on, device 1's got three hops through
