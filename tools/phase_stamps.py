#!/usr/bin/env python3
"""Where do scan_kernel and softbits_kernel spend their cycles?  In-kernel s_memtime stamps against priced issue cycles.

Builds the diagnostic library (libmsk144hip_stamps.so: the product sources with -DMSK144_PHASE_STAMPS, csrc/phase_stamps.h), runs
the bench workload (1024 channels, width 500 / step 1 / depth 6 / threshold 3) on it and reads back, for every 61st workgroup, the
shader cycle at each phase boundary of wave 0 and the cycle at which each of its eight waves ended.  Next to the measured cycles
it prices the VALU instructions between the same two stamps in the diagnostic build's own listing (the s_memtime instructions
delimit the segments) with the issue costs of tools/asm_mix.py.

    python tools/phase_stamps.py --listing      # CPU: compile the listings, print the priced segments
    python tools/phase_stamps.py --out profiles/r04_phase_cycles.txt      # GPU box

A wave's measured cycles are WALL cycles while it shares its SIMD with the other resident waves; priced cycles are what the
wave's own VALU instructions need to issue.  What the table shows is each phase's SHARE of both: a phase whose measured share is
well above its priced share is waiting (barrier, LDS latency, bank conflicts), not issuing.
"""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_mix  # noqa: E402

CSRC = os.path.join(ROOT, "msk144cudecoder_amd", "csrc")

SCAN_PHASES = [  # (name, from slot, to slot)
    ("mix (global loads, sincos, LDS store)", 11, 1), ("barrier after mix", 1, 2), ("correlate (pulse decomposition)", 2, 3),
    ("barrier, C store, barrier", 3, 4), ("E pass (read, barrier, write, barrier)", 4, 5), ("fold + running arg-max (11 positions x D)", 5, 6),
    ("octet merge (DPP max, ballots, stores)", 6, 7), ("barrier after octet merge", 7, 8), ("slice merge 4a + barrier", 8, 9), ("8-slot rule (wave 0 only)", 9, 10)]
SOFTBITS_PHASES = [("mix (global loads, sincos, LDS store)", 11, 1), ("barrier after mix", 1, 2)]


def listing(src, kernel_pat):
    """hipcc -S of the stamps build, device side: the lines of the first kernel whose mangled name matches."""
    out = os.path.join(tempfile.mkdtemp(prefix="msk144_lst_"), os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-DMSK144_PHASE_STAMPS", "-DMSK144_LISTING_EVEN_ONLY", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
                   check=True)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kernel_pat + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def segments(lines):
    """VALU issue cycles, VALU / LDS / VMEM instruction counts and barrier count between consecutive s_memtime instructions."""
    segs, cur = [], dict(cycles=0.0, valu=0, ds=0, vmem=0, barriers=0, backward=False)
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(lines):
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
        if not m:
            continue
        op = m.group(1)
        if op == "s_memtime":
            segs.append(cur)
            cur = dict(cycles=0.0, valu=0, ds=0, vmem=0, barriers=0, backward=False)
        elif op.startswith("v_"):
            cur["cycles"] += asm_mix.cost(op, l.split(";")[0])[1]
            cur["valu"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            cur["vmem"] += 1
        elif op == "s_barrier":
            cur["barriers"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            t = m.group(2).split()[0] if m.group(2) else ""
            if t in labels and labels[t] < i:
                cur["backward"] = True
    segs.append(cur)
    return segs


def fmt_seg(s):
    return f"{s['cycles']:7.0f} priced cycles  ({s['valu']:4d} VALU, {s['ds']:3d} LDS, {s['vmem']:2d} VMEM, {s['barriers']} barriers{', LOOP body priced once' if s['backward'] else ''})"


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--listing", action="store_true", help="only compile and price the listings (no GPU)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--channels", type=int, default=1024)
    a = ap.parse_args()

    scan_seg = segments(listing("scan.hip", "scan_kernelILi6E"))
    sb_seg = segments(listing("softbits.hip", "softbits_kernelILb1ELb1E"))
    report = []
    P = report.append
    P("Phase cycles of scan_kernel<6> and softbits_kernel<true, true> - in-kernel s_memtime stamps (tools/phase_stamps.py, csrc/phase_stamps.h)")
    P("priced = VALU issue cycles of the instructions between the two stamps in the stamps build's listing (tools/asm_mix.py costs)")
    P("")
    P("scan_kernel<6> listing, segments between consecutive s_memtime (stamp order: 0, 11, 1, 2, ... 10, wave-end):")
    for i, s in enumerate(scan_seg):
        P(f"  seg {i:2d}: {fmt_seg(s)}")
    P("softbits_kernel<true, true> listing, segments between consecutive s_memtime:")
    for i, s in enumerate(sb_seg):
        P(f"  seg {i:2d}: {fmt_seg(s)}")
    if a.listing:
        print("\n".join(report))
        return

    import numpy as np
    import torch  # noqa: F401
    import bench
    from msk144cudecoder_amd import build as b
    from msk144cudecoder_amd import hipdecoder as hd
    lib = b.build_stamps_library()
    hd._lib = hd.load_library(lib)
    L = hd._lib
    L.msk144_debug_read_stamps.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    wins, _ = bench.make_inputs(0, a.channels)
    dev = torch.from_numpy(wins).cuda()
    import time
    with hd.HipDecoder(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3, channels=a.channels, max_results=1 << 20) as d:
        t0 = time.perf_counter()
        i = 0
        while time.perf_counter() - t0 < 2.5:         # clock settles under load
            d.submit_audio_device(dev[i % 4].data_ptr())
            d.decode()
            i += 1
            if i % 4 == 0:
                d.synchronize()
        d.synchronize()
        mhz = d.clock_probe(1000)
        d.set_profiling(True)
        d.stage_times(reset=True)
        for k in range(4):
            d.submit_audio_device(dev[k % 4].data_ptr())
            d.decode()
        d.synchronize()
        st = d.stage_times()
        rows = 16384
        slots, every = C.c_int(0), C.c_int(0)
        data = {}
        for kern, name in ((0, "scan"), (1, "softbits")):
            buf = np.zeros((rows, 24), dtype=np.uint64)
            n = L.msk144_debug_read_stamps(kern, buf.ctypes.data_as(C.c_void_p), rows, C.byref(slots), C.byref(every))
            assert n > 0, n
            data[name] = buf[:n][buf[:n, 0] != 0].astype(np.int64)
    P("")
    P(f"measured on {torch.cuda.get_device_name(0)}: shader clock {mhz:.0f} MHz after 2.5 s of back-to-back steps; stamps build stage times per 1024-channel step: "
      f"scan {st['scan'][0]:.2f} ms, softbits {st['softbits'][0]:.2f} ms (the stamps themselves cost a few %: product build figures are in the bench line)")

    # ---- scan ----
    s = data["scan"]
    self_cost = median((s[:, 11] - s[:, 0]).tolist())
    life = (s[:, 16:24].max(axis=1) - s[:, 0])
    w0_life = s[:, 16] - s[:, 0]
    others_end = s[:, 17:24].max(axis=1) - s[:, 0]
    P("")
    P(f"scan_kernel<6>: {len(s)} sampled workgroups; one stamp costs {self_cost} cycles (subtracted per phase); median workgroup lifetime {median(life.tolist())} cycles "
      f"(waves 1-7 end at {median(others_end.tolist())}, wave 0 - which runs the slot rule - at {median(w0_life.tolist())})")
    # priced cycles per phase: segments are delimited in stamp order 0, 11, 1, 2, 3, ..., 10: segment k+1 lies between the k-th and (k+1)-th stamp
    order = [0, 11, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    seg_of = {(order[k], order[k + 1]): scan_seg[k + 1] for k in range(len(order) - 1)}
    # between stamps 9 and 10 the listing also holds the wave-end stamp of waves 1-7: segment 11 (start of the slot rule) and
    # segment 12 (the rule's loop body, run 13 times after the one-step start)
    seg_of[(9, 10)] = dict(cycles=scan_seg[11]["cycles"] + 13 * scan_seg[12]["cycles"], valu=scan_seg[11]["valu"] + 13 * scan_seg[12]["valu"],
                           ds=scan_seg[11]["ds"] + 13 * scan_seg[12]["ds"], vmem=0, barriers=0, backward=True)
    tot_meas = median((s[:, 10] - s[:, 11]).tolist()) - 10 * self_cost
    rows_out = []
    tot_priced = 0.0
    for name, f, t in SCAN_PHASES:
        meas = median((s[:, t] - s[:, f]).tolist()) - self_cost
        seg = seg_of[(f, t)]
        priced = seg["cycles"]
        tot_priced += priced
        rows_out.append((name, meas, priced, seg))
    P(f"  {'phase (wave 0)':48s} {'measured':>9s} {'share':>6s} {'priced':>8s} {'share':>6s}  measured/priced   LDS instr, barriers")
    for name, meas, priced, seg in rows_out:
        ratio = f"{meas / priced:5.2f}" if priced > 50 else "    -"
        P(f"  {name:48s} {meas:9d} {100.0 * meas / tot_meas:5.1f}% {priced:8.0f} {100.0 * priced / tot_priced:5.1f}%  {ratio}             {seg['ds']:3d}, {seg['barriers']}")
    P(f"  {'sum (wave 0, entry -> slot rule done)':48s} {tot_meas:9d}        {tot_priced:8.0f}         {tot_meas / tot_priced:5.2f}")

    # ---- softbits ----
    s = data["softbits"]
    self_cost = median((s[:, 11] - s[:, 0]).tolist())
    life = (s[:, 16:24].max(axis=1) - s[:, 0])
    P("")
    P(f"softbits_kernel<true, true>: {len(s)} sampled workgroups; stamp cost {self_cost}; median workgroup lifetime {median(life.tolist())} cycles; wave 0 ends at "
      f"{median((s[:, 16] - s[:, 0]).tolist())}, the last wave at {median(life.tolist())}, the first at {median((s[:, 16:24].min(axis=1) - s[:, 0]).tolist())}")
    n2 = s[:, 5]
    # listing order of softbits stamps: 0, 11, 1, 2, [loop: t0, t1, t2], 6, wave-end
    sb_names = ["before stamp 0", "stamp pair", "mix (global loads, sincos, LDS store)", "barrier after mix", "candidate loop set-up", "part one: fold slots 0+2, filter, phase, sync check",
                "gate + part two: fold slot 1, filter, normalise, LLR row", "loop tail -> end"]
    mix = median((s[:, 1] - s[:, 11]).tolist()) - self_cost
    bar = median((s[:, 2] - s[:, 1]).tolist()) - self_cost
    p1 = median((s[:, 3] // 6).tolist()) - self_cost
    p2 = median((s[:, 4] // np.maximum(n2, 1))[n2 > 0].tolist()) - self_cost
    kept = float(n2.mean()) / 6.0
    total = median((s[:, 6] - s[:, 11]).tolist())
    P(f"  candidates per wave: 6; part two ran for {100 * kept:.1f} % of them (nbadsync <= 3)")
    P(f"  {'phase (wave 0)':60s} {'measured':>9s} {'priced':>8s}  measured/priced")
    for nm, meas, seg, mult in ((sb_names[2], mix, sb_seg[2], 1), (sb_names[3], bar, sb_seg[3], 1), (sb_names[5] + " (per candidate)", p1, sb_seg[5], 1),
                                (sb_names[6] + " (per kept candidate)", p2, sb_seg[6], 1)):
        ratio = f"{meas / seg['cycles']:5.2f}" if seg["cycles"] > 50 else "    -"
        P(f"  {nm:60s} {meas:9d} {seg['cycles']:8.0f}  {ratio}    ({seg['ds']} LDS, {seg['vmem']} VMEM instr)")
    est = mix + bar + 6 * p1 + 6 * kept * p2
    pr = sb_seg[2]["cycles"] + 6 * sb_seg[5]["cycles"] + 6 * kept * sb_seg[6]["cycles"]
    P(f"  {'wave 0 total = mix + barrier + 6 part one + 6 x kept x part two':60s} {int(est):9d} {pr:8.0f}  {est / pr:5.2f}    (entry -> end measured: {total})")
    text = "\n".join(report)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
