"""Turns the PMC passes of tools/collect_profiles.sh into profiles/counters.json (what bench.py reads for roofline.traffic and
valu_issue) - per kernel and per launch: HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE/WRITE_SIZE are in KB; on
gfx950 FETCH_SIZE reports half of the bytes read: MI355X_MICROARCH.md, HBM section; checked on known byte counts of our own
access patterns, see profiles/README.md), SQ instruction counts and wait shares.  Records the kernel-source hash and commit the
counters were collected on so that bench.py never mixes them with other kernels.

    python tools/make_counters_json.py <tag>      (reads gpurun_out/<tag>_pmc/*.csv, writes profiles/<tag>_counters.json + counters.json)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import pmc_summary  # noqa: E402


def short(name):
    for k in ("frontend_fir_kernel", "frontend_fft_kernel", "scan_kernel", "softbits_kernel", "index_kernel", "ldpc_kernel", "collect_count_kernel", "collect_scatter_kernel"):
        if k in name:
            return "frontend_kernel" if k.startswith("frontend") else k
    return None


def main():
    tag = sys.argv[1]
    d = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc")
    paths = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith("_counter_collection.csv")]
    summ = pmc_summary.summarise(paths)
    # launches per bench step of every kernel (blocked staging launches softbits/index/ldpc once per channel block)
    per_step = {}
    import csv
    for row in csv.DictReader(open(os.path.join(d, "sq1_kernel_trace.csv"))):
        k = short(row["Kernel_Name"])
        if k:
            per_step[k] = per_step.get(k, 0) + 1
    steps = 3.0  # --steps 2 --warmup 1
    kernels = {}
    for name, ctr in summ.items():
        k = short(name)
        if not k:
            continue
        n = per_step.get(k, steps) / steps
        e = kernels.setdefault(k, {"launches_per_step": n})
        for c, v in ctr.items():
            e[c] = e.get(c, 0.0) + v * n          # per STEP: average per launch x launches per step
    cand = 1024 * 24048
    for k, e in kernels.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_step"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
            e["hbm_bytes_per_launch"] = e["hbm_bytes_per_step"] / e["launches_per_step"]
            e["hbm_bytes_per_candidate"] = e["hbm_bytes_per_step"] / cand
        if e.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in e:
                    e[c + "_share_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
        if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_WAVE_CYCLES"):
            e["avg_waves_per_simd"] = e["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
    out = {
        "_how": "tools/collect_profiles.sh on one MI355X: separate rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` (1024 channels, width 500 / "
                "step 1 / depth 6 / threshold 3); values are per bench STEP (per-launch average x launches per step; blocked staging launches softbits/index/ldpc once per "
                "channel block: 128 channels by default).  hbm_bytes_per_step = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the kernel's launches of one step; hbm_bytes_per_launch = that / launches_per_step.",
        "kernel_source_sha": bench.kernel_source_sha(),
        "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
        "candidates_per_step": cand,
        "kernels": kernels,
    }
    for p in (os.path.join(ROOT, "profiles", f"{tag}_counters.json"), os.path.join(ROOT, "profiles", "counters.json")):
        with open(p, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")
    for k, e in sorted(kernels.items()):
        print(f"{k:24s} launches/step {e['launches_per_step']:.0f}  HBM {e.get('hbm_bytes_per_step', 0) / 1e9:8.3f} GB/step  VALU {e.get('SQ_INSTS_VALU', 0):.3e}  waves/SIMD {e.get('avg_waves_per_simd', 0):.2f}")


if __name__ == "__main__":
    main()
