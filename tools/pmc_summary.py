"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter, the average over dispatches."""
import csv
import sys
from collections import defaultdict


def summarise(paths, kernel_filter=None):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(p)):
            k = row["Kernel_Name"]
            if kernel_filter and kernel_filter not in k:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = k
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


if __name__ == "__main__":
    filt = None
    paths = []
    for a in sys.argv[1:]:
        if a.startswith("--kernel="):
            filt = a.split("=", 1)[1]
        else:
            paths.append(a)
    for k, d in summarise(paths, filt).items():
        print(k[:60])
        for c in sorted(d):
            print(f"   {c:28s} {d[c]:.4g}")
