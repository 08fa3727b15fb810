#!/usr/bin/env python3
"""Sums the per-test parity reports a `pytest -m gpu` run leaves in gpurun_out/parity_report.json (tests/conftest.py): scan slots
and near-ties, nbadsync marginals, BP decodes and verified-marginal flips, the largest LLR difference.

    python tools/summarise_parity_report.py [gpurun_out/parity_report.json]
"""
import json
import sys


def walk(node, acc):
    if isinstance(node, dict):
        if "near_ties" in node and "total" in node:
            acc["scan_slots"] += node["total"]
            acc["scan_near_ties"] += node["near_ties"]
            acc["scan_periodic_groups"] += node.get("periodic_groups", 0)
        if "nbadsync_marginal" in node:
            acc["nbadsync_marginal"] += node["nbadsync_marginal"] or 0
            acc["llr_max_abs_diff"] = max(acc["llr_max_abs_diff"], node.get("llr_max_abs_diff", 0.0))
        if "marginal_flips" in node:
            acc["bp_compared"] += node.get("checked", node.get("compared", 0))
            acc["bp_marginal_flips"] += node["marginal_flips"]
        for v in node.values():
            walk(v, acc)
    elif isinstance(node, list):
        for v in node:
            walk(v, acc)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_report.json"
    rep = json.load(open(path))
    acc = dict(reports=len(rep), scan_slots=0, scan_near_ties=0, scan_periodic_groups=0, nbadsync_marginal=0, llr_max_abs_diff=0.0, bp_compared=0, bp_marginal_flips=0)
    walk(rep, acc)
    print(json.dumps(acc, indent=1))


if __name__ == "__main__":
    main()
