#!/usr/bin/env python3
"""Build an A/B partner of libmsk144hip.so: the current tree with some kernel sources taken from an earlier git revision.

    python tools/ab_build.py <rev> <name> file.hip [file2.hip ...]   ->  tools/ab/libmsk144hip_<name>.so

tools/ab_bench.sh then alternates bench.py between the tree's library and that one on ONE box (boxes of the pool differ by +-4 %)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msk144cudecoder_amd import build as b  # noqa: E402

rev, name, files = sys.argv[1], sys.argv[2], sys.argv[3:]
tmp = os.path.join(ROOT, "tools", "ab", "src_" + name)
os.makedirs(tmp, exist_ok=True)
over = {}
for f in files:
    text = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:msk144cudecoder_amd/csrc/{f}"], check=True, capture_output=True).stdout
    open(os.path.join(tmp, f), "wb").write(text)
    over[f] = os.path.join(tmp, f)
print(b.build_library(out=os.path.join(ROOT, "tools", "ab", f"libmsk144hip_{name}.so"), overrides=over))
