#!/usr/bin/env python3
"""Build A/B partners of libmsk144hip.so that differ from the tree by preprocessor defines only (experiments kept in the sources behind
#ifdef, never in the product build):

    python tools/ab_variants.py name=DEFINE[=VALUE][,DEFINE2...] [name2=...]   ->  tools/ab/libmsk144hip_<name>.so

tools/ab_bench.sh then alternates bench.py between the tree's library and those on ONE box (boxes of the pool differ by +-4 %)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msk144cudecoder_amd import build as b  # noqa: E402

os.makedirs(os.path.join(ROOT, "tools", "ab"), exist_ok=True)
for spec in sys.argv[1:]:
    name, defs = spec.split("=", 1)
    print(b.build_library(out=os.path.join(ROOT, "tools", "ab", f"libmsk144hip_{name}.so"), defines=defs.split(",")))
