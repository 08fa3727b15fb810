"""Build libmsk144hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

`python -m msk144cudecoder_amd.build` or build_library().  hipcc cross-compiles for gfx950 without a
GPU present.  Objects are rebuilt only when their sources changed.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_BUILD = os.path.join(_CSRC, "build")
LIB_PATH = os.path.join(_PKG, "libmsk144hip.so")

ARCH = "gfx950"
COMMON = ["-std=c++17", "-O3", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# (source, extra flags).  frontend.hip and the host API keep mul and add separate so that their float
# arithmetic is the reference's expression for expression.
SOURCES = [
    ("frontend.hip", ["-ffp-contract=off"]),
    ("scan.hip", []),
    ("softbits.hip", []),
    ("index.hip", []),
    ("hopring.hip", []),
    ("ldpc.hip", []),
    ("msk144_api.cpp", ["-x", "hip", "-ffp-contract=off"]),
]
HEADERS = sorted(f for f in os.listdir(_CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "msk144hip.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(_BUILD, exist_ok=True)
    hdrs = [os.path.join(_CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    relink = force
    for src, extra in SOURCES:
        s = os.path.join(_CSRC, src)
        o = os.path.join(_BUILD, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            relink = True
    if relink or _newer(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
