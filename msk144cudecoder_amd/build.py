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


def build_library(force: bool = False, verbose: bool = False, out: str = None, defines=(), extra_sources=(), overrides=None) -> str:
    """The product library.  `out` / `defines` / `extra_sources` / `overrides`: a differently configured copy under its own name and
    object directory (tools/phase_stamps.py builds the -DMSK144_PHASE_STAMPS diagnostic library this way; tools/ab_build.py swaps
    single sources for an earlier revision's, {"ldpc.hip": "/path/to/old/ldpc.hip"}, for same-box A/B runs); the product build
    takes none."""
    overrides = overrides or {}
    hipcc = _hipcc()
    lib_path = out or LIB_PATH
    build_dir = _BUILD if out is None else os.path.join(_BUILD, "variant_" + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(build_dir, exist_ok=True)
    hdrs = [os.path.join(_CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    flags = ["-D" + d for d in defines]
    objs = []
    relink = force
    for src, extra in list(SOURCES) + [(s, []) for s in extra_sources]:
        s = overrides.get(src, os.path.join(_CSRC, src))
        o = os.path.join(build_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + flags + extra + ["-I", _CSRC, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            relink = True
    if relink or _newer(lib_path, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib_path


def build_stamps_library(verbose: bool = False) -> str:
    """libmsk144hip_stamps.so: the same sources with -DMSK144_PHASE_STAMPS (csrc/phase_stamps.h) - diagnostic only."""
    return build_library(verbose=verbose, out=os.path.join(_PKG, "libmsk144hip_stamps.so"), defines=["MSK144_PHASE_STAMPS"], extra_sources=["phase_stamps.hip"])


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
