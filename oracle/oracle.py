"""ctypes binding of the CPU oracle (libmsk144_oracle.so) - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (msk144cudecoder_amd) never does.  See oracle/msk144_oracle.h for provenance
("parity unpinned") and the reference file:line each function follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libmsk144_oracle.so")

ITEM_DTYPE = np.dtype([
    ("block_idx", "<u4"), ("pattern_idx", "<u4"), ("pos", "<u4"), ("f0", "<f4"), ("nbadsync", "<i4"), ("xb", "<f4"),
    ("num_avg", "<i4"), ("softbits_wo_sync", "<f4", (128,)), ("is_message_present", "u1"), ("_pad0", "u1", (3,)),
    ("ldpc_num_iterations", "<i4"), ("ldpc_num_hard_errors", "<i4"), ("message", "i1", (77,)), ("_pad1", "u1", (3,)),
])
assert ITEM_DTYPE.itemsize == 632


class Ctx(C.Structure):
    _fields_ = [("center_freq", C.c_float), ("step", C.c_float), ("if1", C.c_float), ("num_blocks", C.c_int), ("scan_depth", C.c_int),
                ("nbadsync_threshold", C.c_int), ("total_items", C.c_int), ("num_threads", C.c_int)]


class SnrTracker(C.Structure):
    _fields_ = [("noise_power", C.c_float), ("snr", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_DIR, "msk144_oracle.cpp"), os.path.join(_DIR, "msk144_oracle.h"),
            os.path.join(_DIR, "..", "msk144cudecoder_amd", "csrc", "msk144_protocol.h")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", _DIR, "-B" if force else "-s"], check=True, stdout=subprocess.DEVNULL)
    return _SO


BENCH_FLAGS = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp"]
_BENCH_DIR = os.path.join(_DIR, "_bench")
_BENCH_SO = os.path.join(_BENCH_DIR, "libmsk144_oracle_bench.so")


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_bench_library() -> str:
    """Second build of the SAME source for bench.py's cpu_baseline leg only: -O3 -march=native (the parity tests keep the -O2
    build of the Makefile).  -march=native is host-specific, so the file is compiled on the machine that times it: oracle/_bench/
    is git-ignored AND gpurun-ignored, and a stamp records the CPU model and flags it was built for."""
    import json
    src = os.path.join(_DIR, "msk144_oracle.cpp")
    stamp_path = _BENCH_SO + ".json"
    stamp = {"cpu": _cpu_model(), "flags": BENCH_FLAGS, "src_mtime": os.path.getmtime(src)}
    try:
        if os.path.exists(_BENCH_SO) and json.load(open(stamp_path)) == stamp:
            return _BENCH_SO
    except (OSError, ValueError):
        pass
    os.makedirs(_BENCH_DIR, exist_ok=True)
    subprocess.run([os.environ.get("CXX", "g++")] + BENCH_FLAGS + ["-shared", "-o", _BENCH_SO, src, "-lm"], check=True)
    json.dump(stamp, open(stamp_path, "w"))
    return _BENCH_SO


def _load(path: str):
    L = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    L.orc_sizeof_item.restype = ip
    L.orc_ctx_init.argtypes = [C.POINTER(Ctx), fp, fp, fp, ip, ip]
    L.orc_set_threads.argtypes = [C.POINTER(Ctx), ip]
    L.orc_frequency.argtypes = [C.POINTER(Ctx), ip]
    L.orc_frequency.restype = fp
    for name in ("orc_get_cb42", "orc_get_pp12"):
        getattr(L, name).argtypes = [vp]
    L.orc_normalize_audio.argtypes = [vp, vp]
    L.orc_convert_iq.argtypes = [vp, vp]
    L.orc_analytic2.argtypes = [vp, vp, ip]
    L.orc_analytic_fft.argtypes = [vp, vp]
    L.orc_frontend_audio.argtypes = [vp, ip, vp]
    L.orc_frontend_iq.argtypes = [vp, vp]
    L.orc_clear_items.argtypes = [C.POINTER(Ctx), vp]
    L.orc_scan.argtypes = [C.POINTER(Ctx), vp, vp]
    L.orc_softbits.argtypes = [C.POINTER(Ctx), vp, vp]
    L.orc_index.argtypes = [C.POINTER(Ctx), vp, vp]
    L.orc_index.restype = ip
    L.orc_ldpc.argtypes = [C.POINTER(Ctx), vp, vp, ip]
    L.orc_decode_window.argtypes = [C.POINTER(Ctx), vp, vp, vp]
    L.orc_decode_window.restype = ip
    L.orc_scan_xb.argtypes = [C.POINTER(Ctx), vp, ip, ip, vp]
    L.orc_softbits_at.argtypes = [C.POINTER(Ctx), vp, ip, ip, C.c_uint32, vp, vp, vp]
    L.orc_ldpc_one.argtypes = [vp, vp, vp, vp]
    L.orc_ldpc_one.restype = ip
    L.orc_crc13.argtypes = [vp, ip]
    L.orc_crc13.restype = C.c_uint16
    L.orc_check_crc_bits.argtypes = [vp]
    L.orc_check_crc_bits.restype = ip
    L.orc_snr_init.argtypes = [C.POINTER(SnrTracker)]
    L.orc_snr_process.argtypes = [C.POINTER(SnrTracker), vp, C.c_uint]
    L.orc_snr_int.argtypes = [C.POINTER(SnrTracker)]
    L.orc_snr_int.restype = ip
    L.orc_segment_power.argtypes = [vp, C.c_uint, vp]
    L.orc_message_gate.argtypes = [vp]
    L.orc_message_gate.restype = ip
    assert L.orc_sizeof_item() == 632
    return L


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _load(_SO)
    return _lib


def bench_lib():
    """The -O3 -march=native build (bench.py's cpu_baseline only)."""
    return _load(build_bench_library())


_FMA_DIR = os.path.join(_DIR, "_fma")
FMA_VARIANTS = {"contract-fast": "libmsk144_oracle_contract.so", "forced-fma": "libmsk144_oracle_fmaf.so",
                "cuda-libm": "libmsk144_oracle_libm.so", "cuda-like": "libmsk144_oracle_cudalike.so"}


def host_has_fma() -> bool:
    try:
        return any(" fma " in line + " " for line in open("/proc/cpuinfo") if line.startswith("flags"))
    except OSError:
        return False


def fma_lib(variant: str):
    """One of the two FMA-contracting builds of the oracle source (oracle/Makefile target `fma`; tests/test_oracle_fma_bracket.py)."""
    subprocess.run(["make", "-s", "-C", _DIR, "fma"], check=True, stdout=subprocess.DEVNULL)
    L = _load(os.path.join(_FMA_DIR, FMA_VARIANTS[variant]))
    assert build_variant(L) == variant
    return L


def build_variant(L=None) -> str:
    L = lib() if L is None else L
    L.orc_build_variant.restype = C.c_char_p
    return L.orc_build_variant().decode()


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One search configuration of the reference decoder, evaluated on the CPU."""

    def __init__(self, center=1500.0, width=200.0, step=2.0, depth=4, nbadsync_threshold=1, threads=1, library=None):
        self.L = lib() if library is None else library
        self.ctx = Ctx()
        self.L.orc_ctx_init(C.byref(self.ctx), center, width, step, depth, nbadsync_threshold)
        self.L.orc_set_threads(C.byref(self.ctx), threads)

    # ---- geometry ----
    @property
    def F(self):
        return self.ctx.num_blocks

    @property
    def D(self):
        return self.ctx.scan_depth

    @property
    def total_items(self):
        return self.ctx.total_items

    def frequency(self, b):
        return float(self.L.orc_frequency(C.byref(self.ctx), int(b)))

    # ---- front ends ----
    def frontend_audio(self, win_i16: np.ndarray, method: int = 2) -> np.ndarray:
        w = np.ascontiguousarray(win_i16, dtype=np.int16)
        assert w.shape == (5184,)
        out = np.empty(5184, dtype=np.complex64)
        self.L.orc_frontend_audio(_p(w), method, _p(out))
        return out

    def frontend_iq(self, win_i8: np.ndarray) -> np.ndarray:
        w = np.ascontiguousarray(win_i8, dtype=np.int8)
        assert w.shape == (2 * 5184,)
        out = np.empty(5184, dtype=np.complex64)
        self.L.orc_frontend_iq(_p(w), _p(out))
        return out

    def normalize_audio(self, win_i16):
        w = np.ascontiguousarray(win_i16, dtype=np.int16)
        out = np.empty(5184, dtype=np.complex64)
        self.L.orc_normalize_audio(_p(w), _p(out))
        return out

    def analytic2(self, x, with_shift=True):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.empty(5184, dtype=np.complex64)
        self.L.orc_analytic2(_p(x), _p(out), 1 if with_shift else 0)
        return out

    def analytic_fft(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.empty(5184, dtype=np.complex64)
        self.L.orc_analytic_fft(_p(x), _p(out))
        return out

    # ---- kernels ----
    def new_items(self):
        return np.zeros(self.total_items, dtype=ITEM_DTYPE)

    def scan(self, cdat, items=None):
        cdat = np.ascontiguousarray(cdat, dtype=np.complex64)
        items = self.new_items() if items is None else items
        self.L.orc_scan(C.byref(self.ctx), _p(cdat), _p(items))
        return items

    def softbits(self, cdat, items):
        cdat = np.ascontiguousarray(cdat, dtype=np.complex64)
        self.L.orc_softbits(C.byref(self.ctx), _p(cdat), _p(items))
        return items

    def index(self, items):
        idx = np.empty(self.total_items, dtype=np.int32)
        n = self.L.orc_index(C.byref(self.ctx), _p(items), _p(idx))
        return idx[:n].copy()

    def ldpc(self, items, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.L.orc_ldpc(C.byref(self.ctx), _p(items), _p(idx), len(idx))
        return items

    def decode_window(self, cdat):
        """Full do_decode GPU part: returns (items, index list)."""
        cdat = np.ascontiguousarray(cdat, dtype=np.complex64)
        items = self.new_items()
        idx = np.empty(self.total_items, dtype=np.int32)
        n = self.L.orc_decode_window(C.byref(self.ctx), _p(cdat), _p(items), _p(idx))
        return items, idx[:n].copy()

    def scan_xb(self, cdat, block_idx, pattern_idx):
        cdat = np.ascontiguousarray(cdat, dtype=np.complex64)
        xb = np.empty(5376, dtype=np.float32)
        self.L.orc_scan_xb(C.byref(self.ctx), _p(cdat), block_idx, pattern_idx, _p(xb))
        return xb

    def softbits_at(self, cdat, block_idx, pattern_idx, pos):
        cdat = np.ascontiguousarray(cdat, dtype=np.complex64)
        soft = np.empty(144, dtype=np.float32)
        llr = np.empty(128, dtype=np.float32)
        nb = np.zeros(1, dtype=np.int32)
        self.L.orc_softbits_at(C.byref(self.ctx), _p(cdat), block_idx, pattern_idx, int(pos), _p(soft), _p(llr), _p(nb))
        return soft, llr, int(nb[0])


def ldpc_one(llr128):
    llr = np.ascontiguousarray(llr128, dtype=np.float32)
    msg = np.zeros(77, dtype=np.int8)
    it = np.zeros(1, dtype=np.int32)
    nh = np.zeros(1, dtype=np.int32)
    ok = lib().orc_ldpc_one(_p(llr), _p(msg), _p(it), _p(nh))
    return bool(ok), msg, int(it[0]), int(nh[0])


def crc13(buf: bytes) -> int:
    a = np.frombuffer(bytes(buf), dtype=np.uint8).copy()
    return int(lib().orc_crc13(_p(a), len(a)))


def check_crc_bits(cw) -> bool:
    a = np.ascontiguousarray(cw, dtype=np.int8)
    assert a.size >= 90
    return bool(lib().orc_check_crc_bits(_p(a)))


def cb42():
    a = np.empty(42, dtype=np.complex64)
    lib().orc_get_cb42(_p(a))
    return a


def pp12():
    a = np.empty(12, dtype=np.float32)
    lib().orc_get_pp12(_p(a))
    return a


def segment_power(x):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty(8, dtype=np.float32)
    lib().orc_segment_power(_p(x), len(x), _p(out))
    return out


def message_gate(msg77) -> bool:
    a = np.ascontiguousarray(msg77, dtype=np.int8)
    return bool(lib().orc_message_gate(_p(a)))


class Snr:
    def __init__(self):
        self.t = SnrTracker()
        lib().orc_snr_init(C.byref(self.t))

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        lib().orc_snr_process(C.byref(self.t), _p(x), len(x))
        return int(lib().orc_snr_int(C.byref(self.t)))
