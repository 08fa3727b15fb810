"""ctypes view of libmsk144hip.so (include/msk144hip.h), one method per ABI entry point.

No CPU fallback: if the library is missing or no HIP device is present the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
from typing import Optional

import numpy as np

from .protocol import ITEM_BYTES

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmsk144hip.so")

STAGE_SCAN, STAGE_SOFTBITS, STAGE_INDEX, STAGE_LDPC, STAGE_COLLECT, STAGE_ALL = 1, 2, 4, 8, 16, 31
T_NAMES = ("frontend", "scan", "softbits", "index", "ldpc", "collect", "h2d", "d2h")

# every symbol include/msk144hip.h declares (tests check the library exports each of them)
ABI_SYMBOLS = (
    "msk144_default_params", "msk144_create", "msk144_destroy", "msk144_last_error", "msk144_geometry", "msk144_frequency",
    "msk144_set_stream", "msk144_submit_audio", "msk144_submit_iq", "msk144_submit_audio_device", "msk144_submit_iq_device",
    "msk144_submit_analytic", "msk144_decode", "msk144_decode_stages", "msk144_synchronize", "msk144_results",
    "msk144_result_count", "msk144_results_device", "msk144_set_channel_base", "msk144_segment_power", "msk144_dump_analytic", "msk144_dump_candidates",
    "msk144_dump_indexes", "msk144_load_candidates", "msk144_set_profiling", "msk144_stage_times",
    "msk144_input_slot", "msk144_submit_slot", "msk144_submit_slot_n", "msk144_fetch_async", "msk144_fetch_wait", "msk144_hop_slot", "msk144_push_hops",
    "msk144_device_count", "msk144_clock_probe", "msk144_set_copy_handover", "msk144_copy_handover", "msk144_copy_count",
    "msk144_set_llr_retention", "msk144_llr_block_channels",
)


class Params(C.Structure):
    _fields_ = [("center_hz", C.c_float), ("width_hz", C.c_float), ("step_hz", C.c_float), ("scan_depth", C.c_int32),
                ("nbadsync_threshold", C.c_int32), ("read_mode", C.c_int32), ("analytic_method", C.c_int32), ("channels", C.c_int32),
                ("device", C.c_int32), ("max_results", C.c_int32), ("llr_block_channels", C.c_int32)]


RESULT_DTYPE = np.dtype([
    ("channel", "<i4"), ("item", "<i4"), ("f0", "<f4"), ("pattern_idx", "<i4"), ("num_avg", "<i4"), ("pos", "<u4"), ("xb", "<f4"),
    ("nbadsync", "<i4"), ("ldpc_iterations", "<i4"), ("ldpc_hard_errors", "<i4"), ("message", "u1", (10,)), ("reserved", "u1", (2,)),
])
assert RESULT_DTYPE.itemsize == 52

CANDIDATE_DTYPE = np.dtype([
    ("block_idx", "<u4"), ("pattern_idx", "<u4"), ("pos", "<u4"), ("f0", "<f4"), ("nbadsync", "<i4"), ("xb", "<f4"),
    ("num_avg", "<i4"), ("softbits_wo_sync", "<f4", (128,)), ("is_message_present", "u1"), ("_pad0", "u1", (3,)),
    ("ldpc_num_iterations", "<i4"), ("ldpc_num_hard_errors", "<i4"), ("message", "i1", (77,)), ("_pad1", "u1", (3,)),
])
assert CANDIDATE_DTYPE.itemsize == ITEM_BYTES


class Msk144Error(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"msk144hip error {code}: {text}")
        self.code = code


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libmsk144hip.so and declare prototypes.  Raises if it is absent - there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MSK144HIP_LIBRARY") or LIB_PATH      # MSK144HIP_LIBRARY: same-box A/B of two builds (tools/)
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7.  If torch were
    # imported AFTER this library, two runtimes would be live and the second finds no GPU.  Importing
    # torch first makes the dynamic linker bind libmsk144hip.so to the runtime torch already loaded.
    if "torch" not in sys.modules and not os.environ.get("MSK144_NO_TORCH_PRELOAD") and importlib.util.find_spec("torch"):
        import torch  # noqa: F401
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: build it with `python -m msk144cudecoder_amd.build` (hipcc, gfx950)")
    L = C.CDLL(p)
    vp, i32 = C.c_void_p, C.c_int32
    L.msk144_default_params.argtypes = [C.POINTER(Params)]
    L.msk144_default_params.restype = None
    L.msk144_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.msk144_destroy.argtypes = [vp]
    L.msk144_destroy.restype = None
    L.msk144_last_error.argtypes = [vp]
    L.msk144_last_error.restype = C.c_char_p
    L.msk144_geometry.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.msk144_frequency.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.msk144_set_stream.argtypes = [vp, vp]
    for n in ("msk144_submit_audio", "msk144_submit_iq", "msk144_submit_audio_device", "msk144_submit_iq_device", "msk144_submit_analytic"):
        getattr(L, n).argtypes = [vp, vp]
    L.msk144_decode.argtypes = [vp]
    L.msk144_decode_stages.argtypes = [vp, C.c_uint32]
    L.msk144_synchronize.argtypes = [vp]
    L.msk144_results.argtypes = [vp, vp, i32, C.POINTER(i32)]
    L.msk144_result_count.argtypes = [vp, C.POINTER(i32)]
    L.msk144_results_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.msk144_set_channel_base.argtypes = [vp, i32]
    L.msk144_segment_power.argtypes = [vp, vp]
    L.msk144_dump_analytic.argtypes = [vp, i32, vp]
    L.msk144_dump_candidates.argtypes = [vp, i32, vp]
    L.msk144_dump_indexes.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.msk144_load_candidates.argtypes = [vp, i32, vp]
    L.msk144_set_profiling.argtypes = [vp, i32]
    L.msk144_stage_times.argtypes = [vp, vp, vp, i32]
    L.msk144_input_slot.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.msk144_submit_slot.argtypes = [vp, i32]
    L.msk144_submit_slot_n.argtypes = [vp, i32, i32]
    L.msk144_hop_slot.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.msk144_push_hops.argtypes = [vp, i32, i32]
    L.msk144_fetch_async.argtypes = [vp, i32]
    L.msk144_fetch_wait.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp)]
    L.msk144_device_count.argtypes = [C.POINTER(i32)]
    L.msk144_clock_probe.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.msk144_set_copy_handover.argtypes = [vp, i32]
    L.msk144_set_llr_retention.argtypes = [vp, i32]
    L.msk144_llr_block_channels.argtypes = [vp, C.POINTER(i32)]
    L.msk144_copy_handover.argtypes = [vp, C.POINTER(i32)]
    L.msk144_copy_count.argtypes = [vp, C.POINTER(C.c_int64)]
    if path is None:
        _lib = L
    return L


def default_params() -> Params:
    p = Params()
    load_library().msk144_default_params(C.byref(p))
    return p


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class HipDecoder:
    """One msk144_handle: `channels` independent windows per decode on one MI355X."""

    def __init__(self, center=1500.0, width=200.0, step=2.0, depth=4, nbadsync_threshold=1, read_mode=1, analytic_method=2,
                 channels=1, device=0, max_results=0, llr_block_channels=0):
        self.L = load_library()
        p = Params(center, width, step, depth, nbadsync_threshold, read_mode, analytic_method, channels, device, max_results, llr_block_channels)
        self.params = p
        self.h = C.c_void_p()
        rc = self.L.msk144_create(C.byref(p), C.byref(self.h))
        if rc != 0:
            raise Msk144Error(rc, (self.L.msk144_last_error(None) or b"").decode())
        f, d, k = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self.L.msk144_geometry(self.h, C.byref(f), C.byref(d), C.byref(k)))
        self.F, self.D, self.K = f.value, d.value, k.value
        self.channels = channels
        self.read_mode = read_mode
        b = C.c_int32()
        self._chk(self.L.msk144_llr_block_channels(self.h, C.byref(b)))
        self.llr_block = b.value            # channels per softbits -> index -> LDPC block (the library's choice when llr_block_channels = 0)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.msk144_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise Msk144Error(rc, (self.L.msk144_last_error(self.h) or b"").decode())
        return rc

    def frequency(self, b: int) -> float:
        v = C.c_float()
        self._chk(self.L.msk144_frequency(self.h, b, C.byref(v)))
        return v.value

    def set_stream(self, hip_stream: int):
        self._chk(self.L.msk144_set_stream(self.h, C.c_void_p(hip_stream)))

    # ---- front end ----
    def submit_audio(self, windows: np.ndarray):
        w = np.ascontiguousarray(windows, dtype=np.int16).reshape(self.channels, 5184)
        self._chk(self.L.msk144_submit_audio(self.h, _ptr(w)))

    def submit_iq(self, windows: np.ndarray):
        w = np.ascontiguousarray(windows, dtype=np.int8).reshape(self.channels, 2 * 5184)
        self._chk(self.L.msk144_submit_iq(self.h, _ptr(w)))

    def submit_audio_device(self, dev_ptr: int):
        self._chk(self.L.msk144_submit_audio_device(self.h, C.c_void_p(dev_ptr)))

    def submit_iq_device(self, dev_ptr: int):
        self._chk(self.L.msk144_submit_iq_device(self.h, C.c_void_p(dev_ptr)))

    def submit_analytic(self, windows: np.ndarray):
        w = np.ascontiguousarray(windows, dtype=np.complex64).reshape(self.channels, 5184)
        self._chk(self.L.msk144_submit_analytic(self.h, _ptr(w)))

    # ---- decode ----
    def decode(self, stages: int = STAGE_ALL):
        self._chk(self.L.msk144_decode_stages(self.h, stages))

    def synchronize(self):
        self._chk(self.L.msk144_synchronize(self.h))

    def result_count(self) -> int:
        n = C.c_int32()
        self._chk(self.L.msk144_result_count(self.h, C.byref(n)))
        return n.value

    def results(self) -> np.ndarray:
        n = self.result_count()
        out = np.zeros(max(n, 1), dtype=RESULT_DTYPE)
        got = C.c_int32()
        self._chk(self.L.msk144_results(self.h, _ptr(out), len(out), C.byref(got)), allow=(-5,))
        return out[:min(n, len(out))]

    def results_device(self):
        rec, cnt = C.c_void_p(), C.c_void_p()
        self._chk(self.L.msk144_results_device(self.h, C.byref(rec), C.byref(cnt)))
        return rec.value, cnt.value

    def set_channel_base(self, base: int):
        """Result records carry channel = base + local channel (global ids for the multi-GPU gather)."""
        self._chk(self.L.msk144_set_channel_base(self.h, base))

    def set_llr_retention(self, retain: bool):
        """A one-block handle (llr_block_channels = channels) keeps every LLR row readable (dumps).  retain=False: behave like a
        blocked handle - early nbadsync gate, copies handed over, dumps refused (what msk144hipdecoder asks for)."""
        self._chk(self.L.msk144_set_llr_retention(self.h, 1 if retain else 0))

    def set_copy_handover(self, on: bool):
        """Blocked staging only: whether a slot that folds the same frames as a lower slot of its group reports that slot's result
        (default) or is demodulated and decoded on its own, as the reference does."""
        self._chk(self.L.msk144_set_copy_handover(self.h, 1 if on else 0))

    def copy_handover(self) -> bool:
        v = C.c_int32()
        self._chk(self.L.msk144_copy_handover(self.h, C.byref(v)))
        return bool(v.value)

    def copy_count(self) -> int:
        """Slots of the last decode that were handed to a lower slot of their group."""
        v = C.c_int64()
        self._chk(self.L.msk144_copy_count(self.h, C.byref(v)))
        return int(v.value)

    def segment_power(self) -> np.ndarray:
        out = np.empty((self.channels, 8), dtype=np.float32)
        self._chk(self.L.msk144_segment_power(self.h, _ptr(out)))
        return out

    # ---- pinned staging slots (pipelined hops) ----
    def input_slot(self, slot: int) -> np.ndarray:
        """The slot's pinned window buffer as a numpy view: int16 [channels][5184] or int8 [channels][2*5184]."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        self._chk(self.L.msk144_input_slot(self.h, slot, C.byref(ptr), C.byref(nbytes)))
        raw = (C.c_uint8 * nbytes.value).from_address(ptr.value)
        a = np.frombuffer(raw, dtype=np.int8 if self.read_mode == 2 else np.int16)
        return a.reshape(self.channels, -1)

    def submit_slot(self, slot: int, n_channels: int = 0):
        """n_channels > 0: the hop covers only the first n_channels windows of the slot."""
        self._chk(self.L.msk144_submit_slot_n(self.h, slot, n_channels) if n_channels else self.L.msk144_submit_slot(self.h, slot))

    def hop_slot(self, slot: int):
        """(hops, first_halves, streams, is_first): numpy views of the slot's pinned hop-ring inputs - hops and first_halves as
        [channels][2592] int16 or [channels][2*2592] int8, streams int32[channels], is_first uint8[channels]."""
        p = [C.c_void_p() for _ in range(4)]
        self._chk(self.L.msk144_hop_slot(self.h, slot, *[C.byref(x) for x in p]))
        half = 5184  # bytes of half a window in both read modes
        dt = np.int8 if self.read_mode == 2 else np.int16
        hops = np.frombuffer((C.c_uint8 * (half * self.channels)).from_address(p[0].value), dtype=dt).reshape(self.channels, -1)
        first = np.frombuffer((C.c_uint8 * (half * self.channels)).from_address(p[1].value), dtype=dt).reshape(self.channels, -1)
        streams = np.frombuffer((C.c_int32 * self.channels).from_address(p[2].value), dtype=np.int32)
        is_first = np.frombuffer((C.c_uint8 * self.channels).from_address(p[3].value), dtype=np.uint8)
        return hops, first, streams, is_first

    def push_hops(self, slot: int, n: int):
        self._chk(self.L.msk144_push_hops(self.h, slot, n))

    def fetch_async(self, slot: int):
        self._chk(self.L.msk144_fetch_async(self.h, slot))

    def fetch_wait(self, slot: int):
        """(records, segment powers [channels][8]) of the slot, copied out of its pinned output."""
        rec, seg, n = C.c_void_p(), C.c_void_p(), C.c_int32()
        self._chk(self.L.msk144_fetch_wait(self.h, slot, C.byref(rec), C.byref(n), C.byref(seg)))
        records = np.frombuffer((C.c_uint8 * (n.value * RESULT_DTYPE.itemsize)).from_address(rec.value), dtype=RESULT_DTYPE).copy() if n.value else np.zeros(0, dtype=RESULT_DTYPE)
        powers = np.frombuffer((C.c_float * (8 * self.channels)).from_address(seg.value), dtype=np.float32).reshape(self.channels, 8).copy()
        return records, powers

    # ---- parity / debug ----
    def dump_analytic(self, channel=0) -> np.ndarray:
        out = np.empty(5184, dtype=np.complex64)
        self._chk(self.L.msk144_dump_analytic(self.h, channel, _ptr(out)))
        return out

    def dump_candidates(self, channel=0) -> np.ndarray:
        out = np.zeros(self.K, dtype=CANDIDATE_DTYPE)
        self._chk(self.L.msk144_dump_candidates(self.h, channel, _ptr(out)))
        return out

    def dump_indexes(self, channel=0) -> np.ndarray:
        out = np.empty(self.K, dtype=np.int32)
        n = C.c_int32()
        self._chk(self.L.msk144_dump_indexes(self.h, channel, _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def load_candidates(self, items: np.ndarray, channel=0):
        a = np.ascontiguousarray(items)
        assert a.dtype.itemsize == ITEM_BYTES and len(a) == self.K
        self._chk(self.L.msk144_load_candidates(self.h, channel, _ptr(a)))

    def set_profiling(self, on: bool):
        self._chk(self.L.msk144_set_profiling(self.h, 1 if on else 0))

    def stage_times(self, reset=False):
        """{stage: (avg_ms, launches)} measured with HIP events on the decode stream."""
        ms = np.zeros(len(T_NAMES), dtype=np.float32)
        cnt = np.zeros(len(T_NAMES), dtype=np.int32)
        self._chk(self.L.msk144_stage_times(self.h, _ptr(ms), _ptr(cnt), 1 if reset else 0))
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(T_NAMES)}

    def clock_probe(self, spin_us: int = 1000) -> float:
        """Shader clock in MHz read by a one-wave kernel beside whatever the decode stream is running (blocks ~spin_us)."""
        mhz = C.c_float(0.0)
        self._chk(self.L.msk144_clock_probe(self.h, int(spin_us), C.byref(mhz)))
        return float(mhz.value)


def device_count() -> int:
    """HIP devices the library sees (raises without one: there is no CPU fallback)."""
    n = C.c_int32(0)
    L = load_library()
    rc = L.msk144_device_count(C.byref(n))
    if rc != 0:
        raise Msk144Error(rc, (L.msk144_last_error(None) or b"").decode())
    return int(n.value)


def unpack_message(msg10: np.ndarray) -> np.ndarray:
    """10 packed bytes (MSB first) -> 77 bits."""
    return np.unpackbits(np.asarray(msg10, dtype=np.uint8))[:77]
