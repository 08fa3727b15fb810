"""The C-ABI library: loads, exports every symbol include/msk144hip.h declares, struct sizes match,
and it fails loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "msk144hip.h")


@pytest.fixture(scope="module")
def lib():
    from msk144cudecoder_amd import hipdecoder
    if not os.path.exists(hipdecoder.LIB_PATH):
        from msk144cudecoder_amd import build
        build.build_library()
    return hipdecoder.load_library(), hipdecoder


def _declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(msk144_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    L, hd = lib
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert set(declared) == set(hd.ABI_SYMBOLS)
    for s in declared:
        assert getattr(L, s) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", hd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (msk144_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_struct_layouts(lib):
    _, hd = lib
    assert C.sizeof(hd.Params) == 44
    assert hd.RESULT_DTYPE.itemsize == 52
    assert hd.CANDIDATE_DTYPE.itemsize == 632
    offs = {n: hd.CANDIDATE_DTYPE.fields[n][1] for n in hd.CANDIDATE_DTYPE.names}
    # offsets of the reference's ResultItem
    assert (offs["block_idx"], offs["pattern_idx"], offs["pos"], offs["f0"], offs["nbadsync"], offs["xb"], offs["num_avg"]) == (0, 4, 8, 12, 16, 20, 24)
    assert (offs["softbits_wo_sync"], offs["is_message_present"], offs["ldpc_num_iterations"], offs["ldpc_num_hard_errors"], offs["message"]) == (28, 540, 544, 548, 552)


def test_defaults_are_the_reference_code_defaults(lib):
    _, hd = lib
    p = hd.default_params()
    assert (p.center_hz, p.width_hz, p.step_hz, p.scan_depth, p.nbadsync_threshold, p.read_mode, p.analytic_method, p.channels) == \
        (1500.0, 200.0, 2.0, 4, 1, 1, 2, 1)      # main.cu:124-133, not the help text (100 / 3 / 2)


def test_create_validates_and_fails_loudly_without_gpu(lib):
    L, hd = lib
    h = C.c_void_p()
    bad = hd.default_params()
    bad.step_hz = 0.0
    assert L.msk144_create(C.byref(bad), C.byref(h)) == -1 and not h.value
    assert b"step" in L.msk144_last_error(None)
    bad = hd.default_params()
    bad.read_mode = 7
    assert L.msk144_create(C.byref(bad), C.byref(h)) == -1
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(hd.Msk144Error) as e:
            hd.HipDecoder()
        assert e.value.code == -2 and "no CPU fallback" in str(e.value)
