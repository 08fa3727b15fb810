"""ctypes views of (a) oracle/_ref/libmsk144_ref_host.so - the REFERENCE's own result_filter.cpp and snr_tracker.cu compiled
unmodified (oracle/ref/Makefile; build container only, the prebuilt .so travels) - and (b) the product's host library.
Test infrastructure."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmsk144_ref_host.so")
HOST_SO = os.path.join(ROOT, "msk144cudecoder_amd", "libmsk144host.so")


def ref_available() -> bool:
    return os.path.exists(REF_SO)


def load_ref():
    L = C.CDLL(REF_SO)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.ref_filter_new.restype = vp
    L.ref_filter_free.argtypes = [vp]
    L.ref_filter_block_begin.argtypes = [vp]
    L.ref_filter_block_end.argtypes = [vp]
    L.ref_filter_put.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.ref_filter_count.argtypes = [vp]
    L.ref_filter_get.argtypes = [vp, C.c_int, ip, fp, ip, ip, ip, C.c_char_p, C.c_int]
    L.ref_snr_new.restype = vp
    L.ref_snr_free.argtypes = [vp]
    L.ref_snr_process.argtypes = [vp, vp, C.c_uint]
    L.ref_snr_float.argtypes = [vp]
    L.ref_snr_float.restype = C.c_float
    L.ref_ldpc_reverse_map.argtypes = [ip]
    L.ref_ldpc_reverse_map.restype = C.POINTER(C.c_byte)
    return L


class Filtered(C.Structure):
    _fields_ = [("snr", C.c_int), ("f0", C.c_float), ("num_avg", C.c_int), ("nbadsync", C.c_int), ("pattern_idx", C.c_int), ("text", C.c_char * 64)]


def load_host():
    L = C.CDLL(HOST_SO)
    vp = C.c_void_p
    L.msk144host_filter_new.restype = vp
    L.msk144host_filter_free.argtypes = [vp]
    L.msk144host_filter_begin.argtypes = [vp]
    L.msk144host_filter_put.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.msk144host_filter_end.argtypes = [vp, C.POINTER(Filtered), C.c_int]
    L.msk144host_snr_new.restype = vp
    L.msk144host_snr_free.argtypes = [vp]
    L.msk144host_snr_update.argtypes = [vp, C.POINTER(C.c_float)]
    L.msk144host_snr_db.argtypes = [vp]
    L.msk144host_snr_db.restype = C.c_float
    return L


def ref_filter_window(L, items):
    """items: [(snr, f0, num_avg, nbadsync, pattern_idx, text)] in arrival order -> the reference's block result."""
    f = L.ref_filter_new()
    L.ref_filter_block_begin(f)
    for snr, f0, na, nb, pi, text in items:
        L.ref_filter_put(f, snr, f0, na, nb, pi, text.encode())
    L.ref_filter_block_end(f)
    out = []
    for i in range(L.ref_filter_count(f)):
        snr, na, nb, pi = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        f0 = C.c_float()
        buf = C.create_string_buffer(64)
        stamp_len = L.ref_filter_get(f, i, C.byref(snr), C.byref(f0), C.byref(na), C.byref(nb), C.byref(pi), buf, 64)
        assert stamp_len == 14
        out.append([snr.value, f0.value, na.value, nb.value, pi.value, buf.value.decode()])
    L.ref_filter_free(f)
    return out


def host_filter_window(L, items):
    f = L.msk144host_filter_new()
    L.msk144host_filter_begin(f)
    for snr, f0, na, nb, pi, text in items:
        L.msk144host_filter_put(f, snr, f0, na, nb, pi, text.encode())
    arr = (Filtered * 256)()
    n = L.msk144host_filter_end(f, arr, 256)
    out = [[arr[i].snr, arr[i].f0, arr[i].num_avg, arr[i].nbadsync, arr[i].pattern_idx, arr[i].text.decode()] for i in range(n)]
    L.msk144host_filter_free(f)
    return out


def segment_powers(win: np.ndarray) -> np.ndarray:
    """8 segment powers of a complex64 window with the reference's float32 accumulation order (snr_tracker.cu:23-31):
    arr[pos] += re*re - (-im)*im, sequentially - what the GPU front end hands to the host (msk144_segment_power)."""
    w = np.asarray(win, dtype=np.complex64)
    re, im = w.real.astype(np.float32), w.imag.astype(np.float32)
    y = (re * re - (-im) * im).astype(np.float32)
    bs = len(w) // 8
    return np.array([np.cumsum(y[k * bs:(k + 1) * bs], dtype=np.float32)[-1] for k in range(8)], dtype=np.float32)


def filter_cases(seed=20241008, n_cases=60):
    """Arrival sequences for the per-window filter: few distinct texts, many duplicates, heavy exact ties (one ping decoded by
    dozens of candidates with equal num_avg/nbadsync), group sizes on both sides of std::sort's 16-element insertion-sort
    threshold."""
    rng = np.random.default_rng(seed)
    texts = ["CQ K1ABC FN42", "K1ABC W9XYZ EN37", "W9XYZ K1ABC -11", "K1ABC W9XYZ R-09", "W9XYZ K1ABC RRR", "K1ABC W9XYZ 73", "TNX BOB 73 GL", "<...> W9XYZ R-12", "CQ DX PJ4/K1ABC", ""]
    cases = []
    for c in range(n_cases):
        n_texts = int(rng.integers(1, 6))
        chosen = list(rng.choice(len(texts), size=n_texts, replace=False))
        n_items = int(rng.choice([1, 3, 9, 17, 40, 120]))
        narrow = bool(rng.integers(0, 2))            # narrow value ranges -> many exact ties
        items = []
        for _ in range(n_items):
            t = texts[chosen[int(rng.integers(0, n_texts))]]
            na = int(rng.integers(1, 3 if narrow else 7))
            nb = int(rng.integers(0, 2 if narrow else 5))
            items.append([int(rng.integers(-8, 25)), float(np.float32(1250 + int(rng.integers(0, 501)))), na, nb, int(rng.integers(0, 8)), t])
        cases.append(items)
    return cases


def snr_sequences(seed=77, n_seq=6, n_win=12):
    """Window sequences for the SNR tracker: complex noise at slowly and abruptly changing levels with occasional bursts;
    plus an all-zero window (NaN path) at the end of the last sequence."""
    rng = np.random.default_rng(seed)
    seqs = []
    for s in range(n_seq):
        wins = []
        level = 1.0
        for w in range(n_win):
            level *= float(rng.choice([1.0, 1.0, 1.3, 0.5, 4.0, 0.9]))
            x = (rng.normal(0, level, 5184) + 1j * rng.normal(0, level, 5184)).astype(np.complex64)
            if rng.integers(0, 3) == 0:
                a = int(rng.integers(0, 4500))
                x[a:a + 600] *= np.float32(rng.uniform(2.0, 30.0))
            wins.append(x)
        if s == n_seq - 1:
            wins.append(np.zeros(5184, dtype=np.complex64))
            wins.append((rng.normal(0, 1, 5184) + 1j * rng.normal(0, 1, 5184)).astype(np.complex64))
        seqs.append(wins)
    return seqs
