"""Golden vectors from the REFERENCE's own compiled host classes -> tests/golden/ref_host_fixtures.json.

BUILD CONTAINER ONLY.  Runs oracle/_ref/libmsk144_ref_host.so (result_filter.cpp and snr_tracker.cu of /root/reference/src,
compiled unmodified by oracle/ref/Makefile) on the seeded inputs of tests/ref_host.py and records its outputs.  The JSON
holds inputs' seeds and expected outputs only (data), so the pins hold wherever the reference is absent.

    make -C oracle/ref && python tests/golden/make_ref_host_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_host as rh  # noqa: E402


def main():
    L = rh.load_ref()
    out = {"_source": "oracle/_ref/libmsk144_ref_host.so = /root/reference/src/result_filter.cpp (g++) + snr_tracker.cu (clang host-only), unmodified",
           "filter": {"seed": 20241008, "n_cases": 60, "expected": []}, "snr": {"seed": 77, "n_seq": 6, "n_win": 12, "expected_int": [], "expected_float_hex": []}}
    for items in rh.filter_cases(20241008, 60):
        out["filter"]["expected"].append(rh.ref_filter_window(L, items))
    for wins in rh.snr_sequences(77, 6, 12):
        t = L.ref_snr_new()
        ints, hexes = [], []
        for w in wins:
            iq = np.ascontiguousarray(w).view(np.float32)
            ints.append(int(L.ref_snr_process(t, iq.ctypes.data, len(w))))
            hexes.append(float(L.ref_snr_float(t)).hex())
        L.ref_snr_free(t)
        out["snr"]["expected_int"].append(ints)
        out["snr"]["expected_float_hex"].append(hexes)
    n = rh.C.c_int()
    p = L.ref_ldpc_reverse_map(rh.C.byref(n))
    out["ldpc_reverse_map_compiled"] = [int(p[i]) for i in range(n.value)]
    with open(os.path.join(HERE, "ref_host_fixtures.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote ref_host_fixtures.json:", len(out["filter"]["expected"]), "filter windows,", sum(len(s) for s in out["snr"]["expected_int"]), "snr windows")


if __name__ == "__main__":
    main()
