import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from msk144cudecoder_amd import synth
from msk144cudecoder_amd.hipdecoder import HipDecoder
from oracle import oracle as orc
rng = np.random.default_rng(3)
m = synth.random_message(rng)
p = synth.Ping(m, start=700, n_frames=7, freq_hz=1506.0, snr_db=3.0, phase=1.0)
x = synth.synth_audio(5184, [p], 1000.0, rng)
cfg = dict(center=1500.0, width=20.0, step=2.0, depth=8, nbadsync_threshold=2)
o = orc.Oracle(threads=8, **cfg); d = HipDecoder(channels=1, **cfg)
cd = o.frontend_audio(x, 2)
d.submit_audio(x); d.decode()
io, _ = o.decode_window(cd); ig = d.dump_candidates(0)
bad = np.nonzero(io['pos'] != ig['pos'])[0]
print(len(bad), "mismatches of", len(io))
for k in bad:
    b, pi = io['block_idx'][k], io['pattern_idx'][k]
    xbo = o.scan_xb(cd, int(b), int(pi))
    print(k, "b", b, "p", pi, "slot", k % 8, "oracle pos", io['pos'][k], io['pos'][k] % 864, "xb", io['xb'][k], "| gpu pos", ig['pos'][k], ig['pos'][k] % 864, "xb", ig['xb'][k], "| oracle xb at gpu pos", xbo[ig['pos'][k]])
# per pattern multiset comparison
for pi in range(d.D):
    sel = io['pattern_idx'] == pi
    print("pattern", pi, "pos equal frac", (io['pos'][sel] == ig['pos'][sel]).mean())
