"""Developer tool: find decodes on un-pinged bench channels and check them against the oracle (deep config)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from msk144cudecoder_amd.hipdecoder import HipDecoder
from oracle import oracle as orc

wins, truth = bench.make_inputs(0, 1024)
cfg = dict(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3)
d = HipDecoder(channels=1024, **cfg)
o = orc.Oracle(threads=16, **cfg)
for t in range(4):
    d.submit_audio(wins[t]); d.decode()
    res = d.results()
    bad = [r for r in res if truth.get(int(r['channel'])) != bytes(r['message'])]
    print("window", t, "decodes", len(res), "unexpected", len(bad))
    for r in bad[:3]:
        ch = int(r['channel'])
        print("  gpu:", ch, r['item'], r['f0'], r['pattern_idx'], r['pos'], r['nbadsync'], r['ldpc_iterations'], r['ldpc_hard_errors'], bytes(r['message']).hex())
        cd = o.frontend_audio(wins[t, ch], 2)
        items, idx = o.decode_window(cd)
        pres = np.nonzero(items['is_message_present'])[0]
        print("  oracle decodes on that channel:", len(pres))
        for k in pres[:5]:
            it = items[k]
            print("   ", k, it['f0'], it['pattern_idx'], it['pos'], it['nbadsync'], it['ldpc_num_iterations'], it['ldpc_num_hard_errors'], np.packbits(np.concatenate([it['message'].astype(np.uint8), np.zeros(3, np.uint8)])).tobytes().hex())
        ig = d.dump_candidates(ch)
        k = int(r['item'])
        print("   oracle item at same k:", items[k]['pos'], items[k]['nbadsync'], items[k]['is_message_present'], "gpu:", ig[k]['pos'], ig[k]['nbadsync'], "llr maxdiff", np.abs(items[k]['softbits_wo_sync'] - ig[k]['softbits_wo_sync']).max())
