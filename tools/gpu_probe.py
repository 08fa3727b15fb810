"""Exploratory GPU check (developer tool): stage-by-stage comparison against the oracle + timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from msk144cudecoder_amd import synth
from msk144cudecoder_amd.hipdecoder import HipDecoder, unpack_message
from oracle import oracle as orc

rng = np.random.default_rng(3)
m = synth.random_message(rng)
p = synth.Ping(m, start=700, n_frames=7, freq_hz=1506.0, snr_db=3.0, phase=1.0)
x = synth.synth_audio(5184, [p], 1000.0, rng)

cfg = dict(center=1500.0, width=20.0, step=2.0, depth=6, nbadsync_threshold=2)
o = orc.Oracle(threads=8, **cfg)
d = HipDecoder(channels=1, **cfg)
print("geometry", d.F, d.D, d.K, o.F, o.D, o.total_items)

# front end
cd_o = o.frontend_audio(x, 2)
d.submit_audio(x)
cd_g = d.dump_analytic(0)
print("frontend max abs diff", np.abs(cd_o - cd_g).max(), "bit-exact:", np.array_equal(cd_o.view(np.uint32), cd_g.view(np.uint32)))
print("seg power", d.segment_power()[0], orc.segment_power(cd_o))

items_o, idx_o = o.decode_window(cd_o)
d.decode()
items_g = d.dump_candidates(0)
idx_g = d.dump_indexes(0)
print("pos equal:", (items_o['pos'] == items_g['pos']).mean(), "xb rel diff max:", np.max(np.abs(items_o['xb'] - items_g['xb']) / np.maximum(items_o['xb'], 1e-9)))
same = items_o['pos'] == items_g['pos']
llr_d = np.abs(items_o['softbits_wo_sync'][same] - items_g['softbits_wo_sync'][same])
print("llr max abs diff (same pos):", llr_d.max(), "nbadsync equal:", (items_o['nbadsync'][same] == items_g['nbadsync'][same]).mean())
print("idx equal:", np.array_equal(idx_o, idx_g), len(idx_o), len(idx_g))
print("present equal:", (items_o['is_message_present'] == items_g['is_message_present']).mean(), items_o['is_message_present'].sum(), items_g['is_message_present'].sum())
pres = (items_o['is_message_present'] == 1) & (items_g['is_message_present'] == 1)
print("messages equal:", (items_o['message'][pres] == items_g['message'][pres]).all(), "iters equal:", (items_o['ldpc_num_iterations'][pres] == items_g['ldpc_num_iterations'][pres]).mean(),
      "nhard equal:", (items_o['ldpc_num_hard_errors'][pres] == items_g['ldpc_num_hard_errors'][pres]).mean())
res = d.results()
print("results:", len(res), "all match tx:", all((unpack_message(r['message']) == m).all() for r in res))
print(res[:3])

# fft front end
d1 = HipDecoder(channels=1, analytic_method=1, **cfg)
d1.submit_audio(x)
g1 = d1.dump_analytic(0)
o1 = o.frontend_audio(x, 1)
print("fft frontend rel err:", np.abs(g1 - o1).max() / np.sqrt(np.mean(np.abs(o1) ** 2)))

# timing: deep config, 64 channels
nch = int(os.environ.get("PROBE_CH", "64"))
dd = HipDecoder(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3, channels=nch)
xs = np.stack([synth.stream_s2(i, seconds=0.432)[0][:5184] for i in range(nch)])
dd.set_profiling(True)
for it in range(3):
    dd.submit_audio(xs)
    dd.decode()
    dd.synchronize()
dd.stage_times(reset=True)
t = time.time()
for it in range(3):
    dd.submit_audio(xs)
    dd.decode()
dd.synchronize()
el = (time.time() - t) / 3
print("deep config", nch, "channels: %.2f ms/step" % (el * 1e3), "cand/s %.3e" % (nch * dd.K / el))
print(dd.stage_times())
print("n results", dd.result_count())
