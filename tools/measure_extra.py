"""Developer tool: single-window latency (BASELINE configs[1] shape) and the 4096-channel IQ config (configs[4])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from msk144cudecoder_amd import synth
from msk144cudecoder_amd.hipdecoder import HipDecoder

# --- latency: one channel, deep config, host buffer in -> result count out ---
x, _ = synth.stream_s1(0, seconds=1.0)
w = x[:5184]
with HipDecoder(center=1500.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3, channels=1) as d:
    for _ in range(5):
        d.submit_audio(w); d.decode(); d.result_count()
    t = []
    for _ in range(50):
        t0 = time.perf_counter(); d.submit_audio(w); d.decode(); n = d.result_count(); t.append(time.perf_counter() - t0)
    d.set_profiling(True)
    for _ in range(20):
        d.submit_audio(w); d.decode(); d.synchronize()
    print("single window (24048 candidates): median %.3f ms host wall, min %.3f ms; stages: %s" % (np.median(t) * 1e3, np.min(t) * 1e3, {k: round(v[0], 4) for k, v in d.stage_times().items()}))

# --- configs[4]: IQ, 4096 low-SNR channels, depth 6, threshold 3, width 500 step 1 ---
nch = int(os.environ.get("IQ_CH", "4096"))
wins, truth = synth.iq_low_snr_batch(nch, 5)
dev = torch.from_numpy(wins).cuda()
with HipDecoder(center=0.0, width=500.0, step=1.0, depth=6, nbadsync_threshold=3, read_mode=2, channels=nch, max_results=1 << 20) as d:
    d.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        d.submit_iq_device(dev.data_ptr()); d.decode()
    torch.cuda.synchronize()
    d.set_profiling(True)
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        d.submit_iq_device(dev.data_ptr()); d.decode()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    res = d.results()
    good = sum(1 for r in res if truth.get(int(r["channel"])) == bytes(r["message"]))
    print("IQ %d channels: %.1f ms/step, %.3e candidates/s; stages %s; decodes %d (matching tx %d, channels %d of %d pinged)" % (
        nch, el * 1e3, nch * d.K / el, {k: round(v[0], 3) for k, v in d.stage_times().items()}, len(res), good,
        len({int(r["channel"]) for r in res if truth.get(int(r["channel"])) == bytes(r["message"])}), len(truth)))
