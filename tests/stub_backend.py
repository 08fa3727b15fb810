"""TEST-ONLY stand-in for bench.HipBackend: lets the N-rank launcher, the gloo process group and the device-buffer
gather of bench.py run on a machine without a GPU.  It decodes nothing: every step it publishes a fixed, rank-dependent
record list in the same device-side layout the HIP decoder exposes (records | int32 count).  bench.py labels any line
produced through it ("backend": "stub", "data": "stub (no decoding, test only)")."""
import numpy as np
import torch

from msk144cudecoder_amd.hipdecoder import RESULT_DTYPE, T_NAMES


class Backend:
    name = "stub"
    dist_backend = "gloo"
    data = "stub (no decoding, test only)"

    def __init__(self, rank, local_rank, channels, channel_base, llr_block=0):
        self.T_NAMES = T_NAMES
        self.device = torch.device("cpu")
        self.F, self.D, self.K = 3, 2, 48
        self.cand_per_step = channels * self.K
        self.channel_base = channel_base
        self.truth = {}
        import os
        if os.environ.get("MSK144_STUB_REAL_INPUTS"):
            # rehearsal of the real ranks' start-up cost on the CPU: every rank synthesises its own 1024-channel input set, as
            # bench.HipBackend does before it stages it in HBM
            import bench
            self.wins_host, self.truth = bench.make_inputs(rank, channels)
            self.truth = {}
        else:
            self.wins_host = np.zeros((1, channels, 5184), dtype=np.int16)
        n = min(channels, 5 + rank)
        rec = np.zeros(n, dtype=RESULT_DTYPE)
        rec["channel"] = channel_base + np.arange(n)
        rec["item"] = 1000 * rank + np.arange(n)
        rec["message"][:, 0] = 0xA0 + rank
        self._rec = rec
        buf = np.zeros(max(channels * 32, 1024) * RESULT_DTYPE.itemsize, dtype=np.uint8)
        buf[:rec.nbytes] = rec.view(np.uint8)
        self.rec_view = torch.from_numpy(buf)
        # MSK144_STUB_OVERFLOW: rank 1 claims more records than any capacity (tests the overflow check)
        import os
        claimed = 10 ** 6 if (os.environ.get("MSK144_STUB_OVERFLOW") and rank == 1) else n
        self.cnt_view = torch.tensor([claimed], dtype=torch.int32)
        self.steps = 0

    def step(self, i):
        self.steps += 1

    def fence(self):
        pass

    def start_profiling(self):
        pass

    def stage_times(self):
        return {n: (1.0 if n == "ldpc" else 0.5, self.steps) for n in T_NAMES}

    def results(self):
        return self._rec.copy()

    class _Mark:
        def __init__(self):
            import time
            self.t = time.perf_counter()

        def synchronize(self):
            pass

    def marker(self):
        return Backend._Mark()

    @staticmethod
    def ms_between(a, b):
        return (b.t - a.t) * 1e3

    def close(self):
        pass
