"""Channel sharding across GPUs and the one exchange step of the path.

Channels are independent: rank r of N decodes a contiguous block of channels on its own
GPU; the only communication is a gather of fixed-size decoded-record buffers to rank 0 after each
batch step (torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
The record payload is latency-sized (KBs), far from the per-link xGMI bandwidth.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .hipdecoder import RESULT_DTYPE

__all__ = ["shard_channels", "pack_records", "unpack_records", "gather_records", "gather_capacity", "RecordGather"]


def shard_channels(n_channels: int, rank: int, world: int) -> Tuple[int, int]:
    """(first channel, number of channels) of `rank`: contiguous blocks, sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world) or n_channels < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_channels, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def pack_records(records: np.ndarray, cap: int, channel_offset: int = 0) -> np.ndarray:
    """Fixed-size send buffer: uint8[cap*52 + 8] = records (local channel ids made global) | count | total.
    `total` may exceed `cap` (overflow is visible to the receiver)."""
    rec = np.asarray(records, dtype=RESULT_DTYPE)
    buf = np.zeros(cap * RESULT_DTYPE.itemsize + 8, dtype=np.uint8)
    n = min(len(rec), cap)
    if n:
        r = rec[:n].copy()
        r["channel"] += channel_offset
        buf[:n * RESULT_DTYPE.itemsize] = r.view(np.uint8).reshape(-1)
    buf[-8:-4] = np.array([n], dtype="<i4").view(np.uint8)
    buf[-4:] = np.array([len(rec)], dtype="<i4").view(np.uint8)
    return buf


def unpack_records(buf: np.ndarray) -> Tuple[np.ndarray, int]:
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n = int(buf[-8:-4].view("<i4")[0])
    total = int(buf[-4:].view("<i4")[0])
    rec = buf[:n * RESULT_DTYPE.itemsize].view(RESULT_DTYPE).copy()
    return rec, total


def gather_records(records: np.ndarray, cap: int, channel_offset: int, device: Optional[str] = None) -> Optional[List[np.ndarray]]:
    """Gather every rank's decoded records on rank 0 (ordered by rank, i.e. by global channel).
    Returns the per-rank record arrays on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    send = torch.from_numpy(pack_records(records, cap, channel_offset))
    if device is not None:
        send = send.to(device)
    world = dist.get_world_size()
    rank = dist.get_rank()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    out = []
    for t in recv:
        rec, total = unpack_records(t.cpu().numpy())
        if total > len(rec):
            raise OverflowError(f"a rank decoded {total} records, more than the gather capacity {cap}")
        out.append(rec)
    return out


def gather_capacity(channels: int) -> int:
    """Records gathered per rank and step: 32 per channel (a ping is decoded by several of its candidates:
    the 1024-channel bench sees ~7 records per channel and step), at least 1024."""
    return max(1024, 32 * int(channels))


class RecordGather:
    """The per-step exchange on DEVICE buffers: every rank sends uint8[cap*52 + 8] = records | n | total to
    rank 0, the same layout as pack_records (records already carry global channel ids:
    msk144_set_channel_base).  Works on any torch.distributed backend (nccl = RCCL on the GPUs, gloo in the
    CPU tests).  Nothing is read back on the host inside step(); rank 0 keeps a running maximum of every
    rank's `total` on the device and finish() validates it against the capacity."""

    def __init__(self, cap: int, device, world: int, rank: int):
        import torch
        self.cap, self.world, self.rank = int(cap), int(world), int(rank)
        self.nbytes = self.cap * RESULT_DTYPE.itemsize
        self.send = torch.zeros(self.nbytes + 8, dtype=torch.uint8, device=device)
        self._trailer = self.send[self.nbytes:].view(torch.int32)                # [n, total]
        self.recv = [torch.zeros_like(self.send) for _ in range(world)] if rank == 0 else None
        self._recv_trailers = [t[self.nbytes:].view(torch.int32) for t in self.recv] if rank == 0 else None
        self.max_total = torch.zeros(world, dtype=torch.int32, device=device) if rank == 0 else None
        self.steps = 0
        self._spans = []          # timed steps: (start, end) device events, or host seconds on a CPU backend

    def step(self, rec_bytes, count_i32, timed: bool = False):
        """rec_bytes: uint8 view of the decoder's device record list (>= cap*52 bytes); count_i32: int32[1]
        view of its device-side record count.  timed: bracket this exchange (copy into the send buffer + gather) with events on
        the current stream - the collective's own stream is joined to it by the blocking dist.gather - read later by mean_ms()."""
        import time

        import torch
        import torch.distributed as dist
        on_gpu = self.send.is_cuda
        if timed:
            if on_gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
        self._step(rec_bytes, count_i32, torch, dist)
        if timed:
            if on_gpu:
                e1.record()
                self._spans.append((e0, e1))
            else:
                self._spans.append((t0, time.perf_counter()))

    def mean_ms(self) -> Optional[float]:
        """Average duration of the timed exchanges on this rank (call after the stream has been synchronised)."""
        if not self._spans:
            return None
        if self.send.is_cuda:
            return float(sum(a.elapsed_time(b) for a, b in self._spans) / len(self._spans))
        return float(sum(b - a for a, b in self._spans) / len(self._spans) * 1e3)

    def _step(self, rec_bytes, count_i32, torch, dist):
        self.send[:self.nbytes].copy_(rec_bytes[:self.nbytes], non_blocking=True)
        self._trailer[1:2].copy_(count_i32, non_blocking=True)
        torch.clamp(count_i32, max=self.cap, out=self._trailer[0:1])
        dist.gather(self.send, self.recv, dst=0)
        if self.rank == 0:
            tot = torch.stack([t[1] for t in self._recv_trailers])
            torch.maximum(self.max_total, tot, out=self.max_total)
        self.steps += 1

    def finish(self) -> Optional[List[np.ndarray]]:
        """Rank 0: per-rank record arrays of the last step; raises OverflowError if any rank ever produced
        more records than the capacity.  Other ranks: None."""
        if self.rank != 0:
            return None
        worst = self.max_total.cpu().numpy()
        if (worst > self.cap).any():
            raise OverflowError(f"decoded records per rank and step peaked at {worst.tolist()}, gather capacity is {self.cap}")
        out = []
        for t in self.recv:
            rec, total = unpack_records(t.cpu().numpy())
            assert total == len(rec)
            out.append(rec)
        return out
