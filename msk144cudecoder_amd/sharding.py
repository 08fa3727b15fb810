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

__all__ = ["shard_channels", "pack_records", "unpack_records", "gather_records"]


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
