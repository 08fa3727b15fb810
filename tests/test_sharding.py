"""N > 1 path on the CPU: channel sharding + the gather of decoded records, world_size 2, gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_channels_partition():
    from msk144cudecoder_amd.sharding import shard_channels
    for n in (0, 1, 7, 8, 1024, 8191, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_channels(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert spans[-1][0] + spans[-1][1] == n
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_channels(8192, 3, 8) == (3072, 1024)
    with pytest.raises(ValueError):
        shard_channels(8, 2, 2)


def test_pack_unpack_records():
    from msk144cudecoder_amd.hipdecoder import RESULT_DTYPE
    from msk144cudecoder_amd.sharding import pack_records, unpack_records
    rec = np.zeros(5, dtype=RESULT_DTYPE)
    rec["channel"] = np.arange(5)
    rec["message"][:, 0] = 0xAB
    buf = pack_records(rec, cap=8, channel_offset=100)
    assert buf.size == 8 * 52 + 8
    got, total = unpack_records(buf)
    assert total == 5 and np.array_equal(got["channel"], np.arange(100, 105)) and (got["message"][:, 0] == 0xAB).all()
    got, total = unpack_records(pack_records(rec, cap=3))
    assert len(got) == 3 and total == 5                                # overflow is visible


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from msk144cudecoder_amd.hipdecoder import RESULT_DTYPE
    from msk144cudecoder_amd.sharding import gather_records, shard_channels
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, count = shard_channels(10, rank, world)
        # stand-in for this rank's decoder output: one record per local channel (channel ids are LOCAL)
        rec = np.zeros(count, dtype=RESULT_DTYPE)
        rec["channel"] = np.arange(count)
        rec["item"] = 1000 * rank + np.arange(count)
        out = gather_records(rec, cap=16, channel_offset=start)
        if rank == 0:
            merged = np.concatenate(out)
            q.put((merged["channel"].tolist(), merged["item"].tolist()))
        else:
            assert out is None
            q.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_records_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chans, items = next(r for r in results if r != "ok")
    assert chans == list(range(10))                                    # global channel ids, in order
    assert items == [0, 1, 2, 3, 4, 1000, 1001, 1002, 1003, 1004]
