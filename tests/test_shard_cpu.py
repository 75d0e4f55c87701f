"""Multi-process (gloo, world_size 2) test of the sharding path: contiguous
byte-balanced partition, size exchange and variable-length gather."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_partition_by_bytes(built):
    from rust_snappy_amd.shard import partition_by_bytes
    lens = [100, 1, 1, 1, 97, 50, 50, 0, 0, 100]
    for world in (1, 2, 3, 4, 8, 16):
        parts = partition_by_bytes(lens, world)
        assert len(parts) == world
        assert parts[0][0] == 0 and parts[-1][1] == len(lens)
        for (a, b), (c, d) in zip(parts, parts[1:]):
            assert b == c and a <= b
    two = partition_by_bytes(lens, 2)
    s0 = sum(lens[two[0][0]:two[0][1]])
    assert abs(s0 - sum(lens) / 2) <= 100
    assert partition_by_bytes([], 4) == [(0, 0)] * 4


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from rust_snappy_amd.shard import (exchange_sizes, gatherv,
                                       partition_by_bytes)
    rnd = [d for _, d in O.corpus_round()] * 2
    lens = [len(d) for d in rnd]
    a, b = partition_by_bytes(lens, world)[rank]
    # stand-in for the GPU codec on this CPU-only test: the oracle's bytes
    mine = b"".join(O.compress(d) for d in rnd[a:b])
    local = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    sizes, offs = exchange_sizes(local.numel())
    assert sizes[rank] == len(mine) and offs[0] == 0
    out = gatherv(local, dst=0)
    if rank == 0:
        want = b"".join(O.compress(d) for d in rnd)
        assert out.numpy().tobytes() == want
        # max-over-ranks timing reduction used by bench.py
    # config 4: a framed stream sharded by chunk range; parts after the
    # first drop their stream identifier and the gathered bytes are the
    # single-stream framing (bench_configs.py cfg4)
    stream = b"".join(rnd)[:40 * 65536]
    chunks = len(stream) // 65536
    lo, hi = chunks * rank // world, chunks * (rank + 1) // world
    part = O.frame_compress(stream[lo * 65536:hi * 65536])
    part = part if rank == 0 else part[10:]
    out = gatherv(torch.frombuffer(bytearray(part), dtype=torch.uint8), dst=0)
    if rank == 0:
        assert out.numpy().tobytes() == O.frame_compress(stream)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_gloo_world2_shard_and_gather(built, tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world,
             join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
