"""The N > 1 code before a multi-GPU box runs it: the RCCL gather
of the C ABI at world size 1, and bench.py's real rank code with two ranks on
ONE GPU (--oversubscribe: gloo instead of RCCL, marked INVALID)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_gatherv_over_the_c_abi_world_1(ctx):
    """snapmi_comm_unique_id / _init / snapmi_gatherv / _destroy on a
    one-rank communicator: sizes, total, the root's own part in place, and
    the capacity check that every rank makes."""
    import torch
    from rust_snappy_amd import shard
    from rust_snappy_amd.error import Error
    ident = shard.Comm.unique_id()
    assert len(ident) == 128
    comm = shard.Comm(ctx, ident, 0, 1)
    part = torch.arange(100_003, dtype=torch.int64, device="cuda").to(
        torch.uint8)
    whole, sizes = comm.gatherv(part, dst=0, cap=200_000)
    assert sizes == [100_003] and whole.numel() == 100_003
    assert bool((whole == part).all())
    empty = part[:0]
    whole, sizes = comm.gatherv(empty, dst=0, cap=16)
    assert sizes == [0] and whole.numel() == 0
    with pytest.raises(Error):                    # root buffer too small
        comm.gatherv(part, dst=0, cap=100_002)
    comm.close()


def test_two_ranks_oversubscribed_on_one_gpu():
    """`python bench.py --gpus 2 --oversubscribe`: the N > 1 branches of
    bench.py and of bench_configs.py::cfg4 (spawn, process group, barriers,
    MAX over ranks, per-rank gather, the sharded frame encode and its gather,
    the children's process group of their own), with real kernels."""
    p = subprocess.run(
        [sys.executable, str(ROOT / "bench.py"), "--gpus", "2",
         "--oversubscribe", "--gib", "0.5", "--steps", "2", "--warmup", "1",
         "--no-cpu"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]     # one line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == [0, 1]
    assert len(rec["per_rank_ms"]) == 2
    assert all(len(r) == 4 and r[3] > 0 for r in rec["per_rank_ms"])
    assert "oversubscribed" in rec["INVALID"]
    assert rec["config"]["parallelism"] == "shard-by-stream x2"
    cfg4 = rec["extras"]["cfg4"]
    assert "error" not in cfg4, cfg4              # includes the oracle check
    assert cfg4["n_gpus"] == 2 and cfg4["ranks_seen"] == [0, 1]
    assert cfg4["gathered_bytes_from_peers"] > 0
    assert cfg4["framed_bytes"] > cfg4["gathered_bytes_from_peers"]
