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


def test_gatherv_around_a_communicator_the_host_already_has(ctx):
    """snapmi_comm_wrap: the host's own ncclComm_t (made here with RCCL's C
    API, as a host that links RCCL would - torch does not hand its
    communicator out) at world size 1: the same gather, and destroying the
    wrapper leaves the host's communicator usable."""
    import ctypes as C
    import glob
    import os
    import torch
    from rust_snappy_amd import shard
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib",
                                   "librccl.so*")) + ["librccl.so"]
    R = None
    for c in cands:
        try:
            R = C.CDLL(c)
            break
        except OSError:
            continue
    assert R is not None, "librccl not found"
    class UniqueId(C.Structure):           # ncclUniqueId: passed BY VALUE
        _fields_ = [("internal", C.c_char * 128)]
    ident = UniqueId()
    R.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    assert R.ncclGetUniqueId(C.byref(ident)) == 0
    comm = C.c_void_p()
    R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId,
                                   C.c_int]
    assert R.ncclCommInitRank(C.byref(comm), 1, ident, 0) == 0
    try:
        for _ in range(2):          # wrap, use, destroy the wrapper - twice
            w = shard.Comm.wrap(ctx, comm.value, 0, 1)
            part = (torch.arange(70_001, dtype=torch.int64, device="cuda")
                    * 7).to(torch.uint8)
            whole, sizes = w.gatherv(part, dst=0, cap=80_000)
            assert sizes == [70_001] and bool((whole == part).all())
            w.close()
    finally:
        R.ncclCommDestroy.argtypes = [C.c_void_p]
        assert R.ncclCommDestroy(comm) == 0


def test_two_ranks_without_two_devices_leave_a_record():
    """The driver's N > 1 launch line on a box with ONE device: every rank
    sees it at once, rank 0 prints one JSON line with "error" (what the
    driver records) and nobody waits in a rendezvous."""
    import socket
    import time
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    t0 = time.perf_counter()
    p = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2",
         "--steps", "1", "--warmup", "0", "--gib", "0.25", "--no-cpu",
         "--no-extras", "--no-pmc"], capture_output=True, text=True,
        timeout=300)
    assert p.returncode != 0
    assert time.perf_counter() - t0 < 120
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:] + p.stderr[-1500:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] is None
    assert "device(s) visible" in rec["error"]


def test_the_line_survives_a_dead_collective(monkeypatch):
    """bench.py's guard around what is not the measurement: with_deadline
    returns its record when the call does not come back, and passes results
    and exceptions of calls that do."""
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    import time
    assert bench.with_deadline(lambda: {"ok": 1}, 5, {"error": "x"}) == {
        "ok": 1}
    r = bench.with_deadline(lambda: time.sleep(30), 0.5, {"error": "late"})
    assert r == {"error": "late", "hung": True}
    r = bench.with_deadline(lambda: 1 / 0, 5, {"error": "x"})
    assert "ZeroDivisionError" in r["error"]
