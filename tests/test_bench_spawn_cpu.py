"""bench.py --gpus N launch plumbing, without a GPU (gloo).  The real run needs
N MI355X; what can be checked here is that `python bench.py --gpus N` alone
turns into N ranks with a working process group, that rank 0 prints one JSON
line with n_gpus = N, and that a WORLD_SIZE / --gpus mismatch is refused."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
              "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args,
                          capture_output=True, text=True, env=e,
                          timeout=timeout)


def json_lines(p):
    return [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]


def test_gpus_n_spawns_n_ranks():
    p = run(["--gpus", "2", "--plumbing-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p)
    assert len(lines) == 1, p.stdout          # rank 0 only, one line
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == [0, 1]
    assert lines[0]["max_over_ranks"] == 2.0   # MAX over ranks, not rank 0's
    assert "spawning 2 ranks" in p.stderr
    # the N > 1 extras: one child per rank, a process group of their own on
    # a port of their own (rank_children); rank 0's child reports
    child = lines[0]["children"]
    assert "error" not in child, child
    assert child["sum"] == 3.0                  # 1 + 2: both children met


def test_single_rank_needs_no_launcher():
    p = run(["--plumbing-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json_lines(p)[0]["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    p = run(["--gpus", "4", "--plumbing-check"],
            env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=2" in p.stderr
    assert not json_lines(p)


def test_more_gpus_than_devices_is_refused():
    # no GPU in the CPU container: --gpus 2 without the plumbing flag must
    # refuse before it spawns anything
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    p = run(["--gpus", "2"])
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr


def test_committed_final_bench_line_is_of_this_tree():
    """Round 5 landed placement code after its last full bench run, and the
    driver's record then held a row nobody had seen.  The final line of a
    round (profiles/r6_final_bench.json: one default `python bench.py` on one
    box) says which sources it measured - `source_sha16`, sha256 over the
    kernels, the C ABI, the Python mirror, bench.py and bench_configs.py - and
    this test compares it with the tree: code that lands behind the last full
    run fails here."""
    import json
    import pytest
    from conftest import ROOT
    import bench
    path = ROOT / "profiles" / "r6_final_bench.json"
    if not path.exists():
        pytest.skip("no final bench line committed yet")
    line = json.loads(path.read_text().strip().splitlines()[-1])
    assert line["source_sha16"] == bench.source_sha16(), \
        "sources changed after the final bench run: run " \
        "tests/hw/final_profile.sh again and commit profiles/r6_final_*"
    assert line["n_gpus"] == 1 and line["roofline"]["traffic_measured"]
    assert line["cpu_baseline"]["kind"] == "port"
    assert "sweep" in line["extras"] and "budget" in line["extras"]
