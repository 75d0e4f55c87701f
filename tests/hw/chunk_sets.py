"""The lane tables' chunk sets: fresh contexts one after the other in one
process (each frees what the last one held: what the next is handed
differs), bench.py's workload at 8 GiB, lane_table_tries 4 - the probe of
every set that was tried, and the compress ms the context then runs at.
usage: python tests/hw/chunk_sets.py [contexts] [tries]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for i in range(n):
    c = raw.Context(0)
    c.set_option("lane_table_budget_pct", 75)
    c.set_option("lane_table_tries", tries)
    diag = {}
    ub, cb, nn, te, td = B.round_tiles(c, dev, 8.0, 3, diag)
    print(f"context {i}: {te * 1e3:7.2f} ms  first call "
          f"{diag['first_call_ms']:7.1f} ms  "
          f"{diag['placement'].split(' | held')[0]}", flush=True)
    c.close()
    torch.cuda.empty_cache()
