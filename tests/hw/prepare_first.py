"""Does it matter WHEN a process places its lane tables?  A fresh process:
(1) the context's tables first (snapmi_ctx_prepare before anything else is
allocated on the device), then bench.py's workload; (0) the workload's
buffers first, the tables inside the first compress call (bench.py's order).
The placement probe and the compress ms the context then runs at.
(2) like (1) with SNAPMI_PREPARE_TOP_OF_MEMORY.
usage: python tests/hw/prepare_first.py 0|1|2"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.cuda.init()
c = raw.Context(0)
c.set_option("lane_table_budget_pct", 75)
import time
t0 = time.perf_counter()
if first:
    c.prepare(int(round(8.0 * 2**30 / 2928571)) * 50, top_of_memory=first == 2)
prep_s = time.perf_counter() - t0
diag = {}
ub, cb, n, te, td = B.round_tiles(c, dev, 8.0, 3, diag)
print(f"tables {['in the first call', 'first', 'first, top of memory'][first]} "
      f"(prepare {prep_s:.1f} s, first call {diag['first_call_ms']:.0f} ms): "
      f"{te * 1e3:7.2f} ms"
      f"  {ub / 2**30 / te:5.1f} GiB/s  {c.table_probe_log().split(' | held')[0]}",
      flush=True)
