"""The lane tables' placement as the library chooses it: bench.py's workload
on fresh contexts at lane_table_budget_pct 33 (the default), 20 and 10 -
compress ms, GiB/s, the probe log; a negative percentage: the far end of the
memory first (snapmi_ctx_prepare with SNAPMI_PREPARE_TOP_OF_MEMORY; round 5's
lane_table_high option is gone)
(no filler in front of the first candidate).
usage: python tests/hw/table_budget.py [gib] [pct ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import _lib, raw  # noqa: E402

dev = torch.device("cuda", 0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
pcts = [int(x) for x in sys.argv[2:]] or [33, 33, 33, 20, 20, 10, 10, -33,
                                          -33]
print(f"# bench.py's workload at {gib:g} GiB, compress, a fresh context per "
      "row: budget -> ms per pass, GiB/s, placements probed (ms each)")
for pct in pcts:
    c = raw.Context(0)
    c.set_option("lane_table_budget_pct", abs(pct))
    if pct < 0:  # 8 GiB of the corpus round: 50 blocks per 2 928 571 bytes
        c.prepare(int(gib * 2**30 / 2928571) * 50, top_of_memory=True)
    free0 = torch.cuda.mem_get_info(dev)[0]
    ub, cb, n, te, td = B.round_tiles(c, dev, gib, 3)
    log = _lib.load().snapmi_table_probe_log(c._h).decode()
    print(f"budget {pct:2d} %: {te*1e3:8.2f} ms {ub/2**30/te:6.1f} GiB/s  "
          f"probes: {log}", flush=True)
    c.close()
    torch.cuda.empty_cache()
