"""The lane kernel's tables in HBM: compress rate of bench.py's workload
against the distance between two lanes' tables (test option
lane_table_stride_kib; the library derives it from lane_table_budget_pct),
with the probe's own timings of the placements it tried.
usage: SNAPMI_TESTING=1 python tests/hw/table_stride.py [gib] [stride_kib ...]"""
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
strides = [int(x) for x in sys.argv[2:]] or [256, 320, 384, 456, 512, 640,
                                              768, 1024, 256, 456, 1024]
print(f"# bench.py's workload at {gib:g} GiB, compress: stride between lane "
      "tables -> ms per pass, GiB/s, kernel, placements probed (ms each)")
for kib in strides:
    c = raw.Context(0)
    c.set_option("lane_table_budget_pct", 75)
    c.set_test_option("lane_table_stride_kib", kib)
    ub, cb, n, te, td = B.round_tiles(c, dev, gib, 3)
    log = _lib.load().snapmi_table_probe_log(c._h).decode()
    print(f"stride {kib:5d} KiB: {te*1e3:8.2f} ms {ub/2**30/te:6.1f} GiB/s  "
          f"{c.last_kernel()}  probes: {log}", flush=True)
    c.close()
    torch.cuda.empty_cache()
