"""Token scratch against the lane kernel's launch size: bench.py's workload
on a fresh context per row with the block list matched and encoded in 1, 2, 3,
4 and 6 segments (option lane_segment_blocks) - compress ms per pass, GiB/s,
the device memory the context holds afterwards.
usage: python tests/hw/segment_ab.py [gib] [segments ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

dev = torch.device("cuda", 0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
segs = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4, 6, 1, 2, 3]
blocks = int(round(gib * 2**30 / 2928571)) * 50
print(f"# bench.py's workload at {gib:g} GiB = {blocks} blocks, compress, a "
      "fresh context per row: segments -> blocks per segment, ms per pass, "
      "GiB/s, bytes the context holds, kernel")
for ns in segs:
    c = raw.Context(0)
    c.set_option("lane_table_budget_pct", 75)
    import os
    for kv in filter(None, os.environ.get("AB_OPTS", "").split(",")):
        k, v = kv.split("=")
        c.set_option(k, int(v))
    per = -(-blocks // ns)
    c.set_option("lane_segment_blocks", max(64, per))
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    diag = {}
    ub, cb, n, te, td = B.round_tiles(c, dev, gib, 4, diag)
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info(dev)[0]
    print(f"{ns} segments of {per:6d}: {te*1e3:8.2f} ms {ub/2**30/te:6.1f} "
          f"GiB/s  context {free0-free1:12d} B = {(free0-free1)/ub:.2f} x "
          f"input  {diag.get('kernel')}  calls {diag.get('call_ms')}  "
          f"probe {diag.get('placement', '')[:24]}",
          flush=True)
    c.close()
    torch.cuda.empty_cache()
