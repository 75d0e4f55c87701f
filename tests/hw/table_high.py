"""Does WHERE in the device's memory the lane tables lie decide their kind?
A filler of F GiB is allocated first (and freed once the tables exist), so the
tables land behind it: compress ms of bench.py's workload per F.
usage: SNAPMI_TESTING=1 python tests/hw/table_high.py [stride_kib] [F ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import _lib, raw  # noqa: E402

dev = torch.device("cuda", 0)
kib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fills = [int(x) for x in sys.argv[2:]] or [0, 64, 128, 192, 0, 128]
print(f"# stride {kib} KiB, one unprobed region behind a filler of F GiB")
for F in fills:
    filler = torch.empty(F << 30, dtype=torch.uint8, device=dev) if F else None
    c = raw.Context(0)
    c.set_option("lane_table_budget_pct", 75)
    c.set_option("lane_table_tries", 1)
    c.set_test_option("lane_table_probe", 1)
    c.set_test_option("lane_table_stride_kib", kib)
    ub, cb, n, te, td = B.round_tiles(c, dev, 8.0, 3)
    log = _lib.load().snapmi_table_probe_log(c._h).decode()
    print(f"filler {F:4d} GiB: {te*1e3:8.2f} ms {ub/2**30/te:6.1f} GiB/s  "
          f"probe: {log}  free now {torch.cuda.mem_get_info(dev)[0]>>30} GiB",
          flush=True)
    c.close()
    del filler
    torch.cuda.empty_cache()
