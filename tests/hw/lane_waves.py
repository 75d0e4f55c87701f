"""k_match_blocks alone by lane wavefronts per CU (test option
lane_waves_per_cu), bench.py's workload at 8 GiB: compress ms, the match
finder alone, and the probe's ms of the tables' placement (the rate follows
it: compare rows of one kind).
usage: SNAPMI_TESTING=1 python tests/hw/lane_waves.py [waves ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import _lib, raw  # noqa: E402

dev = torch.device("cuda", 0)
waves = [int(x) for x in sys.argv[1:]] or [6, 4, 3, 5, 6, 4, 3, 5]
for w in waves:
    c = raw.Context(0)
    c.set_option("lane_coresident", 0)
    c.set_test_option("lane_waves_per_cu", w)
    ub, cb, n, te, td = B.round_tiles(c, dev, 8.0, 3)
    log = _lib.load().snapmi_table_probe_log(c._h).decode().split("|")[0]
    print(f"{w} lane wavefronts per CU: {te*1e3:8.2f} ms {ub/2**30/te:6.1f} "
          f"GiB/s  probes (ms per {w * 256 * 64} lanes x 768 pairs): {log}",
          flush=True)
    c.close()
    torch.cuda.empty_cache()
