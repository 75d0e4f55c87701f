"""One page size of tests/hw/small_blocks.py, for a kernel trace:
small_blocks_one.py <bytes> [gib]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
import oracle_lib as O  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

size = int(sys.argv[1])
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
blob = (O.CORPUS / "alice29.txt").read_bytes()[:size]
ctx = raw.Context(0)
n, c, reps, te, td = B.raw_tiles(ctx, torch.device("cuda", 0), blob, gib, 3,
                                 O.compress(blob))
print(f"{size} bytes x {reps}: compress {te*1e3:.3f} ms {n/2**30/te:.1f} GiB/s")
