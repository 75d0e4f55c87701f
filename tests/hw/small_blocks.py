"""Page-sized streams: the head of alice29.txt in streams of 1 000 .. 16 384
bytes tiled to 1 GiB, compress GiB/s with the small-table window kernels
(k_match_spans_8k, default) and without (small_table_kernel 0)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
import oracle_lib as O  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

dev = torch.device("cuda", 0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
text = (O.CORPUS / "alice29.txt").read_bytes()
for size in (1000, 1500, 2000, 3000, 4096, 6000, 8192, 12000):
    blob = text[:size]
    want = O.compress(blob)
    row = f"{size:6d} bytes:"
    for name, opt in (("small tables", 1), ("off", 0)):
        ctx = raw.Context(0)
        ctx.set_option("small_table_kernel", opt)
        n, c, reps, te, td = B.raw_tiles(ctx, dev, blob, gib, 3, want)
        row += f"  {name} {te*1e3:8.3f} ms {n/2**30/te:7.1f} GiB/s"
        ctx.close()
    print(row, flush=True)
