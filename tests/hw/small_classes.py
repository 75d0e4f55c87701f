"""Streams of 300 .. 1 500 bytes: k_compress_small (a few lanes per wavefront,
state in LDS; small_stream_kernel 1, default) against the small-block window
kernel k_match_spans_8k (small_stream_kernel 0: such streams become blocks)."""
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
text = (O.CORPUS / "alice29.txt").read_bytes()
for size in (300, 400, 600, 800, 1000, 1023):
    blob = text[:size]
    want = O.compress(blob)
    row = f"{size:6d} bytes:"
    for name, opt in (("lanes in LDS", 1), ("window kernel", 0)):
        ctx = raw.Context(0)
        ctx.set_option("small_stream_kernel", opt)
        n, c, reps, te, td = B.raw_tiles(ctx, dev, blob, 0.5, 3, want)
        row += f"  {name} {te*1e3:8.3f} ms {n/2**30/te:7.1f} GiB/s"
        ctx.close()
    print(row, flush=True)
