"""Streams of argv[1] bytes (the head of alice29.txt) tiled to argv[2] GiB:
compress and decompress rates (for rocprofv3 kernel stats, small_prof.sh)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
import oracle_lib as O  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

size, gib = int(sys.argv[1]), float(sys.argv[2])
dev = torch.device("cuda", 0)
ctx = raw.Context(0)
ctx.set_option("lane_table_budget_pct", 75)
for item in sys.argv[3:]:
    k, v = item.split("=")
    ctx.set_option(k, int(v))
# (size 0: a sparse page - 4 KiB of zeros, 196 compressed bytes)
text = (O.CORPUS / "alice29.txt").read_bytes()[:size] if size else bytes(4096)
n, c, reps, te, td = B.raw_tiles(ctx, dev, text, gib, 3, O.compress(text))
print(json.dumps({"size": size, "streams": reps, "ratio": round(c / n, 4),
                  "compress_gibs": round(n / B.GIB / te, 2),
                  "decompress_gibs": round(n / B.GIB / td, 2),
                  "compress_ms": round(te * 1e3, 2),
                  "decompress_ms": round(td * 1e3, 2)}))
