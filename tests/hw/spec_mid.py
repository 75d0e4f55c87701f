"""k_match_blocks_spec against the plain kernel where a block's latency is
what is waited for: alice29.txt tiled to 0.125 .. 1 GiB (2048 .. 16384
blocks), lane kernel forced, speculation off / on."""
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

dev = torch.device("cuda", 0)
text = (O.CORPUS / "alice29.txt").read_bytes()
want = O.compress(text)
out = {}
for spec in (0, 1):
    ctx = raw.Context(0)
    ctx.set_option("lane_min_blocks", 1)
    ctx.set_option("compress_mode", 1)
    ctx.set_option("lane_table_tries", 1)
    ctx.set_option("lane_speculate", spec)
    for gib in (0.125, 0.25, 0.5, 1.0):
        n, c, reps, te, td = B.raw_tiles(ctx, dev, text, gib, 3, want)
        out[f"spec{spec}_{gib}"] = round(te * 1e3, 2)
    ctx.close()
print(json.dumps(out))
