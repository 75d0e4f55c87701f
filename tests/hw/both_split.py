"""compress_mode 2 ("both": k_compress_spans on a share of the CUs, the lane
kernel on the others, one two-ended ticket) against the lane kernel alone:
compress ms per pass, the corpus round (one stream per round) and alice29.txt
tiled to 2 / 8 GiB."""
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
rnd = b"".join(d for _, d in O.corpus_round())
text = (O.CORPUS / "alice29.txt").read_bytes()
cases = [("round", rnd, 8.0), ("round", rnd, 2.0), ("alice29", text, 2.0),
         ("alice29", text, 1.0), ("alice29", text, 0.5)]
want = {"round": O.compress(rnd), "alice29": O.compress(text)}
out = {}
for label, opts in (
        ("lanes", dict(compress_mode=1, lane_min_blocks=1)),
        ("both64", dict(compress_mode=2, lane_min_blocks=1, both_wave_cus=64)),
        ("both96", dict(compress_mode=2, lane_min_blocks=1, both_wave_cus=96)),
        ("both128", dict(compress_mode=2, lane_min_blocks=1,
                         both_wave_cus=128)),
        ("both160", dict(compress_mode=2, lane_min_blocks=1,
                         both_wave_cus=160))):
    ctx = raw.Context(0)
    ctx.set_option("lane_table_budget_pct", 60)
    for k, v in opts.items():
        ctx.set_option(k, v)
    for name, blob, gib in cases:
        n, c, reps, te, td = B.raw_tiles(ctx, dev, blob, gib, 3, want[name])
        out[f"{label}:{name}:{gib:g}"] = round(te * 1e3, 2)
        print(f"{label:8s} {name:8s} {gib:4g} GiB {te*1e3:9.2f} ms "
              f"{n/2**30/te:7.1f} GiB/s", flush=True)
    ctx.close()
    torch.cuda.empty_cache()
print(json.dumps(out))
