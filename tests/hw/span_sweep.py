"""k_compress_spans (window steps) against k_compress_blocks (one copy per
step) and the lane-per-block kernel: compress ms per pass for one corpus file
tiled to 1/64 .. 2 GiB, every first/last stream checked against the oracle.
usage: span_sweep.py [file ...]  (default alice29.txt urls.10K html)"""
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
names = sys.argv[1:] or ["alice29.txt", "urls.10K", "html"]
sizes = (1 / 64, 1 / 16, 0.25, 1.0, 2.0)
configs = {
    "spans": dict(compress_mode=0, span_kernel=1, small_batch_kernel=0),
    "spans_auto": dict(compress_mode=0, span_kernel=1, small_batch_kernel=1),
    "waves": dict(compress_mode=0, span_kernel=0, small_batch_kernel=1),
    "lanes": dict(compress_mode=1, lane_min_blocks=1, lane_table_tries=1),
}
out = {}
for name in names:
    blob = (O.CORPUS / name).read_bytes()
    want = O.compress(blob)
    for cname, opts in configs.items():
        ctx = raw.Context(0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        for gib in sizes:
            if cname == "waves" and gib > 1.0:
                continue
            n, c, reps, te, td = B.raw_tiles(ctx, dev, blob, gib, 3, want)
            out[f"{name}:{cname}:{gib:g}"] = [round(te * 1e3, 3),
                                              round(n / 2**30 / te, 1)]
            print(f"{name:14s} {cname:10s} {gib:8.4f} GiB  {te*1e3:9.3f} ms"
                  f"  {n/2**30/te:7.1f} GiB/s", flush=True)
        ctx.close()
print(json.dumps(out))
