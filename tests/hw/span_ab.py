"""A/B of the window kernel's walk: compress ms per pass of corpus files tiled
to 1/16 .. 1 GiB through k_compress_spans, and the scalar Encoder::compress
latency (k_compress_span_lds), for the library SNAPMI_LIB names.
usage: SNAPMI_LIB=... span_ab.py [file ...]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
import oracle_lib as O  # noqa: E402
import rust_snappy_amd as R  # noqa: E402
from rust_snappy_amd import raw  # noqa: E402

dev = torch.device("cuda", 0)
names = sys.argv[1:] or ["alice29.txt", "urls.10K", "html", "kppkn.gtb"]
print("library", os.environ.get("SNAPMI_LIB", "default"))
for name in names:
    blob = (O.CORPUS / name).read_bytes()
    want = O.compress(blob)
    for cfg in ("inline", "tokens"):
        ctx = raw.Context(0)
        if cfg == "inline":  # k_compress_spans: encodes while it matches
            ctx.set_option("compress_mode", 0)
            ctx.set_option("small_batch_kernel", 0)
        else:  # k_match_spans + k_encode_tokens, whatever the batch size
            ctx.set_option("lane_min_blocks", 1 << 30)
            ctx.set_option("small_batch_kernel", 0)
        for gib in (1 / 16, 0.25, 1.0):
            n, c, reps, te, td = B.raw_tiles(ctx, dev, blob, gib, 5, want)
            print(f"{name:14s} {cfg:6s} {gib:7.4f} GiB {te*1e3:9.3f} ms "
                  f"{n/2**30/te:7.1f} GiB/s", flush=True)
        ctx.close()
enc = R.raw.Encoder()
for name, data in O.corpus_round():
    comp = enc.compress_vec(data)
    assert comp == O.compress(data), name
    t0 = time.perf_counter()
    for _ in range(20):
        enc.compress_vec(data)
    tc = (time.perf_counter() - t0) / 20
    print(f"scalar {name:18s} {len(data):8d} {tc*1e3:8.3f} ms "
          f"{len(data)/tc/1e6:7.0f} MB/s", flush=True)
