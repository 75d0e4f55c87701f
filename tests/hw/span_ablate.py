"""Timing only (no verification: the ablation builds give wrong bytes):
k_compress_spans on corpus files tiled to 0.25 GiB, for the library
SNAPMI_LIB names.  usage: span_ablate.py [file ...]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
import oracle_lib as O  # noqa: E402
from rust_snappy_amd import batch, raw  # noqa: E402

dev = torch.device("cuda", 0)
names = sys.argv[1:] or ["alice29.txt", "html", "kppkn.gtb"]
row = [f"{os.path.basename(os.environ.get('SNAPMI_LIB', 'default')):24s} {os.environ.get('SPAN_CFG', 'spans'):6s}"]
for name in names:
    blob = (O.CORPUS / name).read_bytes()
    ctx = raw.Context(0)
    if os.environ.get("SPAN_CFG") == "match":
        # the window kernel as the token path's match finder + k_encode_tokens
        ctx.set_option("compress_mode", 1)
        ctx.set_option("lane_min_blocks", 1)
        ctx.set_option("match_kernel", 1)
    else:
        ctx.set_option("compress_mode", 0)
        ctx.set_option("small_batch_kernel", 0)
    reps = max(1, int(0.25 * B.GIB / len(blob)))
    stride = (len(blob) + 15) // 16 * 16
    one = np.zeros(stride, dtype=np.uint8)
    one[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
    data = torch.from_numpy(one).to(dev).repeat(reps)
    src = batch.StreamBatch(data, np.arange(reps, dtype=np.int64) * stride,
                            np.full(reps, len(blob), dtype=np.int64))
    cap = raw.max_compress_len(len(blob))
    comp = batch.StreamBatch.empty(np.full(reps, cap, dtype=np.int64), dev)
    clens = torch.zeros(reps, dtype=torch.int64, device=dev)

    def enc():
        raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs,
                           comp.d_lens, clens, None, host_in_lens=src.h_lens)
    B.time_it(enc, 5, ctx)
    te = B.time_it(enc, 10, ctx)
    row.append(f"{name} {te*1e3:8.3f} ms")
    ctx.close()
print("  ".join(row), flush=True)
