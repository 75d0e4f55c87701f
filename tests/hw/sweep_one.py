"""One size of bench.py's batch-size sweep (the corpus round tiled to <gib>):
decompress ms, for kernel traces.  usage: sweep_one.py <gib>
(SCAN_SEGS in the environment, with SNAPMI_TESTING=1: the test option
stream_scan_segs)"""
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
gib = float(sys.argv[1])
rnd = O.corpus_round()
mix = b"".join(d for _, d in rnd)
reps = max(1, int(gib * B.GIB / len(mix)))
comp_one = [O.compress(d) for _, d in rnd]
cmix = b"".join(comp_one)
data = torch.frombuffer(bytearray(cmix), dtype=torch.uint8).to(dev).repeat(reps)
offs, lens, caps, pos = [], [], [], 0
for _ in range(reps):
    for c, (_, d) in zip(comp_one, rnd):
        offs.append(pos)
        lens.append(len(c))
        caps.append(len(d))
        pos += len(c)
src = batch.StreamBatch(data, np.array(offs, dtype=np.int64),
                        np.array(lens, dtype=np.int64))
dst = batch.StreamBatch.empty(caps, dev)
olens = torch.zeros(len(lens), dtype=torch.int64, device=dev)
ctx = raw.Context(0)
if os.environ.get("SCAN_SEGS"):
    ctx.set_test_option("stream_scan_segs", int(os.environ["SCAN_SEGS"]))


def dec():
    raw.decompress_batch(ctx, src.d_ptrs, src.d_lens, dst.d_ptrs, dst.d_lens,
                         olens, None)


td = B.time_it(dec, 5, ctx)
n = sum(caps)
assert dst.stream_bytes(1) == rnd[1][1]
print(f"{gib} GiB: {len(lens)} streams, decompress {td*1e3:.3f} ms "
      f"{n/2**30/td:.1f} GiB/s")
