"""k_match_both (three lane + two window wavefronts per CU, one two-ended
ticket) against k_match_blocks alone: the corpus round (bench.py's workload)
tiled to 2 / 4 / 8 GiB and alice29.txt at 4 GiB, compress ms per pass (all
kernels) and the match finder alone (HIP events of the library)."""
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
rnd = O.corpus_round()
mix = b"".join(d for _, d in rnd)
want0 = O.compress(rnd[0][1])


def round_batch(gib):
    reps = max(1, int(gib * B.GIB / len(mix)))
    data = torch.frombuffer(bytearray(mix), dtype=torch.uint8).to(dev).repeat(reps)
    offs, lens, pos = [], [], 0
    for _ in range(reps):
        for _, d in rnd:
            offs.append(pos)
            lens.append(len(d))
            pos += len(d)
    src = batch.StreamBatch(data, np.array(offs, dtype=np.int64),
                            np.array(lens, dtype=np.int64))
    comp = batch.StreamBatch.empty([raw.max_compress_len(n) for n in lens], dev)
    clens = torch.zeros(len(lens), dtype=torch.int64, device=dev)
    return src, comp, clens, pos


for gib in (8.0,):
    src, comp, clens, n = round_batch(gib)
    import os
    row = f"{os.path.basename(os.environ.get('SNAPMI_LIB', 'default')):14s} round {gib:4.1f} GiB:"
    for label, on in (("lanes", 0), ("both", 1), ("lanes", 0), ("both", 1)):
        ctx = raw.Context(0)
        ctx.set_option("lane_coresident", on)
        ctx.set_option("lane_coresident_min_blocks", 1)

        def enc():
            raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs,
                               comp.d_lens, clens, None,
                               host_in_lens=src.h_lens)
        te = B.time_it(enc, 3, ctx)
        t = ctx.last_timing()
        assert comp.stream_bytes(0, int(clens[0])) == want0
        row += (f"  {label} {te*1e3:7.2f} ms (match {t['dominant_ms']:7.2f})"
                f" {n/2**30/te:5.1f}")
        ctx.close()
    print(row, flush=True)
    del src, comp, clens
    torch.cuda.empty_cache()
