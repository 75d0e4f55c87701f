"""Where the lane-per-block kernel takes over from the window kernel: the
corpus round (bench.py's workload) and two single files tiled to 0.5 .. 4 GiB,
compress ms per pass with lane_min_blocks at 1 (lanes) and 2^30 (windows)."""
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
rnd = O.corpus_round()
mix = b"".join(d for _, d in rnd)
sets = {"round": None, "alice29.txt": (O.CORPUS / "alice29.txt").read_bytes(),
        "html": (O.CORPUS / "html").read_bytes()}
for name, blob in sets.items():
    for gib in (0.5, 1.0, 1.5, 2.0, 3.0, 4.0):
        row = f"{name:12s} {gib:4.1f} GiB:"
        for label, lmb in (("windows", 1 << 30), ("lanes", 1)):
            ctx = raw.Context(0)
            ctx.set_option("lane_min_blocks", lmb)
            ctx.set_option("lane_table_tries", 3)
            if blob is None:
                # the 12 streams of the round, tiled
                import numpy as np
                from rust_snappy_amd import batch
                reps = max(1, int(gib * B.GIB / len(mix)))
                data = torch.frombuffer(bytearray(mix),
                                        dtype=torch.uint8).to(dev).repeat(reps)
                offs, lens, pos = [], [], 0
                for _ in range(reps):
                    for _, d in rnd:
                        offs.append(pos)
                        lens.append(len(d))
                        pos += len(d)
                src = batch.StreamBatch(data, np.array(offs, dtype=np.int64),
                                        np.array(lens, dtype=np.int64))
                caps = [raw.max_compress_len(n) for n in lens]
                comp = batch.StreamBatch.empty(caps, dev)
                clens = torch.zeros(len(lens), dtype=torch.int64, device=dev)

                def enc():
                    raw.compress_batch(ctx, src.d_ptrs, src.d_lens,
                                       comp.d_ptrs, comp.d_lens, clens, None,
                                       host_in_lens=src.h_lens)
                te = B.time_it(enc, 3, ctx)
                n = pos
                del data, comp
            else:
                n, c, reps, te, td = B.raw_tiles(ctx, dev, blob, gib, 3, None)
            row += f"  {label} {te*1e3:8.2f} ms {n/2**30/te:6.1f} GiB/s"
            ctx.close()
            torch.cuda.empty_cache()
        print(row, flush=True)
