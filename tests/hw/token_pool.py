"""The token pool against the data: a fresh context, five compress calls of
one batch - call ms, blocks spilled (compressed a second time by
k_redo_spilled), pool pages, what the pool has grown to, token scratch bytes
against the input - for bench.py's workload, English text alone (plrabn12.txt)
and the densest file of the corpus alone (kppkn.gtb), at the default
token_pool_pct (39) and at 100 (no block can spill).
usage: python tests/hw/token_pool.py [gib]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
from rust_snappy_amd import batch, raw  # noqa: E402

dev = torch.device("cuda", 0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
rnd = dict(O.corpus_round())
names = list(rnd)


def tiled(files):
    """the files, 16-byte aligned, repeated to `gib` as independent streams"""
    offs, pos = [], 0
    for d in files:
        offs.append(pos)
        pos += (len(d) + 15) // 16 * 16
    one = np.zeros(pos, dtype=np.uint8)
    for d, o in zip(files, offs):
        one[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
    lens1 = np.array([len(d) for d in files], dtype=np.int64)
    rounds = max(1, int(round(gib * 2**30 / int(lens1.sum()))))
    data = torch.from_numpy(one).to(dev).repeat(rounds)
    o_all = (np.arange(rounds, dtype=np.int64)[:, None] * pos
             + np.array(offs, dtype=np.int64)[None, :]).reshape(-1)
    lens = np.tile(lens1, rounds)
    src = batch.StreamBatch(data, o_all, lens)
    caps = np.array([raw.max_compress_len(int(x)) for x in lens1],
                    dtype=np.int64)
    comp = batch.StreamBatch.empty(np.tile(caps, rounds), dev)
    return src, comp, int(lens.sum()), len(lens)


sets = {"corpus round": [rnd[k] for k in names],
        "plrabn12.txt": [next(v for k, v in rnd.items() if "txt4" in k)],
        "kppkn.gtb": [next(v for k, v in rnd.items() if "gaviota" in k)]}
for label, files in sets.items():
    src, comp, ub, n = tiled(files)
    clens = torch.zeros(n, dtype=torch.int64, device=dev)
    want = [O.compress(d) for d in files]
    for pct in (0, 100):          # 0: the library's default
        c = raw.Context(0)
        c.set_option("lane_table_budget_pct", 75)
        if pct:
            c.set_option("token_pool_pct", pct)
        print(f"## {label}, {ub / 2**30:.2f} GiB, {n} streams, "
              f"token_pool_pct {pct or 'default'}")
        for call in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            raw.compress_batch(c, src.d_ptrs, src.d_lens, comp.d_ptrs,
                               comp.d_lens, clens, None,
                               host_in_lens=src.h_lens)
            c.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            cl = clens[-len(files):].cpu().numpy()
            for j, w in enumerate(want):
                k = n - len(files) + j
                assert comp.stream_bytes(k, int(cl[j])) == w, (label, j)
            print(f"call {call}: {ms:8.2f} ms {ub / 2**30 / ms * 1e3:6.1f} "
                  f"GiB/s  spilled {c.info('token_blocks_spilled'):6d}  pool "
                  f"{c.info('token_pool_pages'):8d} pages (now "
                  f"{c.info('token_pool_pct_now')} %), asked "
                  f"{c.info('token_pages_asked'):8d}  token scratch "
                  f"{c.info('token_scratch_bytes') / ub:.3f} x input, "
                  f"context {c.info('scratch_bytes') / 1e9:.2f} GB  "
                  f"{c.last_kernel()}", flush=True)
        c.close()
        torch.cuda.empty_cache()
    del src, comp, clens
    torch.cuda.empty_cache()
