"""What the ORDER of a batch's blocks is worth to the lane kernel: bench.py's
workload at 8 GiB with its streams (a) as bench.py has them (the 12 files
round after round), (b) grouped by file, the files whose blocks cost a lane
most first (urls, the texts, gaviota, html, pb, pdf, jpg), (c) the same the
other way round.  One context, compress ms per order.
usage: python tests/hw/order_ab.py [gib]"""
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
rnd = O.corpus_round()
names = [k for k, _ in rnd]
files = [d for _, d in rnd]
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
heavy = ["urls", "txt4", "txt3", "txt2", "txt1", "gaviota", "html4", "html",
         "pb", "pdf", "jpg_200", "jpg"]


def rank(name):
    for i, h in enumerate(heavy):
        if name.endswith(h):
            return i
    raise KeyError(name)


by_weight = sorted(range(12), key=lambda j: rank(names[j]))
orders = {
    "as bench.py has them": [(r, j) for r in range(rounds) for j in range(12)],
    "heavy files first": [(r, j) for j in by_weight for r in range(rounds)],
    "light files first": [(r, j) for j in reversed(by_weight)
                          for r in range(rounds)],
}
c = raw.Context(0)
c.set_option("lane_table_budget_pct", 75)
for label, order in list(orders.items()) + [list(orders.items())[0]]:
    o_all = np.array([r * pos + offs[j] for r, j in order], dtype=np.int64)
    lens = np.array([lens1[j] for _, j in order], dtype=np.int64)
    src = batch.StreamBatch(data, o_all, lens)
    caps = np.array([raw.max_compress_len(int(x)) for x in lens],
                    dtype=np.int64)
    comp = batch.StreamBatch.empty(caps, dev)
    clens = torch.zeros(len(lens), dtype=torch.int64, device=dev)
    ms = []
    for call in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        raw.compress_batch(c, src.d_ptrs, src.d_lens, comp.d_ptrs, comp.d_lens,
                           clens, None, host_in_lens=src.h_lens)
        c.synchronize()
        ms.append((round((time.perf_counter() - t0) * 1e3, 2),
                   c.info("token_blocks_spilled"),
                   c.info("token_pages_asked"), c.info("token_pool_pages")))
    cl = clens.cpu().numpy()
    for k in (0, len(lens) - 1):
        assert comp.stream_bytes(k, int(cl[k])) == O.compress(
            files[order[k][1]]), k
    print(f"{label:22s} (ms, spilled, pages asked, pool) {ms}  "
          f"{c.last_kernel()}", flush=True)
    del src, comp, clens
    torch.cuda.empty_cache()
