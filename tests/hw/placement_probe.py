#!/usr/bin/env python3
"""Does the lane kernel's duration depend on WHICH memory its tables got?
One process, the cfg2 batch built once; then N fresh contexts (each allocates
its own lane tables, token scratch, slots), optionally holding the previous
context's memory so the next one lands elsewhere.  Prints the dominant-kernel
ms of two launches per context and the device pointers involved.

  python tests/hw/placement_probe.py [contexts] [hold_previous 0/1] [spread 0/1]
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import importlib.util
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from rust_snappy_amd import batch, raw

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hold = int(sys.argv[2]) if len(sys.argv) > 2 else 0
spread = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
rnd, host_round, r_offs, r_lens, shas = bench.build_round()
rounds = 2934
d_round = torch.from_numpy(host_round).to(dev)
data = d_round.repeat(rounds)
offs = (np.arange(rounds, dtype=np.int64)[:, None] * int(host_round.size) + r_offs[None, :]).reshape(-1)
lens = np.tile(r_lens, rounds)
src = batch.StreamBatch(data, offs, lens)
caps = np.array([raw.max_compress_len(int(x)) for x in r_lens], dtype=np.int64)
comp = batch.StreamBatch.empty(np.tile(caps, rounds), dev)
n = len(lens)
comp_lens = torch.zeros(n, dtype=torch.int64, device=dev)
print(f"in {data.data_ptr():#x} out {comp.data.data_ptr():#x}", flush=True)
kept = []
for i in range(n_ctx):
    ctx = raw.Context(0)
    ctx.set_option("lane_table_spread", spread)
    ms = []
    for _ in range(3):
        raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs, comp.d_lens, comp_lens, None, host_in_lens=src.h_lens)
        ms.append(ctx.last_timing()["dominant_ms"])
    free, total = torch.cuda.mem_get_info()
    print(f"ctx {i}: lane kernel ms {ms[0]:.1f} {ms[1]:.1f} {ms[2]:.1f}   free {free/2**30:.0f} GiB", flush=True)
    if hold and i % 2 == 0:
        kept.append(ctx)      # keep its memory: the next context lands elsewhere
    else:
        ctx.close()
