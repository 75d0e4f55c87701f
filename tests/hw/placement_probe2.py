#!/usr/bin/env python3
"""Does k_probe_tables rank table placements the way k_match_blocks does?
One process, the cfg2 batch, ONE context: the lane tables are dropped and
re-placed several times (everything else stays where it is); for every
placement: the probe's ms (snapmi_table_probe_log) and the lane kernel's ms.
Then the same with lane_table_tries = 4 (the library picks)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import importlib.util
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from rust_snappy_amd import batch, raw, _lib

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
comp_lens = torch.zeros(len(lens), dtype=torch.int64, device=dev)
L = _lib.load()

def run(ctx, k=2):
    ms = []
    for _ in range(k):
        raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs, comp.d_lens, comp_lens, None, host_in_lens=src.h_lens)
        ms.append(ctx.last_timing()["dominant_ms"])
    return ms

hold = []   # dummy allocations that push the next placement elsewhere
for tries, reps in ((1, 6), (4, 3), (8, 2)):
    ctx = raw.Context(0)
    ctx.set_test_option("lane_table_probe", 1)
    ctx.set_option("lane_table_tries", tries)
    for i in range(reps):
        ms = run(ctx)
        log = L.snapmi_table_probe_log(ctx._h).decode()
        free, _ = torch.cuda.mem_get_info()
        print(f"tries {tries} placement {i}: probe ms [{log}]  lane kernel ms {ms[0]:.1f} {ms[1]:.1f}  free {free/2**30:.0f} GiB", flush=True)
        ctx.set_test_option("lane_tables_renew", 1)
        if tries == 1 and i % 2 == 1:   # shift what the allocator hands out next
            hold.append(torch.empty((3 + i) << 30, dtype=torch.uint8, device=dev))
    ctx.close()
    hold.clear()
    torch.cuda.empty_cache()
