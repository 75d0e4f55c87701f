#!/bin/bash
# (prepared for the next round, not run yet) where k_match_blocks_spec stops paying: alice29.txt and urls.10K tiled to
# 1 / 1.5 / 2 GiB with the speculation limit at 0 (off) and at 65536 blocks (on for all three sizes)
R=$PWD
mkdir -p gpurun_out
timeout 120 python - <<'PY' | tee gpurun_out/spec_sweep.txt
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench_configs as B
import oracle_lib as O
from rust_snappy_amd import raw
dev = torch.device("cuda", 0)
out = {}
for name in ("alice29.txt", "urls.10K"):
    text = (O.CORPUS / name).read_bytes()
    want = O.compress(text)
    for limit in (0, 65536):
        ctx = raw.Context(0)
        ctx.set_option("lane_min_blocks", 1)
        ctx.set_option("lane_table_budget_pct", 75)
        ctx.set_test_option("lane_speculate_max_blocks", limit)
        for gib in (1.0, 1.5, 2.0):
            n, c, reps, te, td = B.raw_tiles(ctx, dev, text, gib, 3, want)
            out[f"{name}_limit{limit}_{gib}"] = round(te * 1e3, 2)
        ctx.close()
print(json.dumps(out))
PY
