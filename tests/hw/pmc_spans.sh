#!/bin/bash
# SQ counters of k_compress_spans on alice29.txt tiled to 0.25 GiB (4096
# blocks of ~1148 steps): one rocprofv3 --pmc pass per group, per-kernel sums
R=$PWD
cd /tmp && export TMPDIR=/tmp
cat > /tmp/spans_one.py <<'PY'
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import bench_configs as B, oracle_lib as O
from rust_snappy_amd import raw
ctx = raw.Context(0)
ctx.set_option("compress_mode", 0); ctx.set_option("small_batch_kernel", 0)
blob = (O.CORPUS / "alice29.txt").read_bytes()
B.raw_tiles(ctx, torch.device("cuda", 0), blob, 0.25, 1, O.compress(blob))
PY
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_spans_$i
  rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- python /tmp/spans_one.py > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_compress_spans" in k:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c, v in sorted(acc.items()):
    print(f"k_compress_spans {c} {v/len(n[c]):.5g} per launch ({len(n[c])} launches)")
PY
  rm -rf $out
done
