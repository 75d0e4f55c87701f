#!/bin/bash
# usage: pmc_dec.sh <tag> <gib> "<counter group 1>" ...   (SNAPMI_DECODE_KERNEL selects the kernel)
# one rocprofv3 --pmc pass per counter group; prints per-kernel sums for the decoder
tag=$1; gib=$2; shift 2
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- python $R/bench.py --gib $gib --steps 1 --warmup 0 --no-cpu --no-extras --no-verify --no-pmc > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_decompress_streams" in k:
        acc[(k.split("(")[0][-21:], r["Counter_Name"])] += float(r["Counter_Value"])
for (k, c), v in sorted(acc.items()):
    print(f"{tag} {k} {c} {v:.4g}")
PY
  rm -rf $out
done
