#!/bin/bash
# usage: pmc.sh <variant|-> <lane waves|-> <tag> "<counter group 1>" "<counter group 2>" ...
# one rocprofv3 --pmc pass per counter group; prints per-kernel sums for k_match_blocks
v=$1; w=$2; tag=$3; shift 3
R=$PWD
[ "$v" != "-" ] && export SNAPMI_LIB=$R/rust-snappy_amd/variants/$v.so
[ "$w" != "-" ] && export SNAPMI_TESTING=1 SNAPMI_LANE_WAVES=$w
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-verify --no-pmc > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "snapmi::" in k and "k_plan" not in k and "k_scan" not in k:
        acc[(k.split("(")[0][-16:], r["Counter_Name"])] += float(r["Counter_Value"])
for (k, c), v in sorted(acc.items()):
    print(f"{tag} {k} {c} {v:.4g}")
PY
done
