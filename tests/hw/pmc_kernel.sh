#!/bin/bash
# SQ counters of ONE kernel of a command, one rocprofv3 --pmc pass per group,
# per-launch averages: pmc_kernel.sh <kernel substring> <command ...>
k=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  out=$R/gpurun_out/pmc_k_$i
  rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o p -- "$@" > $out.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" "$k" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c, v in sorted(acc.items()):
    print(f"{sys.argv[2]} {c} {v/len(n[c]):.5g} per launch ({len(n[c])} launches)")
PY
  rm -rf $out $out.log
done
