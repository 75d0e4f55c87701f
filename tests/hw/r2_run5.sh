#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
make -C rust-snappy_amd/csrc -s profile 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" | head -3
timeout 300 python tests/hw/prof_decode2.py 200 2>&1 | grep -v amdgpu.ids | tee $O/r2_prof_decode2.txt
cd /tmp && export TMPDIR=/tmp
for k in 1 2; do
 for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  tag=k${k}_$(echo $grp | tr ' ' '+' | cut -c1-40)
  rm -rf $O/pmc_dec_$tag
  SNAPMI_DECODE_KERNEL=$k timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_dec_$tag -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $O/pmc_dec_$tag.log 2>&1
  f=$(find $O/pmc_dec_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" "k$k" <<'PY' >> $O/r2_pmc_decoders.txt
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_decompress_streams" not in k: continue
    k = k.split("snapmi::")[1].split("(")[0]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for (k, c), v in sorted(acc.items()):
    print(f"{sys.argv[2]} {k:24s} {c:24s} {v / len(cnt[k]):.4g}")
PY
  find $O/pmc_dec_$tag -type f -size +1M -delete
 done
done
cat $O/r2_pmc_decoders.txt
