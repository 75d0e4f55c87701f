#!/bin/bash
# round 2, second GPU call: evidence for the lane kernel's roofline argument
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "side_index or long_stream" 2>&1 | tail -3
cd tests/hw
for p in random_rw16 random_wsize random_rmw; do hipcc --offload-arch=gfx950 -O3 -w -o $p $p.hip; done
( echo "== random_rw16 packed (tables 256 KiB apart, 40 GiB)"; timeout 120 ./random_rw16 0 40
  echo "== random_rw16 spread over 160 GiB"; timeout 120 ./random_rw16 0 160
  echo "== random_rw16 packed, behind 100 GiB of other memory"; timeout 120 ./random_rw16 100 40
  echo "== random_wsize"; timeout 120 ./random_wsize
  echo "== random_rmw"; timeout 120 ./random_rmw ) > $O/r2_random_probes.txt 2>&1
tail -25 $O/r2_random_probes.txt
cd $R
( echo "== placement_probe 5 contexts, release each"; timeout 300 python tests/hw/placement_probe.py 5 0 1
  echo "== placement_probe 6 contexts, hold every other one"; timeout 300 python tests/hw/placement_probe.py 6 1 1 ) > $O/r2_placement.txt 2>&1
cat $O/r2_placement.txt | grep -v amdgpu.ids
bash tests/hw/modes.sh 10 > $O/r2_modes.txt 2>&1; cat $O/r2_modes.txt
# transaction counters of one cfg2 step (separate pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '+')
  rm -rf $O/pmc_r2_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_r2_$tag -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $O/pmc_r2_$tag.log 2>&1
  f=$(find $O/pmc_r2_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' >> $O/r2_pmc_requests.txt
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "snapmi::" not in k: continue
    k = k.split("snapmi::")[1].split("(")[0]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for (k, c), v in sorted(acc.items()):
    print(f"{k:24s} {c:28s} {v / len(cnt[k]):.4g} per dispatch ({len(cnt[k])} dispatches)")
PY
  find $O/pmc_r2_$tag -type f -size +2M -delete
done
cat $O/r2_pmc_requests.txt
