#!/bin/bash
# cache policy / uncached allocation for random 16-byte table accesses
R=$PWD
mkdir -p $R/gpurun_out/r34
$R/tests/hw/random_policy 2>&1 | tee $R/gpurun_out/r34/timing.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r34/$c -o p -- $R/tests/hw/random_policy > /dev/null 2>&1
  f=$(find $R/gpurun_out/r34/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2])
for r in rows:
    k=r.get('Kernel_Name','')
    if 'probe' in k: print('  ', k[:40], r.get('Grid_Size'), r.get('Counter_Value'))
PY
done 2>&1 | tee $R/gpurun_out/r34/pmc.txt
find $R/gpurun_out/r34 -name "*.csv" -size +1M -delete
