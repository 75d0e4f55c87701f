#!/bin/bash
# usage: scan_timeline.sh  (GPU box, repo root): kernel timeline of one 64 MiB
# decompress call with 64, 16 and 8 segments per scan wavefront
R=$PWD
export SNAPMI_TESTING=1
cd /tmp; export TMPDIR=/tmp
for v in 64 16 8; do
  set -- $v
  rm -rf $R/gpurun_out/prof_tl
  SCAN_SEGS=$1 timeout 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o p -- python $R/tests/hw/sweep_one.py 0.0625 > $R/gpurun_out/tl_run.txt 2>&1
  f=$(find $R/gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
  echo "## scan_segs $1"; tail -1 $R/gpurun_out/tl_run.txt
  python $R/tests/hw/timeline.py $f k_long_plan | grep -v "plan_decompress\|fillBuffer\|tiny\|small"
done
rm -rf $R/gpurun_out/prof_tl
