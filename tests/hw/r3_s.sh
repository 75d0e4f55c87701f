#!/bin/bash
# round 3, call S: where the time of 10.7 M tiny streams goes (kernel trace)
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tiny
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tiny -o tiny -- python $R/bench_configs.py --plan tiny:2 > $R/gpurun_out/r3_s_tiny.log 2>&1
db=$(find $R/gpurun_out/prof_tiny -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $R/gpurun_out/r3_s_tiny_stats.md
rm -rf $R/gpurun_out/prof_tiny
grep "^{" $R/gpurun_out/r3_s_tiny.log; head -14 $R/gpurun_out/r3_s_tiny_stats.md
