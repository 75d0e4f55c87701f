#!/bin/bash
# cfg5 (incompressible) per-kernel breakdown
R=$PWD
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r29
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r29 -o cfg5 -- python $R/bench_configs.py --plan cfg5:16 --steps 5 > $R/gpurun_out/r29/log.txt 2>&1
db=$(find $R/gpurun_out/r29 -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $R/gpurun_out/r29/kernel_stats.md; head -16 $R/gpurun_out/r29/kernel_stats.md
find $R/gpurun_out/r29 -name "*.db" -delete
