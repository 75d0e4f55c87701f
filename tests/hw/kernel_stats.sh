#!/bin/bash
# usage: kernel_stats.sh <tag> <bench_configs plan>   e.g. cfg5 cfg5:16
# rocprofv3 --kernel-trace --stats of one extras config; the per-kernel table
# goes to gpurun_out/<tag>_kernel_stats.md (profiles/db_stats.py)
tag=$1; plan=$2
R=/root/repo
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
rm -rf $out
timeout 900 rocprofv3 --kernel-trace --stats -d $out -o p -- python $R/bench_configs.py --plan $plan > $R/gpurun_out/${tag}_run.txt 2>&1
db=$(find $out -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $R/gpurun_out/${tag}_kernel_stats.md
head -20 $R/gpurun_out/${tag}_kernel_stats.md
rm -rf $out
