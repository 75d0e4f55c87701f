#!/bin/bash
# where the time of 1 KiB / 4 KiB streams goes: kernel stats of compress + decompress of 1 GiB each
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for size in 1024 4096; do
  rm -rf $R/gpurun_out/prof_small
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_small -o t -- python $R/tests/hw/small_streams.py $size 1 > $R/gpurun_out/prof_small_$size.log 2>&1
  db=$(find $R/gpurun_out/prof_small -name "*.db" | head -1)
  echo "== streams of $size bytes" | tee -a $R/gpurun_out/small_prof.txt
  grep "^{" $R/gpurun_out/prof_small_$size.log | tee -a $R/gpurun_out/small_prof.txt
  python $R/profiles/db_stats.py $db | head -16 | tee -a $R/gpurun_out/small_prof.txt
  rm -rf $R/gpurun_out/prof_small
done
