#!/bin/bash
# usage: final_profile.sh <tag>   (on the GPU box, from the repo root)
# tests -> bench line -> rocprofv3 kernel stats -> PMC traffic passes
tag=$1
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.log
tail -2 $R/gpurun_out/bench_$tag.log; cat $R/gpurun_out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras > $R/gpurun_out/prof_$tag.log 2>&1
db=$(find $R/gpurun_out/prof_$tag -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $R/gpurun_out/kernel_stats_$tag.md; head -8 $R/gpurun_out/kernel_stats_$tag.md
grep "^{\"metric" $R/gpurun_out/prof_$tag.log > $R/gpurun_out/prof_${tag}_bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${tag}_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
f=$(find $R/gpurun_out/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $R/gpurun_out/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python $R/profiles/pmc_traffic.py $f $w "cfg2: 12-stream zflat/uflat round x2934 = 8.002 GiB, 35208 raw streams" > $R/gpurun_out/pmc_traffic_$tag.json
# keep the merge small: the raw csv/db files are large
find $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE $R/gpurun_out/prof_$tag -type f -size +4M -delete
cat $R/gpurun_out/pmc_traffic_$tag.json | head -40
