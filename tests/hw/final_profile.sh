#!/bin/bash
# usage: final_profile.sh <tag>   (on the GPU box, from the repo root)
# the default bench line (headline, PMC traffic by its own child runs, CPU
# baseline, all extras) -> rocprofv3 kernel stats of the same workload and
# the line that process printed (HIP events vs rocprof averages)
tag=$1
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.log
tail -4 $R/gpurun_out/bench_$tag.log; cut -c1-600 $R/gpurun_out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras --no-pmc > $R/gpurun_out/prof_$tag.log 2>&1
db=$(find $R/gpurun_out/prof_$tag -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $R/gpurun_out/kernel_stats_$tag.md; head -8 $R/gpurun_out/kernel_stats_$tag.md
grep "^{\"metric" $R/gpurun_out/prof_$tag.log > $R/gpurun_out/prof_${tag}_bench.json
find $R/gpurun_out/prof_$tag -type f -size +4M -delete
# one Encoder::compress / Decoder::decompress call per bench input: the
# product library's rule, then the test build at four long-stream thresholds
cd $R
{ echo "# product library (pieces for inputs of 32 KiB and more that expand by half, and of 256 KiB and more)";
  python tests/hw/scalar_latency.py 2>/dev/null;
  for t in 16384 65536 262144; do echo "# test build, SNAPMI_LONG_STREAM=$t";
    SNAPMI_TESTING=1 SNAPMI_LONG_STREAM=$t python tests/hw/scalar_latency.py 2>/dev/null | cut -c1-76; done; } > $R/gpurun_out/scalar_latency_$tag.txt
head -14 $R/gpurun_out/scalar_latency_$tag.txt
# kernel trace of one 2 GiB raw stream
bash tests/hw/kernel_stats.sh stream_$tag stream:2 > /dev/null 2>&1
grep -E "k_stream|k_decompress_streams3" $R/gpurun_out/stream_${tag}_kernel_stats.md | cut -c1-150
grep -h decompress_stream_gibs $R/gpurun_out/stream_${tag}_run.txt | cut -c1-250
