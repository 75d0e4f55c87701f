#!/bin/bash
# round 3: tests -> full bench line (extras, PMC traffic) -> rocprofv3 kernel stats -> decoder SQ counters
tag=${1:-r3_v1}
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3 | tee $R/gpurun_out/tests_$tag.txt
( time timeout 1200 python bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.log ) 2> $R/gpurun_out/bench_${tag}_time.txt
tail -3 $R/gpurun_out/bench_$tag.log; tail -3 $R/gpurun_out/bench_${tag}_time.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras --no-pmc > $R/gpurun_out/prof_$tag.log 2>&1
db=$(find $R/gpurun_out/prof_$tag -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $R/gpurun_out/kernel_stats_$tag.md; head -12 $R/gpurun_out/kernel_stats_$tag.md
grep "^{\"metric" $R/gpurun_out/prof_$tag.log > $R/gpurun_out/prof_${tag}_bench.json
rm -rf $R/gpurun_out/prof_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${tag}_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-pmc > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
f=$(find $R/gpurun_out/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $R/gpurun_out/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python $R/profiles/pmc_traffic.py $f $w "cfg2: 12-stream zflat/uflat round x2934 = 8.002 GiB, 35208 raw streams" > $R/gpurun_out/pmc_traffic_$tag.json
rm -rf $R/gpurun_out/pmc_${tag}_FETCH_SIZE $R/gpurun_out/pmc_${tag}_WRITE_SIZE
cd $R
[ "$2" = nosq ] && exit 0   # (the decoder kernels' SQ counters: only when those kernels changed)
for k in 2 3; do
  SNAPMI_TESTING=1 SNAPMI_DECODE_KERNEL=$k bash tests/hw/pmc_dec.sh ${tag}_k$k 8 \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
    "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE" \
    >> gpurun_out/pmc_dec_$tag.txt 2>&1
done
cat gpurun_out/pmc_dec_$tag.txt
