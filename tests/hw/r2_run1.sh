#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite (all failures, not -x),
# then the default bench line (with extras) 
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $R/gpurun_out/r2a_pytest.log 2>&1
tail -40 $R/gpurun_out/r2a_pytest.log
timeout 900 python bench.py > $R/gpurun_out/r2a_bench.json 2> $R/gpurun_out/r2a_bench.log
tail -5 $R/gpurun_out/r2a_bench.log | cut -c1-600
cat $R/gpurun_out/r2a_bench.json | cut -c1-6000
