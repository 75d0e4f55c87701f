#!/bin/bash
# round 3, call J: 256-byte windows as the default: all GPU tests, bench, per-file decoder times (profile build)
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_j_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_j_tests.log
tail -3 gpurun_out/r3_j_tests.log
timeout 200 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 > gpurun_out/r3_j_bench.json 2> gpurun_out/r3_j_bench.log
tail -1 gpurun_out/r3_j_bench.log
make -C rust-snappy_amd/csrc profile > /dev/null 2>&1
timeout 250 python tests/hw/prof_decode2.py 200 > gpurun_out/r3_j_prof.txt 2>&1
cat gpurun_out/r3_j_prof.txt
