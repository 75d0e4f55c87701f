#!/bin/bash
# round 3, call U: wave-aggregated tickets in the lane kernel: parity, tiny streams, the headline
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compress or many_small or tiled or epoch" > gpurun_out/r3_u_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_u_tests.log
tail -4 gpurun_out/r3_u_tests.log
timeout 300 python bench_configs.py --plan tiny:2,cfg5:8 2>/dev/null | grep "^{" | tee gpurun_out/r3_u_tiny.json
timeout 300 python bench.py --no-extras --no-cpu --no-pmc --steps 6 --warmup 2 2>&1 >/dev/null | tail -1 | tee gpurun_out/r3_u_bench.txt
