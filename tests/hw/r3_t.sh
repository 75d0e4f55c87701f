#!/bin/bash
# round 3, call T: many-workgroup plan / scan / sort kernels: parity, then the tiny-stream regime again
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_small or budget or tiled_corpus or reference" > gpurun_out/r3_t_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_t_tests.log
tail -4 gpurun_out/r3_t_tests.log
timeout 300 python bench_configs.py --plan tiny:2,cfg5:8 2>/dev/null | grep "^{" | tee gpurun_out/r3_t_tiny.json
