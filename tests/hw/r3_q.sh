#!/bin/bash
# round 3, call Q: cfg3 with the SURVEY generator; bench.py's own PMC passes (2 GiB)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_frame.py -m gpu -x -q -k "cfg3 or chunks_on_device" > gpurun_out/r3_q_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_q_tests.log
tail -3 gpurun_out/r3_q_tests.log
timeout 300 python bench_configs.py --plan cfg3:8 2>&1 | tail -1 | tee gpurun_out/r3_q_cfg3.json
timeout 600 python bench.py --gib 2 --steps 3 --warmup 1 --no-cpu --no-extras 2> gpurun_out/r3_q_bench.log | tee gpurun_out/r3_q_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']); print(d['roofline_decompress'])"
tail -3 gpurun_out/r3_q_bench.log
