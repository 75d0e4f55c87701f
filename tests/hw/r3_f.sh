#!/bin/bash
# round 3, call F: 256 compressed bytes per window (kG3 = 4) against 128: parity (decoder tests) + time
mkdir -p gpurun_out
export SNAPMI_LIB=$PWD/rust-snappy_amd/variants/dec3_g4.so
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_suite.py -m gpu -x -q -k "decomp or decode or roundtrip or foreign or long_stream or error" > gpurun_out/r3_f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_f_tests.log
tail -3 gpurun_out/r3_f_tests.log
timeout 200 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 > gpurun_out/r3_f_bench.json 2> gpurun_out/r3_f_bench.log
tail -1 gpurun_out/r3_f_bench.log
