#!/bin/bash
# round 3, call N: pipelined host-buffer frame calls: frame tests, then pcie + adapters extras
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_tools.py tests/test_gpu_reference_suite.py -m gpu -x -q > gpurun_out/r3_n_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_n_tests.log
tail -5 gpurun_out/r3_n_tests.log
timeout 600 python bench_configs.py --plan pcie:4,adapters:4 > gpurun_out/r3_n_pcie.json 2> gpurun_out/r3_n_pcie.log
cat gpurun_out/r3_n_pcie.json; tail -3 gpurun_out/r3_n_pcie.log
