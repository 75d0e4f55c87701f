#!/bin/bash
# round 3, call K: the N > 1 code on one GPU (two oversubscribed ranks) and snapmi_gatherv at world 1
mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -s > gpurun_out/r3_k_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_k_tests.log
tail -30 gpurun_out/r3_k_tests.log
