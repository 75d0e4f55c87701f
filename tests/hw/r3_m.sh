#!/bin/bash
# round 3, call M: frame adapters (reader errors, no waiting for full batches)
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_frame.py -m gpu -x -q > gpurun_out/r3_m_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_m_tests.log
tail -30 gpurun_out/r3_m_tests.log
