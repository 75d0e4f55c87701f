#!/bin/bash
# round 3, call L: tests of the reference branches that had none, the context pool of the snappy C API
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "too_big or 4_gib or short_buffer or many_threads" > gpurun_out/r3_l_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_l_tests.log
tail -30 gpurun_out/r3_l_tests.log
