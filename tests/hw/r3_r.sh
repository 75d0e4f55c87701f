#!/bin/bash
# round 3, call R: whole GPU suite after the option split / placement budget / host pipeline
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_r_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_r_tests.log
tail -15 gpurun_out/r3_r_tests.log
