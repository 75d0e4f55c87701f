#!/bin/bash
# whole GPU suite; call rate of the snappy C API from 1 and 8 threads
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12 | tee gpurun_out/r3_check_tests.log
