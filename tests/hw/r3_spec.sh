#!/bin/bash
# k_match_blocks_spec (lane-kernel launches with blocks <= lanes also fetch the next probe's entry):
# the compress side of the GPU suite, the per-file rates at 2 GiB (32768 blocks each: the speculating kernel), the rest
mkdir -p gpurun_out
F=gpurun_out/r3_spec.txt
: > $F
filt() { grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_suite.py -x -q -k "compress or lane or atomic or small or tiny or golden or corpus or press" 2>&1 | filt | tail -4 | tee -a $F
timeout 40 python bench_configs.py --plan files:2 2>/dev/null | grep "^{" | tee -a $F
timeout 45 python -m pytest tests/test_gpu_frame.py tests/test_gpu_tools.py tests/test_gpu_multi.py -x -q 2>&1 | filt | tail -3 | tee -a $F
