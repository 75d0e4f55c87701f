#!/bin/bash
# after k_compress_tiny / k_compress_small / k_decompress_streams3_many: GPU suite, then the tiny + small-stream plan
# (with the small-stream kernel on and off); "files" adds the per-file rates at 8 GiB
R=$PWD
mkdir -p gpurun_out
F=gpurun_out/r3_late.txt
: > $F
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | tee -a $F
timeout 300 python bench_configs.py --plan tiny:2 2>/dev/null | grep "^{" | tee -a $F
echo "small_stream_kernel=0" | tee -a $F
timeout 300 python bench_configs.py --plan tiny:2 --option small_stream_kernel=0 2>/dev/null | grep "^{" | tee -a $F
if [ "$1" = files ]; then timeout 400 python bench_configs.py --plan files:8 2>/dev/null | grep "^{" | tee -a $F; fi
