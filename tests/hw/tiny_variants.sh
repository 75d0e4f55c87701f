#!/bin/bash
# GPU suite on the default build, then the tiny + small-stream plan on the default (k_compress_tiny stores its output
# itself: four wavefronts per CU) and on variants/tiny_s.so (output column in LDS: three)
R=$PWD
mkdir -p gpurun_out
F=gpurun_out/tiny_variants.txt
: > $F
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | tee -a $F
echo "default (output straight to the caller's buffer)" | tee -a $F
timeout 300 python bench_configs.py --plan tiny:2 2>/dev/null | grep "^{" | tee -a $F
echo "tiny_s (output column in LDS)" | tee -a $F
SNAPMI_LIB=$R/rust-snappy_amd/variants/tiny_s.so timeout 300 python bench_configs.py --plan tiny:2 2>/dev/null | grep "^{" | tee -a $F
