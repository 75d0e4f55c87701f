#!/bin/bash
# lane-per-stream path of the decoder for tiny streams: whole GPU suite, then tiny streams and cfg2
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4 | tee gpurun_out/tiny_dec.txt
timeout 200 python bench_configs.py --plan tiny:2 2>/dev/null | grep "^{" | tee -a gpurun_out/tiny_dec.txt
timeout 200 python bench.py --no-extras --no-cpu --no-pmc --steps 6 --warmup 2 2>&1 >/dev/null | tail -1 | tee -a gpurun_out/tiny_dec.txt
