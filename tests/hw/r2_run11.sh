#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_frame.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15
timeout 600 python bench_configs.py --only cfg3 --gib 64 2>&1 | grep -v amdgpu.ids | tail -2
