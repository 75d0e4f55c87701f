#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "stream" 2>&1 | tail -5
timeout 300 python bench_configs.py --only stream --gib 2 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python bench_configs.py --only stream --gib 3 2>&1 | grep -v amdgpu.ids | tail -1
