#!/bin/bash
R=$PWD; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "compress" 2>&1 | tail -5
timeout 200 python tests/hw/scalar_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/r2_scalar_latency_v2.txt
