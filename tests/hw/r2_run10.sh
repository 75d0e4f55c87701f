#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 200 python tests/hw/scalar_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/r2_scalar_latency.txt
bash tests/hw/final_profile.sh r2_v1 2>&1 | tail -60
