#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 600 python tests/hw/placement_probe2.py > $O/r2_placement2.txt 2>&1; grep -v amdgpu.ids $O/r2_placement2.txt
