#!/bin/bash
# A/B in one box: direct encode (csize in the match kernel) vs the old path
mkdir -p gpurun_out
for i in 1 2; do
  echo "== direct"; timeout 600 python bench.py --no-extras --steps 10 2>&1 | grep -o '"kernel_ms": {[^}]*}\|probe ms.*'
  echo "== old";    SNAPMI_LANE_DIRECT=0 SNAPMI_LIB=$PWD/rust-snappy_amd/variants/nocsize.so timeout 600 python bench.py --no-extras --steps 10 2>&1 | grep -o '"kernel_ms": {[^}]*}\|probe ms.*'
done
