#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
./tests/hw/lds_unaligned
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=15 > $O/r2d_pytest.log 2>&1; tail -30 $O/r2d_pytest.log
for k in 2 1; do
  SNAPMI_DECODE_KERNEL=$k timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-extras 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tail -2
done
