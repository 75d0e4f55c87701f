#!/bin/bash
# plan_class (tiny compressed streams with a large output go to the wavefront decoder): GPU suite, then 2 GiB of
# sparse 4 KiB pages (196 compressed bytes each) with the old plan (variants/plan_old.so, built from the commit before)
# and the new one, and the 200-byte streams (the plan now reads every tiny stream's header)
R=$PWD
mkdir -p gpurun_out
F=gpurun_out/r3_plan_class.txt
: > $F
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | tee -a $F
for lib in variants/plan_old.so libsnapmi.so; do
  for size in 0 200; do
    echo -n "$lib " | tee -a $F
    SNAPMI_LIB=$R/rust-snappy_amd/$lib timeout 120 python tests/hw/small_streams.py $size 2 2>&1 | grep "^{\|Error\|error" | tail -1 | tee -a $F
  done
done
