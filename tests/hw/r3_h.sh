#!/bin/bash
# round 3, call H: pipelined third-generation decoder, waves per SIMD x window groups
mkdir -p gpurun_out
for v in w8g2 w7g2 w6g2 w7g4 w6g4; do
  echo -n "$v: " >> gpurun_out/r3_h.txt
  SNAPMI_LIB=$PWD/rust-snappy_amd/variants/dec3_$v.so timeout 200 python bench.py --no-extras --no-cpu --steps 6 --warmup 2 2>&1 >/dev/null | grep -o "decompress: .*" >> gpurun_out/r3_h.txt
done
cat gpurun_out/r3_h.txt
