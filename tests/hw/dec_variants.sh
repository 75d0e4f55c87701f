#!/bin/bash
# decoder build variants (rust-snappy_amd/variants/dec3_*.so) against the default: cfg2 decompress ms per step
mkdir -p gpurun_out; rm -f gpurun_out/dec_variants.txt
for v in default "$@"; do
  lib=$PWD/rust-snappy_amd/libsnapmi.so; [ $v != default ] && lib=$PWD/rust-snappy_amd/variants/dec3_$v.so
  echo -n "$v: " >> gpurun_out/dec_variants.txt
  SNAPMI_LIB=$lib timeout 200 python bench.py --no-extras --no-cpu --no-pmc --steps 6 --warmup 2 2>&1 >/dev/null | grep -o "decompress: .*" >> gpurun_out/dec_variants.txt
done
cat gpurun_out/dec_variants.txt
