#!/bin/bash
# decoder occupancy after the instruction diet: waves per SIMD 8 (default) / 7 / 6 / 5
for w in 8 7 6 5; do
  lib=$PWD/rust-snappy_amd/libsnapmi.so; [ $w != 8 ] && lib=$PWD/rust-snappy_amd/variants/dec_w$w.so
  echo -n "waves $w: "; SNAPMI_LIB=$lib timeout 300 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 2>&1 | grep -o '"decompress": [0-9.]*}' | tail -1
done
