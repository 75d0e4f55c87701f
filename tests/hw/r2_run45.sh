#!/bin/bash
# decoder: ring source as one unaligned ds_read_b128 instead of two ds_read_b64
for v in default rd128 default rd128; do
  lib=$PWD/rust-snappy_amd/libsnapmi.so; [ $v != default ] && lib=$PWD/rust-snappy_amd/variants/dec_$v.so
  echo -n "$v: "; SNAPMI_LIB=$lib timeout 300 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 2>&1 | grep -o 'parity ok\|"decompress": [0-9.]*}' | tr '\n' ' '; echo
done
