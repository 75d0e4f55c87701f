#!/bin/bash
R=$PWD
for v in ONETRIP W8 ONETRIP_W8; do
  echo -n "dec2 $v :"; SNAPMI_LIB=$R/rust-snappy_amd/variants/dec2_$v.so timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras 2>&1 | grep "kernel ms per step\|Error\|assert" | sed 's/.*decompress://'
done
