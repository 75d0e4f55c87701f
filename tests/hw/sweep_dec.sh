#!/bin/bash
# usage: sweep_dec.sh "<variant .so names>" [repeats] [steps]: decompress ms per step
R=${2:-2}; K=${3:-4}
for v in $1; do
  echo -n "$v :"
  for i in $(seq $R); do
  SNAPMI_LIB=$PWD/rust-snappy_amd/variants/$v.so timeout 150 python bench.py --steps $K --warmup 1 --no-cpu 2>&1 | grep "kernel ms per step" | sed 's/.*decompress://' | tr '\n' '|'
  done; echo
done
