#!/bin/bash
# usage: sweep.sh "<variant .so names>" "<lane waves list>" [repeats] [steps]
# prints the lane-kernel ms of every timed step of every run
R=${3:-3}; K=${4:-4}
for v in $1; do for w in $2; do
  echo -n "$v waves=$w :"
  for i in $(seq $R); do
  SNAPMI_LIB=$PWD/rust-snappy_amd/variants/$v.so SNAPMI_LANE_WAVES=$w timeout 150 python bench.py --steps $K --warmup 1 --no-cpu 2>&1 | grep "kernel ms per step" | sed 's/.*per step://; s/|.*//' | tr '\n' '|'
  done; echo
done; done
