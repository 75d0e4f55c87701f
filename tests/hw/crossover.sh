#!/bin/bash
# lane kernel vs wavefront kernel by batch size (GiB of the cfg2 corpus)
for g in 0.25 0.5 1 2 4; do
  for m in waves lanes; do
    echo -n "gib=$g $m :"
    SNAPMI_TESTING=1 SNAPMI_LANE_MIN_BLOCKS=1 SNAPMI_COMPRESS=$m timeout 150 python bench.py --gib $g --steps 3 --warmup 1 --no-cpu 2>&1 | grep "kernel ms per step" | sed 's/.*per step://; s/|.*//'
  done
done
