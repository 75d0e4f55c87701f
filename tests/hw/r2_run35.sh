#!/bin/bash
# A/B in one box: lane tables from an uncached allocation
for i in 1 2; do
  for u in 0 1; do
    echo "== uncached=$u"; SNAPMI_LANE_UNCACHED=$u timeout 600 python bench.py --no-extras --no-cpu --steps 10 2>&1 | grep -o '"kernel_ms": {[^}]*}\|probe ms.*\|parity.*'
  done
done
