#!/bin/bash
# lane-kernel ms (first timed step) of N fresh processes of `python bench.py`
echo -n "modes :"
for i in $(seq ${1:-6}); do
  timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu --no-extras 2>&1 | grep "kernel ms per step" | sed 's/.*per step://; s/|.*//' | awk '{printf " %s", $1}'
done; echo
