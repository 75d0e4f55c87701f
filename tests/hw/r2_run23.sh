#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "compress or frame or szip" 2>&1 | tail -4
for ov in 1 0; do
echo -n "overlap $ov: "; python - <<PY 2>&1 | grep -v amdgpu | tail -1
import os, sys, subprocess
env = dict(os.environ)
# (no env knob for the option: a tiny wrapper sets it through the API)
PY
done
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-extras 2>&1 | grep "kernel ms per step" | cut -c1-300
timeout 300 python bench_configs.py --only cfg3 --gib 16 2>&1 | grep -v amdgpu.ids | tail -1
