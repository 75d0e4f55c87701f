#!/bin/bash
# decoder: uniform-base addressing, no zero-inits
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python bench.py --no-extras 2>gpurun_out/r43_bench.err | grep '^{"metric' > gpurun_out/r43_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r43_bench.json'))
print(d['value'], d['kernel_ms'], d['compress_gibs'])
PY
