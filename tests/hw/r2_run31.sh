#!/bin/bash
# encoder: literals of 17..64 bytes as 16-byte pieces, loads first
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py 2>gpurun_out/r31_bench.err | grep '^{"metric' > gpurun_out/r31_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r31_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline'])
print(json.dumps(d.get('extras'))[:3000])
PY
