#!/bin/bash
# round 3, call G (software pipeline: parse N+1 before copy N): k_decompress_streams3 (parity + time)
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_g_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_g_tests.log
tail -3 gpurun_out/r3_g_tests.log
timeout 200 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 > gpurun_out/r3_g_bench.json 2> gpurun_out/r3_g_bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_g_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('kernels'), d.get('passes'))
PY
