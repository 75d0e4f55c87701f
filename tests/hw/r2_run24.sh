#!/bin/bash
for side in 1 0 1 0; do
  echo -n "crc side stream $side: "
  SNAPMI_FRAME_CRC_SIDE=$side timeout 300 python bench_configs.py --only cfg3 --gib 16 --steps 3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['encode_ms'], d['frame_encode_gibs'])"
done
