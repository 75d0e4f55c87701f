#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_frame.py -m gpu -q -p no:cacheprovider -x -k "chunks_on_device" 2>&1 | tail -2
for seg in 262144 73350 36675; do
  echo -n "lane_segment_blocks $seg: "
  SNAPMI_LANE_SEGMENT_BLOCKS=$seg timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu --no-extras 2>&1 | grep "kernel ms per step" | sed 's/.*per step://; s/|.*//'
done
