#!/bin/bash
./tests/hw/lds_unaligned
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "decompress or decode or roundtrip or stream or lds" 2>&1 | tail -3
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-extras 2>&1 | grep "kernel ms per step\|probe" | cut -c1-400
