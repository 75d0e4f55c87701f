#!/bin/bash
R=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "decompress or decode or roundtrip or stream" 2>&1 | tail -2
echo -n "dec2 fixed :"; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras 2>&1 | grep "kernel ms per step" | sed 's/.*decompress://'
for v in NOTRIP NOSWEEP NOFLUSH; do
  echo -n "dec2 $v :"; SNAPMI_LIB=$R/rust-snappy_amd/variants/dec2_$v.so timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-extras --no-verify 2>&1 | grep "kernel ms per step" | sed 's/.*decompress://'
done
