#!/bin/bash
# round 3, call I: third generation with one copy step + late elements in order; 128- and 256-byte windows
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/r3_i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_i_tests.log
tail -3 gpurun_out/r3_i_tests.log
timeout 200 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 2>&1 >/dev/null | grep -o "decompress: .*" > gpurun_out/r3_i.txt
SNAPMI_LIB=$PWD/rust-snappy_amd/variants/dec3_g4.so timeout 200 python bench.py --no-extras --no-cpu --steps 8 --warmup 2 2>&1 >/dev/null | grep -o "decompress: .*" >> gpurun_out/r3_i.txt
cat gpurun_out/r3_i.txt
