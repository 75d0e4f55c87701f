#!/bin/bash
# round 3, call P: host pipeline without copy-engine round trips on the kernel stream
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_tools.py -m gpu -x -q > gpurun_out/r3_p_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_p_tests.log
tail -3 gpurun_out/r3_p_tests.log
for dc in 1024 4096 16384; do
  echo -n "decode chunks $dc: " >> gpurun_out/r3_p.txt
  SNAPMI_TESTING=1 SNAPMI_HOST_DECODE_CHUNKS=$dc SNAPMI_HOST_ENCODE_SLICE=2147483648 timeout 300 python bench_configs.py --plan pcie:4 2>/dev/null | grep -o '"frame_encode_gibs.*decode_ms": [0-9.]*' >> gpurun_out/r3_p.txt
done
cat gpurun_out/r3_p.txt
