#!/bin/bash
# round 3, call O: slice sizes of the pipelined host-buffer frame calls
mkdir -p gpurun_out
for dc in 4096 16384; do for es in 1073741824 2147483648 3000000000; do
  echo -n "decode chunks $dc encode slice $es: " >> gpurun_out/r3_o.txt
  SNAPMI_TESTING=1 SNAPMI_HOST_DECODE_CHUNKS=$dc SNAPMI_HOST_ENCODE_SLICE=$es timeout 300 python bench_configs.py --plan pcie:4 2>/dev/null | grep -o '"frame_encode_gibs.*decode_ms": [0-9.]*' >> gpurun_out/r3_o.txt
done; done
timeout 300 python bench_configs.py --plan adapters:4 2>&1 | tail -2 >> gpurun_out/r3_o.txt
cat gpurun_out/r3_o.txt
