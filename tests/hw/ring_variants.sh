#!/bin/bash
# the decoder with a 4 / 8 / 16 KiB LDS ring at 8 / 6 / 4 / 2 wavefronts per
# SIMD (make -C rust-snappy_amd/csrc ring_variants): cfg2 decompress ms and the
# kernel's PMC traffic (bench.py's own FETCH_SIZE / WRITE_SIZE child runs)
cd /root/repo
for v in default ring4k_w6 ring4k_w4 ring8k_w4 ring16k_w2; do
  lib=rust-snappy_amd/variants/$v.so
  [ $v = default ] && lib=rust-snappy_amd/libsnapmi.so
  SNAPMI_LIB=$PWD/$lib timeout 400 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); r=j['roofline_decompress']
        print('$v', 'decompress_ms', r['avg_launch_ms'], 'traffic_GB', (r['traffic'] or 0)/1e9, 'measured', r['traffic_measured'], 'compress_ms', j['kernel_ms']['compress'])
"
done
