#!/bin/bash
# k_stream_scan's window line size and the least number of hopping lanes a
# round goes on for (make hop_variants): one 2 GiB stream, decode ms
cd /root/repo
for v in rust-snappy_amd/variants/hop_*.so; do
  echo -n "$(basename $v .so)  "
  SNAPMI_TESTING=1 SNAPMI_LIB=$PWD/$v timeout 120 python bench_configs.py --plan stream:2 2>/dev/null | grep -o '"decompress_stream_ms": [0-9.]*'
done
