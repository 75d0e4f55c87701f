#!/bin/bash
# k_compress_tiny (streams under 256 bytes, one per lane, all state in LDS): the GPU suite,
# then 2 GiB of 200-byte streams with the kernel on and off, and the kernel's own duration
R=$PWD
mkdir -p gpurun_out
F=gpurun_out/tiny_comp.txt
: > $F
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | tee -a $F
for v in 1 0; do
  echo "tiny_stream_kernel=$v" | tee -a $F
  timeout 200 python bench_configs.py --plan tiny:2 --option tiny_stream_kernel=$v 2>/dev/null | grep "^{" | tee -a $F
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tiny
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tiny -o t -- python $R/bench_configs.py --plan tiny:1 > $R/gpurun_out/prof_tiny.log 2>&1
db=$(find $R/gpurun_out/prof_tiny -name "*.db" | head -1)
python $R/profiles/db_stats.py $db | head -14 | tee -a $R/$F
rm -rf $R/gpurun_out/prof_tiny
