#!/bin/bash
# usage: r5_profile.sh   (on the GPU box, from the repo root)
# everything profiles/r5_* holds that is not a one-off experiment: the default
# bench line + rocprofv3 kernel stats of the same workload + scalar latency +
# one 2 GiB stream (final_profile.sh), then the window kernel (SQ counters,
# phase counters of the profile build, the sweep by size), the small-block
# kernel, and the timeline of a 64 MiB decompress
R=$PWD
bash tests/hw/final_profile.sh r5_v1 > $R/gpurun_out/final_profile_r5.log 2>&1
bash tests/hw/pmc_spans.sh > $R/gpurun_out/pmc_spans_r5.txt 2>&1
{ for l in libsnapmi_profile.so libsnapmi_profile2.so; do echo "== $l"; SNAPMI_LIB=$R/rust-snappy_amd/$l timeout 300 python tests/hw/prof_spans.py 60 2>&1 | grep -v amdgpu.ids; done; } > $R/gpurun_out/prof_spans_r5.txt
SNAPMI_TESTING=1 timeout 300 python tests/hw/span_ab.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/span_ab_r5.txt
timeout 400 python tests/hw/small_blocks.py 2>&1 | grep -v amdgpu.ids > $R/gpurun_out/small_blocks_r5.txt
bash tests/hw/pmc_kernel.sh k_match_spans_8k python $R/tests/hw/small_blocks_one.py 8192 0.25 > $R/gpurun_out/pmc_small8k_r5.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o p -- python $R/tests/hw/sweep_one.py 0.0625 > $R/gpurun_out/tl_run.txt 2>&1
f=$(find $R/gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
{ tail -1 $R/gpurun_out/tl_run.txt; python $R/tests/hw/timeline.py $f k_long_plan; } > $R/gpurun_out/timeline_dec64_r5.txt
rm -rf $R/gpurun_out/prof_tl
cd $R
tail -3 gpurun_out/final_profile_r5.log
