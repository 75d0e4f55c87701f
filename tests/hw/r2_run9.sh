#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_tools.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
D=/dev/shm
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import oracle_lib as O
one = b"".join(d for _, d in O.corpus_round())
with open('/dev/shm/szip_in.bin','wb') as f:
    for _ in range(16 * (1 << 30) // len(one)):
        f.write(one)
PY
( for j in 1 2 3 4 6; do
    rm -f $D/szip_in.bin.sz
    ./tools/szip -k -v -j $j $D/szip_in.bin 2>&1 | sed "s/^/compress   -j $j: /"
  done
  cp $D/szip_in.bin.sz $D/copy.bin.sz
  for j in 1 2 3 4; do
    rm -f $D/copy.bin
    ./tools/szip -d -k -v -j $j $D/copy.bin.sz 2>&1 | sed "s/^/decompress -j $j: /"
  done
  cmp $D/copy.bin $D/szip_in.bin && echo "round trip identical (16 GiB)" ) | tee $O/r2_szip_pipeline.txt
rm -f $D/szip_in.bin* $D/copy.bin*
# where the frame encoder's time goes: kernel trace of cfg3 at 8 GiB
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_cfg3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 -o cfg3 -- python $R/bench_configs.py --only cfg3 --gib 8 --steps 3 > $O/prof_cfg3.log 2>&1
db=$(find $O/prof_cfg3 -name "*.db" | head -1)
python $R/profiles/db_stats.py $db > $O/r2_cfg3_kernel_stats.md; head -30 $O/r2_cfg3_kernel_stats.md
tail -1 $O/prof_cfg3.log
find $O/prof_cfg3 -type f -size +2M -delete
