#!/bin/bash
# usage: ab_old_new.sh <out dir> <pairs>   (on the GPU box, from the repo root)
# Same-box A/B of two builds of the library: _ab_old/ holds another commit of
# this repository, built (git archive <commit> | tar -x -C _ab_old; make -C
# _ab_old/rust-snappy_amd/csrc).  Alternating processes, each under rocprofv3:
# bench.py's workload through tests/hw/segment_ab.py, the placement probe of
# the context's lane tables (what a process is handed differs by 5 %), and
# the kernels' average durations.
out=$1; pairs=${2:-3}
R=$PWD
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
for i in $(seq $pairs); do
  for side in old new; do
    d=$R; [ $side = old ] && d=$R/_ab_old
    rm -rf /tmp/ab_$side
    (cd $d && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ab_$side -o $side -- python tests/hw/segment_ab.py 8 1 2>&1 | grep "segments of" | cut -c1-60,130-400 | sed "s/^/$side: /") >> $R/$out/ab.txt
    db=$(find /tmp/ab_$side -name "*.db" | head -1)
    python $R/profiles/db_stats.py $db | grep -E "k_match_both|k_match_spans|k_match_blocks|k_encode_tokens|k_redo" | cut -d"|" -f2,3,5 | sed "s/^/$side: /" >> $R/$out/ab.txt
  done
done
cat $R/$out/ab.txt
