#!/bin/bash
# usage: ab_opts.sh <out dir> <pairs> <AB_OPTS>  - ab_old_new.sh without the
# profiler, with options for both sides (name=value,name=value)
out=$1; pairs=${2:-3}; export AB_OPTS=$3
R=$PWD
mkdir -p $R/$out
for i in $(seq $pairs); do
  for side in old new; do
    d=$R; [ $side = old ] && d=$R/_ab_old
    (cd $d && timeout 300 python tests/hw/segment_ab.py 8 1 2>&1 | grep "segments of" | cut -c1-60,130-400 | sed "s/^/$side [$AB_OPTS]: /") >> $R/$out/ab_opts.txt
  done
done
