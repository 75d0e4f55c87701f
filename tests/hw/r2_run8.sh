#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > $O/r2e_pytest.log 2>&1; tail -25 $O/r2e_pytest.log
# szip host pipeline: 4 GiB of the corpus in a RAM file system
D=/dev/shm; df -h $D | tail -1
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import oracle_lib as O
one = b"".join(d for _, d in O.corpus_round())
with open('/dev/shm/szip_in.bin','wb') as f:
    for _ in range(4 * (1 << 30) // len(one)):
        f.write(one)
PY
ls -la $D/szip_in.bin
( for j in 1 2 3 4; do
    rm -f $D/szip_in.bin.sz
    ./tools/szip -k -v -j $j $D/szip_in.bin 2>&1 | sed "s/^/compress   -j $j: /"
  done
  cp $D/szip_in.bin.sz $D/copy.bin.sz
  for j in 1 2 3; do
    rm -f $D/copy.bin
    ./tools/szip -d -k -v -j $j $D/copy.bin.sz 2>&1 | sed "s/^/decompress -j $j: /"
  done
  cmp $D/copy.bin $D/szip_in.bin && echo "round trip identical" ) | tee $O/r2_szip_pipeline.txt
rm -f $D/szip_in.bin* $D/copy.bin*
