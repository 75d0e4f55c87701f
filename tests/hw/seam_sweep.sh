#!/bin/bash
# tests/seam_consumer.c against libsnapmi.so (as libsnappy.so): MB/s per bench
# input with 1 / 4 / 16 / 64 callers, for 2 / 4 / 8 contexts in the seam's pool
R=$PWD
d=$(mktemp -d)
ln -s $R/rust-snappy_amd/libsnapmi.so $d/libsnappy.so
gcc -O2 -I/opt/conda/include -o $d/consumer $R/tests/seam_consumer.c -L$d -lsnappy -lpthread -Wl,-rpath,$d -Wl,-rpath,$R/rust-snappy_amd
mkdir $d/in
python - "$d/in" <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
for name, data in O.corpus_round():
    if name in ("zflat00_html", "zflat06_txt1", "zflat01_urls", "zflat11_gaviota"):
        open(f"{sys.argv[1]}/{name}.in", "wb").write(data)
        open(f"{sys.argv[1]}/{name}.snappy", "wb").write(O.compress(data))
PY
for ctxs in 2 4 8; do for t in 1 4 16 64; do
  echo "== contexts $ctxs callers $t"
  SNAPMI_SEAM_CONTEXTS=$ctxs $d/consumer bench $d/in $t 150 2>&1 | grep -v amdgpu.ids
done; done
rm -rf $d
