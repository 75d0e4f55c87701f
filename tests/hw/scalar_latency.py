#!/usr/bin/env python3
"""Per-call latency of the scalar entry points (snap::raw::Encoder::compress /
Decoder::decompress through snapmi_raw_* = what the reference's own
`--features cpp` bench drives through snappy_compress / snappy_uncompress):
host buffer in, host buffer out, one stream per call.  Compared with the CPU
libsnappy 1.1.8 on the same inputs (the library those calls replace)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O
import rust_snappy_amd as R
enc, dec = R.raw.Encoder(), R.raw.Decoder()
print(f"{'input':18s} {'bytes':>8s} | GPU compress  ms   MB/s | GPU decompress ms   MB/s | libsnappy 1.1.8 compress / uncompress MB/s")
for name, data in O.corpus_round():
    comp = enc.compress_vec(data)
    assert comp == O.compress(data)
    def t(fn, reps=20):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps
    tc = t(lambda: enc.compress_vec(data))
    td = t(lambda: dec.decompress_vec(comp))
    row = f"{name:18s} {len(data):8d} | {tc*1e3:12.3f} {len(data)/tc/1e6:7.0f} | {td*1e3:14.3f} {len(data)/td/1e6:7.0f}"
    if O.libsnappy() is not None:
        sc = t(lambda: O.libsnappy_compress(data)); sd = t(lambda: O.libsnappy_uncompress(comp))
        row += f" | {len(data)/sc/1e6:7.0f} / {len(data)/sd/1e6:7.0f}"
    print(row, flush=True)
