import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch
ctx = R.raw.Context(0)
for data in [b"a" * 120, b"hello world" * 10, (O.CORPUS / "html").read_bytes()[:20000], (O.CORPUS / "html").read_bytes()]:
    src = batch.StreamBatch.from_bytes([data])
    print("compress", len(data), flush=True)
    dst, lens, errs = batch.compress(ctx, src)
    print(" ->", lens, errs, dst.stream_bytes(0, lens[0]) == O.compress(data), flush=True)
