import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch
ctx = R.raw.Context(0)
html = (O.CORPUS / "html").read_bytes()
for streams in [[b"a" * 120], [html], [html] * 700, [d for _, d in O.corpus_round()] * 40]:
    src = batch.StreamBatch.from_bytes(streams)
    print("compress", len(streams), flush=True)
    dst, lens, errs = batch.compress(ctx, src)
    ok = all(dst.stream_bytes(i, lens[i]) == O.compress(s) for i, s in list(enumerate(streams))[:24])
    print(" ->", ok, ctx.last_timing()["codec_ms"], flush=True)
