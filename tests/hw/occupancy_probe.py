"""Experiment: how many k_compress_blocks workgroups (32 KiB LDS each) are
resident per CU?  Identical blocks; time vs grid size shows the step."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch
ctx = R.raw.Context(0)
blk = (O.CORPUS / "plrabn12.txt").read_bytes()[:65536]
for n in (256, 512, 768, 1024, 1025, 1280, 1281, 1536, 2048, 2560, 5120):
    src = batch.StreamBatch.from_bytes([blk] * n)
    best = 1e9
    for _ in range(3):
        batch.compress(ctx, src)
        best = min(best, ctx.last_timing()["codec_ms"])
    print(f"blocks {n:5d}: {best:8.3f} ms  ({best / ((n + 1279) // 1280):.3f} per 1280-round, {best / ((n + 1023) // 1024):.3f} per 1024-round)")
