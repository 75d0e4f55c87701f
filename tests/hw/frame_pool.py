"""cfg3's encoder against the token pool: the synthetic text of SURVEY 8d,
[gib] GiB as one framed stream, five snapmi_frame_compress calls on a fresh
context at the default token_pool_pct and at 100 - call ms, blocks spilled in
the call's last launch, pool pages, what the pool has grown to.
usage: python tests/hw/frame_pool.py [gib]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from rust_snappy_amd import frame, raw  # noqa: E402

dev = torch.device("cuda", 0)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
period = B.synth_text(dev, 1 << 30)
data = period.repeat(int(gib))
for pct in (0, 100, 0):
    c = raw.Context(0)
    c.set_option("lane_table_budget_pct", 75)
    if pct:
        c.set_option("token_pool_pct", pct)
    print(f"## {gib:g} GiB of text framed, token_pool_pct {pct or 'default'}")
    for call in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, flen, index = frame.compress_device(c, data, want_index=False)
        c.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        print(f"call {call}: {ms:8.2f} ms {gib / ms * 1e3:6.1f} GiB/s  "
              f"spilled {c.info('token_blocks_spilled'):6d}  pool "
              f"{c.info('token_pool_pages'):8d} pages (now "
              f"{c.info('token_pool_pct_now')} %), asked "
              f"{c.info('token_pages_asked'):8d}  context "
              f"{c.info('scratch_bytes') / 1e9:.2f} GB  {c.last_kernel()}",
              flush=True)
        del out
    c.close()
    torch.cuda.empty_cache()
