"""Round 5's driver record held one row nobody could explain: extras.sweep
4 GiB compress at 2 308 ms per call (59 ms on the builder's boxes).  This is
that row call by call: bench_configs.sweep's sizes on ONE fresh context, every
compress call timed on its own (wall clock around call + synchronize, and the
library's own event timing), the placement log, and hipMemGetInfo around it.

  python tests/hw/sweep_repro.py [repeats] [name=value ...]   (options of the context)
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench_configs as B  # noqa: E402


def one(rep, options):
    from rust_snappy_amd import _lib, raw
    dev = torch.device("cuda", 0)
    own = raw.Context(0)
    for item in options:
        name, value = item.split("=")
        own.set_option(name, int(value))
    for label, gib in (("64MiB", 1 / 16), ("256MiB", 0.25), ("1GiB", 1.0),
                       ("4GiB", 4.0)):
        calls = []
        orig = raw.compress_batch

        def timed(ctx, *a, **k):
            torch.cuda.synchronize()
            f0 = torch.cuda.mem_get_info(dev)[0]
            t0 = time.perf_counter()
            orig(ctx, *a, **k)
            t1 = time.perf_counter()
            ctx.synchronize()
            t2 = time.perf_counter()
            tm = ctx.last_timing()
            calls.append({"call_ms": round((t1 - t0) * 1e3, 2),
                          "sync_ms": round((t2 - t1) * 1e3, 2),
                          "kernels_ms": round(tm["total_ms"], 2),
                          "kernel": ctx.last_kernel(),
                          "free_before_gib": round(f0 / 2**30, 1),
                          "free_after_gib": round(
                              torch.cuda.mem_get_info(dev)[0] / 2**30, 1)})
        raw.compress_batch = timed
        try:
            t0 = time.perf_counter()
            ub, cb, n, te, td = B.round_tiles(own, dev, gib, 3)
            wall = time.perf_counter() - t0
        finally:
            raw.compress_batch = orig
        row = {"rep": rep, "size": label, "wall_s": round(wall, 2),
               "compress_ms_as_bench_reports": round(te * 1e3, 2),
               "decompress_ms": round(td * 1e3, 2),
               "probe_log": _lib.load().snapmi_table_probe_log(own._h).decode(),
               "calls": calls}
        print(json.dumps(row), flush=True)
        torch.cuda.empty_cache()
    own.close()
    torch.cuda.empty_cache()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
    opts = [a for a in sys.argv[1:] if "=" in a]
    import __graft_entry__ as g
    g.build()
    for r in range(reps):
        one(r, opts)
