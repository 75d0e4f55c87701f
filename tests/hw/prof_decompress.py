"""Experiment: per-phase cycle split of k_decompress_streams (profile build)."""
import ctypes as C, os, sys
os.environ["SNAPMI_LIB"] = "/root/repo/rust-snappy_amd/libsnapmi_profile.so"
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch, _lib
ctx = R.raw.Context(0)
rnd = [d for _, d in O.corpus_round()]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sets = {"all": rnd * rounds}
for name, idx in (("html", 0), ("urls", 1), ("txt4", 9), ("jpg", 2), ("pdf", 4), ("gtb", 11)):
    sets[name] = [rnd[idx]] * (rounds * 4)
L = _lib.load()
L.snapmi_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
names = ["loop", "load", "decode", "walk", "scan+chk", "compact", "map+fetch", "classify+load", "resolve+store", "end"]
for k, streams in sets.items():
    comp = batch.StreamBatch.from_bytes([O.compress(s) for s in streams[:12]] * (len(streams)//12) if k == "all" else [O.compress(streams[0])] * len(streams))
    for _ in range(2):
        dst, lens, errs = batch.decompress(ctx, comp)
    t = ctx.last_timing()
    out = (C.c_uint64 * 16)()
    L.snapmi_debug_profile(ctx._h, out)
    v = list(out)
    tot = sum(v[:10]); nw, npass, ne, nf, nr, nst = v[10:16]
    ub = sum(len(s) for s in streams)
    print(f"== {k}: {ub/2**30:.2f} GiB, codec {t['codec_ms']:.1f} ms -> {ub/2**30/(t['codec_ms']/1e3):.2f} GiB/s; streams {nst}, windows {nw}, elem/win {ne/nw:.1f}, pass/win {npass/nw:.2f}, resolve iters/pass {nr/max(npass,1):.2f}, fences {nf}, cycles/win {tot/nw:.0f}")
    print("   per window: " + "  ".join(f"{names[i]}={v[i]/nw:.0f}" for i in range(10)))
