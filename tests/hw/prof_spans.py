"""Per-phase cycle split of k_compress_spans (profile build: make -C
rust-snappy_amd/csrc profile)."""
import ctypes as C, os, sys
os.environ.setdefault("SNAPMI_LIB", "/root/repo/rust-snappy_amd/libsnapmi_profile.so")
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch, _lib
ctx = R.raw.Context(0)
ctx.set_option("compress_mode", 0)
ctx.set_option("small_batch_kernel", 0)
rnd = [d for _, d in O.corpus_round()]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sets = {"all": rnd * rounds}
for name, idx in (("html", 0), ("urls", 1), ("txt1", 6), ("txt4", 9), ("jpg", 2), ("pdf", 4), ("gaviota", 11)):
    sets[name] = [rnd[idx]] * (rounds * 4)
L = _lib.load()
L.snapmi_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
names = ["top", "bperm+x issue", "hash+xchg", "gather", "cmp+walk", "fixup", "long/extend", "window slide", "tail", "-"]
for k, streams in sets.items():
    src = batch.StreamBatch.from_bytes(streams)
    for _ in range(2):
        dst, lens, errs = batch.compress(ctx, src)
    t = ctx.last_timing()
    out = (C.c_uint64 * 16)()
    L.snapmi_debug_profile(ctx._h, out)
    v = list(out)
    tot = sum(v[:9]) + sum(v[13:16]); nb, nc, nblk = v[10], v[11], v[12]
    ub = sum(len(s) for s in streams)
    print(f"== {k}: {ub/2**30:.2f} GiB, codec {t['codec_ms']:.1f} ms -> {ub/2**30/(t['codec_ms']/1e3):.2f} GiB/s; blocks {nblk}, steps/blk {nb/max(nblk,1):.0f}, cycles/blk {tot/max(nblk,1)/1e6:.2f}M, cycles/step {tot/max(nb,1):.0f}")
    print("   " + "  ".join(f"{names[i]}={v[i]/max(nb,1):.0f}" for i in range(9)))
    print(f"   (cmp+walk = push/rest; before it: wait x,y + compare + ballots {v[13]/max(nb,1):.0f}, walk {v[14]/max(nb,1):.0f})")
