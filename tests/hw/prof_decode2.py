"""Per-file decode time of the decoder kernels (second and third generation) and their
counters (profile build: make -C rust-snappy_amd/csrc profile)."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
os.environ["SNAPMI_LIB"] = str(ROOT / "rust-snappy_amd" / "libsnapmi_profile.so")
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch, _lib
rnd = [d for _, d in O.corpus_round()]
names = [b for b, _ in O.corpus_round()]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = _lib.load()
L.snapmi_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
sets = {"all": None}
for i in (0, 1, 2, 4, 6, 9, 10, 11):
    sets[names[i]] = i
for k, idx in sets.items():
    if idx is None:
        comp = batch.StreamBatch.from_bytes([O.compress(s) for s in rnd] * rounds)
        ub = sum(len(s) for s in rnd) * rounds
    else:
        comp = batch.StreamBatch.from_bytes([O.compress(rnd[idx])] * (rounds * 6))
        ub = len(rnd[idx]) * rounds * 6
    row = f"{k:18s} {ub/2**30:5.2f} GiB:"
    for kern in (2, 3):
        ctx = R.raw.Context(0)
        ctx.set_option("decode_kernel", kern)
        for _ in range(2):
            dst, lens, errs = batch.decompress(ctx, comp)
        t = ctx.last_timing()["codec_ms"]
        row += f"  k{kern} {t:7.2f} ms {ub/2**30/(t/1e3):7.1f} GiB/s"
        out = (C.c_uint64 * 16)()
        L.snapmi_debug_profile(ctx._h, out)
        v = list(out)
        nw = max(v[10] + v[7], 1)
        row += (f" | windows {v[7]}+{v[10]} runs/win {v[8]/nw:.2f} trips/win {v[11]/nw:.2f} elem/win {v[12]/nw:.1f} "
                f"swept/win {v[14]/nw:.2f} far/win {v[9]/nw:.2f} fences/win {v[13]/nw:.3f} out/win {ub/nw:.0f}\n" + " " * 28)
        ctx.close()
    print(row, flush=True)
