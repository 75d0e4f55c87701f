import ctypes as C, os, sys
os.environ["SNAPMI_LIB"] = "/root/repo/rust-snappy_amd/libsnapmi_dbg.so"
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch, _lib
ctx = R.raw.Context(0)
L = _lib.load(); L.snapmi_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
src = batch.StreamBatch.from_bytes([b"a" * 120])
dst, lens, errs = batch.compress(ctx, src)
out = (C.c_uint64 * 16)(); L.snapmi_debug_profile(ctx._h, out)
for w in range(5):
    print("wave", w, [(v & 0xffffffff, v >> 32) for v in list(out)[w*3:w*3+3]])
print(lens, errs)
