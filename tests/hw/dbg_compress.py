import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as O
import rust_snappy_amd as R
from rust_snappy_amd import batch
ctx = R.raw.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else 'urls.10K'
data = (O.CORPUS / name).read_bytes()
# per 64 KiB block, as independent streams, to localise
blocks = [data[i:i + 65536] for i in range(0, len(data), 65536)]
src = batch.StreamBatch.from_bytes(blocks)
dst, lens, errs = batch.compress(ctx, src)
for i, b in enumerate(blocks):
    want = O.compress(b)
    got = dst.stream_bytes(i, lens[i])
    if got != want:
        j = next(k for k in range(min(len(got), len(want))) if got[k] != want[k])
        print(f"block {i}: len got {len(got)} want {len(want)} first diff at {j}")
        print(" got ", got[max(0,j-12):j+20].hex())
        print(" want", want[max(0,j-12):j+20].hex())
        # decode want up to j to find element context
        pos = 3 if len(b) >= 16384 else (2 if len(b) >= 128 else 1)
        d = 0
        last = None
        while pos <= j:
            t = want[pos]
            start = pos
            if t & 3 == 0:
                ln = (t >> 2) + 1
                pos += 1
                if ln > 60:
                    nb = ln - 60
                    ln = int.from_bytes(want[pos:pos+nb], 'little') + 1
                    pos += nb
                last = ('lit', ln, start, d)
                pos += ln; d += ln
            elif t & 3 == 1:
                ln = 4 + ((t >> 2) & 7); off = ((t >> 5) << 8) | want[pos+1]
                last = ('copy1', ln, off, start, d); pos += 2; d += ln
            else:
                ln = (t >> 2) + 1; off = want[pos+1] | (want[pos+2] << 8)
                last = ('copy2', ln, off, start, d); pos += 3; d += ln
        print(" element containing diff:", last)
        break
else:
    print("all blocks equal")
