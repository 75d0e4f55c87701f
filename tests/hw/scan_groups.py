"""Segments per wavefront of the long-stream scan (StreamArgs::scan_segs;
test option stream_scan_segs) against the call it serves: bench.py's batch-size sweep at 64 / 256 MiB (decompress),
one 126 MB stream as a batch of one, and the scalar Decoder::decompress of
two corpus files.  Needs libsnapmi_test.so (SNAPMI_TESTING=1).
usage: SNAPMI_TESTING=1 python tests/hw/scan_groups.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
import oracle_lib as O  # noqa: E402
from rust_snappy_amd import batch, raw  # noqa: E402

dev = torch.device("cuda", 0)
ctx = raw.Context(0)
rnd = O.corpus_round()
comp_one = [O.compress(d) for _, d in rnd]
VARIANTS = (64, 32, 16, 8, 0)


def tiled(gib):
    mix = sum(len(d) for _, d in rnd)
    reps = max(1, int(gib * B.GIB / mix))
    cmix = b"".join(comp_one)
    data = torch.frombuffer(bytearray(cmix), dtype=torch.uint8).to(dev) \
        .repeat(reps)
    offs, lens, caps, pos = [], [], [], 0
    for _ in range(reps):
        for c, (_, d) in zip(comp_one, rnd):
            offs.append(pos)
            lens.append(len(c))
            caps.append(len(d))
            pos += len(c)
    src = batch.StreamBatch(data, np.array(offs, dtype=np.int64),
                            np.array(lens, dtype=np.int64))
    dst = batch.StreamBatch.empty(caps, dev)
    olens = torch.zeros(len(lens), dtype=torch.int64, device=dev)

    def dec():
        raw.decompress_batch(ctx, src.d_ptrs, src.d_lens, dst.d_ptrs,
                             dst.d_lens, olens, None)

    def check():
        for j, (_, d) in enumerate(rnd):
            assert dst.stream_bytes(len(lens) - 12 + j) == d, j
    return dec, check, sum(caps)


def one_stream(mb):
    blob = b"".join(d for _, d in rnd)
    reps = max(1, mb * 1000000 // len(blob))
    big = blob * reps
    comp = O.compress(big)
    cin = torch.frombuffer(bytearray(comp), dtype=torch.uint8).to(dev)
    out = torch.zeros(len(big), dtype=torch.uint8, device=dev)
    iptr = torch.tensor([cin.data_ptr()], dtype=torch.int64, device=dev)
    ilen = torch.tensor([len(comp)], dtype=torch.int64, device=dev)
    optr = torch.tensor([out.data_ptr()], dtype=torch.int64, device=dev)
    ocap = torch.tensor([len(big)], dtype=torch.int64, device=dev)
    olen = torch.zeros(1, dtype=torch.int64, device=dev)
    want = torch.frombuffer(bytearray(big), dtype=torch.uint8)

    def dec():
        raw.decompress_batch(ctx, iptr, ilen, optr, ocap, olen, None)

    def check():
        assert int(olen.item()) == len(big)
        assert torch.equal(out.cpu(), want)
    return dec, check, len(big)


def scalar(name):
    data = (O.CORPUS / name).read_bytes()
    comp = O.compress(data)
    d = raw.Decoder(ctx)

    def dec():
        dec.got = d.decompress_vec(comp)

    def check():
        assert dec.got == data
    return dec, check, len(data)


def timed(fn, reps):
    fn()
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


cases = (("sweep 64 MiB", tiled(1 / 16), 10), ("sweep 256 MiB", tiled(0.25), 6),
         ("batch of one, 126 MB", one_stream(126), 6),
         ("Decoder::decompress urls.10K", scalar("urls.10K"), 20),
         ("Decoder::decompress lcet10.txt", scalar("lcet10.txt"), 20))
print("# scan_segs (0 = by size, stream_scan_segs): ms per call, GiB/s")
for label, (dec, check, nbytes), reps in cases:
    row = []
    for ss in VARIANTS:
        ctx.set_test_option("stream_scan_segs", ss)
        t = timed(dec, reps)
        check()
        row.append(f"{ss:2d}: {t*1e3:7.3f} ms {nbytes/2**30/t:6.1f}")
    print(f"{label:32s} " + "  ".join(row), flush=True)
