"""GPU suite: the HIP path, called through the C ABI, against the oracle.

Bit-exact is the bar: compressed bytes equal the oracle's bytes, decompressed
bytes equal the original, error variants and field values equal the
reference's (test/tests.rs:345-466)."""
import hashlib
import time
import random

import numpy as np
import pytest

import kats
from conftest import ROOT, SCAN_GEOMETRIES, set_scan_geometry
import oracle_lib as O

pytestmark = pytest.mark.gpu


def gpu_compress(ctx, streams):
    from rust_snappy_amd import batch
    src = batch.StreamBatch.from_bytes(streams)
    dst, lens, errs = batch.compress(ctx, src)
    out = []
    for i in range(len(streams)):
        assert errs[i][0] == 0, (i, errs[i])
        out.append(dst.stream_bytes(i, lens[i]))
    return out


def gpu_decompress(ctx, comps, caps=None):
    from rust_snappy_amd import batch
    src = batch.StreamBatch.from_bytes(comps)
    dst, lens, errs = batch.decompress(ctx, src, caps)
    return [dst.stream_bytes(i, lens[i]) for i in range(len(comps))], errs


def random_inputs(seed, count):
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        alpha = rng.choice([1, 2, 3, 4, 16, 256])
        n = rng.choice([0, 1, 2, 15, 16, 17, 18, 63, 64, 65, 255, 256, 257,
                        1000, 5000, 65535, 65536, 65537, 70000,
                        rng.randrange(0, 200000)])
        out.append(bytes(rng.choices(range(alpha), k=n)))
    return out


def test_compress_corpus_bit_exact(cctx):
    ctx = cctx
    rnd = O.corpus_round()
    got = gpu_compress(ctx, [d for _, d in rnd])
    for (bench_id, data), c in zip(rnd, got):
        n_in, n_out, sha = kats.CORPUS_SHA256[bench_id]
        assert len(c) == n_out, bench_id
        assert hashlib.sha256(c).hexdigest() == sha, bench_id


def test_compress_corpus_tiled_batch(cctx):
    """200 blocks in one batch: several lane-kernel segments, lanes that take
    more than one block, multi-block streams next to one-block streams."""
    ctx = cctx
    rnd = O.corpus_round()
    streams = [d for _, d in rnd] * 4
    got = gpu_compress(ctx, streams)
    for i, c in enumerate(got):
        bench_id = rnd[i % len(rnd)][0]
        n_in, n_out, sha = kats.CORPUS_SHA256[bench_id]
        assert len(c) == n_out, (i, bench_id)
        assert hashlib.sha256(c).hexdigest() == sha, (i, bench_id)


def test_compress_golden_and_kats(cctx):
    ctx = cctx
    txt = (O.CORPUS / "Mark.Twain-Tom.Sawyer.txt").read_bytes()
    snp = (O.CORPUS / "Mark.Twain-Tom.Sawyer.txt.rawsnappy").read_bytes()
    ins = [txt, b"a" * 120, b"", b"\x00", kats.RANDOM1, kats.RANDOM2,
           kats.RANDOM3, kats.RANDOM4]
    got = gpu_compress(ctx, ins)
    assert got[0] == snp                       # test/tests.rs:200-205
    assert got[1].hex() == "780061fe0100da0100"
    assert got[2] == b"\x00"
    for d, c in zip(ins, got):
        assert c == O.compress(d)


def test_compress_structured_bit_exact(cctx):
    ctx = cctx
    ins = kats.small_copy_inputs() + kats.small_regular_inputs()
    got = gpu_compress(ctx, ins)
    for d, c in zip(ins, got):
        assert c == O.compress(d), len(d)


def test_compress_random_bit_exact(cctx):
    ctx = cctx
    ins = random_inputs(7, 600)
    got = gpu_compress(ctx, ins)
    for d, c in zip(ins, got):
        assert c == O.compress(d), (len(d), d[:16])


def test_compress_block_edges(cctx):
    ctx = cctx
    rng = random.Random(3)
    base = (O.CORPUS / "alice29.txt").read_bytes()
    ins = []
    for n in [1, 14, 15, 16, 17, 18, 31, 32, 33, 255, 256, 257, 511, 512,
              8191, 8192, 8193, 16383, 16384, 16385, 65535, 65536, 65537,
              65536 + 16, 65536 + 17, 131071, 131072, 131073]:
        ins.append(base[:n])
        ins.append(bytes([rng.randrange(2) for _ in range(n)]))
        ins.append(b"\x00" * n)
    got = gpu_compress(ctx, ins)
    for d, c in zip(ins, got):
        assert c == O.compress(d), len(d)


def test_decompress_corpus_and_kats(ctx):
    rnd = O.corpus_round()
    comps = [O.compress(d) for _, d in rnd]
    comps += [k[1] for k in kats.DECODE_KATS]
    comps += [(O.CORPUS / "Mark.Twain-Tom.Sawyer.txt.rawsnappy").read_bytes()]
    want = [d for _, d in rnd] + [k[2] for k in kats.DECODE_KATS]
    want += [(O.CORPUS / "Mark.Twain-Tom.Sawyer.txt").read_bytes()]
    got, errs = gpu_decompress(ctx, comps)
    for i, (w, g) in enumerate(zip(want, got)):
        assert errs[i][0] == 0, (i, errs[i])
        assert g == w, i


def test_decompress_random_roundtrip(ctx):
    ins = random_inputs(11, 600) + kats.small_copy_inputs()
    ins += kats.small_regular_inputs()[::5]
    comps = [O.compress(d) for d in ins]
    got, errs = gpu_decompress(ctx, comps)
    for i, (d, g) in enumerate(zip(ins, got)):
        assert errs[i][0] == 0, (i, errs[i])
        assert g == d, (i, len(d))


def test_gpu_roundtrip_gpu(ctx):
    ins = random_inputs(13, 200) + [d for _, d in O.corpus_round()]
    comps = gpu_compress(ctx, ins)
    got, errs = gpu_decompress(ctx, comps)
    for i, (d, g) in enumerate(zip(ins, got)):
        assert errs[i][0] == 0 and g == d, i


def test_decompress_error_kats_exact_fields(ctx):
    import rust_snappy_amd as R
    names = R.error.KINDS
    datas, caps = [], []
    for name, data, want, bad_header in kats.ERROR_KATS:
        datas.append(data)
        caps.append(1024 if bad_header else O.decompress_len(data))
    got, errs = gpu_decompress(ctx, datas, caps)
    for (name, data, want, bad_header), e in zip(kats.ERROR_KATS, errs):
        variant = names[e[0]][0]
        assert variant == want[0], (name, e)
        assert tuple(e[1:1 + len(want) - 1]) == tuple(want[1:]), (name, e)
        # and the oracle agrees field for field
        try:
            O.decompress(data, 1024 if bad_header else O.decompress_len(data))
            raise AssertionError("oracle accepted " + name)
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == e, (name, e)


def test_decompress_corrupt_never_faults_and_matches_oracle(ctx):
    rng = random.Random(5)
    base = [O.compress(d) for _, d in O.corpus_round()[:4]]
    base += [(O.CORPUS / n).read_bytes() for n in
             ("baddata1.snappy", "baddata2.snappy", "baddata3.snappy")]
    muts = list(base[4:])
    for c in base[:4]:
        for _ in range(40):
            b = bytearray(c)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
            if rng.random() < 0.3:
                b = b[:rng.randrange(1, len(b))]
            muts.append(bytes(b))
    caps = []
    for m in muts:
        try:
            caps.append(min(O.decompress_len(m), 1 << 22))
        except O.SnapError:
            caps.append(1024)
    got, errs = gpu_decompress(ctx, muts, caps)
    for m, cap, g, e in zip(muts, caps, got, errs):
        try:
            want = O.decompress(m, cap)
            assert e[0] == 0 and g == want
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == e, (e, oe)


def test_decompress_fuzz_against_oracle(ctx):
    """2 400 mutated streams in one batch - corpus files, RLE, incompressible
    data and foreign-encoder streams (copy-4, far offsets, overlapping copies,
    non-minimal literal lengths), each with 1-6 bytes changed, bytes inserted
    or removed, or truncated: the decoded bytes or the error variant with all
    its fields must be the oracle's, stream by stream."""
    import foreign
    rng = random.Random(77)
    rnd = O.corpus_round()
    base = [O.compress(d[:200000]) for _, d in rnd]
    base += [O.compress(bytes(70000)), O.compress(b"abcd" * 30000),
             O.compress(bytes(rng.randrange(256) for _ in range(70000)))]
    base += [c for c, _ in foreign.cases()[:6]]
    muts = []
    for _ in range(2400):
        b = bytearray(rng.choice(base))
        kind = rng.random()
        if kind < 0.6:
            for _ in range(rng.randrange(1, 7)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif kind < 0.75:
            p = rng.randrange(len(b))
            b[p:p] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 5)))
        elif kind < 0.9:
            p = rng.randrange(len(b))
            del b[p:p + rng.randrange(1, 5)]
        else:
            b = b[:rng.randrange(1, len(b))]
        muts.append(bytes(b))
    caps = []
    for m in muts:
        try:
            caps.append(min(O.decompress_len(m), 1 << 20))
        except O.SnapError:
            caps.append(1024)
    got, errs = gpu_decompress(ctx, muts, caps)
    bad = ok = 0
    for m, cap, g, e in zip(muts, caps, got, errs):
        try:
            want = O.decompress(m, cap)
            assert e[0] == 0 and g == want
            ok += 1
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == e, (e, oe)
            bad += 1
    assert bad > 1000 and ok > 20      # both outcomes are well represented


def test_buffer_too_small_batch(ctx):
    from rust_snappy_amd import batch
    import torch
    data = [b"hello world, hello world", b"abc"]
    src = batch.StreamBatch.from_bytes(data)
    dst = batch.StreamBatch.empty([10, 64])
    out_lens = torch.zeros(2, dtype=torch.int64, device="cuda")
    errs = torch.zeros(64, dtype=torch.uint8, device="cuda")
    import rust_snappy_amd as R
    R.raw.compress_batch(ctx, src.d_ptrs, src.d_lens, dst.d_ptrs, dst.d_lens,
                         out_lens, errs, host_in_lens=src.h_lens)
    ctx.synchronize()
    e = batch.read_errors(errs)
    assert e[0] == (2, 10, 32 + 24 + 4, 0)   # BufferTooSmall{given,min}
    assert e[1][0] == 0
    assert dst.stream_bytes(1, int(out_lens[1])) == O.compress(b"abc")


def test_scalar_mirror_and_snappy_c_api(ctx):
    import ctypes as C
    import rust_snappy_amd as R
    from rust_snappy_amd import _lib
    enc, dec = R.raw.Encoder(ctx), R.raw.Decoder(ctx)
    html = (O.CORPUS / "html").read_bytes()
    c = enc.compress_vec(html)
    assert c == O.compress(html)
    assert dec.decompress_vec(c) == html
    with pytest.raises(R.Error) as ei:
        dec.decompress_vec(b"\x11\x00a\x01\xFF")
    assert ei.value == R.Error(9, 255, 1)       # Offset{offset:255,dst_pos:1}
    with pytest.raises(R.Error) as ei:
        dec.decompress(b"", bytearray(4))
    assert ei.value.key() == ("Empty",)
    with pytest.raises(R.Error) as ei:
        enc.compress(b"abc", bytearray(10))
    assert ei.value.key() == ("BufferTooSmall", 10, 35)
    # the four symbols the reference binds (snappy-cpp/src/lib.rs:66-88)
    L = _lib.load()
    cap = C.c_size_t(L.snappy_max_compressed_length(len(html)))
    out = C.create_string_buffer(cap.value)
    assert L.snappy_compress(html, len(html), out, C.byref(cap)) == 0
    assert out.raw[:cap.value] == c
    n = C.c_size_t(0)
    assert L.snappy_uncompressed_length(c, len(c), C.byref(n)) == 0
    back = C.create_string_buffer(n.value)
    assert L.snappy_uncompress(c, len(c), back, C.byref(n)) == 0
    assert back.raw[:n.value] == html
    assert L.snappy_validate_compressed_buffer(c, len(c)) == 0
    bad = (O.CORPUS / "baddata1.snappy").read_bytes()
    assert L.snappy_validate_compressed_buffer(bad, len(bad)) == 1


def test_tiled_corpus_properties(ctx):
    """cfg2-shaped batch at reduced tiling: every round's outputs equal round
    0 (independence of streams) and round 0 equals the oracle."""
    from rust_snappy_amd import batch
    rnd = [d for _, d in O.corpus_round()]
    rounds = 24
    src = batch.StreamBatch.from_bytes(rnd * rounds)
    dst, lens, errs = batch.compress(ctx, src)
    want = [O.compress(d) for d in rnd]
    for r in range(rounds):
        for j in range(12):
            i = r * 12 + j
            assert errs[i][0] == 0
            assert dst.stream_bytes(i, lens[i]) == want[j], (r, j)
    comp = batch.StreamBatch.from_bytes(want * rounds)
    out, olens, errs = batch.decompress(ctx, comp)
    for r in range(rounds):
        for j in range(12):
            i = r * 12 + j
            assert errs[i][0] == 0
            assert out.stream_bytes(i, olens[i]) == rnd[j], (r, j)


def test_hw_lds_atomic_lane_order(ctx):
    """The wavefront-per-block compressor reproduces 64 sequential table
    updates with ONE LDS atomic; that is exact only if gfx950 applies the
    lanes of one DS atomic in ascending lane order.  Checked on the hardware
    the tests run on (tests/hw/lds_atomic_order.hip)."""
    import subprocess
    from conftest import ROOT, SCAN_GEOMETRIES, set_scan_geometry
    exe = ROOT / "tests" / "hw" / "lds_atomic_order"
    assert exe.exists(), "run __graft_entry__.build() first"
    out = subprocess.run([str(exe)], capture_output=True, text=True,
                         timeout=60).stdout
    assert "PASS ascending-lane order" in out, out


def test_hw_lds_unaligned_access(built):
    """The element-major decoder reads 8 bytes and writes 8 / 4 / 2 / 1 bytes
    at arbitrary byte addresses of its LDS ring with single DS instructions;
    checked bytewise on the hardware (tests/hw/lds_unaligned.hip)."""
    import subprocess
    from conftest import ROOT, SCAN_GEOMETRIES, set_scan_geometry
    exe = ROOT / "tests" / "hw" / "lds_unaligned"
    assert exe.exists(), "run __graft_entry__.build() first"
    out = subprocess.run([str(exe)], capture_output=True, text=True,
                         timeout=60).stdout
    assert "PASS unaligned DS access is bytewise" in out, out


def test_hw_lds_store_lane_order(built):
    """k_decompress_streams2 lets a lane store a whole 16 bytes for a shorter
    element: the lanes above it (the following elements) must win where the
    ranges overlap, i.e. one DS store instruction must apply its lanes in
    ascending order (tests/hw/lds_write_order.hip; also self-checked at
    context creation, which falls back to the byte-per-lane decoder)."""
    import subprocess
    from conftest import ROOT, SCAN_GEOMETRIES, set_scan_geometry
    exe = ROOT / "tests" / "hw" / "lds_write_order"
    assert exe.exists(), "run __graft_entry__.build() first"
    out = subprocess.run([str(exe)], capture_output=True, text=True,
                         timeout=60).stdout
    assert "PASS overlapping stores" in out, out


def test_decode_streams_that_end_at_the_allocation_end(ctx):
    """Compressed input and decoded output both end exactly where their
    device allocations end: the decoder's 16-byte loads (speculative literal
    bytes, far back-references) must never touch a byte behind them."""
    import torch
    from rust_snappy_amd import batch, raw
    rnd = O.corpus_round()
    jpg, txt = rnd[2][1], rnd[6][1]
    for data in (txt, jpg[:70000] + txt[:3000], txt[:70] + jpg[:66000],
                 b"x" * 100 + jpg[:100], txt[:131072]):
        comp = O.compress(data)
        n, m = len(comp), len(data)
        # allocations of exactly n and m bytes (torch rounds up internally;
        # the tensors are carved from the END of bigger ones, whose last byte
        # is the allocation's last byte for a 2 MiB multiple)
        big_in = torch.empty(2 << 20, dtype=torch.uint8, device="cuda")
        big_out = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
        d_in = big_in[(2 << 20) - n:]
        d_in.copy_(torch.frombuffer(bytearray(comp), dtype=torch.uint8))
        d_out = big_out[(4 << 20) - m:]
        ptr = lambda t: torch.tensor([t.data_ptr()], dtype=torch.int64,
                                     device="cuda")
        lens = lambda v: torch.tensor([v], dtype=torch.int64, device="cuda")
        out_len = torch.zeros(1, dtype=torch.int64, device="cuda")
        errs = torch.zeros(32, dtype=torch.uint8, device="cuda")
        raw.decompress_batch(ctx, ptr(d_in), lens(n), ptr(d_out), lens(m),
                             out_len, errs)
        ctx.synchronize()
        assert batch.read_errors(errs)[0][0] == 0
        assert int(out_len.item()) == m
        assert d_out.cpu().numpy().tobytes() == data


def test_compress_stream_at_end_of_allocation(cctx):
    """The input ends exactly where its device allocation ends: the kernels
    may read whole cache lines, but never a line that starts behind the
    stream (memory safety of the lane kernel's input window)."""
    import torch
    from rust_snappy_amd import batch
    ctx = cctx
    n = 8 << 20  # its own allocator segment: the stream ends at its end
    rnd = O.corpus_round()
    blob = (b"".join(d for _, d in rnd) * 4)[:n]
    assert len(blob) == n
    torch.cuda.empty_cache()
    data = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    src = batch.StreamBatch(data, np.array([0], dtype=np.int64),
                            np.array([n], dtype=np.int64))
    dst, lens, errs = batch.compress(ctx, src)
    assert errs[0][0] == 0
    assert dst.stream_bytes(0, lens[0]) == O.compress(blob)


def test_decompress_foreign_encoder_streams(ctx):
    """Streams this encoder never writes: copy-4 elements,
    offsets beyond 64 KiB, overlapping copies, 4-byte literal lengths; also
    truncated and corrupted variants of them against the oracle's errors."""
    import foreign
    cases = foreign.cases()
    got, errs = gpu_decompress(ctx, [s for s, _ in cases])
    for i, ((s, want), g) in enumerate(zip(cases, got)):
        assert errs[i][0] == 0, (i, errs[i])
        assert g == want, i
    rng = random.Random(9)
    muts, caps = [], []
    for s, want in cases:
        for _ in range(12):
            b = bytearray(s)
            for _ in range(rng.randrange(1, 3)):
                b[rng.randrange(len(b))] = rng.randrange(256)
            if rng.random() < 0.3:
                b = b[:rng.randrange(1, len(b))]
            muts.append(bytes(b))
            try:
                caps.append(min(O.decompress_len(muts[-1]), 1 << 22))
            except O.SnapError:
                caps.append(1024)
    got, errs = gpu_decompress(ctx, muts, caps)
    for m, cap, g, e in zip(muts, caps, got, errs):
        try:
            w = O.decompress(m, cap)
            assert e[0] == 0 and g == w
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == e, (e, oe)


def stream_decode(ctx, comp, cap):
    """snapmi_decompress_stream on one device-resident stream."""
    import torch
    from rust_snappy_amd import raw
    d_in = torch.frombuffer(bytearray(comp) or bytearray(1),
                            dtype=torch.uint8).cuda()
    d_out = torch.zeros(max(cap, 16), dtype=torch.uint8, device="cuda")[:cap]
    out_len = torch.zeros(1, dtype=torch.int64, device="cuda")
    err = torch.zeros(32, dtype=torch.uint8, device="cuda")
    raw.decompress_stream(ctx, d_in, len(comp), d_out, out_len, err)
    ctx.synchronize()
    e = np.frombuffer(err.cpu().numpy().tobytes(), dtype=np.dtype(
        [("kind", "<i4"), ("r", "<u4"), ("a", "<u8"), ("b", "<u8"),
         ("c", "<u8")]))[0]
    n = int(out_len.item())
    return d_out[:n].cpu().numpy().tobytes(), (int(e["kind"]), int(e["a"]),
                                               int(e["b"]), int(e["c"]))


@pytest.mark.parametrize("geom", SCAN_GEOMETRIES)
def test_long_stream_parallel_decode(ctx, geom):
    """One raw stream on many wavefronts: equal to the original, for streams
    of this format's encoders (64 KiB blocks: pieces independent), for
    foreign streams (copies across pieces: sequential path), tiny and empty
    streams, and with the oracle's error on corrupted ones."""
    import foreign
    # (both segment sizes of the scan: 1 KiB is what streams of this size get
    # by themselves, 4 KiB what those of 256 MiB and more do)
    seg = set_scan_geometry(ctx, geom)
    rnd = O.corpus_round()
    big = b"".join(d for _, d in rnd) * 3            # 8.8 MB, 137 blocks
    cases = [big, rnd[2][1] * 5, bytes(300000), b"", b"a", rnd[0][1],
             bytes(range(256)) * 1000]
    from rust_snappy_amd import raw
    for i, data in enumerate(cases):
        comp = O.compress(data)
        got, e = stream_decode(ctx, comp, len(data))
        assert e[0] == 0, (i, e)
        assert got == data, i
        # block-structured streams never need the sequential path
        assert raw.stream_decode_path(ctx) == 0, i
    paths = []
    for i, (st, want) in enumerate(foreign.cases()):
        got, e = stream_decode(ctx, st, len(want))
        assert e[0] == 0 and got == want, (i, e)
        paths.append(raw.stream_decode_path(ctx))
    assert 1 in paths  # copies across pieces: sequential path exercised
    # errors: same variant and fields as the oracle (= the reference)
    rng = random.Random(21)
    comp = O.compress(big[:3000000])
    muts = [comp[:len(comp) // 2], comp[:-1], comp + b"\x00"]
    for _ in range(10):
        b = bytearray(comp)
        for _ in range(rng.randrange(1, 3)):
            b[rng.randrange(len(b))] = rng.randrange(256)
        muts.append(bytes(b))
    for m in muts:
        try:
            cap = min(O.decompress_len(m), 1 << 23)
        except O.SnapError:
            cap = 1024
        got, e = stream_decode(ctx, m, cap)
        try:
            want = O.decompress(m, cap)
            assert e[0] == 0 and got == want
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == e, (e, oe)
    # too small an output buffer
    got, e = stream_decode(ctx, O.compress(big), len(big) - 1)
    try:
        O.decompress(O.compress(big), len(big) - 1)
        raise AssertionError
    except O.SnapError as oe:
        assert (oe.kind, oe.a, oe.b, oe.c) == e
    set_scan_geometry(ctx, None)


@pytest.mark.parametrize("geom", SCAN_GEOMETRIES)
def test_long_stream_scan_shapes(ctx, geom):
    """The structure k_stream_scan / k_stream_cuts work in (8 .. 64 segments
    of 1 or 4 KiB per scan wavefront, 512 per cuts wavefront, walks handed out from a
    pool, chains that join the next segment's trunk): streams whose segment
    count sits at and around those boundaries, streams where chains run
    through the bytes of long literals (they never join: one walk each),
    streams of two-byte copies and of one-byte literals (the most hops a
    round can hold), and literals with one, two and three length bytes at
    every alignment.  All must come from the parallel path."""
    from rust_snappy_amd import raw
    seg = set_scan_geometry(ctx, geom)
    seg_bytes = 1 << seg
    rng = random.Random(77)
    text = b"".join(d for n, d in O.corpus_round() if "txt" in n or "html" in n)

    def with_segments(nseg):
        # a prefix of text x k whose stream is nseg segments long (bisection)
        src = text * (nseg * seg_bytes * 3 // len(text) + 2)
        lo, hi = 1, len(src)
        while lo < hi:
            mid = (lo + hi) // 2
            if (len(O.compress(src[:mid])) + seg_bytes - 1) // seg_bytes \
                    < nseg:
                lo = mid + 1
            else:
                hi = mid
        return src[:lo]

    cases = [with_segments(k) for k in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65,
                                        128, 129, 511, 512, 513, 1025)]
    noise = lambda n: bytes(rng.randrange(256) for _ in range(n))  # noqa: E731
    mixed = bytearray()
    for n in (200_000, 70_000, 61, 62, 300, 65_536, 65_537, 4096, 100_000):
        mixed += noise(n) + text[rng.randrange(100_000):][:rng.randrange(
            5_000, 150_000)]
    cases.append(bytes(mixed))
    small = bytes(rng.choice(b"abcd") for _ in range(700_000))   # tiny elements
    cases.append(small)
    cases.append(bytes(rng.randrange(2) for _ in range(900_000)))
    # literals of 61 .. 70 000 bytes between copies, at drifting alignment
    lit = bytearray()
    for k in range(400):
        lit += noise(rng.choice((61, 62, 100, 255, 256, 257, 1000, 4095, 4096,
                                 4097, 20_000, 70_000)))
        lit += lit[-rng.randrange(4, 60):] * rng.randrange(1, 4)
    cases.append(bytes(lit))
    for i, data in enumerate(cases):
        comp = O.compress(data)
        got, e = stream_decode(ctx, comp, len(data))
        assert e[0] == 0, (i, e)
        assert got == data, (i, len(data), len(comp))
        assert raw.stream_decode_path(ctx) == 0, i
    set_scan_geometry(ctx, None)


@pytest.mark.parametrize("geom", SCAN_GEOMETRIES)
def test_batch_with_long_streams(ctx, geom):
    """A small batch gives its long streams their pieces (k_long_plan,
    k_bstream_*): long and short streams side by side, long ones that are
    corrupt, truncated, lie in their header, come from a foreign encoder
    (copies across pieces: the wavefront decoder in the launch behind), or
    do not fit the caller's buffer - bytes and errors are the oracle's stream
    by stream, with the option on (default) and off."""
    import foreign
    import rust_snappy_amd as R
    rng = random.Random(5)
    rnd = O.corpus_round()
    big = b"".join(d for _, d in rnd) * 3
    plain = [d for _, d in rnd] + [big, b"", b"a", bytes(500000),
                                   bytes(rng.randrange(256)
                                         for _ in range(300000))]
    comps = [O.compress(d) for d in plain]
    assert sum(len(c) >= (128 << 10) for c in comps) >= 5
    long_ones = [c for c in comps if len(c) >= (128 << 10)]
    muts = []
    for c in long_ones[:4]:
        for _ in range(6):
            b = bytearray(c)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
            muts.append(bytes(b))
        muts += [c[:len(c) // 2], c[:-1], c + b"\x00"]
        # headers that announce more / less than the elements produce
        n, hd = O.decompress_len(c), 0
        while c[hd] & 0x80:
            hd += 1
        hd += 1
        muts += [foreign.varint(n + 70000) + c[hd:],
                 foreign.varint(n - 70000) + c[hd:]]
    far = [foreign.build(100 + k, 600000) for k in range(3)]   # 3 long ones
    streams = comps + muts + [st for st, _ in far]
    caps = []
    for m in streams:
        try:
            caps.append(min(O.decompress_len(m), 1 << 25))
        except O.SnapError:
            caps.append(1024)
    # ... and long streams whose output buffer is too small
    streams += long_ones[:2]
    caps += [O.decompress_len(long_ones[0]) - 1, 70000]
    assert len(streams) < 200
    plain_ctx = R.raw.Context(0)
    plain_ctx.set_option("batch_long_streams", 0)
    seg = set_scan_geometry(ctx, geom)
    try:
        for c in (ctx, plain_ctx):
            got, errs = gpu_decompress(c, streams, caps)
            for i, (m, cap, g, e) in enumerate(zip(streams, caps, got, errs)):
                try:
                    want = O.decompress(m, cap)
                    assert e[0] == 0 and g == want, (i, e, len(m))
                except O.SnapError as oe:
                    assert (oe.kind, oe.a, oe.b, oe.c) == e, (i, e, oe)
    finally:
        plain_ctx.close()
        set_scan_geometry(ctx, None)


def test_scalar_decompress_uses_long_stream_path(ctx):
    import rust_snappy_amd as R
    data = b"".join(d for _, d in O.corpus_round()) * 2
    dec = R.raw.Decoder(ctx=ctx)
    assert dec.decompress_vec(O.compress(data)) == data


def _lane_ctx(**opts):
    import rust_snappy_amd as R
    c = R.raw.Context(0)
    c.set_option("compress_mode", 1)
    c.set_option("lane_min_blocks", 1)
    for k, v in opts.items():
        try:
            c.set_option(k, v)
        except Exception:  # noqa: BLE001 - a knob of include/snapmi_test.h
            c.set_test_option(k, v)
    return c


def test_lane_table_epoch_wrap(built):
    """The lane kernel never zeroes its HBM hash tables: entries carry a
    16-bit epoch and a lane bumps its epoch per block.  After 65 535 blocks on
    one lane the epoch wraps: the table is really cleared and the epoch
    restarts at 1 (snapmi_compress.hip, `if (epoch == 0)`).  Epochs persist in
    the context, so a long-lived context gets there.  The option
    lane_epoch_preset starts every lane shortly before the wrap; stale entries
    of epochs 1, 2, ... left by the earlier batch must not come back to life
    after it.  One wavefront (64 lanes) for ~300 blocks: about five blocks per
    lane in one launch, across the wrap."""
    rnd = O.corpus_round()
    streams = [d for _, d in rnd] * 6                    # 300 blocks
    want = [O.compress(d) for _, d in rnd] * 6
    for preset in (0xFFFD, 0xFFFE, 0xFFFF):
        c = _lane_ctx(lane_max_waves=1)
        assert gpu_compress(c, streams) == want          # epochs 1..5 in use
        c.set_test_option("lane_epoch_preset", preset)
        assert gpu_compress(c, streams) == want, hex(preset)
        assert gpu_compress(c, streams[::-1]) == want[::-1], hex(preset)
        c.close()
    # the same across launches (one block per lane and launch, segments of 64)
    c = _lane_ctx(lane_segment_blocks=64)
    assert gpu_compress(c, streams) == want
    c.set_test_option("lane_epoch_preset", 0xFFFE)
    assert gpu_compress(c, streams) == want
    c.close()


def test_compress_without_lds_atomic_order(built):
    """A context whose LDS-atomic-order self-check failed (forced here) must
    not launch the wavefront-per-block kernel: small batches then go to the
    lane kernel and still give the reference's bytes."""
    import rust_snappy_amd as R
    c = R.raw.Context(0)
    c.set_test_option("lds_order_ok", 0)
    ins = [d for _, d in O.corpus_round()] + random_inputs(11, 60)
    got = gpu_compress(c, ins)
    for d, g in zip(ins, got):
        assert g == O.compress(d), len(d)
    c.close()


def test_compress_too_big_by_descriptor(cctx):
    """Error::TooBig on compress (reference src/compress.rs:104-110) without a
    4 GiB buffer: a batch descriptor CLAIMS 2^32 bytes.  The stream is
    rejected with the reference's fields and never touched; its neighbours
    compress to the oracle's bytes."""
    import torch
    from rust_snappy_amd import batch, raw
    ins = [b"neighbour " * 500, b"x" * 64, bytes(range(256)) * 40]
    src = batch.StreamBatch.from_bytes(ins)
    fake = src.lens.copy()
    fake[1] = 1 << 32
    d_lens = torch.from_numpy(fake).to(src.data.device)
    caps = [raw.max_compress_len(len(s)) for s in ins]
    dst = batch.StreamBatch.empty(caps, src.data.device)
    out_lens = torch.zeros(3, dtype=torch.int64, device=src.data.device)
    errs = torch.zeros(32 * 3, dtype=torch.uint8, device=src.data.device)
    # (the capacity check comes second in the reference: pass no capacities)
    raw.compress_batch(cctx, src.d_ptrs, d_lens, dst.d_ptrs, None, out_lens,
                       errs, host_in_lens=torch.from_numpy(fake.copy()))
    cctx.synchronize()
    e = batch.read_errors(errs)
    assert e[1] == (O.KIND_NAMES.index("TooBig"), 1 << 32, 0xFFFFFFFF, 0)
    assert int(out_lens[1]) == 0
    for i in (0, 2):
        assert e[i][0] == 0
        assert dst.stream_bytes(i, int(out_lens[i])) == O.compress(ins[i])


def test_decode_descriptor_of_4_gib_goes_to_the_sequential_decoder(ctx):
    """A compressed stream of 4 GiB or more is legal (every element tiny) and
    is left to the sequential decoder as a whole (positions in the window
    loops are 32-bit).  Exercised with a descriptor that CLAIMS 2^32 + 100
    bytes over a tiny buffer whose first element already fails: the error
    carries the claimed length, like the reference's would
    (src/decompress.rs:209-217)."""
    import torch
    from rust_snappy_amd import batch, raw
    # header: 5 bytes of output; a literal of 10 bytes -> Literal{len 10,
    # src_len (what is left behind the tag), dst_len 5}
    body = bytes([5, (10 - 1) << 2]) + b"0123456789" + bytes(64)
    src = batch.StreamBatch.from_bytes([body, O.compress(b"fine" * 100)])
    claimed = (1 << 32) + 100
    fake = src.lens.copy()
    fake[0] = claimed
    d_lens = torch.from_numpy(fake).to(src.data.device)
    dst = batch.StreamBatch.empty([5, 400], src.data.device)
    out_lens = torch.zeros(2, dtype=torch.int64, device=src.data.device)
    errs = torch.zeros(64, dtype=torch.uint8, device=src.data.device)
    raw.decompress_batch(ctx, src.d_ptrs, d_lens, dst.d_ptrs, dst.d_lens,
                         out_lens, errs)
    ctx.synchronize()
    e = batch.read_errors(errs)
    assert e[0] == (O.KIND_NAMES.index("Literal"), 10, claimed - 2, 5), e[0]
    assert e[1][0] == 0 and dst.stream_bytes(1, 400) == b"fine" * 100


def test_snappy_uncompress_into_a_short_buffer(ctx):
    """snappy_uncompress with *uncompressed_length too small: the status of
    libsnappy 1.1.8 (the library the reference's seam binds), and the length
    is left alone."""
    import ctypes as C
    from rust_snappy_amd import _lib
    L = _lib.load()
    data = b"status parity " * 300
    comp = O.compress(data)
    buf = C.create_string_buffer(len(data))
    for cap in (0, 1, len(data) - 1):
        n = C.c_size_t(cap)
        got = L.snappy_uncompress(comp, len(comp), buf, C.byref(n))
        assert got == 2 and n.value == cap          # SNAPPY_BUFFER_TOO_SMALL
        ref = O.libsnappy()
        if ref is not None:
            m = C.c_size_t(cap)
            ref.snappy_uncompress.restype = C.c_int
            want = ref.snappy_uncompress(comp, C.c_size_t(len(comp)), buf,
                                         C.byref(m))
            assert got == want, (cap, got, want)
    n = C.c_size_t(len(data))
    assert L.snappy_uncompress(comp, len(comp), buf, C.byref(n)) == 0
    assert buf.raw[:n.value] == data


def test_snappy_c_api_from_many_threads(ctx):
    """The reference's native wrappers are stateless and callable from any
    thread at once (snappy-cpp/src/lib.rs:13-64): 8 threads x 200 calls of
    snappy_compress + snappy_uncompress, bytes equal to the oracle's, and the
    calls really overlap (a pool of contexts, not one behind a mutex)."""
    import ctypes as C
    import threading
    import time
    from rust_snappy_amd import _lib
    L = _lib.load()
    rng = random.Random(77)
    inputs = [bytes(rng.choices(range(rng.choice([2, 4, 16, 256])),
                                k=rng.randrange(1, 40000))) for _ in range(25)]
    want = [O.compress(x) for x in inputs]
    bad = []

    def worker(tid, calls):
        cap = max(L.snappy_max_compressed_length(len(x)) for x in inputs)
        out = C.create_string_buffer(cap)
        back = C.create_string_buffer(40000)
        for k in range(calls):
            i = (tid * 7 + k) % len(inputs)
            n = C.c_size_t(cap)
            if L.snappy_compress(inputs[i], len(inputs[i]), out, C.byref(n)) \
                    or out.raw[:n.value] != want[i]:
                bad.append(("compress", tid, k))
            m = C.c_size_t(40000)
            if L.snappy_uncompress(want[i], len(want[i]), back, C.byref(m)) \
                    or back.raw[:m.value] != inputs[i]:
                bad.append(("uncompress", tid, k))

    worker(0, 20)                                  # warm: first context
    t0 = time.perf_counter()
    worker(0, 200)
    t1 = time.perf_counter() - t0
    ths = [threading.Thread(target=worker, args=(t, 200)) for t in range(8)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    t8 = time.perf_counter() - t0
    assert not bad, bad[:5]
    rate1, rate8 = 400 / t1, 8 * 400 / t8
    print(f"\nsnappy C API calls/s: 1 thread {rate1:.0f}, 8 threads {rate8:.0f}"
          f" ({rate8 / rate1:.2f}x)")
    assert rate8 > 2.0 * rate1, (rate1, rate8)


class _FreeMemoryPoll:
    """hipMemGetInfo from a second thread, every half millisecond, while the
    body runs: the least free memory anyone saw."""

    def __enter__(self):
        import threading
        import torch
        self.low = torch.cuda.mem_get_info()[0]
        self.samples = 0
        self._stop = False

        def poll():
            while not self._stop:
                f = torch.cuda.mem_get_info()[0]
                self.low = min(self.low, f)
                self.samples += 1
                time.sleep(0.0005)
        self._t = threading.Thread(target=poll, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._t.join()


def _budget_batch():
    from rust_snappy_amd import batch
    blob = b"".join(d for _, d in O.corpus_round())
    blocks = [blob[o:o + 65536] for o in range(0, 40 * 65536, 65536)]
    ins = [blocks[i % 40] for i in range(16500)]          # ~1 GiB, one block each
    return ins, batch.StreamBatch.from_bytes(ins)


@pytest.mark.parametrize("lib", ["test", "product", "product-plain"])
@pytest.mark.parametrize("pct", [10, 33])
def test_lane_table_placement_stays_within_its_budget(built, pct, lib):
    """The GPU may be shared: NO compress call holds more than
    lane_table_budget_pct of the memory that was free when it began - not
    afterwards, and not for a moment while the placement of the lane tables
    is being chosen (round 5's placement held the whole device for the
    duration of two hipMallocs per candidate).  A second thread polls
    hipMemGetInfo while a context's first chip-filling launch (16 384 lanes
    or more) places its tables; the bytes are the oracle's as ever.  On the
    test build and on the shipped library with its default options (tables
    as mapped chunks), and on the shipped library without spreading (one
    plain hipMalloc region)."""
    import re
    import torch
    from conftest import product_context
    from rust_snappy_amd import batch
    if lib.startswith("product"):
        c = product_context()
        c.set_option("lane_min_blocks", 1)
        if lib == "product-plain":
            # no spreading: one packed region from hipMalloc, the path the
            # placement also takes when the virtual-memory calls fail
            c.set_option("lane_table_spread", 0)
    else:
        c = _lane_ctx()
    c.set_option("lane_table_budget_pct", pct)
    ins, src = _budget_batch()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    with _FreeMemoryPoll() as poll:
        dst, lens, errs = batch.compress(c, src)
        c.synchronize()
    log = c.table_probe_log()
    m = re.search(r"^(.*)\| held at most (\d+) of budget (\d+) \| kept (\d+) KiB "
                  r"apart, (\d+) lanes, (\d+) bytes \| placement ([0-9.]+) ms",
                  log)
    assert m, log
    n_cand = len(re.findall(r"\d+\.\d+\(", m.group(1)))
    assert (n_cand == 0 if lib == "product-plain" else 1 <= n_cand <= 2), log
    assert ("chunks:" in log) == (lib != "product-plain"), log
    held, budget, kept = int(m.group(2)), int(m.group(3)), int(m.group(6))
    assert 0 < kept <= held <= budget <= pct / 100 * free1 + (1 << 20), log
    assert float(m.group(7)) < 3000, log          # no seconds of placement
    # what anyone else saw: the tables' candidates within the budget, plus
    # this batch's own scratch and output (1 GiB in: tokens 2.2 GB, out 1.3)
    assert poll.samples > 20
    assert free1 - poll.low <= budget + (6 << 30), \
        (free1, poll.low, budget, log)
    for i in (0, 39, 16499):
        assert errs[i][0] == 0
        assert dst.stream_bytes(i, lens[i]) == O.compress(ins[i])
    free2, _ = torch.cuda.mem_get_info()
    assert free1 - free2 <= kept + (6 << 30), (free1, free2, kept)
    c.close()


def test_prepare_places_the_tables_ahead_of_the_first_batch(built):
    """snapmi_ctx_prepare: the tables a batch of that many blocks needs exist
    when it returns (the first compress call places nothing), within the
    budget; with SNAPMI_PREPARE_TOP_OF_MEMORY - the explicit opt-in - one
    packed candidate behind a filler, probed, and the memory is free again
    afterwards but for the tables.  Bytes are the oracle's."""
    import re
    import torch
    from conftest import product_context
    from rust_snappy_amd import batch
    ins, src = _budget_batch()
    for top in (False, True):
        c = product_context()
        c.set_option("lane_min_blocks", 1)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        c.prepare(len(ins), top_of_memory=top)
        log = c.table_probe_log()
        m = re.search(r"kept (\d+) KiB apart, (\d+) lanes, (\d+) bytes", log)
        assert m and int(m.group(2)) >= 16384, log
        assert ("top of memory" in log) == top, log
        if top:
            assert int(m.group(1)) == 256, log    # packed
        kept = int(m.group(3))
        free2, _ = torch.cuda.mem_get_info()
        assert free1 - free2 <= kept + (1 << 30), (free1, free2, log)
        dst, lens, errs = batch.compress(c, src)
        c.synchronize()
        assert c.table_probe_log() == log         # nothing placed again
        for i in (0, 39, 16499):
            assert errs[i][0] == 0
            assert dst.stream_bytes(i, lens[i]) == O.compress(ins[i])
        c.close()
    # below lane_min_blocks there is nothing to prepare
    c = product_context()
    c.prepare(100)
    assert c.table_probe_log() == ""
    c.close()


def test_token_pool_spills_are_compressed_again_and_the_pool_grows(built):
    """The token pool (CompressArgs::tok_pool): a batch whose blocks need
    more pages than the pool holds loses nothing - the blocks that find no
    page are compressed a second time by k_redo_spilled, to the oracle's
    bytes - the counts say how many did, and the context's next batches get a
    larger pool until none does.  Shipped library, its default routing (1 GiB
    of one-block streams: the lane kernel beside the window kernel)."""
    from conftest import product_context
    from rust_snappy_amd import batch
    ins, src = _budget_batch()
    n = len(ins)
    c = product_context()
    c.set_option("lane_min_blocks", 1)
    c.set_option("token_pool_min_pages", 0)
    c.set_option("token_pool_pct", 10)   # the corpus needs about 35
    seen = []
    for _ in range(9):                   # 10, 16, 25, 38 per cent ...
        dst, lens, errs = batch.compress(c, src)
        spilled = c.info("token_blocks_spilled")
        seen.append((c.info("token_pool_pages"), spilled,
                     c.info("token_pages_asked")))
        for i in list(range(0, n, 397)) + [n - 1]:
            assert errs[i][0] == 0
            assert dst.stream_bytes(i, lens[i]) == O.compress(ins[i]), i
        if len(seen) > 1 and seen[-2][1] == 0:
            break                        # two calls without a spilled block
    pages = [p for p, _, _ in seen]
    assert 0 < seen[0][1] < n, seen            # some spilled, some did not
    assert pages[1] > pages[0], seen           # ... so the pool grew
    assert pages == sorted(pages), seen        # (it never shrinks)
    assert seen[-1][1] == 0 and seen[-2][1] == 0, seen  # until nothing spilled
    assert pages[-1] == pages[-2], seen        # ... and then it stays
    assert seen[-1][2] <= pages[-1], seen
    assert c.info("token_pool_pct_now") > 10
    assert c.info("token_scratch_bytes") <= c.info("scratch_bytes")
    with pytest.raises(Exception) as e:        # a name that is not in the list
        c.info("no_such_thing")
    assert "unknown info" in str(e.value) or "101" in str(e.value), e.value
    # 100 %: no block can spill, whatever the data
    c.set_option("token_pool_pct", 100)
    dst, lens, errs = batch.compress(c, src)
    assert c.info("token_blocks_spilled") == 0
    assert dst.stream_bytes(n - 1, lens[n - 1]) == O.compress(ins[n - 1])
    c.close()


def test_many_small_streams_plan_on_many_workgroups(cctx, ctx):
    """Batches of more than 16 384 streams are planned, scanned and sorted by
    many workgroups (k_plan_compress_a/b/c, k_scan_sizes_a/b/c,
    k_plan_decompress_a/b/c): 40 000 streams of mixed sizes - empty ones, tiny
    ones, some of several blocks - against the oracle, both directions."""
    from rust_snappy_amd import batch
    rng = random.Random(2024)
    blob = b"".join(d for _, d in O.corpus_round())
    kinds = [0, 1, 7, 200, 200, 200, 1000, 4096, 70000, 140000]
    ins = []
    for i in range(40000):
        n = kinds[i % len(kinds)] if i % 97 else rng.randrange(0, 3000)
        o = rng.randrange(0, len(blob) - 150000)
        ins.append(blob[o:o + n])
    src = batch.StreamBatch.from_bytes(ins)
    dst, lens, errs = batch.compress(cctx, src)
    assert all(e[0] == 0 for e in errs)
    check = list(range(0, 40000, 613)) + [0, 1, 39998, 39999]
    want = {i: O.compress(ins[i]) for i in check}
    for i in check:
        assert dst.stream_bytes(i, lens[i]) == want[i], i
    # every stream's size: the oracle's size function is its compressor, so
    # compare sizes of equal inputs among themselves instead
    by_input = {}
    for i, x in enumerate(ins):
        by_input.setdefault(x, set()).add(int(lens[i]))
    assert all(len(v) == 1 for v in by_input.values())
    comp = batch.StreamBatch(dst.data, dst.offsets, lens)
    # (second pass: k_decompress_streams3_many, 16 streams of the sorted order
    # per workgroup - the kernel of batches of more than a million streams)
    for many_min in (1 << 20, 1000):
        ctx.set_test_option("decode_many_min", many_min)
        back, blens, derrs = batch.decompress(ctx, comp)
        assert all(e[0] == 0 for e in derrs)
        assert [int(x) for x in blens] == [len(x) for x in ins]
        for i in check:
            assert back.stream_bytes(i, blens[i]) == ins[i], i
    # errors come through it as well: damage three streams (1 000, 4 096 and
    # 70 000 bytes: their headers then promise more than the buffers hold)
    bad = [16, 20007, 39988]
    raw_comp = comp.data.clone()
    for i in bad:
        o = int(comp.offsets[i])
        raw_comp[o] = 0xFF                     # header byte: wrong length
    comp_bad = batch.StreamBatch(raw_comp, comp.offsets, lens)
    results = []
    for many_min in (1 << 20, 1000):
        ctx.set_test_option("decode_many_min", many_min)
        _, blens2, derrs2 = batch.decompress(
            ctx, comp_bad, caps=[len(x) for x in ins])
        results.append(([int(x) for x in blens2], derrs2))
    ctx.set_test_option("decode_many_min", 1 << 20)
    assert results[0] == results[1]
    assert all(results[0][1][i][0] != 0 for i in bad)
    assert sum(1 for e in results[0][1] if e[0] != 0) == len(bad)


def test_decoder_boundaries_of_the_third_generation(ctx):
    """Streams sized around every threshold the third-generation path has -
    256 compressed bytes (lane-per-stream kernel, staged in LDS when the
    output is at most 256 bytes too), 337 bytes of input left (kTail3: the
    window loops change hands), literals of 60 / 61 bytes (the latter end a
    window), windows cut at 2 048 output bytes, copies around the 4 KiB ring
    and its 2 KiB of safe history - valid and mutated, against the oracle:
    bytes, or the error variant with all of its fields."""
    import foreign
    rng = random.Random(20263)
    txt = (O.CORPUS / "alice29.txt").read_bytes()
    html = (O.CORPUS / "html").read_bytes()
    comps = []
    # compressed sizes around 256 and 337 + k * 256: take prefixes of text
    # whose compressed length lands on the wanted size
    want_sizes = set(list(range(240, 275)) + list(range(325, 350)) +
                     [592, 593, 594, 848, 849, 850, 1104, 1105, 1106])
    n = 200
    seen = set()
    while want_sizes - seen and n < 4000:
        c = O.compress(txt[:n])
        if len(c) in want_sizes and len(c) not in seen:
            seen.add(len(c))
            comps.append(c)
        n += 1
    assert len(seen) > 40
    # tiny inputs with tiny and not so tiny outputs (RLE: 64-byte copies)
    for k in (1, 2, 5, 60, 61, 64, 65, 200, 255, 256, 257, 300, 4000, 20000):
        comps.append(O.compress(b"x" * k))
        comps.append(O.compress(bytes(rng.randrange(256) for _ in range(k))))
        comps.append(O.compress((b"ab" * k)[:k]))
    # literal lengths around 60 / 61 between copies, hand-made
    for ll in (59, 60, 61, 62, 64, 65, 100, 255, 256, 257):
        body = bytes(rng.randrange(256) for _ in range(ll))
        out = body + body[:40]
        el = foreign.lit(body) + foreign.copy(ll, 40 if ll >= 40 else 4, 2) \
            if ll >= 4 else foreign.lit(body)
        if ll >= 40:
            comps.append(foreign.varint(len(out)) + el)
    # far / near copies around the ring's limits, many per window
    data = bytearray(bytes(rng.randrange(256) for _ in range(9000)))
    els, outb = [foreign.lit(bytes(data[:60]))], bytearray(data[:60])
    pos = 60
    while pos < 9000:
        els.append(foreign.lit(bytes(data[pos:pos + 30])))
        outb += data[pos:pos + 30]
        pos += 30
        for off in (1, 7, 15, 16, 17, 63, 64, 2000, 2031, 2032, 2033, 2047,
                    2048, 2049, 4079, 4080, 4081, 4095, 4096, 4097, 8000):
            if off <= len(outb):
                ln = rng.choice([4, 11, 16, 17, 33, 64])
                els.append(foreign.copy(off, ln, 2))
                for _ in range(ln):
                    outb.append(outb[-off])
    comps.append(foreign.varint(len(outb)) + b"".join(els))
    assert O.decompress(comps[-1]) == bytes(outb)
    # html: wide windows (many long copies -> the 2 048-byte cut)
    comps.append(O.compress(html))
    valid = list(comps)
    # mutations of everything above
    for c in valid:
        for _ in range(6):
            b = bytearray(c)
            kind = rng.random()
            if kind < 0.5 and len(b) > 1:
                for _ in range(rng.randrange(1, 4)):
                    b[rng.randrange(len(b))] = rng.randrange(256)
            elif kind < 0.8 and len(b) > 2:
                b = b[:rng.randrange(1, len(b))]
            else:
                p = rng.randrange(len(b) + 1)
                b[p:p] = bytes([rng.randrange(256)])
            comps.append(bytes(b))
    caps = []
    for m in comps:
        try:
            caps.append(min(O.decompress_len(m), 1 << 20))
        except O.SnapError:
            caps.append(512)
    got, errs = gpu_decompress(ctx, comps, caps)
    ok = bad = 0
    for m, cap, g, e in zip(comps, caps, got, errs):
        try:
            want = O.decompress(m, cap)
            assert e[0] == 0 and g == want, (len(m), e)
            ok += 1
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == e, (len(m), e, oe)
            bad += 1
    assert ok >= len(valid) and bad > 100


def test_tiny_streams_one_per_lane(built):
    """k_compress_tiny (streams of 1..255 bytes, one per LANE with input,
    table and output in LDS) and k_compress_small (256..2047 bytes, a few per
    wavefront; three size classes) - the algorithm itself is checked on the
    CPU by test_tiny_lane_cpu.py.  Every length up to 300, around every class
    limit and a sample between, every kind of data, in ONE batch next to
    empty and larger streams, inputs and outputs packed back to back (every
    alignment; a byte written past a stream's end would damage its
    neighbour), some capacities one byte short.  The same batch with the
    kernels off (such streams are then one-block streams of the block
    kernels) must give the same bytes."""
    import torch
    import rust_snappy_amd as R
    from rust_snappy_amd import batch
    rng = random.Random(99)
    blob = b"".join(d for _, d in O.corpus_round())
    ins = []
    sizes = list(range(0, 300)) + list(range(300, 2100, 37)) + [
        510, 511, 512, 513, 1022, 1023, 1024, 1025, 2046, 2047, 2048, 2049,
        4096, 65536, 70000]
    for n in sizes:
        o = rng.randrange(0, len(blob) - 70000)
        ins.append(blob[o:o + n])
        ins.append(bytes(rng.randrange(2) for _ in range(n)))
        ins.append(bytes(n))
        ins.append(bytes(rng.randrange(256) for _ in range(n)))
        unit = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
        ins.append((unit * (n // len(unit) + 1))[:n])
    rng.shuffle(ins)
    lens = [len(x) for x in ins]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    host = np.frombuffer(b"".join(ins) + b"\0", dtype=np.uint8).copy()
    src = batch.StreamBatch(torch.from_numpy(host).cuda(), offs, lens)
    need = [R.raw.max_compress_len(n) for n in lens]
    short = {i for i in range(len(ins)) if i % 7 == 3}
    caps = [need[i] - 1 if i in short else need[i] for i in range(len(ins))]
    coffs = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.int64)
    want = [O.compress(x) for x in ins]
    for tiny in (3, 1, 0):
        c = R.raw.Context(0)
        c.set_option("tiny_stream_kernel", tiny & 1)
        c.set_option("small_stream_kernel", tiny >> 1)
        dst = batch.StreamBatch(
            torch.full((int(sum(caps)) + 16,), 0xEE, dtype=torch.uint8,
                       device="cuda"), coffs, caps)
        out_lens = torch.full((len(ins),), -1, dtype=torch.int64,
                              device="cuda")
        errs = torch.zeros(32 * len(ins), dtype=torch.uint8, device="cuda")
        R.raw.compress_batch(c, src.d_ptrs, src.d_lens, dst.d_ptrs,
                             dst.d_lens, out_lens, errs,
                             host_in_lens=src.h_lens)
        c.synchronize()
        e = batch.read_errors(errs)
        flat = dst.data.cpu().numpy().tobytes()
        ol = out_lens.cpu().numpy()
        for i in range(len(ins)):
            if i in short:
                assert e[i] == (2, caps[i], need[i], 0), (tiny, i, e[i])
                assert ol[i] == 0
                # nothing of a rejected stream is written
                assert flat[coffs[i]:coffs[i] + caps[i]] == b"\xEE" * caps[i]
            else:
                assert e[i][0] == 0, (tiny, i, e[i])
                assert flat[coffs[i]:coffs[i] + ol[i]] == want[i], \
                    (tiny, i, lens[i])
        c.close()
    # (bytes behind a stream's end inside its own capacity may differ between
    # kernels - the block kernels over-copy there - so only lengths + content
    # were compared above)


@pytest.mark.parametrize("count", [60, 20000])
def test_tiny_compressed_streams_with_large_output(ctx, count):
    """Streams of under 256 COMPRESSED bytes whose output is larger (runs,
    sparse pages: 4 KiB of zeros are 197 bytes) are sorted with the wavefront
    decoder's first class by the plan (plan_class reads their header); the
    lane-per-stream kernel keeps the ones whose output fits its LDS column.
    Both plan kernels (one workgroup / many), next to ordinary tiny streams
    and to broken headers of the same size."""
    from rust_snappy_amd import batch
    rng = random.Random(count)
    plain = [bytes(n) for n in (255, 256, 257, 300, 1000, 4096, 5000)]
    plain += [b"ab" * 2000, b"xyz" * 90, b"q" * 256, b"q" * 257]
    ins = []
    for i in range(count):
        if i % 3 == 0:
            ins.append(plain[(i // 3) % len(plain)])
        else:
            ins.append(bytes(rng.randrange(4) for _ in range(
                rng.randrange(0, 240))))
    comps = [O.compress(x) for x in ins]
    assert max(len(c) for c in comps[::3]) < 256
    # broken: a header that promises 4 GiB - 1, and one that never ends
    comps[1] = b"\xff\xff\xff\xff\x0f" + comps[1][1:]
    comps[2] = b"\xff" * 7
    got, errs = gpu_decompress(ctx, comps, caps=[max(len(x), 1) for x in ins])
    for i, x in enumerate(ins):
        if i in (1, 2):
            assert errs[i][0] != 0
            try:
                O.decompress(comps[i], max(len(x), 1))
                raise AssertionError("the oracle accepts the broken stream")
            except O.SnapError as e:
                assert errs[i][0] == e.kind, (i, errs[i], e.key())
        else:
            assert errs[i][0] == 0, (i, errs[i])
            assert got[i] == x, i


@pytest.mark.gpu
def test_product_library_refuses_the_cross_check_options(built):
    """libsnapmi.so (built without SNAPMI_TESTING) beside the test build this
    suite runs on: it compresses and decompresses like it, and the options
    that select the cross-check kernels are arguments it does not know."""
    import ctypes as C
    from rust_snappy_amd import _lib
    P = C.CDLL(str(ROOT / "rust-snappy_amd" / "libsnapmi.so"),
               mode=getattr(__import__("os"), "RTLD_LOCAL", 0))
    assert not hasattr(P, "snapmi_ctx_set_test_option")
    P.snapmi_ctx_create.argtypes = [C.c_int, C.c_void_p,
                                    C.POINTER(C.c_void_p)]
    P.snapmi_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    P.snapmi_ctx_destroy.argtypes = [C.c_void_p]
    P.snapmi_raw_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t),
                                      C.POINTER(_lib.SnapmiError)]
    P.snapmi_raw_decompress.argtypes = P.snapmi_raw_compress.argtypes
    h = C.c_void_p()
    assert P.snapmi_ctx_create(0, None, C.byref(h)) == 0
    try:
        E_ARG = 101
        for name, bad, good in ((b"compress_mode", 2, 1),
                                (b"span_kernel", 0, 1),
                                (b"decode_kernel", 2, 3)):
            assert P.snapmi_ctx_set_option(h, name, bad) == E_ARG, name
            assert P.snapmi_ctx_set_option(h, name, good) == 0, name
        data = (O.CORPUS / "alice29.txt").read_bytes()
        cap = len(data) + len(data) // 6 + 64
        out = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        err = _lib.SnapmiError()
        assert P.snapmi_raw_compress(h, data, len(data), out, cap,
                                     C.byref(n), C.byref(err)) == 0
        comp = out.raw[:n.value]
        assert comp == O.compress(data)
        back = C.create_string_buffer(len(data))
        assert P.snapmi_raw_decompress(h, comp, len(comp), back, len(data),
                                       C.byref(n), C.byref(err)) == 0
        assert back.raw[:n.value] == data
    finally:
        P.snapmi_ctx_destroy(h)
