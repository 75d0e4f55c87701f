"""GPU suite: nothing is ever written outside a caller's buffer.

The reference decodes into `&mut output[..dlen]` (src/decompress.rs:84-94) and
compresses into `output` (src/compress.rs:99-127): a Rust slice cannot be
overrun.  The kernels here store whole 16-byte pieces for shorter elements
("the lanes above repair the excess") and over-copy literals, so the
same property has to be shown: every output of a batch sits between bands of
0xA5 bytes - one in front of the first, one behind every buffer, caps exactly
decompress_len / max_compress_len, the bands' lengths chosen so that the
buffers fall on every alignment - and the bands must be intact after the call.
Run through both decoder generations (`ctx`), the lane-per-stream, sequential,
long-stream and frame paths, and every encoder configuration (`cctx`)."""
import random

import numpy as np
import pytest
import torch

import kats
from conftest import SCAN_GEOMETRIES, set_scan_geometry
import oracle_lib as O

pytestmark = pytest.mark.gpu

GUARD = 0xA5
ERR_DT = np.dtype([("kind", "<i4"), ("r", "<u4"), ("a", "<u8"), ("b", "<u8"),
                   ("c", "<u8")])


class Guarded:
    """Buffers of exactly caps[i] bytes in one slab, a band of >= 64 guard
    bytes in front of the first and behind each one.  `aligned`: every buffer
    starts on a 16-byte boundary (the guard behind it begins at its very last
    byte all the same); otherwise the bands vary by 0..15 bytes so that the
    buffers start at every residue."""

    def __init__(self, caps, seed=0, aligned=False, fill=None):
        rng = random.Random(seed)
        self.caps = [int(c) for c in caps]
        offs, pos = [], 64
        for c in self.caps:
            if aligned:
                pos = (pos + 15) // 16 * 16
            else:
                pos += rng.randrange(16)
            offs.append(pos)
            pos += c + 64
        self.size = pos + 64
        self.offs = np.array(offs, dtype=np.int64)
        host = np.full(self.size, GUARD, dtype=np.uint8)
        if fill is not None:
            for o, b in zip(offs, fill):
                host[o:o + len(b)] = np.frombuffer(bytes(b), dtype=np.uint8)
        self.data = torch.from_numpy(host).cuda()
        self.d_ptrs = torch.from_numpy(self.offs).cuda() + self.data.data_ptr()
        self.d_caps = torch.tensor(self.caps, dtype=torch.int64, device="cuda")
        self.h_caps = torch.tensor(self.caps, dtype=torch.int64)

    def fetch(self):
        self.host = self.data.cpu().numpy()
        return self.host

    def bytes(self, i, n):
        o = int(self.offs[i])
        return self.host[o:o + int(n)].tobytes()

    def assert_guards(self, what=""):
        host = self.fetch()
        inside = np.zeros(self.size + 1, dtype=np.int32)
        np.add.at(inside, self.offs, 1)
        np.add.at(inside, self.offs + np.array(self.caps, dtype=np.int64), -1)
        inside = np.cumsum(inside[:-1]) > 0
        bad = np.flatnonzero(~inside & (host != GUARD))
        if bad.size:
            p = int(bad[0])
            i = int(np.searchsorted(self.offs, p, side="right")) - 1
            o = int(self.offs[max(i, 0)])
            raise AssertionError(
                f"{what}: {bad.size} guard bytes overwritten, first at slab "
                f"offset {p} = buffer {i} (offset {o}, cap "
                f"{self.caps[max(i, 0)]}) {p - o:+d}; value {host[p]:#x}")


def read_errs(t):
    rec = np.frombuffer(t.cpu().numpy().tobytes(), dtype=ERR_DT)
    return [(int(r["kind"]), int(r["a"]), int(r["b"]), int(r["c"]))
            for r in rec]


def decode_guarded(ctx, comps, caps, seed, aligned, in_aligned=True):
    from rust_snappy_amd import raw
    n = len(comps)
    src = Guarded([max(len(c), 1) for c in comps], seed + 1, in_aligned, comps)
    d_in_lens = torch.tensor([len(c) for c in comps], dtype=torch.int64,
                             device="cuda")
    dst = Guarded(caps, seed, aligned)
    out_lens = torch.zeros(n, dtype=torch.int64, device="cuda")
    errs = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    raw.decompress_batch(ctx, src.d_ptrs, d_in_lens, dst.d_ptrs, dst.d_caps,
                         out_lens, errs)
    ctx.synchronize()
    dst.assert_guards(f"decode aligned={aligned}")
    return dst, out_lens.cpu().numpy(), read_errs(errs)


def check_against_oracle(comps, caps, dst, lens, errs):
    ok = bad = 0
    for i, (m, cap) in enumerate(zip(comps, caps)):
        try:
            want = O.decompress(m, cap)
            assert errs[i][0] == 0 and dst.bytes(i, lens[i]) == want, \
                (i, errs[i])
            ok += 1
        except O.SnapError as oe:
            assert (oe.kind, oe.a, oe.b, oe.c) == errs[i], (i, errs[i], oe)
            bad += 1
    return ok, bad


def fuzzed(seed, count, cap_limit=1 << 20):
    import foreign
    rng = random.Random(seed)
    rnd = O.corpus_round()
    base = [O.compress(d[:150000]) for _, d in rnd]
    base += [O.compress(bytes(70000)), O.compress(b"abcd" * 30000),
             O.compress(bytes(rng.randrange(256) for _ in range(70000)))]
    base += [O.compress(d[:n]) for _, d in rnd[:8]
             for n in (40, 150, 300, 700, 1500, 4000)]
    base += [c for c, _ in foreign.cases()[:6]]
    muts = []
    for _ in range(count):
        b = bytearray(rng.choice(base))
        kind = rng.random()
        if kind < 0.15:
            pass  # untouched: a valid stream among the broken ones
        elif kind < 0.65:
            for _ in range(rng.randrange(1, 7)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif kind < 0.78:
            p = rng.randrange(len(b))
            b[p:p] = bytes(rng.randrange(256)
                           for _ in range(rng.randrange(1, 5)))
        elif kind < 0.9:
            p = rng.randrange(len(b))
            del b[p:p + rng.randrange(1, 5)]
        else:
            b = b[:rng.randrange(1, len(b))]
        muts.append(bytes(b))
    caps = []
    for m in muts:
        try:
            caps.append(min(O.decompress_len(m), cap_limit))
        except O.SnapError:
            caps.append(1024)
    return muts, caps


def valid_cases():
    import foreign
    rnd = O.corpus_round()
    datas = [d for _, d in rnd]
    # every tail length mod 16, short streams of every class (lane-per-stream
    # decoder below 256 compressed bytes, second-generation tail, one window)
    txt, jpg = rnd[6][1], rnd[2][1]
    for n in list(range(0, 70)) + [100, 255, 256, 257, 300, 336, 337, 338,
                                   400, 1000, 1023, 1024, 4095, 4097, 65535,
                                   65536, 65537, 70001, 131071]:
        datas.append(txt[7:7 + n])
        datas.append(jpg[:n])
        datas.append((b"ab" * n)[:n])
        datas.append(bytes(n))
    comps = [O.compress(d) for d in datas]
    for c, w in foreign.cases():
        comps.append(c)
        datas.append(w)
    for _, c, w in kats.DECODE_KATS:
        comps.append(c)
        datas.append(w)
    return comps, datas


@pytest.mark.parametrize("aligned", [True, False])
def test_decoders_write_nothing_outside_their_buffers(ctx, aligned):
    comps, datas = valid_cases()
    caps = [len(d) for d in datas]
    dst, lens, errs = decode_guarded(ctx, comps, caps, 3, aligned,
                                     in_aligned=aligned)
    for i, d in enumerate(datas):
        assert errs[i][0] == 0, (i, errs[i])
        assert dst.bytes(i, lens[i]) == d, i


@pytest.mark.parametrize("aligned", [True, False])
def test_decoders_keep_to_their_buffers_on_errors_and_fuzz(ctx, aligned):
    comps = [k[1] for k in kats.ERROR_KATS]
    caps = []
    for name, data, want, bad_header in kats.ERROR_KATS:
        caps.append(1024 if bad_header else O.decompress_len(data))
    m, c = fuzzed(101, 2000)
    comps += m
    caps += c
    dst, lens, errs = decode_guarded(ctx, comps, caps, 5, aligned)
    ok, bad = check_against_oracle(comps, caps, dst, lens, errs)
    assert ok > 200 and bad > 800


def test_decoders_keep_to_short_buffers(ctx):
    """Buffers SHORTER than the stream's output (the reference answers
    BufferTooSmall before it decodes, src/decompress.rs:84-89) and buffers
    that are larger than the output: the bands hold either way."""
    comps, datas = valid_cases()
    rng = random.Random(8)
    caps = []
    for d in datas:
        r = rng.random()
        if r < 0.4 and len(d) > 0:
            caps.append(rng.randrange(0, len(d)))
        elif r < 0.7:
            caps.append(len(d) + rng.randrange(1, 40))
        else:
            caps.append(len(d))
    dst, lens, errs = decode_guarded(ctx, comps, caps, 9, False)
    check_against_oracle(comps, caps, dst, lens, errs)


def test_sequential_decoder_keeps_to_its_buffers(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import rust_snappy_amd as R
    c = R.raw.Context(0)
    c.set_option("decode_kernel", 0)
    try:
        comps, caps = fuzzed(55, 600, cap_limit=1 << 18)
        dst, lens, errs = decode_guarded(c, comps, caps, 2, False)
        check_against_oracle(comps, caps, dst, lens, errs)
    finally:
        c.close()


@pytest.mark.parametrize("geom", SCAN_GEOMETRIES)
def test_long_stream_paths_keep_to_their_buffers(ctx, geom):
    """snapmi_decompress_stream (scan, cuts, pieces) and the long streams of a
    batch (k_bstream_*): outputs of exactly the announced length between
    bands; inputs in allocations of exactly their size (the scan's last
    workgroup begins behind a stream of k * 256 KiB - 100 bytes: ADVICE
    round 4) - carved from the end of a 2 MiB-multiple allocation, so a read
    behind them leaves the allocation."""
    import foreign
    from rust_snappy_amd import raw
    seg = set_scan_geometry(ctx, geom)
    rnd = O.corpus_round()
    rng = random.Random(4)
    big = b"".join(d for _, d in rnd)
    noise = bytes(rng.randrange(256) for _ in range(1 << 20))

    def with_compressed_len(target):
        # text, then incompressible bytes whose count sets the length
        body = big[:600000]
        base = len(O.compress(body))
        pad = max(target - base - 64, 0)
        for _ in range(64):
            c = O.compress(body + noise[:pad])
            if len(c) == target:
                return body + noise[:pad]
            pad += target - len(c)
            assert pad >= 0
        raise AssertionError("no input of that compressed length")

    # (the scan's last workgroup begins behind the input when the segment
    # count is a multiple of 64: k * 256 KiB - 100 with 4 KiB segments,
    # k * 64 KiB - 100 with 1 KiB)
    datas = [with_compressed_len(k * 262144 - 100) for k in (2, 3)]
    datas += [with_compressed_len(k * 65536 - 100) for k in (7,)]
    datas += [with_compressed_len(2 * 262144 - 4095),
              with_compressed_len(2 * 262144 - 1),
              with_compressed_len(2 * 262144 + 1),
              big * 2, rnd[2][1] * 3, bytes(300001),
              bytes(range(256)) * 1001 + b"x"]
    comps = [O.compress(d) for d in datas]
    assert len(comps[0]) == 2 * 262144 - 100
    for c, w in foreign.cases()[3:]:
        comps.append(c)
        datas.append(w)
    # one stream at a time through snapmi_decompress_stream
    for i, (comp, data) in enumerate(zip(comps, datas)):
        n = len(comp)
        room = (n + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        big_in = torch.empty(room, dtype=torch.uint8, device="cuda")
        d_in = big_in[room - n:]
        d_in.copy_(torch.frombuffer(bytearray(comp), dtype=torch.uint8))
        dst = Guarded([len(data)], i, aligned=bool(i & 1))
        out_len = torch.zeros(1, dtype=torch.int64, device="cuda")
        err = torch.zeros(32, dtype=torch.uint8, device="cuda")
        d_out = dst.data[int(dst.offs[0]):int(dst.offs[0]) + len(data)]
        raw.decompress_stream(ctx, d_in, n, d_out, out_len, err)
        ctx.synchronize()
        dst.assert_guards(f"decompress_stream case {i}")
        assert read_errs(err)[0][0] == 0, i
        assert dst.bytes(0, int(out_len.item())) == data, i
        del big_in
    # ... and all of them, with short streams between, as one batch
    ctx.set_option("batch_long_streams", 1)
    short = [O.compress(rnd[6][1][:n]) for n in (10, 300, 5000, 70000)]
    allc = []
    alld = []
    for c, d in zip(comps, datas):
        allc += [c] + short
        alld += [d] + [rnd[6][1][:n] for n in (10, 300, 5000, 70000)]
    for aligned in (True, False):
        dst, lens, errs = decode_guarded(ctx, allc, [len(d) for d in alld], 12,
                                         aligned, in_aligned=aligned)
        for i, d in enumerate(alld):
            assert errs[i][0] == 0 and dst.bytes(i, lens[i]) == d, i
    # corrupted long streams: the fallback decoders inside the same bands
    muts, caps = [], []
    for c in comps[:6]:
        for _ in range(6):
            b = bytearray(c)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
            muts.append(bytes(b))
    for m in muts:
        try:
            caps.append(min(O.decompress_len(m), 1 << 22))
        except O.SnapError:
            caps.append(1024)
    dst, lens, errs = decode_guarded(ctx, muts, caps, 13, False)
    check_against_oracle(muts, caps, dst, lens, errs)
    set_scan_geometry(ctx, None)


def encode_guarded(ctx, datas, seed, aligned, caps=None):
    from rust_snappy_amd import raw
    n = len(datas)
    src = Guarded([max(len(d), 1) for d in datas], seed + 1, aligned, datas)
    h_lens = torch.tensor([len(d) for d in datas], dtype=torch.int64)
    d_lens = h_lens.cuda()
    if caps is None:
        caps = [raw.max_compress_len(len(d)) for d in datas]
    dst = Guarded(caps, seed, aligned)
    out_lens = torch.zeros(n, dtype=torch.int64, device="cuda")
    errs = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    raw.compress_batch(ctx, src.d_ptrs, d_lens, dst.d_ptrs, dst.d_caps,
                       out_lens, errs, host_in_lens=h_lens)
    ctx.synchronize()
    dst.assert_guards(f"encode aligned={aligned}")
    src.assert_guards("encode: the INPUT slab")
    return dst, out_lens.cpu().numpy(), read_errs(errs)


def encoder_inputs():
    rnd = O.corpus_round()
    rng = random.Random(31)
    txt, jpg = rnd[6][1], rnd[2][1]
    datas = [d for _, d in rnd]
    for n in list(range(0, 40)) + [100, 255, 256, 257, 511, 512, 1023, 1024,
                                   1025, 2047, 2048, 4096, 8191, 8192, 8193,
                                   16384, 65535, 65536, 65537, 131073]:
        datas.append(txt[3:3 + n])
        datas.append(jpg[:n])       # incompressible: output longest vs cap
        datas.append(bytes(n))
    for _ in range(60):
        alpha = rng.choice([1, 2, 3, 4, 16, 256])
        n = rng.choice([15, 17, 63, 65, 1000, 5000, 70000,
                        rng.randrange(0, 200000)])
        datas.append(bytes(rng.choices(range(alpha), k=n)))
    return datas


@pytest.mark.parametrize("aligned", [True, False])
def test_encoders_write_nothing_outside_their_buffers(cctx, aligned):
    datas = encoder_inputs()
    dst, lens, errs = encode_guarded(cctx, datas, 21, aligned)
    for i, d in enumerate(datas):
        assert errs[i][0] == 0, (i, errs[i])
        assert dst.bytes(i, lens[i]) == O.compress(d), (i, len(d))


def test_encoders_leave_refused_buffers_alone(cctx):
    """A capacity under max_compress_len is BufferTooSmall
    (src/compress.rs:111-116) and nothing is written, whatever the other
    streams of the batch do."""
    from rust_snappy_amd import raw
    datas = encoder_inputs()[:80]
    rng = random.Random(2)
    caps = []
    for d in datas:
        full = raw.max_compress_len(len(d))
        caps.append(full if rng.random() < 0.5 else rng.randrange(0, full))
    dst, lens, errs = encode_guarded(cctx, datas, 23, False, caps)
    for i, d in enumerate(datas):
        full = raw.max_compress_len(len(d))
        if caps[i] < full:
            assert errs[i] == (2, caps[i], full, 0), (i, errs[i])
            assert lens[i] == 0
        else:
            assert errs[i][0] == 0
            assert dst.bytes(i, lens[i]) == O.compress(d), i


def test_frame_calls_keep_to_their_buffers(ctx):
    """snapmi_frame_compress / _decompress on buffers of exactly
    snapmi_frame_max_len / the decoded length, between bands."""
    import ctypes as C
    from rust_snappy_amd import _lib, frame, raw
    L = _lib.load()
    rnd = O.corpus_round()
    rng = random.Random(6)
    datas = [rnd[0][1], rnd[2][1], rnd[6][1][:65536], rnd[6][1][:65537],
             rnd[1][1], b"", b"x", bytes(200001),
             bytes(rng.randrange(256) for _ in range(70001))]
    for i, d in enumerate(datas):
        n = len(d)
        src = Guarded([max(n, 1)], i, aligned=not (i & 1), fill=[d])
        cap = frame.frame_max_len(n)
        dst = Guarded([cap], i + 50, aligned=not (i & 2))
        out_len = torch.zeros(1, dtype=torch.int64, device="cuda")
        rc = L.snapmi_frame_compress(
            ctx._h, C.c_void_p(int(src.d_ptrs[0].item())) if n else None, n,
            C.c_void_p(int(dst.d_ptrs[0].item())), cap,
            C.c_void_p(out_len.data_ptr()), None)
        assert rc == 0
        ctx.synchronize()
        dst.assert_guards(f"frame_compress case {i}")
        m = int(out_len.item())
        framed = dst.bytes(0, m)
        assert framed == O.frame_compress(d), i
        # decode: framed bytes in an exact buffer, output exactly n bytes
        fin = Guarded([m], i + 70, aligned=bool(i & 1), fill=[framed])
        out = Guarded([n], i + 90, aligned=bool(i & 2))
        err = torch.zeros(32, dtype=torch.uint8, device="cuda")
        for index in (False,):
            rc = L.snapmi_frame_decompress(
                ctx._h, C.c_void_p(int(fin.d_ptrs[0].item())), m,
                C.c_void_p(int(out.d_ptrs[0].item())) if n else None, n,
                C.c_void_p(out_len.data_ptr()), C.c_void_p(err.data_ptr()),
                None, 0)
            assert rc == 0
            ctx.synchronize()
            out.assert_guards(f"frame_decompress case {i}")
            assert read_errs(err)[0][0] == 0, i
            assert out.bytes(0, int(out_len.item())) == d, i
        # a corrupted frame (bad CRC / bad body): still inside the bands
        if m > 30:
            b = bytearray(framed)
            for _ in range(3):
                b[rng.randrange(10, m)] ^= 1 + rng.randrange(255)
            fin = Guarded([m], i + 110, aligned=False, fill=[bytes(b)])
            out = Guarded([n], i + 130, aligned=False)
            rc = L.snapmi_frame_decompress(
                ctx._h, C.c_void_p(int(fin.d_ptrs[0].item())), m,
                C.c_void_p(int(out.d_ptrs[0].item())) if n else None, n,
                C.c_void_p(out_len.data_ptr()), C.c_void_p(err.data_ptr()),
                None, 0)
            assert rc == 0
            ctx.synchronize()
            out.assert_guards(f"frame_decompress of a corrupted case {i}")


def test_host_frame_calls_keep_to_their_buffers(ctx):
    """snapmi_frame_encode_host / _decode_host (the copy kernel k_to_host
    writes mapped pinned memory): pinned buffers with bands around the
    regions handed to the calls."""
    from rust_snappy_amd import frame
    rnd = O.corpus_round()
    data = b"".join(d for _, d in rnd)[:3000001]
    lens = [65536] * (len(data) // 65536)
    if len(data) % 65536:
        lens.append(len(data) % 65536)
    want = O.frame_compress(data)
    hin = frame.HostBuffer(len(data) + 256)
    from rust_snappy_amd import _lib
    import ctypes as C
    L = _lib.load()
    bound = L.snapmi_frame_encode_bound(len(data), len(lens))
    hout = frame.HostBuffer(bound + 256)
    vin = hin.array
    vout = hout.array
    vin[:] = GUARD
    vout[:] = GUARD
    vin[128:128 + len(data)] = np.frombuffer(data, dtype=np.uint8)
    arr = np.array(lens, dtype=np.uint32)
    written = C.c_size_t(0)
    rc = L.snapmi_frame_encode_host(
        ctx._h, C.c_void_p(hin.ptr + 128),
        arr.ctypes.data_as(C.c_void_p), len(lens), 0,
        C.c_void_p(hout.ptr + 67), bound, C.byref(written))
    assert rc == 0
    assert vout[67:67 + written.value].tobytes() == want
    assert (vout[:67] == GUARD).all()
    assert (vout[67 + bound:] == GUARD).all()
    assert (vin[:128] == GUARD).all() and (vin[128 + len(data):] == GUARD).all()
    # decode at odd addresses; the call wants 65 536 bytes of room per data
    # chunk (include/snapmi.h), so the output region is len(lens) * 65536
    room = len(lens) * 65536
    fin = frame.HostBuffer(len(want) + 256)
    fout = frame.HostBuffer(room + 256)
    wi = fin.array
    wo = fout.array
    wi[:] = GUARD
    wo[:] = GUARD
    wi[61:61 + len(want)] = np.frombuffer(want, dtype=np.uint8)
    stale = (C.c_uint8 * 10)()
    consumed = C.c_size_t(0)
    err = _lib.SnapmiError()
    rc = L.snapmi_frame_decode_host(
        ctx._h, C.c_void_p(fin.ptr + 61), len(want), 2, stale,
        C.c_void_p(fout.ptr + 77), room, C.byref(written),
        C.byref(consumed), C.byref(err))
    assert rc == 0, (rc, err.kind)
    assert written.value == len(data) and consumed.value == len(want)
    assert wo[77:77 + len(data)].tobytes() == data
    assert (wo[:77] == GUARD).all() and (wo[77 + room:] == GUARD).all()
    del vin, vout, wi, wo
    for b in (hin, hout, fin, fout):
        b.close()


@pytest.mark.parametrize("count", [1, 4000])
def test_small_streams_decoded_32_per_wavefront(ctx, count):
    """k_decompress_small (round 5): raw streams of under 512 compressed bytes
    whose output is at most 512 bytes, decoded one per lane on half the lanes
    of a wavefront with input and output in LDS - every length around the
    class limits (255 / 256 / 511 / 512 bytes in, 256 / 257 / 512 / 513 out),
    text, runs, incompressible bytes and foreign elements, next to tiny and
    larger streams in one batch, between guard bands; then the same streams
    broken, truncated and with buffers one byte short, against the oracle's
    errors."""
    import foreign
    rng = random.Random(17 + count)
    txt = (O.CORPUS / "alice29.txt").read_bytes()
    jpg = (O.CORPUS / "fireworks.jpeg").read_bytes()
    datas = []
    for n in list(range(240, 530, 3)) + [255, 256, 257, 511, 512, 513, 600]:
        o = rng.randrange(len(txt) - 1000)
        datas += [txt[o:o + n], jpg[100:100 + n], bytes(n),
                  (txt[o:o + 7] * 80)[:n]]
    comps = [O.compress(d) for d in datas]
    # foreign elements in that size class: copy-4, 4-byte literal lengths
    for seed in range(40):
        c, w = foreign.build(100 + seed, rng.randrange(200, 520))
        comps.append(c)
        datas.append(w)
    # ... and enough of them to fill many wavefronts
    while len(comps) < count:
        k = rng.randrange(len(datas))
        comps.append(comps[k])
        datas.append(datas[k])
    classes = sum(1 for c, d in zip(comps, datas)
                  if len(c) < 512 and len(d) <= 512
                  and not (len(c) < 256 and len(d) <= 256))
    assert classes > 100
    caps = [len(d) for d in datas]
    for aligned in (True, False):
        dst, lens, errs = decode_guarded(ctx, comps, caps, 31, aligned,
                                         in_aligned=aligned)
        for i, d in enumerate(datas):
            assert errs[i][0] == 0, (i, errs[i])
            assert dst.bytes(i, lens[i]) == d, i
    # broken streams of the class, short buffers
    muts, mcaps = [], []
    for c, d in list(zip(comps, datas))[:400]:
        b = bytearray(c)
        r = rng.random()
        if r < 0.5 and len(b) > 3:
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        elif r < 0.75 and len(b) > 2:
            b = b[:rng.randrange(1, len(b))]
        muts.append(bytes(b))
        try:
            cap = min(O.decompress_len(bytes(b)), 1 << 16)
        except O.SnapError:
            cap = 600
        if rng.random() < 0.2 and cap:
            cap -= 1
        mcaps.append(cap)
    dst, lens, errs = decode_guarded(ctx, muts, mcaps, 33, False)
    ok, bad = check_against_oracle(muts, mcaps, dst, lens, errs)
    assert ok > 20 and bad > 100
