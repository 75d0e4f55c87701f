"""GPU suite for the frame layer: CRC32C kernel, framed
bytes equal to the oracle's restatement of write::FrameEncoder, round trips,
and FrameDecoder errors with the reference's variants and fields."""
import io
import random
import struct

import numpy as np
import pytest
import torch

import kats
import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_crc32c_kernel_known_answers(ctx):
    from rust_snappy_amd import frame
    assert frame.crc32c_masked(ctx, b"123456789") == 0xC78AB0E5
    rng = random.Random(1)
    for n in [0, 1, 2, 3, 4, 5, 7, 8, 63, 64, 255, 256, 257, 1000, 4095, 4096,
              65535, 65536]:
        d = bytes(rng.randrange(256) for _ in range(n))
        assert frame.crc32c_masked(ctx, d) == O.crc32c_masked(d), n
    html = (O.CORPUS / "html").read_bytes()
    for off in range(1, 9):  # unaligned starts
        assert frame.crc32c_masked(ctx, html[off:off + 65536 - 8]) == \
            O.crc32c_masked(html[off:off + 65536 - 8])


def framed(ctx, data):
    from rust_snappy_amd import frame
    sink = io.BytesIO()
    enc = frame.FrameEncoder(sink, ctx)
    enc.write_all(data)
    return enc.into_inner().getvalue()


def test_frame_bytes_equal_oracle_on_corpus(ctx):
    # read_and_write_frame_encoder_match / roundtrip_frame of the reference
    # (test/tests.rs:76-88) on its inputs (:180-195,:469-504)
    from rust_snappy_amd import frame
    names = ["html", "urls.10K", "fireworks.jpeg", "paper-100k.pdf",
             "html_x_4", "alice29.txt", "asyoulik.txt", "lcet10.txt",
             "plrabn12.txt", "geo.protodata", "kppkn.gtb",
             "Mark.Twain-Tom.Sawyer.txt"]
    datas = [(O.CORPUS / n).read_bytes() for n in names]
    datas += [b"", b"\x00", kats.RANDOM1, kats.RANDOM2, kats.RANDOM3,
              kats.RANDOM4, b"a" * 65536, b"ab" * 40000]
    for d in datas:
        f = framed(ctx, d)
        assert f == O.frame_compress(d), len(d)
        back = frame.FrameDecoder(io.BytesIO(f), ctx).read_to_end()
        assert back == d
        # read::FrameEncoder produces the same bytes (test/tests.rs:83-88)
        assert frame.ReadFrameEncoder(io.BytesIO(d), ctx).read() == f


def test_frame_structure_expected_sizes(ctx):
    want = {"html": 22872, "urls.10K": 335620, "fireworks.jpeg": 123119,
            "paper-100k.pdf": 85327, "kppkn.gtb": 69566}
    for name, size in want.items():
        assert len(framed(ctx, (O.CORPUS / name).read_bytes())) == size


def test_frame_random_roundtrip(ctx):
    from rust_snappy_amd import frame
    rng = random.Random(9)
    for _ in range(40):
        alpha = rng.choice([1, 2, 4, 16, 256])
        n = rng.choice([1, 17, 65535, 65536, 65537, 131072,
                        rng.randrange(1, 300000)])
        d = bytes(rng.choices(range(alpha), k=n))
        f = framed(ctx, d)
        assert f == O.frame_compress(d), (alpha, n)
        assert frame.FrameDecoder(io.BytesIO(f), ctx).read_to_end() == d


def test_frame_decoder_with_side_index_device(ctx):
    from rust_snappy_amd import frame
    d = (O.CORPUS / "lcet10.txt").read_bytes() * 3
    d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    out, n, index = frame.compress_device(ctx, d_in, want_index=True)
    assert out[:n].cpu().numpy().tobytes() == O.frame_compress(d)
    back, m = frame.decompress_device(ctx, out, n, index=index)
    assert back[:m].cpu().numpy().tobytes() == d
    back, m = frame.decompress_device(ctx, out, n)   # device header walk
    assert back[:m].cpu().numpy().tobytes() == d


def expect_error(ctx, stream, key):
    import rust_snappy_amd as R
    from rust_snappy_amd import frame
    with pytest.raises(R.Error) as ei:
        frame.FrameDecoder(io.BytesIO(stream), ctx).read_to_end()
    got = ei.value
    want_oracle = None
    try:
        O.frame_decompress(stream)
    except O.SnapError as oe:
        want_oracle = oe
    assert want_oracle is not None, "oracle accepted the stream"
    if want_oracle.kind == -1:
        assert got.variant == "UnexpectedEof"
    elif got.variant == "StreamHeaderMismatch":
        # reference field: bytes: Vec<u8>; the ABI packs the 6 bytes LE
        body = stream[4:10]
        assert got.kind == want_oracle.kind
        assert got.fields["bytes"] == int.from_bytes(body, "little")
    else:
        # every field the variant has, and the unused ones must be zero
        assert got.kind == want_oracle.kind
        assert got.abc == (want_oracle.a, want_oracle.b, want_oracle.c), \
            (got, want_oracle)
    assert got.variant == key, (got, key)
    return got


def test_frame_decoder_errors(ctx):
    good = O.frame_compress((O.CORPUS / "html").read_bytes())
    ident = b"\xff\x06\x00\x00sNaPpY"
    # issue #42 regression of the reference (test/tests.rs:536-545)
    expect_error(ctx, b"123", "UnexpectedEof")
    expect_error(ctx, b"\x00\x05\x00\x00abcde", "StreamHeader")
    expect_error(ctx, b"\xff\x06\x00\x00sNaPpX", "StreamHeaderMismatch")
    expect_error(ctx, b"\xff\x05\x00\x00sNaPp", "UnsupportedChunkLength")
    expect_error(ctx, ident + b"\x02\x01\x00\x00a", "UnsupportedChunkType")
    expect_error(ctx, ident + b"\x00\xff\xff\xff", "UnsupportedChunkLength")
    expect_error(ctx, ident + b"\x00\x03\x00\x00abc", "UnsupportedChunkLength")
    # checksum mismatch
    bad = bytearray(good)
    bad[10 + 4] ^= 0x55
    expect_error(ctx, bytes(bad), "Checksum")
    # corrupt payload -> raw decoder error surfaces
    bad = bytearray(good)
    bad[10 + 8 + 3 + 5] ^= 0xFF
    with pytest.raises(O.SnapError) as oe:   # the oracle rejects it ...
        O.frame_decompress(bytes(bad))
    assert oe.value.kind in (5, 6, 7, 8, 9, 14)  # raw decode error or CRC
    expect_error(ctx, bytes(bad), O.KIND_NAMES[oe.value.kind])  # ... same here
    # a sweep of single-byte corruptions of the first payload: whatever the
    # oracle says (error variant + fields, or the decoded bytes), the GPU says
    rng = random.Random(5)
    from rust_snappy_amd import frame as _frame
    for _ in range(40):
        bad = bytearray(good)
        bad[10 + 8 + rng.randrange(3, 2000)] ^= 1 << rng.randrange(8)
        try:
            want = O.frame_decompress(bytes(bad))
        except O.SnapError as oe2:
            expect_error(ctx, bytes(bad), O.KIND_NAMES[oe2.kind])
        else:
            assert _frame.FrameDecoder(io.BytesIO(bytes(bad)),
                                       ctx).read_to_end() == want
    # truncated stream
    expect_error(ctx, good[:-5], "UnexpectedEof")
    # skippable + padding chunks are skipped (src/read.rs:143-158)
    from rust_snappy_amd import frame
    s = ident + b"\x80\x03\x00\x00xyz" + b"\xfe\x02\x00\x00pp" + good[10:] \
        + ident
    assert frame.FrameDecoder(io.BytesIO(s), ctx).read_to_end() == \
        O.frame_decompress(s)


def test_frame_flush_boundaries(ctx):
    """A flush ends the current chunk (src/write.rs:154-161); the stream
    identifier is written once."""
    from rust_snappy_amd import frame
    a = (O.CORPUS / "alice29.txt").read_bytes()[:100000]
    b = (O.CORPUS / "asyoulik.txt").read_bytes()[:70000]
    sink = io.BytesIO()
    enc = frame.FrameEncoder(sink, ctx)
    enc.write(a)
    enc.flush()
    enc.write(b)
    enc.into_inner()
    want = O.frame_compress(a) + O.frame_compress(b)[10:]
    assert sink.getvalue() == want
    assert frame.FrameDecoder(io.BytesIO(want), ctx).read_to_end() == a + b


def test_frame_compress_in_segments(built):
    """A stream longer than the encoder's segment (64 chunks here, 262 144 by
    default) is framed segment by segment; bytes and side index must not
    depend on the segmentation."""
    import numpy as np
    import rust_snappy_amd as R
    from rust_snappy_amd import frame
    data = b"".join(d for _, d in O.corpus_round()) * 4   # 179 chunks
    want = O.frame_compress(data)
    for mode in (0, 1):
        c = R.raw.Context(0)
        c.set_option("compress_mode", mode)
        c.set_option("lane_min_blocks", 1)
        c.set_option("lane_segment_blocks", 64)
        d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        out, flen, index = frame.compress_device(c, d, want_index=True)
        got = out[:flen].cpu().numpy().tobytes()
        assert got == want, mode
        idx = index.cpu().numpy()
        assert idx[0] == 10 and idx[-1] == len(want)
        back, m = frame.decompress_device(c, out, flen, index=index,
                                          out_cap=len(data))
        assert m == len(data)
        assert back[:m].cpu().numpy().tobytes() == data
        c.close()


def test_frame_decoder_host_index_with_other_chunk_types(ctx):
    """FrameDecoder scans the chunk headers on the host; skippable, padding
    and repeated identifier chunks between the data chunks do not matter."""
    from rust_snappy_amd import frame
    data = b"".join(d for _, d in O.corpus_round()[:4])
    f = O.frame_compress(data)
    offs = frame.index_host(f)
    first = int(offs[1] - offs[0])
    g = (f[:10] + bytes([0x80, 3, 0, 0, 1, 2, 3]) + f[10:10 + first]
         + bytes([0xFE, 2, 0, 0, 9, 9]) + f[:10] + f[10 + first:])
    assert frame.FrameDecoder(io.BytesIO(g), ctx).read_to_end() == data
    assert frame.FrameDecoder(io.BytesIO(f), ctx).read_to_end() == data


def test_read_frame_encoder_big_and_little_buffers(ctx):
    """test/tests.rs:321-340: the framed bytes do not depend on how the
    caller reads (1 MB reads vs 5-byte reads); same for FrameDecoder."""
    from rust_snappy_amd import frame
    data = (O.CORPUS / "alice29.txt").read_bytes()
    want = O.frame_compress(data)

    def drain(rd, step):
        out = bytearray()
        while True:
            b = rd.read(step)
            if not b:
                return bytes(out)
            out += b

    assert drain(frame.ReadFrameEncoder(io.BytesIO(data), ctx), 1 << 20) == want
    assert drain(frame.ReadFrameEncoder(io.BytesIO(data), ctx), 5) == want
    assert drain(frame.FrameDecoder(io.BytesIO(want), ctx), 5) == data
    assert drain(frame.FrameDecoder(io.BytesIO(want), ctx), 1 << 20) == data


# ---------------------------------------------------------------------
# streaming semantics of the adapters (reference src/write.rs:123-192,
# src/read.rs:105-238,342-409)
# ---------------------------------------------------------------------
def reference_chunks(ops):
    """Where write::FrameEncoder cuts chunks for a sequence of ("w", bytes) /
    ("f",) operations followed by into_inner(): a literal restatement of
    src/write.rs:123-161 (64 KiB buffer `src`; a write larger than the free
    space goes out directly when the buffer is empty, else fills and flushes
    the buffer)."""
    CAP = 65536
    src, out = bytearray(), []

    def inner_write(buf):
        for o in range(0, len(buf), CAP):
            out.append(bytes(buf[o:o + CAP]))
        return len(buf)

    def flush():
        if src:
            inner_write(bytes(src))
            del src[:]

    for op in ops:
        if op[0] == "f":
            flush()
            continue
        buf = op[1]
        while True:
            free = CAP - len(src)
            if len(buf) <= free:
                break
            if not src:
                n = inner_write(buf)
            else:
                src.extend(buf[:free])
                flush()
                n = free
            buf = buf[n:]
        src.extend(buf)
    flush()
    return out


def frame_of_chunks(chunks):
    """The framed stream whose chunks are exactly `chunks`: every chunk
    through the oracle's compress_frame (= frame_compress of <= 64 KiB)."""
    if not chunks:
        return b""
    return b"\xff\x06\x00\x00sNaPpY" + b"".join(
        O.frame_compress(c)[10:] for c in chunks)


def test_frame_encoder_write_state_machine(ctx):
    from rust_snappy_amd import frame
    text = (O.CORPUS / "lcet10.txt").read_bytes()
    jpg = (O.CORPUS / "fireworks.jpeg").read_bytes()
    # the advisor's example: two writes of 100000 bytes, no flush between
    ops = [("w", text[:100000]), ("w", text[100000:200000])]
    assert [len(c) for c in reference_chunks(ops)] == [65536, 34464, 65536,
                                                       34464]
    rng = random.Random(21)
    cases = [ops,
             [("w", text[:65536]), ("w", text[65536:65537])],
             [("w", text[:10]), ("w", text[10:200000])],
             [("w", text[:65535]), ("w", text[65535:65537]), ("f",),
              ("w", jpg), ("w", b"")],
             [("w", b"")], []]
    for _ in range(12):   # random mixes of small / large writes and flushes
        pos, case = 0, []
        blob = text + jpg + text
        while pos < len(blob):
            n = rng.choice([1, 7, 100, 4096, 65535, 65536, 65537, 100000,
                            200000])
            case.append(("w", blob[pos:pos + n]))
            pos += n
            if rng.random() < 0.2:
                case.append(("f",))
        cases.append(case)
    for case in cases:
        for batch in (frame.BATCH_BYTES, 1 << 16, 3 << 16):
            sink = io.BytesIO()
            enc = frame.FrameEncoder(sink, ctx, batch_bytes=batch)
            for op in case:
                if op[0] == "w":
                    assert enc.write(op[1]) == len(op[1])
                else:
                    enc.flush()
            got = enc.into_inner().getvalue()
            want = frame_of_chunks(reference_chunks(case))
            assert got == want, ([len(c) for c in reference_chunks(case)],
                                 batch)
            data = b"".join(op[1] for op in case if op[0] == "w")
            assert frame.FrameDecoder(io.BytesIO(got), ctx).read_to_end() \
                == data


def test_frame_encoder_emits_before_flush_and_into_inner_error(ctx):
    """With a small batch the writer sees chunks while the stream is still
    being written (bounded memory); a failing inner writer surfaces as
    IntoInnerError carrying the encoder (src/write.rs:91-97)."""
    from rust_snappy_amd import frame
    text = (O.CORPUS / "plrabn12.txt").read_bytes()
    sink = io.BytesIO()
    enc = frame.FrameEncoder(sink, ctx, batch_bytes=1 << 17)
    enc.write(text[:300000])
    assert len(sink.getvalue()) > 10     # four whole chunks are out already
    enc.write(text[300000:])
    assert enc.into_inner().getvalue() == frame_of_chunks(reference_chunks(
        [("w", text[:300000]), ("w", text[300000:])]))

    class Broken(io.RawIOBase):
        def write(self, b):
            raise OSError("disk full")

    enc = frame.FrameEncoder(Broken(), ctx)
    enc.write(b"abc")
    with pytest.raises(frame.IntoInnerError) as ei:
        enc.into_inner()
    assert isinstance(ei.value.error(), OSError)
    assert ei.value.into_inner() is enc


class Dribble(io.RawIOBase):
    """A reader that returns at most `step` bytes per read call."""

    def __init__(self, data, step):
        self.b, self.step = io.BytesIO(data), step

    def read(self, n=-1):
        n = self.step if n is None or n < 0 else min(n, self.step)
        return self.b.read(n)


def test_frame_decoder_streams_in_batches(ctx):
    from rust_snappy_amd import frame
    data = b"".join(d for _, d in O.corpus_round()) * 2      # ~90 chunks
    f = O.frame_compress(data)
    for batch, step in ((1 << 17, 1 << 20), (1 << 17, 70000), (1 << 18, 999),
                        (frame.BATCH_BYTES, 1 << 20)):
        dec = frame.FrameDecoder(Dribble(f, step), ctx, batch_bytes=batch)
        out = bytearray()
        while True:
            b = dec.read(100000)
            if not b:
                break
            out += b
        assert bytes(out) == data, (batch, step)
    # the reader is only asked for what a batch needs: after the first read
    # of a small-batch decoder most of the stream is still unread
    rd = io.BytesIO(f)
    dec = frame.FrameDecoder(rd, ctx, batch_bytes=1 << 17)
    assert dec.read(10) == data[:10]
    assert rd.tell() <= (1 << 17) + 10
    # concatenated streams / other chunk types across batch boundaries
    g = f + f[:10] + bytes([0x80, 3, 0, 0, 1, 2, 3]) + f[10:] + \
        bytes([0xFE, 2, 0, 0, 9, 9])
    dec = frame.FrameDecoder(io.BytesIO(g), ctx, batch_bytes=1 << 17)
    assert dec.read_to_end() == data + data == O.frame_decompress(g)


def test_frame_decoder_returns_good_chunks_before_the_error(ctx):
    """src/read.rs:111-118: bytes of earlier chunks are handed out before the
    read that reaches a bad chunk fails."""
    import rust_snappy_amd as R
    from rust_snappy_amd import frame
    data = (O.CORPUS / "alice29.txt").read_bytes()           # 3 chunks
    f = bytearray(O.frame_compress(data))
    offs = frame.index_host(bytes(f))
    f[int(offs[1]) + 8 + 20] ^= 0xFF                          # 2nd chunk body
    with pytest.raises(O.SnapError) as oe:
        O.frame_decompress(bytes(f))
    for batch in (1 << 17, frame.BATCH_BYTES):
        dec = frame.FrameDecoder(io.BytesIO(bytes(f)), ctx, batch_bytes=batch)
        assert dec.read(65536) == data[:65536]   # the good chunk is readable
        with pytest.raises(R.Error) as ei:
            dec.read(1)
        assert ei.value.kind == oe.value.kind
        assert ei.value.abc == (oe.value.a, oe.value.b, oe.value.c)
        dec = frame.FrameDecoder(io.BytesIO(bytes(f)), ctx, batch_bytes=batch)
        with pytest.raises(R.Error) as ei:
            dec.read_to_end()
        assert ei.value.partial == data[:65536]
    # truncated in the third chunk: two chunks readable, then UnexpectedEof
    g = O.frame_compress(data)[:-7]
    dec = frame.FrameDecoder(io.BytesIO(g), ctx, batch_bytes=1 << 17)
    assert dec.read(1 << 20) == data[:131072]
    with pytest.raises(R.Error) as ei:
        dec.read(1)
    assert ei.value.variant == "UnexpectedEof"


def test_device_decode_reports_the_valid_prefix(ctx):
    """The device entry point behind the decoder: the length of the output in
    front of the failing chunk comes back next to the error."""
    from rust_snappy_amd import frame
    data = (O.CORPUS / "alice29.txt").read_bytes()           # 3 chunks
    f = bytearray(O.frame_compress(data))
    offs = frame.index_host(bytes(f))
    f[int(offs[1]) + 8 + 20] ^= 0xFF                          # 2nd chunk body
    with pytest.raises(O.SnapError) as oe:
        O.frame_decompress(bytes(f))
    d_in = torch.frombuffer(bytearray(f), dtype=torch.uint8).cuda()
    for index in (None, torch.from_numpy(offs).cuda()):
        good, err = frame.decompress_batch_device(ctx, d_in, len(f), 3, index)
        assert good == data[:65536] and err.kind == oe.value.kind


def test_read_frame_encoder_chunks_follow_the_reads(ctx):
    """read::FrameEncoder makes ONE read of up to 65536 bytes per chunk
    (src/read.rs:378): a reader that returns short reads gets short chunks."""
    from rust_snappy_amd import frame
    data = (O.CORPUS / "asyoulik.txt").read_bytes()
    for step in (65536, 50000, 1000):
        want = frame_of_chunks([data[o:o + step]
                                for o in range(0, len(data), step)])
        for batch in (1 << 16, frame.BATCH_BYTES):
            enc = frame.ReadFrameEncoder(Dribble(data, step), ctx,
                                         batch_bytes=batch)
            out = bytearray()
            while True:
                b = enc.read(4097)
                if not b:
                    break
                out += b
            assert bytes(out) == want, (step, batch)


class Flaky(io.RawIOBase):
    """A reader whose k-th read call fails (once) with `exc`; short reads of
    `step` bytes otherwise."""

    def __init__(self, data, step, fail_at, exc):
        self.b, self.step, self.calls = io.BytesIO(data), step, 0
        self.fail_at, self.exc = fail_at, exc

    def read(self, n=-1):
        self.calls += 1
        if self.calls == self.fail_at:
            raise self.exc
        n = self.step if n is None or n < 0 else min(n, self.step)
        return self.b.read(n)


def test_adapters_lose_nothing_when_the_reader_fails(ctx):
    """A read error in the middle of a batch (the reference makes one inner
    read per outer read, src/read.rs:378, so an error there costs nothing):
    what was read before it is compressed / decoded and handed out, the error
    comes behind it, and a caller that retries gets the rest - no gap."""
    from rust_snappy_amd import frame
    data = (O.CORPUS / "lcet10.txt").read_bytes()
    step = 65536
    want = frame_of_chunks([data[o:o + step]
                            for o in range(0, len(data), step)])
    for fail_at, exc in ((3, OSError("wire trouble")),
                         (2, InterruptedError())):
        enc = frame.ReadFrameEncoder(Flaky(data, step, fail_at, exc), ctx)
        out, errors = bytearray(), 0
        while True:
            try:
                b = enc.read(50000)
            except OSError:
                errors += 1
                continue                      # retry, like io::copy would
            if not b:
                break
            out += b
        assert bytes(out) == want
        assert errors == (0 if isinstance(exc, InterruptedError) else 1)
    # the decoder keeps what it has read when a later read fails
    f = O.frame_compress(data)
    dec = frame.FrameDecoder(Flaky(f, 30000, 4, OSError("again")), ctx)
    out, errors = bytearray(), 0
    while True:
        try:
            b = dec.read(100000)
        except OSError:
            errors += 1
            continue
        if not b:
            break
        out += b
    assert bytes(out) == data and errors == 1


class Turnstile(io.RawIOBase):
    """A request/response peer: it hands out one piece per read and refuses
    (the test fails) to be asked for the next piece before the consumer has
    acknowledged the bytes of the previous one."""

    def __init__(self, pieces):
        self.pieces, self.i, self.allowed = pieces, 0, 1

    def read(self, n=-1):
        if self.i >= len(self.pieces):
            return b""
        assert self.i < self.allowed, "decoder waits for input it has no use for"
        p = self.pieces[self.i]
        assert len(p) <= n
        self.i += 1
        return p


def test_frame_decoder_does_not_wait_for_a_full_batch(ctx):
    """The reference returns after each chunk (src/read.rs:105-172).  A
    batching decoder must decode as soon as a read came back short and the
    buffer holds a whole chunk - on a pipe or a socket the peer may be waiting
    for our answer before it sends the next chunk."""
    from rust_snappy_amd import frame
    msgs = [bytes([65 + i]) * (1000 + 37 * i) for i in range(5)]
    pieces = []
    for i, m in enumerate(msgs):
        f = O.frame_compress(m)
        pieces.append(f if i == 0 else f[10:])   # one stream, 5 chunks
    peer = Turnstile(pieces)
    dec = frame.FrameDecoder(peer, ctx)           # default 64 MiB batches
    for i, m in enumerate(msgs):
        got = dec.read(len(m))
        assert got == m, i
        peer.allowed = i + 2                      # "answer" received: send on
    assert dec.read(10) == b""


# ---------------------------------------------------------------------
# src/read.rs:216: decompress_len over the reader's whole scratch buffer
# ---------------------------------------------------------------------
def chunk(ty, body, crc=b""):
    n = len(crc) + len(body)
    return bytes([ty, n & 255, (n >> 8) & 255, n >> 16]) + crc + body


def test_frame_short_varint_reads_the_stale_scratch_buffer(ctx):
    """A compressed chunk whose payload has no varint terminator (fewer than
    10 bytes, all >= 0x80, or empty): the reference parses the length from its
    76 490-byte scratch buffer, i.e. continues into this chunk's own header
    bytes and the bodies of earlier chunks.  The oracle models that buffer;
    the device reproduces it (frame_short_varint).  Every outcome: Header,
    Empty, TooBig, UnsupportedChunkLength."""
    import rust_snappy_amd as R
    from rust_snappy_amd import frame
    ident = b"\xff\x06\x00\x00sNaPpY"
    good = O.frame_compress((O.CORPUS / "html").read_bytes())[10:]
    crc = b"\x11\x22\x33\x44"
    skip = lambda body: chunk(0x80, body)
    streams = {
        # payload ff ff, then header bytes 2..3 (00 00): varint ends -> 16383,
        # acceptable -> Decoder::decompress(2 bytes) -> Header
        "hdr": ident + chunk(0, b"\xff\xff", crc),
        # empty payload: phantom length from the header bytes -> Empty
        "empty": ident + chunk(0, b"", crc),
        # 5 continuation bytes, stale[5] = 0x7f from a skippable chunk -> TooBig
        "toobig": ident + skip(b"\xaa" * 5 + b"\x7f" + b"\xaa" * 4)
                  + chunk(0, b"\xff" * 5, crc),
        # 80 80 80 80 then stale[4] = 01 -> 1 << 28 -> UnsupportedChunkLength
        "len": ident + skip(b"\xbb" * 4 + b"\x01" + b"\xbb" * 5)
               + chunk(0, b"\x80" * 4, crc),
        # stale bytes from an earlier COMPRESSED chunk's payload (good data)
        "after_data": ident + good + chunk(0, b"\x80" * 6, crc),
        # stale continuation never ends -> Header
        "never": ident + skip(b"\xcc" * 10) + chunk(0, b"\x80" * 4, crc),
        # a stored chunk in between does not touch the scratch buffer
        "stored_between": ident + skip(b"\xdd" * 5 + b"\x03" + b"\xdd" * 4)
                          + chunk(1, b"x" * 20, struct.pack(
                              "<I", O.crc32c_masked(b"x" * 20)))
                          + chunk(0, b"\x80" * 5, crc),
    }
    seen = set()
    for name, s in streams.items():
        with pytest.raises(O.SnapError) as oe:
            O.frame_decompress(s)
        seen.add(oe.value.name)
        got = expect_error(ctx, s, oe.value.name)
        assert got.abc == (oe.value.a, oe.value.b, oe.value.c), name
        # small batches: the stale bytes travel from batch to batch
        dec = frame.FrameDecoder(Dribble(s, 7), ctx, batch_bytes=1 << 17)
        with pytest.raises(R.Error) as ei:
            dec.read_to_end()
        assert ei.value == got, name
        # a side index does not change the verdict (the walk takes over)
        st, used, offs = frame.scan_host(s)
        d_in = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
        _, err = frame.decompress_batch_device(
            ctx, d_in, len(s), len(offs) - 1, torch.from_numpy(offs).cuda())
        assert err == got, name
    assert seen == {"Header", "Empty", "TooBig", "UnsupportedChunkLength"}
    # the good chunks in front of such a chunk are still delivered
    dec = frame.FrameDecoder(io.BytesIO(streams["after_data"]), ctx)
    html = (O.CORPUS / "html").read_bytes()
    assert dec.read(1 << 20) == html


def test_frame_side_index_is_only_a_hint(ctx):
    """A wrong side index (out of range, not increasing, skipping chunks,
    pointing into a payload) must give the result of decoding without one."""
    from rust_snappy_amd import frame
    data = b"".join(d for _, d in O.corpus_round()[:3])
    f = O.frame_compress(data)
    offs = frame.index_host(f)
    d_in = torch.frombuffer(bytearray(f), dtype=torch.uint8).cuda()
    n = len(offs) - 1
    bad = []
    for mut in range(6):
        o = offs.copy()
        if mut == 0:
            o[1] = np.int64(-8)                 # 2^64 - 8: r + 8 wraps
        elif mut == 1:
            o[2], o[3] = o[3], o[2]
        elif mut == 2:
            o = np.delete(o, 1)                 # skips a chunk
        elif mut == 3:
            o[1] += 3                           # into a payload
        elif mut == 4:
            o[-1] -= 1
        else:
            o[0] = 0
        bad.append(o)
    bad.append(offs[-1:].copy())                # an index with no chunks
    for o in bad:
        good, err = frame.decompress_batch_device(
            ctx, d_in, len(f), n, torch.from_numpy(o.astype(np.int64)).cuda())
        assert err is None and good == data


# ---------------------------------------------------------------------
# BASELINE configs 3 and 5 at reduced size, whole output against the oracle
# ---------------------------------------------------------------------
def test_cfg3_shape_framed_text_1024_chunks(ctx):
    """64 MiB of the cfg3 synthetic text (bench_configs.synth_text, seeded) =
    1024 chunks through snapmi_frame_compress: the WHOLE framed stream
    against the oracle's restatement of write::FrameEncoder, then decoded with
    the side index, without it, and through the streaming FrameDecoder."""
    import bench_configs as BC
    from rust_snappy_amd import frame
    dev = torch.device("cuda", 0)
    text = BC.synth_text(dev, 64 << 20)
    out, flen, index = frame.compress_device(ctx, text, want_index=True)
    got = out[:flen].cpu().numpy().tobytes()
    host = text.cpu().numpy().tobytes()
    want = O.frame_compress(host)
    assert got == want
    assert index.numel() - 1 == 1024
    for idx in (index, None):
        back, m = frame.decompress_device(ctx, out, flen, index=idx)
        assert m == len(host) and torch.equal(back[:m], text)
    dec = frame.FrameDecoder(io.BytesIO(got), ctx, batch_bytes=8 << 20)
    assert dec.read_to_end() == host


def test_cfg5_shape_jpeg_through_the_frame_layer(ctx):
    """fireworks.jpeg tiled to ~31 MiB through the frame layer: every chunk
    takes the Uncompressed branch of compress_frame (src/frame.rs:85) at
    scale - 472 stored chunks, byte-identical to the oracle - plus a stream
    that alternates stored and compressed chunks."""
    from rust_snappy_amd import frame
    jpg = (O.CORPUS / "fireworks.jpeg").read_bytes()
    txt = (O.CORPUS / "alice29.txt").read_bytes()
    data = jpg * 256
    d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out, flen, index = frame.compress_device(ctx, d, want_index=True)
    got = out[:flen].cpu().numpy().tobytes()
    assert got == O.frame_compress(data)
    idx = index.cpu().numpy()
    types = [got[int(o)] for o in idx[:-1]]
    assert len(types) == (len(data) + 65535) // 65536 and set(types) == {1}
    back, m = frame.decompress_device(ctx, out, flen, index=index)
    assert m == len(data) and torch.equal(back[:m], d)
    mixed = (jpg[:65536] + txt[:65536]) * 40 + jpg[:1000]
    f = framed(ctx, mixed)
    assert f == O.frame_compress(mixed)
    offs = frame.index_host(f)
    assert {f[int(o)] for o in offs[:-1]} == {0, 1}
    assert frame.FrameDecoder(io.BytesIO(f), ctx).read_to_end() == mixed
    # the raw path of cfg5: the jpeg as independent raw streams
    from test_gpu_parity import gpu_compress, gpu_decompress
    comp = gpu_compress(ctx, [jpg] * 64)
    assert all(c == O.compress(jpg) for c in comp)
    outs, errs = gpu_decompress(ctx, comp)
    assert all(o == jpg for o in outs) and all(e[0] == 0 for e in errs)


def test_frame_parallel_header_walk(built):
    """Without a side index the chunk headers are found in parallel (k_fw_*:
    plausible headers per segment, chains followed to the segment's end, the
    segments strung together); anything the reference's reader would reject
    sends the stream to the sequential walk.  Forced here for small streams
    with 128 KiB segments, against the sequential walk and the oracle."""
    import rust_snappy_amd as R
    from rust_snappy_amd import frame
    par = R.raw.Context(0)
    par.set_option("frame_parallel_walk_min", 0)
    par.set_test_option("frame_walk_segment", 128 << 10)
    seq = R.raw.Context(0)
    seq.set_option("frame_parallel_walk_min", 1 << 60)
    rnd = O.corpus_round()
    data = b"".join(d for _, d in rnd)                    # 45 chunks
    f = O.frame_compress(data)
    ident = f[:10]
    offs = frame.index_host(f)
    # other chunk types between data chunks, at several places
    g = bytearray(ident)
    for i in range(len(offs) - 1):
        g += f[int(offs[i]):int(offs[i + 1])]
        if i % 7 == 3:
            g += bytes([0x80, 5, 0, 0]) + b"skip!"
        if i % 11 == 5:
            g += bytes([0xFE, 3, 0, 0]) + b"pad" + ident
    jpg = rnd[2][1]
    streams = [f, bytes(g), O.frame_compress(jpg * 3),
               O.frame_compress(b"x" * 70000), ident, b"",
               O.frame_compress(data[:65536])]
    bad = []
    for cut in (1, 5, 9, 300, 70000):
        bad.append(f[:-cut])                               # UnexpectedEof
    b2 = bytearray(f); b2[int(offs[20])] = 0x05; bad.append(bytes(b2))
    b3 = bytearray(f); b3[int(offs[30]) + 3] = 0xFF; bad.append(bytes(b3))
    b4 = bytearray(f); b4[int(offs[12]) + 4] ^= 1; bad.append(bytes(b4))   # crc
    b5 = bytearray(f); b5[int(offs[40]) + 100] ^= 0x40; bad.append(bytes(b5))
    bad.append(b"\x00" + f[1:])                            # StreamHeader
    bad.append(f[:int(offs[9])] + chunk(0, b"\x80" * 5, b"\x01\x02\x03\x04")
               + f[int(offs[9]):])                         # stale-buffer rule
    for s in streams + bad:
        n_chunks = len(s) // 8 + 1
        d_in = torch.frombuffer(bytearray(s or b"\0"),
                                dtype=torch.uint8).cuda()
        got = frame.decompress_batch_device(par, d_in, len(s), n_chunks)
        want = frame.decompress_batch_device(seq, d_in, len(s), n_chunks)
        assert got[0] == want[0] and got[1] == want[1], len(s)
        try:
            truth = O.frame_decompress(s)
            assert got == (truth, None)
        except O.SnapError as oe:
            assert got[1] is not None
            if oe.kind >= 0 and got[1].variant != "StreamHeaderMismatch":
                assert (got[1].kind,) + got[1].abc == \
                    (oe.kind, oe.a, oe.b, oe.c), (got[1], oe)
    # default segment size, a stream of several segments (cfg3 shape)
    import bench_configs as BC
    dev = torch.device("cuda", 0)
    text = BC.synth_text(dev, 96 << 20)
    c = R.raw.Context(0)
    out, flen, index = frame.compress_device(c, text, want_index=True)
    back, m = frame.decompress_device(c, out, flen)        # no index
    assert m == text.numel() and torch.equal(back[:m], text)
    for x in (par, seq, c):
        x.close()


def test_frame_decoder_output_larger_than_its_buffer(ctx):
    """Highly compressible data: a batch of compressed chunks decodes to far
    more than the decoder's output buffer (snapmi_frame_decode_host then
    takes only as many chunks as fit and reports how far it got); also the
    C call directly with a buffer of one and two chunks."""
    import ctypes as C
    from rust_snappy_amd import _lib, frame
    data = bytes(40 << 20) + b"".join(d for _, d in O.corpus_round()) + \
        b"ab" * (3 << 20)
    f = O.frame_compress(data)
    assert len(f) < len(data) // 8
    for batch in (1 << 17, 1 << 20, frame.BATCH_BYTES):
        dec = frame.FrameDecoder(io.BytesIO(f), ctx, batch_bytes=batch)
        out = bytearray()
        while True:
            b = dec.read(1 << 20)
            if not b:
                break
            out += b
        assert bytes(out) == data, batch
    # the C call with room for exactly one / two chunks at a time
    for room in (65536, 131072 + 5):
        pos, out, stale, cont = 0, bytearray(), bytearray(10), False
        buf = bytearray(room)
        while pos < len(f):
            piece = f[pos:pos + (1 << 20)]
            n, used, err = frame.decode_host(ctx, piece, buf, cont,
                                             pos + len(piece) >= len(f), stale)
            assert err is None and used > 0
            assert n <= room // 65536 * 65536
            out += buf[:n]
            pos += used
            cont = True
        assert bytes(out) == data
    # a buffer below one chunk: the identifier is consumed, then the call is
    # refused (never an overrun)
    import rust_snappy_amd as R
    small = bytearray(1000)
    n, used, err = frame.decode_host(ctx, f[:1 << 20], small, False, False,
                                     bytearray(10))
    assert (n, used, err) == (0, 10, None)
    with pytest.raises(R.Error):
        frame.decode_host(ctx, f[10:1 << 20], small, True, False,
                          bytearray(10))


def test_frame_compress_chunks_on_device_buffers(ctx):
    """snapmi_frame_compress_chunks with device-resident input: chunk
    boundaries chosen by the caller (here: what a reader with short reads
    gives read::FrameEncoder), with and without the stream identifier."""
    from rust_snappy_amd import frame
    data = (O.CORPUS / "lcet10.txt").read_bytes()[:300000]
    rng = random.Random(2)
    lens, left = [], len(data)
    while left:
        n = min(left, rng.choice([1, 17, 4096, 65535, 65536]))
        lens.append(n)
        left -= n
    chunks, pos = [], 0
    for n in lens:
        chunks.append(data[pos:pos + n])
        pos += n
    want = frame_of_chunks(chunks)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out, n = frame.compress_chunks_device(ctx, d_in, lens, ident=True)
    assert out[:n].cpu().numpy().tobytes() == want
    out, n = frame.compress_chunks_device(ctx, d_in, lens, ident=False)
    assert out[:n].cpu().numpy().tobytes() == want[10:]


def test_host_calls_on_pinned_buffers(ctx):
    """snapmi_frame_encode_host / _decode_host on pinned memory
    (snapmi_host_alloc): the sliced pipeline (small slices forced), the copy
    kernel that takes decoded bytes home - to an UNALIGNED destination - and
    the bytes against the oracle."""
    import ctypes as C
    from rust_snappy_amd import _lib, frame
    L = _lib.load()
    data = b"".join(d for _, d in O.corpus_round())[:3_000_001]
    n = len(data)
    nch = (n + 65535) // 65536
    lens = np.full(nch, 65536, dtype=np.uint32)
    lens[-1] = n - (nch - 1) * 65536
    h_in = frame.HostBuffer(n)
    h_in.array[:] = np.frombuffer(data, dtype=np.uint8)
    h_out = frame.HostBuffer(10 + n + 8 * nch)
    ctx.set_option("host_encode_slice", 1 << 20)       # three slices
    ctx.set_option("host_decode_slice_chunks", 7)      # seven slices
    try:
        k = frame.encode_host_into(ctx, h_in.view, lens, h_out)
        want = O.frame_compress(data)
        assert bytes(h_out.view[:k]) == want
        for by_kernel in (1, 0):
            ctx.set_option("host_copy_kernel", by_kernel)
            h_back = frame.HostBuffer(nch * 65536 + 64)
            stale = (C.c_uint8 * 10)()
            written, consumed = C.c_size_t(0), C.c_size_t(0)
            err = _lib.SnapmiError()
            rc = L.snapmi_frame_decode_host(
                ctx._h, C.c_void_p(h_out.ptr), k, 2, stale,
                C.c_void_p(h_back.ptr + 3), nch * 65536, C.byref(written),
                C.byref(consumed), C.byref(err))
            assert rc == 0 and written.value == n and consumed.value == k
            assert bytes(h_back.view[3:3 + n]) == data
            h_back.close()
    finally:
        ctx.set_option("host_encode_slice", 2048 << 20)
        ctx.set_option("host_decode_slice_chunks", 8192)
        ctx.set_option("host_copy_kernel", 1)
        h_in.close()
        h_out.close()


def test_frame_decoder_readinto(ctx):
    """io::Read::read as the reference has it (src/read.rs:104): into the
    caller's buffer.  A buffer with room for a batch takes the bytes straight
    from the device call, a small one gets them from the decoder's own room;
    an error comes behind the bytes in front of it."""
    from rust_snappy_amd import frame
    data = (O.CORPUS / "lcet10.txt").read_bytes() * 3
    framed = O.frame_compress(data)
    for size in (1 << 22, 65536, 1000, 7):
        dec = frame.FrameDecoder(io.BytesIO(framed), ctx, batch_bytes=1 << 18)
        buf, got = bytearray(size), bytearray()
        while True:
            k = dec.readinto(buf)
            if k == 0:
                break
            assert 0 < k <= size
            got += buf[:k]
            if size == 7 and len(got) > 5000:
                got += dec.read(-1)          # (the rest in one go)
                break
        assert bytes(got) == data, size
        assert dec.readinto(buf) == 0
    hb = frame.HostBuffer(len(data) + (1 << 21))
    dec = frame.FrameDecoder(io.BytesIO(framed), ctx, batch_bytes=1 << 18)
    pos = 0
    while True:
        k = dec.readinto(hb.view[pos:])
        if k == 0:
            break
        pos += k
    assert bytes(hb.view[:pos]) == data
    hb.close()
    # a damaged chunk behind good ones
    bad = bytearray(framed)
    bad[len(framed) // 2] ^= 0x55
    dec = frame.FrameDecoder(io.BytesIO(bytes(bad)), ctx, batch_bytes=1 << 18)
    buf, got, err = bytearray(1 << 22), bytearray(), None
    while True:
        try:
            k = dec.readinto(buf)
        except frame.Error as e:
            err = e
            break
        if k == 0:
            break
        got += buf[:k]
    assert err is not None
    assert len(got) > 0 and bytes(got) == data[:len(got)]
    try:
        O.frame_decompress(bytes(bad))
        raise AssertionError("the oracle accepts the damaged stream")
    except O.SnapError as e:
        assert err.kind == e.kind and err.abc == (e.a, e.b, e.c), (err, e)
