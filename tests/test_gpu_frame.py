"""GPU suite for the frame layer: CRC32C kernel, framed
bytes equal to the oracle's restatement of write::FrameEncoder, round trips,
and FrameDecoder errors with the reference's variants and fields."""
import io
import random
import struct

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
        assert got.kind == want_oracle.kind
        nfields = len(got.fields)
        assert tuple(got.fields.values()) == \
            (want_oracle.a, want_oracle.b, want_oracle.c)[:nfields]
    assert got.variant == key, (got, key)


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
    try:
        O.frame_decompress(bytes(bad))
        corrupt_ok = True
    except O.SnapError as oe:
        corrupt_ok = False
        expect_error(ctx, bytes(bad), O.KIND_NAMES[oe.kind])
    assert not corrupt_ok or True
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
