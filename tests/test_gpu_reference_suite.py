"""The reference's own test suite (test/tests.rs), test for test and under the
same names, run through the host-side mirror of the reference interface
(rust_snappy_amd.raw.{Encoder, Decoder}, rust_snappy_amd.frame.{FrameEncoder,
FrameDecoder, ReadFrameEncoder}) and therefore through the C ABI and the HIP
kernels.  Where the reference's tests only assert a round trip, these also
assert the bytes against the oracle (the reference's encoder is
deterministic, README.md:87-90).

The `cpp` tests (test/tests.rs:90-160, 547-573, feature "cpp") use Google's
libsnappy 1.1.8 when the image has it (snappy-cpp/src/lib.rs:66-88 binds the
same four functions)."""
import io
import random

import pytest

import kats
import oracle_lib as O

pytestmark = pytest.mark.gpu


# ---- helper functions of test/tests.rs:593-644 ---------------------------
def press(ctx, data):            # :593-595
    from rust_snappy_amd import raw
    return raw.Encoder(ctx).compress_vec(data)


def depress(ctx, data):          # :597-599
    from rust_snappy_amd import raw
    return raw.Decoder(ctx).decompress_vec(data)


def write_frame_press(ctx, data):    # :601-608
    from rust_snappy_amd import frame
    wtr = frame.FrameEncoder(io.BytesIO(), ctx)
    wtr.write_all(data)
    return wtr.into_inner().getvalue()


def read_frame_depress(ctx, data):   # :610-617
    from rust_snappy_amd import frame
    return frame.FrameDecoder(io.BytesIO(data), ctx).read_to_end()


def read_frame_press(ctx, data):     # :619-626
    from rust_snappy_amd import frame
    return frame.ReadFrameEncoder(io.BytesIO(data), ctx).read()


def press_cpp(data):             # :628-636
    return O.libsnappy_compress(data)


def depress_cpp(data):           # :638-644
    return O.libsnappy_uncompress(data)


def corpus(name):
    return (O.CORPUS / name).read_bytes()


# ---- testtrip! instances, test/tests.rs:180-195 and :469-504 ---------------
TESTTRIP = {
    "empty": lambda: b"",
    "one_zero": lambda: b"\x00",
    "data_html": lambda: corpus("html"),
    "data_urls": lambda: corpus("urls.10K"),
    "data_jpg": lambda: corpus("fireworks.jpeg"),
    "data_pdf": lambda: corpus("paper-100k.pdf"),
    "data_html4": lambda: corpus("html_x_4"),
    "data_txt1": lambda: corpus("alice29.txt"),
    "data_txt2": lambda: corpus("asyoulik.txt"),
    "data_txt3": lambda: corpus("lcet10.txt"),
    "data_txt4": lambda: corpus("plrabn12.txt"),
    "data_pb": lambda: corpus("geo.protodata"),
    "data_gaviota": lambda: corpus("kppkn.gtb"),
    "data_golden": lambda: corpus("Mark.Twain-Tom.Sawyer.txt"),
    "random1": lambda: kats.RANDOM1,
    "random2": lambda: kats.RANDOM2,
    "random3": lambda: kats.RANDOM3,
    "random4": lambda: kats.RANDOM4,
}


@pytest.fixture(params=sorted(TESTTRIP))
def data(request):
    return TESTTRIP[request.param]()


def test_roundtrip_raw(ctx, data):                        # :70-74
    comp = press(ctx, data)
    assert comp == O.compress(data)
    assert depress(ctx, comp) == data


def test_roundtrip_frame(ctx, data):                      # :76-81
    assert read_frame_depress(ctx, write_frame_press(ctx, data)) == data


def test_read_and_write_frame_encoder_match(ctx, data):   # :83-88
    w = write_frame_press(ctx, data)
    assert read_frame_press(ctx, data) == w
    assert w == O.frame_compress(data)


def test_cpp_decompresses_rust(ctx, data):                # :92-126
    if O.libsnappy() is None:
        pytest.skip("no libsnappy in this image")
    assert depress_cpp(press(ctx, data)) == data


def test_rust_decompresses_cpp(ctx, data):                # :128-162
    if O.libsnappy() is None:
        pytest.skip("no libsnappy in this image")
    assert depress(ctx, press_cpp(data)) == data


# ---- single tests ---------------------------------------------------------
def test_data_golden_rev(ctx):                            # :199-205
    data = corpus("Mark.Twain-Tom.Sawyer.txt.rawsnappy")
    want = corpus("Mark.Twain-Tom.Sawyer.txt")
    assert depress(ctx, data) == want
    assert press(ctx, want) == data


def test_small_copy(ctx):                                 # :208-216
    for d in kats.small_copy_inputs():
        assert depress(ctx, press(ctx, d)) == d


def test_small_regular(ctx):                              # :218-229
    for d in kats.small_regular_inputs():
        assert depress(ctx, press(ctx, d)) == d


@pytest.mark.parametrize("name,comp,want", kats.DECODE_KATS,
                         ids=[k[0] for k in kats.DECODE_KATS])
def test_decompress_copy_close_to_end(ctx, name, comp, want):   # :232-317
    assert depress(ctx, comp) == want


@pytest.mark.parametrize("name,comp,want,bad_header", kats.ERROR_KATS,
                         ids=[k[0] for k in kats.ERROR_KATS])
def test_errored(ctx, name, comp, want, bad_header):      # :19-58, :345-466
    """errored!: with a bad header decompress_len fails with the same error
    and the buffer is 1024 bytes; otherwise the buffer has exactly
    decompress_len bytes.  Decoder::decompress returns the named variant
    with the named field values."""
    from rust_snappy_amd import raw
    from rust_snappy_amd.error import Error
    if bad_header:
        with pytest.raises(Error) as ei:
            raw.decompress_len(comp)
        assert ei.value.key() == tuple(want), name
        buf = bytearray(1024)
    else:
        buf = bytearray(raw.decompress_len(comp))
    with pytest.raises(Error) as ei:
        raw.Decoder(ctx).decompress(comp, buf)
    assert ei.value.key() == tuple(want), name


def test_qc_roundtrip(ctx):                               # :509-518
    """quickcheck, 1 000 vectors of up to 10 000 bytes; here seeded, a
    hundred through the scalar calls (each one is a device round trip)."""
    rng = random.Random(0x5A4D5350)
    for _ in range(100):
        n = rng.randrange(0, 10_000)
        # quickcheck's Vec<u8> is uniform bytes; mix in a small alphabet so
        # the match finder has something to find
        alpha = rng.choice([256, 256, 4, 2])
        d = bytes(rng.randrange(alpha) for _ in range(n))
        assert depress(ctx, press(ctx, d)) == d


def test_qc_roundtrip_stream(ctx):                        # :520-533
    rng = random.Random(0x5A4D5351)
    for _ in range(50):
        n = rng.randrange(1, 10_000)
        alpha = rng.choice([256, 4, 2])
        d = bytes(rng.randrange(alpha) for _ in range(n))
        assert read_frame_depress(ctx, write_frame_press(ctx, d)) == d


def test_short_input(ctx):                                # :536-545
    from rust_snappy_amd import frame
    import rust_snappy_amd as R
    with pytest.raises(R.Error) as ei:
        frame.FrameDecoder(io.BytesIO(b"123"), ctx).read_to_end()
    assert ei.value.variant == "UnexpectedEof"   # io::ErrorKind::UnexpectedEof


def test_qc_cpp_decompresses_rust(ctx):                   # :547-560
    if O.libsnappy() is None:
        pytest.skip("no libsnappy in this image")
    rng = random.Random(0x5A4D5352)
    for _ in range(50):
        d = bytes(rng.randrange(rng.choice([256, 3]))
                  for _ in range(rng.randrange(0, 10_000)))
        assert depress_cpp(press(ctx, d)) == d


def test_qc_rust_decompresses_cpp(ctx):                   # :562-575
    if O.libsnappy() is None:
        pytest.skip("no libsnappy in this image")
    rng = random.Random(0x5A4D5353)
    for _ in range(50):
        d = bytes(rng.randrange(rng.choice([256, 3]))
                  for _ in range(rng.randrange(0, 10_000)))
        assert depress(ctx, press_cpp(d)) == d
