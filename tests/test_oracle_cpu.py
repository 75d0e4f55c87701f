"""CPU suite, part 1: pin the oracle (oracle/snappy_oracle.c) against every
golden vector / KAT the reference's tests hold for the hot path, and against
libsnappy 1.1.8 when it can be loaded."""
import hashlib
import random

import pytest

import kats
import oracle_lib as O


def test_golden_rawsnappy_both_directions():
    # reference test/tests.rs:200-205 (data_golden_rev)
    txt = (O.CORPUS / "Mark.Twain-Tom.Sawyer.txt").read_bytes()
    snp = (O.CORPUS / "Mark.Twain-Tom.Sawyer.txt.rawsnappy").read_bytes()
    assert len(txt) == 14168 and len(snp) == 9871
    assert O.decompress(snp) == txt
    assert O.compress(txt) == snp
    assert O.compress(O.decompress(snp)) == snp


def test_corpus_known_answers():
    for bench_id, data in O.corpus_round():
        n_in, n_out, sha = kats.CORPUS_SHA256[bench_id]
        c = O.compress(data)
        assert (len(data), len(c)) == (n_in, n_out), bench_id
        assert hashlib.sha256(c).hexdigest() == sha, bench_id
        assert O.decompress(c) == data


def test_tiny_kat_and_empty():
    # exercises the >=68 split of src/compress.rs:339-342
    assert O.compress(b"a" * 120).hex() == "780061fe0100da0100"
    assert O.compress(b"") == b"\x00"          # src/compress.rs:120-125
    assert O.decompress(b"\x00") == b""
    assert O.decompress_len(b"") == 0          # src/decompress.rs:31-33
    assert O.max_compress_len(65536) == 76490  # src/frame.rs:12
    assert O.max_compress_len(2**32) == 0
    assert O.max_compress_len(0xFFFFFFFF) == 0


@pytest.mark.parametrize("name,comp,want", kats.DECODE_KATS,
                         ids=[k[0] for k in kats.DECODE_KATS])
def test_decode_kats(name, comp, want):
    assert O.decompress(comp) == want


@pytest.mark.parametrize("name,data,want,bad_header", kats.ERROR_KATS,
                         ids=[k[0] for k in kats.ERROR_KATS])
def test_error_kats(name, data, want, bad_header):
    # reference errored! macro, test/tests.rs:19-58
    if bad_header:
        with pytest.raises(O.SnapError) as ei:
            O.decompress_len(data)
        got = ei.value.key()
        assert got[:len(want)] == want
        cap = 1024
    else:
        cap = O.decompress_len(data)
    with pytest.raises(O.SnapError) as ei:
        O.decompress(data, cap)
    got = ei.value.key()
    assert got[0] == want[0] and got[1:len(want)] == want[1:], (got, want)


def test_roundtrip_structured():
    for d in ([b"", b"\x00", kats.RANDOM1, kats.RANDOM2, kats.RANDOM3,
               kats.RANDOM4] + kats.small_copy_inputs()
              + kats.small_regular_inputs()[::7]):
        assert O.decompress(O.compress(d)) == d


def test_buffer_too_small_and_too_big():
    with pytest.raises(O.SnapError) as ei:
        O.compress(b"abc", cap=10)
    assert ei.value.key() == ("BufferTooSmall", 10, 35, 0)
    with pytest.raises(O.SnapError) as ei:
        O.decompress(O.compress(b"hello world"), cap=3)
    assert ei.value.key() == ("BufferTooSmall", 3, 11, 0)


@pytest.mark.skipif(O.libsnappy() is None, reason="libsnappy 1.1.8 not here")
def test_oracle_equals_libsnappy():
    rng = random.Random(20260925)
    for bench_id, data in O.corpus_round():
        assert O.compress(data) == O.libsnappy_compress(data), bench_id
    for it in range(400):
        alpha = rng.choice([1, 2, 3, 4, 16, 256])
        n = rng.choice([0, 1, 15, 16, 17, 18, 300, 5000, 65535, 65536, 65537,
                        rng.randrange(0, 150000)])
        data = bytes(rng.choices(range(alpha), k=n))
        c = O.compress(data)
        assert c == O.libsnappy_compress(data), (alpha, n)
        assert O.libsnappy_uncompress(c) == data
    for name in ("baddata1.snappy", "baddata2.snappy", "baddata3.snappy"):
        bad = (O.CORPUS / name).read_bytes()
        assert O.libsnappy_uncompress(bad) is None
        with pytest.raises(O.SnapError):
            O.decompress(bad)


def test_crc32c_known_answer():
    # external pin (the reference's tests do not pin the CRC)
    assert O.crc32c(b"123456789") == 0xE3069283
    assert O.crc32c_masked(b"123456789") == 0xC78AB0E5
    assert O.crc32c(b"") == 0


def test_frame_structure_and_roundtrip():
    # expected framed sizes: SURVEY App-B, rule src/frame.rs:85
    want = {"html": 22872, "urls.10K": 335620, "fireworks.jpeg": 123119,
            "paper-100k.pdf": 85327, "html_x_4": 92318, "alice29.txt": 88074,
            "asyoulik.txt": 77532, "lcet10.txt": 234745,
            "plrabn12.txt": 319362, "geo.protodata": 23364,
            "kppkn.gtb": 69566}
    for name, size in want.items():
        data = (O.CORPUS / name).read_bytes()
        f = O.frame_compress(data)
        assert len(f) == size, name
        assert f[:10] == b"\xff\x06\x00\x00sNaPpY"
        assert O.frame_decompress(f) == data
    assert O.frame_compress(b"") == b""  # src/write.rs:154-170
    with pytest.raises(O.SnapError) as ei:   # test/tests.rs:536-545
        O.frame_decompress(b"123")
    assert ei.value.kind == -1


def test_foreign_encoder_streams():
    """copy-4, offsets beyond 64 KiB, overlapping copies, non-minimal literal
    length forms: oracle == element-level model == libsnappy."""
    import foreign
    for stream, want in foreign.cases():
        assert O.decompress_len(stream) == len(want)
        assert O.decompress(stream, len(want)) == want
        if O.libsnappy() is not None:
            assert O.libsnappy_uncompress(stream) == want


def _port_fast():
    import ctypes as C
    L = O.lib()
    L.snapf_max_compressed_length.restype = C.c_size_t
    L.snapf_max_compressed_length.argtypes = [C.c_size_t]

    def compress(data):
        cap = L.snapf_max_compressed_length(len(data))
        out = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(cap)
        rc = L.snapf_compress(bytes(data), C.c_size_t(len(data)), out,
                              C.byref(n))
        return rc, out.raw[:n.value]

    def uncompress(data, cap):
        out = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(cap)
        rc = L.snapf_uncompress(bytes(data), C.c_size_t(len(data)), out,
                                C.byref(n))
        return rc, out.raw[:n.value] if rc == 0 else b""
    return compress, uncompress


def test_port_fast_computes_the_oracles_results():
    """oracle/snappy_port_fast.c is the restatement WITH the reference's fast
    paths (16-byte blind copies src/compress.rs:440-453, src/decompress.rs:
    170-183; tag table and the three copy strategies :233-343) that bench.py's
    cpu_baseline times; a timing of wrong results would be worthless, so: its
    bytes are the oracle's on the corpus, the golden vector, structured and
    random inputs and the foreign-encoder streams, and it refuses exactly
    what the oracle refuses (every error KAT, the baddata files, truncations
    of a good stream, a short output buffer)."""
    import foreign
    compress, uncompress = _port_fast()
    rng = random.Random(606)
    inputs = [d for _, d in O.corpus_round()] + [
        (O.CORPUS / "Mark.Twain-Tom.Sawyer.txt").read_bytes(), b"", b"a",
        b"a" * 120, kats.RANDOM1, kats.RANDOM2, kats.RANDOM3, kats.RANDOM4]
    inputs += kats.small_copy_inputs() + kats.small_regular_inputs()[::5]
    for _ in range(300):
        alpha = rng.choice([1, 2, 3, 4, 16, 256])
        n = rng.choice([0, 1, 15, 16, 17, 18, 31, 32, 33, 300, 5000, 65535,
                        65536, 65537, rng.randrange(0, 150000)])
        inputs.append(bytes(rng.choices(range(alpha), k=n)))
    for d in inputs:
        rc, c = compress(d)
        assert rc == 0 and c == O.compress(d), len(d)
        rc, back = uncompress(c, len(d))
        assert rc == 0 and back == d, len(d)
        if len(d) > 1:   # a short output buffer: BufferTooSmall, like :84-90
            assert uncompress(c, len(d) - 1)[0] == 2
    for stream, want in foreign.cases():
        rc, back = uncompress(stream, len(want))
        assert rc == 0 and back == want
    for name, data, want, bad_header in kats.ERROR_KATS:
        rc, _ = uncompress(data, 1 << 16)
        assert rc != 0, name
    for name in ("baddata1.snappy", "baddata2.snappy", "baddata3.snappy"):
        assert uncompress((O.CORPUS / name).read_bytes(), 1 << 20)[0] != 0
    good = O.compress((O.CORPUS / "html").read_bytes())
    for cut in list(range(0, 64)) + [len(good) // 2, len(good) - 1]:
        piece = good[:cut]
        try:
            O.decompress(piece, 102400)
            ok = True
        except O.SnapError:
            ok = False
        assert (uncompress(piece, 102400)[0] == 0) == ok, cut
