"""k_compress_spans (a window of 63 consecutive positions per wavefront step)
on the CPU: tests/span_wave_host.cpp runs the walks of
rust-snappy_amd/csrc/snapmi_span.hpp - the very text the kernel compiles:
span_walk() as it is, span_par_walk() and span_fast_ok_w() over arrays of 64
lanes - with the 64 lanes around them emulated in the order gfx950 applies the
lanes of one DS instruction in, and the stream must be the oracle's: every length up to 300
over alphabets that make consecutive positions share table slots (runs, tiny
alphabets: every C-bit case, cuts, deferred inserts), periodic data, blocks of
every corpus file (long miss runs -> schedule steps, long matches), block-size
edges."""
import ctypes as C
import random
import subprocess

import pytest

import oracle_lib as O
from conftest import ROOT


@pytest.fixture(scope="module")
def span(tmp_path_factory):
    so = tmp_path_factory.mktemp("span") / "span_wave_host.so"
    subprocess.check_call(
        ["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
         "-I", str(ROOT / "rust-snappy_amd" / "csrc"),
         str(ROOT / "tests" / "span_wave_host.cpp"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.span_wave_compress.restype = C.c_uint32
    L.span_wave_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p,
                                     C.c_uint32, C.POINTER(C.c_uint64)]

    def run(data):
        cap = len(data) + len(data) // 6 + 64
        out = C.create_string_buffer(cap)
        st = (C.c_uint64 * 16)()
        r = L.span_wave_compress(bytes(data), len(data), out, cap, st)
        assert r < 0x80000000, hex(r)
        return out.raw[:r], list(st)
    return run


def _small_inputs(rng):
    for n in range(1, 301):
        yield bytes(n)
        for alpha in (2, 3, 4, 16, 256):
            yield bytes(rng.randrange(alpha) for _ in range(n))
    for n in (1000, 4095, 4096, 4097, 5000, 16384, 16385, 20000, 32768,
              65535, 65536):
        yield bytes(n)
        for alpha in (2, 3, 16, 256):
            yield bytes(rng.randrange(alpha) for _ in range(n))
        for period in (1, 2, 3, 5, 7, 13, 31, 62, 63, 64, 65, 100, 1000):
            unit = bytes(rng.randrange(256) for _ in range(period))
            yield (unit * (n // period + 1))[:n]


def test_span_steps_give_the_oracle_stream_small_and_synthetic(span):
    rng = random.Random(7)
    for data in _small_inputs(rng):
        got, _ = span(data)
        assert got == O.compress(data), (len(data), data[:32].hex())


def test_span_steps_give_the_oracle_stream_on_the_corpus(span):
    windows = tokens = 0
    for p in sorted(O.CORPUS.iterdir()):
        if p.suffix in (".rawsnappy", ".snappy") or p.name == "COPYING":
            continue
        data = p.read_bytes()
        for at in range(0, len(data), 65536):
            blk = data[at:at + 65536]
            got, st = span(blk)
            assert got == O.compress(blk), (p.name, at)
            windows += st[0] + st[1]
            tokens += st[4]
    # what the kernel is for: many copies per step (k_compress_blocks: one)
    assert tokens > 4 * windows, (tokens, windows)


def test_span_steps_phrases_with_noise(span):
    """literal + copy pairs of every length, offsets near and far, copies
    that end in every lane of a window (deferred inserts at lanes 62, 63)"""
    rng = random.Random(3)
    for _ in range(60):
        phrase = bytes(rng.randrange(256) for _ in range(rng.randrange(4, 90)))
        buf = bytearray()
        n = rng.choice((300, 2000, 9000, 40000))
        while len(buf) < n:
            buf += phrase[:rng.randrange(4, len(phrase) + 1)]
            buf += bytes(rng.randrange(256)
                         for _ in range(rng.randrange(0, 40)))
        data = bytes(buf[:n])
        got, _ = span(data)
        assert got == O.compress(data), (n, data[:24].hex())


def test_fast_walk_equals_the_exact_walk_on_random_windows(tmp_path):
    """span_par_walk (the lane-parallel walk of round 5) against span_walk on
    random per-lane results - hit densities from sparse to every lane, match
    lengths of every class, C bits with preds anywhere below, chain / run
    starts: inserted lanes, tokens and the state left behind must be identical
    wherever the fast walk may run, and span_fast_ok_w must say so where
    span_fast_ok does."""
    so = tmp_path / "span_wave_host.so"
    subprocess.check_call(
        ["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
         "-I", str(ROOT / "rust-snappy_amd" / "csrc"),
         str(ROOT / "tests" / "span_wave_host.cpp"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.span_walk_diff.restype = C.c_uint32
    L.span_walk_diff.argtypes = [C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint64)]
    seen = (C.c_uint64 * 4)()
    for seed in range(1, 9):
        assert L.span_walk_diff(seed, 200000, seen) == 0, seed
    windows, cuts, longs, runs = list(seen)
    assert windows > 500000 and cuts > 50000 and longs > 50000 \
        and runs > 5000, list(seen)


def test_small_block_tables_at_every_length(span):
    """The small-block window kernel (k_match_spans_8k) gives a block of
    n <= 8 192 bytes a table of that many entries at most (the model also
    checks 4 096 for n <= 4 096): the host
    model allocates exactly that room and checks every index, the table must
    be what the reference sizes for n (src/compress.rs:491-518: the power of
    two at or above n, 256 .. 16 384), and the stream the oracle's - every
    length 1 024 .. 8 193, three kinds of data."""
    rng = random.Random(5)
    txt = (O.CORPUS / "alice29.txt").read_bytes()
    noise = bytes(rng.randrange(4) for _ in range(9000))
    for n in range(1024, 8194):
        want_table = 256
        while want_table < 16384 and want_table < n:
            want_table *= 2
        for data in (txt[n:2 * n], noise[:n], (txt[:97] * 90)[:n]):
            got, st = span(data)
            assert st[8] == want_table, (n, st[8])
            assert got == O.compress(data), n
