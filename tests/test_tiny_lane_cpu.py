"""The algorithm of k_compress_tiny (one stream under 256 bytes per GPU lane),
run on the CPU: tests/tiny_lane_host.cpp instantiates the very header the
kernel is built from (rust-snappy_amd/csrc/snapmi_tiny.hpp) over byte arrays,
and its bytes are compared with the oracle's for every length 1..255 over
data that takes every branch (literal only, skip loop, copy chains, long and
overlapping copies, copies of 65..255 bytes that are split)."""
import ctypes as C
import random
import subprocess

import pytest

import oracle_lib as O
from conftest import ROOT


@pytest.fixture(scope="module")
def lane(tmp_path_factory):
    so = tmp_path_factory.mktemp("tiny") / "tiny_lane_host.so"
    subprocess.check_call(
        ["g++", "-O1", "-shared", "-fPIC", "-std=c++17",
         "-I", str(ROOT / "rust-snappy_amd" / "csrc"),
         str(ROOT / "tests" / "tiny_lane_host.cpp"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.tiny_lane_compress.restype = C.c_uint32
    L.tiny_lane_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
    L.tiny_lane_out_max.restype = C.c_uint32
    cap = L.tiny_lane_out_max()

    def run(data):
        out = C.create_string_buffer(cap)
        r = L.tiny_lane_compress(bytes(data), len(data), out)
        assert r < 0x80000000, f"assumption violated: flags {r & 0xFFFF:#x}"
        return out.raw[:r]
    return run


def _inputs(n, rng):
    yield bytes(n)                                        # one long copy
    yield bytes(rng.randrange(256) for _ in range(n))     # literal only
    for alphabet in (2, 3, 4, 16):
        yield bytes(rng.randrange(alphabet) for _ in range(n))
    for period in (1, 2, 3, 4, 5, 7, 8, 13, 16, 31, 64, 100):
        unit = bytes(rng.randrange(256) for _ in range(period))
        yield (unit * (n // period + 1))[:n]
    # a repeated phrase with noise between: literals + copies of every length
    phrase = bytes(rng.randrange(256) for _ in range(rng.randrange(4, 90)))
    buf = bytearray()
    while len(buf) < n:
        buf += phrase[:rng.randrange(4, len(phrase) + 1)]
        buf += bytes(rng.randrange(256) for _ in range(rng.randrange(0, 9)))
    yield bytes(buf[:n])


def test_every_length_against_the_oracle(lane):
    rng = random.Random(20260925)
    cases = 0
    for n in range(1, 256):
        for data in _inputs(n, rng):
            assert lane(data) == O.compress(data), (n, data.hex())
            cases += 1
    assert cases > 4000


def test_corpus_slices_against_the_oracle(lane):
    rng = random.Random(7)
    for path in sorted(O.CORPUS.iterdir()):
        blob = path.read_bytes()
        if len(blob) < 512:
            continue
        for _ in range(150):
            n = rng.randrange(1, 256)
            at = rng.randrange(0, len(blob) - n)
            data = blob[at:at + n]
            got = lane(data)
            assert got == O.compress(data), (path.name, at, n)
            assert O.decompress(got) == data


def test_growth_bound(lane):
    """kTinyOutMax (n + 4) is what sizes the lane's output column in LDS."""
    rng = random.Random(3)
    for n in range(1, 256):
        for _ in range(20):
            # pairs of a short literal and a 4-byte copy: the worst case
            unit = bytes(rng.randrange(256) for _ in range(4))
            buf = bytearray()
            while len(buf) < n:
                buf += unit + bytes([rng.randrange(256)])
            assert len(lane(bytes(buf[:n]))) <= n + 4
