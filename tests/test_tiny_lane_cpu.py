"""The algorithm of k_compress_tiny / k_compress_small (one stream under 256
bytes / under 2 KiB per GPU lane), run on the CPU: tests/tiny_lane_host.cpp
instantiates the very header the kernels are built from
(rust-snappy_amd/csrc/snapmi_tiny.hpp) over byte arrays, and its bytes are
compared with the oracle's for every length 1..2047 (and a sample up to a
whole block) over data that takes every branch (literal only, skip loop, copy
chains, long and overlapping copies, copies that are split, literals of every
tag size).  The output must also stay inside the bound the device's output
columns are sized by (input + 4 under 256 bytes, input + 12 under 2 KiB)."""
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
    L.tiny_lane_compress.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p,
                                     C.c_uint32]
    L.tiny_lane_out_max.restype = C.c_uint32
    L.tiny_lane_out_max.argtypes = [C.c_uint32]

    def run(data):
        cap = L.tiny_lane_out_max(len(data))
        out = C.create_string_buffer(cap + 8)
        r = L.tiny_lane_compress(bytes(data), len(data), out, cap)
        assert r < 0x80000000, \
            f"assumption violated: flags {r & 0xFFFF:#x} at {len(data)} bytes"
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
    for n in range(1, 2048):
        for data in _inputs(n, rng):
            assert lane(data) == O.compress(data), (n, data.hex())
            cases += 1
    assert cases > 36000


def test_larger_blocks_against_the_oracle(lane):
    """The header is written for any one-block stream: tables of 4 096 to
    16 384 entries, three-byte headers, offsets past 2 047."""
    rng = random.Random(5)
    blob = b"".join(p.read_bytes() for p in sorted(O.CORPUS.iterdir())
                    if p.stat().st_size > 4096)
    for n in [2048, 2049, 4095, 4096, 4097, 8192, 8193, 16383, 16384, 16385,
              30000, 65535, 65536]:
        for _ in range(3):
            at = rng.randrange(0, len(blob) - n)
            data = blob[at:at + n]
            assert lane(data) == O.compress(data), (n, at)
        data = bytes(rng.randrange(3) for _ in range(n))
        assert lane(data) == O.compress(data), n


def test_corpus_slices_against_the_oracle(lane):
    rng = random.Random(7)
    for path in sorted(O.CORPUS.iterdir()):
        blob = path.read_bytes()
        if len(blob) < 4096:
            continue
        for _ in range(300):
            n = rng.randrange(1, 2048)
            at = rng.randrange(0, len(blob) - n)
            data = blob[at:at + n]
            got = lane(data)
            assert got == O.compress(data), (path.name, at, n)
            assert O.decompress(got) == data


def test_growth_bound(lane):
    """input + 4 (under 256 bytes) and input + 12 (under 2 KiB) size the
    lanes' output columns in LDS: `lane` fails on a byte outside them.  The
    inputs are the ones that grow most: literals just past a tag-size step
    (61, 257 bytes) in front of the shortest copy."""
    rng = random.Random(3)
    for n in range(1, 2048, 3):
        for lit in (1, 60, 61, 256, 257, 300):
            unit = bytes(rng.randrange(256) for _ in range(4))
            buf = bytearray(unit)
            while len(buf) < n:
                buf += bytes(rng.randrange(256) for _ in range(lit)) + unit
            got = lane(bytes(buf[:n]))
            assert len(got) <= n + (4 if n < 256 else 12)
            assert got == O.compress(bytes(buf[:n]))
