"""ctypes access to oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's algorithm
(oracle/snappy_oracle.c).  It may be imported from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg, never from the
product package.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
CORPUS = ROOT / "tests" / "golden" / "corpus"

KIND_NAMES = [
    "Ok", "TooBig", "BufferTooSmall", "Empty", "Header", "HeaderMismatch",
    "Literal", "CopyRead", "CopyWrite", "Offset", "StreamHeader",
    "StreamHeaderMismatch", "UnsupportedChunkType", "UnsupportedChunkLength",
    "Checksum",
]


class OracleError(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_uint32),
                ("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64)]

    def astuple(self):
        return (self.kind, self.a, self.b, self.c)


class Stats(C.Structure):
    _fields_ = [("probes", C.c_uint64), ("copies", C.c_uint64),
                ("literals", C.c_uint64), ("elements", C.c_uint64)]


def build():
    """Compile oracle/liboracle.so if it is missing or stale."""
    so = ORACLE_DIR / "liboracle.so"
    src = ORACLE_DIR / "snappy_oracle.c"
    if (not so.exists()) or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(ORACLE_DIR), "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(build()))
        L.snapo_max_compress_len.restype = C.c_size_t
        L.snapo_max_compress_len.argtypes = [C.c_size_t]
        for name in ("snapo_compress", "snapo_decompress",
                     "snapo_frame_compress", "snapo_frame_decompress"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                          C.POINTER(C.c_size_t), C.POINTER(OracleError)]
        L.snapo_decompress_len.restype = C.c_int
        L.snapo_decompress_len.argtypes = [C.c_char_p, C.c_size_t,
                                           C.POINTER(C.c_size_t),
                                           C.POINTER(OracleError)]
        L.snapo_crc32c.restype = C.c_uint32
        L.snapo_crc32c.argtypes = [C.c_char_p, C.c_size_t]
        L.snapo_crc32c_masked.restype = C.c_uint32
        L.snapo_crc32c_masked.argtypes = [C.c_char_p, C.c_size_t]
        L.snapo_frame_max_len.restype = C.c_size_t
        L.snapo_frame_max_len.argtypes = [C.c_size_t]
        L.snapo_stats_get.argtypes = [C.POINTER(Stats)]
        _lib = L
    return _lib


class SnapError(Exception):
    """Mirror of snap::Error: (kind name, a, b, c)."""

    def __init__(self, kind, a=0, b=0, c=0):
        name = KIND_NAMES[kind] if 0 <= kind < len(KIND_NAMES) else (
            "UnexpectedEof" if kind == -1 else f"kind{kind}")
        super().__init__(f"{name}({a},{b},{c})")
        self.kind, self.name, self.a, self.b, self.c = kind, name, a, b, c

    def key(self):
        return (self.name, self.a, self.b, self.c)


def max_compress_len(n):
    return lib().snapo_max_compress_len(n)


def _call(fn, data, cap):
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(0)
    e = OracleError()
    k = fn(bytes(data), len(data), out, cap, C.byref(n), C.byref(e))
    if k != 0:
        raise SnapError(e.kind, e.a, e.b, e.c)
    return out.raw[:n.value]


def compress(data, cap=None):
    if cap is None:
        cap = max_compress_len(len(data))
    return _call(lib().snapo_compress, data, cap)


def decompress_len(data):
    n = C.c_size_t(0)
    e = OracleError()
    k = lib().snapo_decompress_len(bytes(data), len(data), C.byref(n),
                                   C.byref(e))
    if k != 0:
        raise SnapError(e.kind, e.a, e.b, e.c)
    return n.value


def decompress(data, cap=None):
    if cap is None:
        cap = decompress_len(data)
    return _call(lib().snapo_decompress, data, cap)


def crc32c(data):
    return lib().snapo_crc32c(bytes(data), len(data))


def crc32c_masked(data):
    return lib().snapo_crc32c_masked(bytes(data), len(data))


def frame_compress(data):
    return _call(lib().snapo_frame_compress, data,
                 lib().snapo_frame_max_len(len(data)))


def frame_decompress(data, cap=None):
    if cap is None:
        cap = max(len(data) * 64, 1 << 16)
    return _call(lib().snapo_frame_decompress, data, cap)


def stats_reset():
    lib().snapo_stats_reset()


def stats():
    s = Stats()
    lib().snapo_stats_get(C.byref(s))
    return {"probes": s.probes, "copies": s.copies, "literals": s.literals,
            "elements": s.elements}


# ---- Google libsnappy 1.1.8 (the library the reference cross-tests against,
# snappy-cpp/src/lib.rs:66-88).  Present in this image under /opt/conda/lib;
# used only to cross-check the oracle when it can be loaded.
_snappy = None


def libsnappy():
    global _snappy
    if _snappy is None:
        for p in ("/opt/conda/lib/libsnappy.so.1.1.8",
                  "/opt/conda/lib/libsnappy.so.1"):
            if os.path.exists(p):
                try:
                    L = C.CDLL(p)
                except OSError:
                    continue
                L.snappy_max_compressed_length.restype = C.c_size_t
                L.snappy_max_compressed_length.argtypes = [C.c_size_t]
                _snappy = L
                break
        else:
            _snappy = False
    return _snappy or None


def libsnappy_compress(data):
    L = libsnappy()
    cap = L.snappy_max_compressed_length(len(data))
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    r = L.snappy_compress(bytes(data), C.c_size_t(len(data)), out, C.byref(n))
    assert r == 0
    return out.raw[:n.value]


def libsnappy_uncompress(data):
    L = libsnappy()
    n = C.c_size_t(0)
    r = L.snappy_uncompressed_length(bytes(data), C.c_size_t(len(data)),
                                     C.byref(n))
    if r != 0:
        return None
    out = C.create_string_buffer(max(n.value, 1))
    r = L.snappy_uncompress(bytes(data), C.c_size_t(len(data)), out,
                            C.byref(n))
    if r != 0:
        return None
    return out.raw[:n.value]


BENCH_FILES = [
    ("zflat00_html", "html", None),
    ("zflat01_urls", "urls.10K", None),
    ("zflat02_jpg", "fireworks.jpeg", None),
    ("zflat03_jpg_200", "fireworks.jpeg", 200),
    ("zflat04_pdf", "paper-100k.pdf", None),
    ("zflat05_html4", "html_x_4", None),
    ("zflat06_txt1", "alice29.txt", None),
    ("zflat07_txt2", "asyoulik.txt", None),
    ("zflat08_txt3", "lcet10.txt", None),
    ("zflat09_txt4", "plrabn12.txt", None),
    ("zflat10_pb", "geo.protodata", None),
    ("zflat11_gaviota", "kppkn.gtb", None),
]


def corpus_round():
    """The 12-stream bench round of reference bench/src/bench.rs:83-114."""
    out = []
    for bench_id, fname, limit in BENCH_FILES:
        data = (CORPUS / fname).read_bytes()
        if limit is not None:
            data = data[:limit]
        out.append((bench_id, data))
    return out
