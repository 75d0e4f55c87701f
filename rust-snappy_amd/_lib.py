"""ctypes binding of libsnapmi.so (include/snapmi.h).

The shared library is the product: HIP kernels for gfx950 behind a C ABI.
This module only declares signatures.  It never falls back to a CPU codec:
if the library is missing, importing the package fails loudly.
"""
import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
# SNAPMI_LIB selects another build of the same ABI (the profile build of
# `make -C rust-snappy_amd/csrc profile`, a fresh build in a scratch
# directory: __graft_entry__.fresh_build_check); never a CPU codec.
# A process that says SNAPMI_TESTING=1 (the test suite: tests/conftest.py; the
# experiment drivers under tests/hw/) gets libsnapmi_test.so - the same
# sources with the test knobs and the cross-check kernels compiled in
# (include/snapmi_test.h), which the product library does not carry.
_DEFAULT = "libsnapmi_test.so" if os.environ.get("SNAPMI_TESTING") \
    else "libsnapmi.so"
LIB_PATH = Path(os.environ.get("SNAPMI_LIB", PKG_DIR / _DEFAULT))


class SnapmiError(C.Structure):
    """snapmi_error: (kind, a, b, c) -- snap::Error variant + fields."""
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_uint32),
                ("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64)]


class SnapmiTiming(C.Structure):
    _fields_ = [("plan_ms", C.c_float), ("codec_ms", C.c_float),
                ("compact_ms", C.c_float), ("total_ms", C.c_float),
                ("codec_launches", C.c_uint64), ("dominant_ms", C.c_float),
                ("reserved", C.c_float)]


# every symbol include/snapmi.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SZ = C.c_size_t
_SZP = C.POINTER(C.c_size_t)
_ERRP = C.POINTER(SnapmiError)
SYMBOLS = [
    ("snappy_compress", C.c_int, [C.c_char_p, _SZ, _P, _SZP]),
    ("snappy_uncompress", C.c_int, [C.c_char_p, _SZ, _P, _SZP]),
    ("snappy_max_compressed_length", _SZ, [_SZ]),
    ("snappy_uncompressed_length", C.c_int, [C.c_char_p, _SZ, _SZP]),
    ("snappy_validate_compressed_buffer", C.c_int, [C.c_char_p, _SZ]),
    ("snapmi_ctx_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("snapmi_ctx_destroy", None, [_P]),
    ("snapmi_last_error", C.c_char_p, [_P]),
    ("snapmi_table_probe_log", C.c_char_p, [_P]),
    ("snapmi_last_kernel", C.c_char_p, [_P]),
    ("snapmi_error_string", _SZ, [_ERRP, C.c_char_p, _SZ]),
    ("snapmi_host_alloc", _P, [_SZ]),
    ("snapmi_host_free", None, [_P]),
    ("snapmi_ctx_stream", _P, [_P]),
    ("snapmi_version", C.c_char_p, []),
    ("snapmi_ctx_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("snapmi_ctx_prepare", C.c_int, [_P, C.c_uint64, C.c_uint32]),
    ("snapmi_ctx_get_info", C.c_int,
     [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    ("snapmi_ctx_set_test_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("snapmi_max_compress_len", _SZ, [_SZ]),
    ("snapmi_decompress_len", C.c_int, [C.c_char_p, _SZ, _SZP, _ERRP]),
    ("snapmi_raw_compress", C.c_int,
     [_P, C.c_char_p, _SZ, _P, _SZ, _SZP, _ERRP]),
    ("snapmi_raw_decompress", C.c_int,
     [_P, C.c_char_p, _SZ, _P, _SZ, _SZP, _ERRP]),
    ("snapmi_compress_batch", C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _SZ]),
    ("snapmi_decompress_batch", C.c_int, [_P, _P, _P, _P, _P, _P, _P, _SZ]),
    ("snapmi_decompress_len_batch", C.c_int, [_P, _P, _P, _P, _P, _SZ]),
    ("snapmi_decompress_stream", C.c_int,
     [_P, _P, C.c_uint64, _P, C.c_uint64, _P, _P]),
    ("snapmi_stream_decode_path", C.c_int, [_P]),
    ("snapmi_ctx_synchronize", C.c_int, [_P]),
    ("snapmi_last_timing", C.c_int, [_P, C.POINTER(SnapmiTiming)]),
    ("snapmi_frame_max_len", _SZ, [_SZ]),
    ("snapmi_frame_compress", C.c_int,
     [_P, _P, C.c_uint64, _P, C.c_uint64, _P, _P]),
    ("snapmi_frame_decompress", C.c_int,
     [_P, _P, C.c_uint64, _P, C.c_uint64, _P, _P, _P, C.c_uint64]),
    ("snapmi_frame_compress_chunks", C.c_int,
     [_P, _P, _P, _SZ, C.c_uint32, _P, C.c_uint64, _P, _P]),
    ("snapmi_frame_decompress_ex", C.c_int,
     [_P, _P, C.c_uint64, _P, C.c_uint64, _P, _P, _P, C.c_uint64,
      C.c_uint32, _P]),
    ("snapmi_frame_scan_host", C.c_int,
     [_P, C.c_uint64, C.c_uint32, _P, _P, C.c_uint64,
      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("snapmi_frame_encode_bound", _SZ, [_SZ, _SZ]),
    ("snapmi_frame_encode_host", C.c_int,
     [_P, _P, _P, _SZ, C.c_uint32, _P, _SZ, _SZP]),
    ("snapmi_frame_decode_host", C.c_int,
     [_P, _P, _SZ, C.c_uint32, _P, _P, _SZ, _SZP, _SZP, _ERRP]),
    ("snapmi_frame_index_host", C.c_int,
     [_P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("snapmi_crc32c_masked_batch", C.c_int, [_P, _P, _P, _P, _SZ]),
    ("snapmi_comm_unique_id", C.c_int, [_P]),
    ("snapmi_comm_init", C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("snapmi_comm_wrap", C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    ("snapmi_comm_destroy", None, [_P]),
    ("snapmi_gatherv", C.c_int,
     [_P, _P, C.c_int, _P, C.c_uint64, _P, C.c_uint64, _P,
      C.POINTER(C.c_uint64)]),
]

# include/snapmi_test.h: exported by libsnapmi_test.so only
TEST_ONLY = {"snapmi_ctx_set_test_option"}

_lib = None
_product = None


def _open(path):
    if not path.exists():
        raise ImportError(
            f"{path} is missing: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(str(path), mode=getattr(os, "RTLD_LOCAL", 0))
    for name, res, args in SYMBOLS:
        if name in TEST_ONLY and not hasattr(L, name):
            continue          # the product build: no test knobs
        f = getattr(L, name)  # AttributeError if the ABI lost a symbol
        f.restype = res
        f.argtypes = args
    return L


def load():
    """Load the process's library (libsnapmi.so; libsnapmi_test.so under
    SNAPMI_TESTING=1); raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # In a torch process both libsnapmi.so and torch need the HIP runtime
    # (SONAME libamdhip64.so.7).  torch dlopens its bundled copy by absolute
    # path, so it must come first: libsnapmi.so then binds to that same copy
    # by SONAME.  Two HIP/HSA runtimes in one process cannot both open the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _open(LIB_PATH)
    return _lib


def load_product():
    """The SHIPPED library, libsnapmi.so, whatever load() gives this process:
    the test suite runs on the test build and, through contexts made with
    Context(lib=load_product()), on this one as well (both can live in one
    process: they are loaded RTLD_LOCAL)."""
    global _product
    if _product is None:
        load()
        prod = PKG_DIR / "libsnapmi.so"
        _product = _lib if LIB_PATH == prod else _open(prod)
    return _product


def of(ctx):
    """The library a context was made by (every call that takes a context
    must go to that one)."""
    return ctx._L if ctx is not None and getattr(ctx, "_L", None) else load()
