"""rust-snappy_amd: MI355X-native Snappy raw block codec.

The product is `libsnapmi.so` (hand-written HIP kernels for gfx950 behind the
C ABI of include/snapmi.h).  This package is the thin host-side mirror of the
reference's `snap` crate surface for that hot path:

    rust_snappy_amd.raw.{Encoder, Decoder, max_compress_len, decompress_len}
    rust_snappy_amd.Error

Import name: `rust_snappy_amd` (the directory is `rust-snappy_amd/`; the
top-level `rust_snappy_amd.py` shim maps one onto the other).
"""
from . import _lib
from .error import DeviceError, Error

_lib.load()  # fail loudly at import if the HIP library has not been built

from . import raw  # noqa: E402
from . import frame  # noqa: E402
from . import frame as read  # snap::read::{FrameDecoder, FrameEncoder}
from . import frame as write  # snap::write::FrameEncoder

__all__ = ["raw", "frame", "read", "write", "Error", "DeviceError"]
