"""Snappy frame format: host-side mirror of the reference's streaming types

    snap::write::FrameEncoder<W>   reference src/write.rs:36-162
    snap::read::FrameDecoder<R>    reference src/read.rs:47-239
    snap::read::FrameEncoder<R>    reference src/read.rs:272-363

on top of the device frame layer of libsnapmi.so (CRC32C kernel, chunk
compress / decode kernels).  The reference compresses one <=64 KiB chunk per
call; here the writer collects everything written between flushes and hands
the device ALL chunks at once (every chunk is an independent raw stream), so
chunk boundaries are exactly the reference's: 65536-byte multiples of the
bytes written since the last flush (src/write.rs:123-152).
"""
import ctypes as C
import io

import numpy as np
import torch

from . import _lib, raw
from .error import Error

STREAM_IDENTIFIER = b"\xff\x06\x00\x00sNaPpY"   # reference src/frame.rs:18
MAX_BLOCK_SIZE = 1 << 16                        # reference src/lib.rs:97


def frame_max_len(n):
    return _lib.load().snapmi_frame_max_len(int(n))


def _err_tuple(t):
    rec = np.frombuffer(t.cpu().numpy().tobytes(), dtype=np.dtype(
        [("kind", "<i4"), ("r", "<u4"), ("a", "<u8"), ("b", "<u8"),
         ("c", "<u8")]))[0]
    return int(rec["kind"]), int(rec["a"]), int(rec["b"]), int(rec["c"])


def compress_device(ctx, d_in, want_index=False):
    """Frame-compress a uint8 CUDA tensor; returns (framed tensor, length,
    chunk offsets tensor or None).  Asynchronous work is synchronised."""
    n = d_in.numel()
    dev = d_in.device
    cap = frame_max_len(n)
    out = torch.empty(max(cap, 16), dtype=torch.uint8, device=dev)
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    chunks = (n + MAX_BLOCK_SIZE - 1) // MAX_BLOCK_SIZE
    index = (torch.zeros(chunks + 1, dtype=torch.int64, device=dev)
             if want_index else None)
    rc = _lib.load().snapmi_frame_compress(
        ctx._h, C.c_void_p(d_in.data_ptr()) if n else None, n,
        C.c_void_p(out.data_ptr()), cap, C.c_void_p(out_len.data_ptr()),
        C.c_void_p(index.data_ptr()) if index is not None else None)
    if rc:
        raw._raise(ctx, rc)
    ctx.synchronize()
    return out, int(out_len.item()), index


def decompress_device(ctx, d_in, n_in, index=None, out_cap=None):
    """Frame-decompress d_in[:n_in]; returns (output tensor, length).
    Raises snap.Error with the reference's variant and fields."""
    dev = d_in.device
    L = _lib.load()
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(32, dtype=torch.uint8, device=dev)
    n_idx = (index.numel() - 1) if index is not None else 0
    idx_p = C.c_void_p(index.data_ptr()) if index is not None else None
    in_p = C.c_void_p(d_in.data_ptr()) if n_in else None
    if out_cap is None:  # first pass: total decompressed length
        rc = L.snapmi_frame_decompress(ctx._h, in_p, n_in, None, 0,
                                       C.c_void_p(out_len.data_ptr()),
                                       C.c_void_p(err.data_ptr()), idx_p,
                                       n_idx)
        if rc:
            raw._raise(ctx, rc)
        ctx.synchronize()
        e = _err_tuple(err)
        if e[0]:
            raise Error(*e)
        out_cap = int(out_len.item())
    out = torch.empty(max(out_cap, 16), dtype=torch.uint8, device=dev)
    rc = L.snapmi_frame_decompress(ctx._h, in_p, n_in,
                                   C.c_void_p(out.data_ptr()), out_cap,
                                   C.c_void_p(out_len.data_ptr()),
                                   C.c_void_p(err.data_ptr()), idx_p, n_idx)
    if rc:
        raw._raise(ctx, rc)
    ctx.synchronize()
    e = _err_tuple(err)
    if e[0]:
        raise Error(*e)
    return out, int(out_len.item())


def index_host(data):
    """Chunk scan of a framed stream in host memory (the hops of
    FrameDecoder::read, src/read.rs:105-172): int64 array of the data chunk
    header offsets plus len(data), or None when the stream is not structurally
    regular (decode it without an index: the device walk reports the error)."""
    L = _lib.load()
    if hasattr(data, "data_ptr"):  # a host (CPU) uint8 tensor, not copied
        assert data.device.type == "cpu"
        buf, size = C.c_void_p(data.data_ptr()), data.numel()
    else:
        data = bytes(data)
        buf, size = data, len(data)
    n = C.c_uint64(0)
    if L.snapmi_frame_index_host(buf, size, None, 0, C.byref(n)):
        return None
    offs = np.zeros(n.value + 1, dtype=np.uint64)
    if L.snapmi_frame_index_host(buf, size, offs.ctypes.data_as(
            C.c_void_p), len(offs), C.byref(n)):
        return None
    return offs.astype(np.int64)


def crc32c_masked(ctx, data):
    """CheckSummer::crc32c_masked (reference src/crc32.rs:35-38) of one
    buffer of at most 65536 bytes, on the device."""
    data = bytes(data)
    dev = torch.device("cuda", ctx.device)
    buf = torch.frombuffer(bytearray(data or b"\0"), dtype=torch.uint8).to(dev)
    ptrs = torch.tensor([buf.data_ptr()], dtype=torch.int64, device=dev)
    lens = torch.tensor([len(data)], dtype=torch.int64, device=dev)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _lib.load().snapmi_crc32c_masked_batch(
        ctx._h, C.c_void_p(ptrs.data_ptr()), C.c_void_p(lens.data_ptr()),
        C.c_void_p(out.data_ptr()), 1)
    if rc:
        raw._raise(ctx, rc)
    ctx.synchronize()
    return int(out.item()) & 0xFFFFFFFF


class FrameEncoder:
    """snap::write::FrameEncoder<W>: `write`, `flush`, `into_inner`,
    `get_ref`; flushes on close like the reference's Drop."""

    def __init__(self, wtr, ctx=None):
        self.w = wtr
        self.ctx = ctx or raw.default_context()
        self._src = bytearray()
        self._wrote_ident = False

    def get_ref(self):
        return self.w

    def write(self, buf):
        self._src += bytes(buf)
        return len(buf)

    def write_all(self, buf):
        self.write(buf)

    def flush(self):
        """Everything written so far becomes chunks (reference: a flush
        emits the partial block, src/write.rs:154-161)."""
        if not self._src:
            return
        dev = torch.device("cuda", self.ctx.device)
        d_in = torch.frombuffer(self._src, dtype=torch.uint8).to(dev)
        out, n, _ = compress_device(self.ctx, d_in)
        framed = out[:n].cpu().numpy().tobytes()
        if self._wrote_ident:  # identifier only once per stream (:167-170)
            framed = framed[len(STREAM_IDENTIFIER):]
        self._wrote_ident = True
        self.w.write(framed)
        self._src = bytearray()

    def into_inner(self):
        self.flush()
        return self.w

    def close(self):
        self.flush()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.flush()


class FrameDecoder:
    """snap::read::FrameDecoder<R>: `read`, `get_ref`, `into_inner`."""

    def __init__(self, rdr, ctx=None):
        self.r = rdr
        self.ctx = ctx or raw.default_context()
        self._out = None
        self._pos = 0

    def get_ref(self):
        return self.r

    def into_inner(self):
        return self.r

    def _fill(self):
        data = self.r.read()
        dev = torch.device("cuda", self.ctx.device)
        if not data:
            self._out = b""
            return
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        # the reader's bytes pass through the host anyway: scan the chunk
        # headers here and spare the device its sequential walk
        offs = index_host(data)
        index = torch.from_numpy(offs).to(dev) if offs is not None else None
        out, n = decompress_device(self.ctx, d_in, len(data), index=index)
        self._out = out[:n].cpu().numpy().tobytes()

    def read(self, size=-1):
        if self._out is None:
            self._fill()
        if size is None or size < 0:
            size = len(self._out) - self._pos
        chunk = self._out[self._pos:self._pos + size]
        self._pos += len(chunk)
        return chunk

    def read_to_end(self):
        return self.read(-1)


class ReadFrameEncoder:
    """snap::read::FrameEncoder<R>: reading yields the framed stream."""

    def __init__(self, rdr, ctx=None):
        self.r = rdr
        self.ctx = ctx or raw.default_context()
        self._buf = None
        self._pos = 0

    def get_ref(self):
        return self.r

    def read(self, size=-1):
        if self._buf is None:
            sink = io.BytesIO()
            enc = FrameEncoder(sink, self.ctx)
            enc.write_all(self.r.read())
            enc.flush()
            self._buf = sink.getvalue()
        if size is None or size < 0:
            size = len(self._buf) - self._pos
        chunk = self._buf[self._pos:self._pos + size]
        self._pos += len(chunk)
        return chunk
