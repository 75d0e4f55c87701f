"""Snappy frame format: host-side mirror of the reference's streaming types

    snap::write::FrameEncoder<W>   reference src/write.rs:36-162
    snap::read::FrameDecoder<R>    reference src/read.rs:47-239
    snap::read::FrameEncoder<R>    reference src/read.rs:272-363

on top of the device frame layer of libsnapmi.so (CRC32C kernel, chunk
compress / decode kernels).  The reference handles one <=64 KiB chunk per
call; these adapters run the reference's state machines on the host to decide
WHERE chunks begin and end (so the bytes are the reference's), but hand the
device a bounded batch of chunks at a time (`batch_bytes`, 64 MiB by default;
every chunk is an independent raw stream).  Memory stays bounded by the batch,
data flows while the stream is still being written / read, and a decoder
returns the bytes of the chunks in front of a bad chunk before it reports the
error, like the reference's reader does.
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
    rc = _lib.of(ctx).snapmi_frame_compress(
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
    L = _lib.of(ctx)
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


def compress_chunks_device(ctx, d_in, chunk_lens, ident=True):
    """Frame-compress the chunks d_in[0:l0], d_in[l0:l0+l1], ... (each 1..65536
    bytes, boundaries chosen by the caller; snapmi_frame_compress_chunks).
    Returns (framed tensor, length)."""
    lens = np.ascontiguousarray(chunk_lens, dtype=np.uint32)
    n = int(lens.size)
    total = int(lens.sum(dtype=np.uint64))
    cap = (10 if ident else 0) + total + 8 * n
    dev = d_in.device
    out = torch.empty(max(cap, 16), dtype=torch.uint8, device=dev)
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = _lib.of(ctx).snapmi_frame_compress_chunks(
        ctx._h, C.c_void_p(d_in.data_ptr()) if n else None,
        lens.ctypes.data_as(C.c_void_p), n, 0 if ident else 1,
        C.c_void_p(out.data_ptr()), cap, C.c_void_p(out_len.data_ptr()), None)
    if rc:
        raw._raise(ctx, rc)
    ctx.synchronize()
    return out, int(out_len.item())


def scan_host(data, continuation=False, stale=None):
    """snapmi_frame_scan_host: (status, consumed, offsets) for the complete,
    well-formed chunks at the start of `data` (host bytes).  status 0 = all of
    it, 2 = the next chunk is cut off, 1 = the next chunk is one the decoder
    rejects.  `stale` (bytearray(10), updated in place) is the reference
    reader's src[0..10) - see include/snapmi.h."""
    L = _lib.load()
    data = bytes(data)
    n = C.c_uint64(0)
    used = C.c_uint64(0)
    st = (C.c_uint8 * 10).from_buffer(stale) if stale is not None else None
    tmp = (C.c_uint8 * 10)(*stale) if stale is not None else None
    # first pass counts (on a copy of the stale bytes), second pass fills
    rc = L.snapmi_frame_scan_host(data, len(data), 1 if continuation else 0,
                                  tmp, None, 0, C.byref(n), C.byref(used))
    if rc > 2:
        raise Error(rc)
    offs = np.zeros(n.value + 1, dtype=np.uint64)
    rc = L.snapmi_frame_scan_host(data, len(data), 1 if continuation else 0,
                                  st, offs.ctypes.data_as(C.c_void_p),
                                  len(offs), C.byref(n), C.byref(used))
    if rc > 2:
        raise Error(rc)
    return rc, int(used.value), offs.astype(np.int64)


def decompress_batch_device(ctx, d_in, n_in, n_chunks, index=None,
                            continuation=False, stale=None):
    """One batch of a stream through snapmi_frame_decompress_ex.  The output
    buffer is sized from the chunk count (a chunk yields at most 65536 bytes).
    Returns (bytes in front of the first error - all of them without one,
    Error or None)."""
    dev = d_in.device
    cap = int(n_chunks) * MAX_BLOCK_SIZE
    out = torch.empty(max(cap, 16), dtype=torch.uint8, device=dev)
    out_len = torch.zeros(1, dtype=torch.int64, device=dev)
    err = torch.zeros(32, dtype=torch.uint8, device=dev)
    st = (C.c_uint8 * 10)(*stale) if stale is not None else None
    rc = _lib.of(ctx).snapmi_frame_decompress_ex(
        ctx._h, C.c_void_p(d_in.data_ptr()) if n_in else None, n_in,
        C.c_void_p(out.data_ptr()), cap, C.c_void_p(out_len.data_ptr()),
        C.c_void_p(err.data_ptr()),
        C.c_void_p(index.data_ptr()) if index is not None else None,
        (index.numel() - 1) if index is not None else 0,
        1 if continuation else 0, st)
    if rc:
        raw._raise(ctx, rc)
    ctx.synchronize()
    e = _err_tuple(err)
    good = out[:int(out_len.item())].cpu().numpy().tobytes()
    return good, (Error(*e) if e[0] else None)


def encode_host(ctx, data, chunk_lens, ident=True):
    """snapmi_frame_encode_host: host bytes -> framed bytes, chunk boundaries
    given by the caller."""
    L = _lib.of(ctx)
    lens = np.ascontiguousarray(chunk_lens, dtype=np.uint32)
    n = int(lens.size)
    cap = L.snapmi_frame_encode_bound(len(data), n)
    out = bytearray(max(cap, 1))
    written = C.c_size_t(0)
    rc = L.snapmi_frame_encode_host(
        ctx._h, (C.c_char * len(data)).from_buffer(data) if len(data) else None,
        lens.ctypes.data_as(C.c_void_p), n, 0 if ident else 1,
        (C.c_char * len(out)).from_buffer(out), cap, C.byref(written))
    if rc:
        raw._raise(ctx, rc)
    return bytes(out[:written.value])


class HostBuffer:
    """Pinned host memory (snapmi_host_alloc): what makes the copies of the
    host-buffer calls asynchronous, so that the three legs of a batch (in,
    kernels, out) overlap.  `.view` is a writable memoryview, `.array` a
    uint8 numpy array over the same bytes."""

    def __init__(self, nbytes):
        L = _lib.load()
        self.nbytes = int(nbytes)
        self.ptr = L.snapmi_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MemoryError(f"snapmi_host_alloc({nbytes})")
        self._c = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)
        self.view = memoryview(self._c).cast("B")[:self.nbytes]
        self.array = np.frombuffer(self._c, dtype=np.uint8,
                                   count=self.nbytes)

    def close(self):
        if self.ptr:
            self.view = self.array = self._c = None
            _lib.load().snapmi_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def _address(buf):
    """(address, length) of a buffer-protocol object, without a copy."""
    a = np.frombuffer(buf, dtype=np.uint8)
    return a.ctypes.data, a.size, a


def encode_host_into(ctx, buf, chunk_lens, out, ident=True):
    """snapmi_frame_encode_host from any buffer into the HostBuffer `out`
    (no copies on the Python side); returns the bytes written."""
    L = _lib.of(ctx)
    lens = np.ascontiguousarray(chunk_lens, dtype=np.uint32)
    addr, n, keep = _address(buf)
    written = C.c_size_t(0)
    rc = L.snapmi_frame_encode_host(
        ctx._h, C.c_void_p(addr), lens.ctypes.data_as(C.c_void_p),
        int(lens.size), 0 if ident else 1, C.c_void_p(out.ptr), out.nbytes,
        C.byref(written))
    del keep
    if rc:
        raw._raise(ctx, rc)
    return written.value


def decode_host(ctx, data, out, continuation, final, stale):
    """snapmi_frame_decode_host: decode the whole chunks at the start of
    `data` (bytes) into the bytearray `out`; returns (written, consumed,
    Error or None).  `stale`: bytearray(10) of decoder state, updated."""
    L = _lib.of(ctx)
    written, consumed = C.c_size_t(0), C.c_size_t(0)
    err = _lib.SnapmiError()
    flags = (1 if continuation else 0) | (2 if final else 0)
    addr, n_in, keep = _address(data)      # bytes, memoryview, pinned view
    rc = L.snapmi_frame_decode_host(
        ctx._h, C.c_void_p(addr), n_in, flags,
        (C.c_uint8 * 10).from_buffer(stale),
        (C.c_char * len(out)).from_buffer(out), len(out), C.byref(written),
        C.byref(consumed), C.byref(err))
    del keep
    if rc >= 100:
        raw._raise(ctx, rc)
    e = Error(err.kind, err.a, err.b, err.c) if rc else None
    return written.value, consumed.value, e


def count_chunks_host(data, continuation):
    """snapmi_frame_scan_host (host code, no GPU) without a copy of `data`:
    (data chunks, bytes) of the complete, well-formed chunks at its start."""
    L = _lib.load()
    addr, n, keep = _address(data)
    nd, used = C.c_uint64(0), C.c_uint64(0)
    rc = L.snapmi_frame_scan_host(C.c_void_p(addr), n,
                                  1 if continuation else 0, None, None, 0,
                                  C.byref(nd), C.byref(used))
    del keep
    if rc > 3:
        raise Error(rc, message="snapmi_frame_scan_host")
    return nd.value, used.value


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
    rc = _lib.of(ctx).snapmi_crc32c_masked_batch(
        ctx._h, C.c_void_p(ptrs.data_ptr()), C.c_void_p(lens.data_ptr()),
        C.c_void_p(out.data_ptr()), 1)
    if rc:
        raw._raise(ctx, rc)
    ctx.synchronize()
    return int(out.item()) & 0xFFFFFFFF


BATCH_BYTES = 64 << 20  # what the streaming adapters hand the device at once


class IntoInnerError(Exception):
    """std::io::IntoInnerError (re-exported by the reference, src/write.rs:17):
    into_inner() could not flush; carries the encoder and the error."""

    def __init__(self, encoder, err):
        super().__init__(f"into_inner: flush failed: {err}")
        self._encoder, self._err = encoder, err

    def error(self):
        return self._err

    def into_inner(self):
        return self._encoder


class FrameEncoder:
    """snap::write::FrameEncoder<W> (reference src/write.rs:36-192): `write`,
    `flush`, `into_inner`, `get_ref`, `get_mut`; flushes on close / with-exit
    like the reference's Drop.

    The chunking is the reference's state machine: a 65536-byte buffer `src`;
    a write that does not fit fills the buffer and emits it - or, when the
    buffer is empty, goes out directly as ceil(len / 65536) chunks INCLUDING
    its partial tail (src/write.rs:123-152,171-190).  Chunks are queued and
    compressed `batch_bytes` at a time; flush() compresses what is queued."""

    # a write of at least this many bytes that arrives with an empty block
    # buffer is compressed where it lies (no queue, no copy), at most
    # DIRECT_MAX bytes per device call (which bounds the pinned staging of
    # the framed bytes: 1.2 GiB)
    DIRECT_MIN = 4 << 20
    DIRECT_MAX = 1 << 30

    def __init__(self, wtr, ctx=None, batch_bytes=BATCH_BYTES):
        self.w = wtr
        self.ctx = ctx or raw.default_context()
        self.batch_bytes = max(int(batch_bytes), MAX_BLOCK_SIZE)
        self._src = bytearray()      # reference `src`, capacity MAX_BLOCK_SIZE
        self._queue = []             # chunks cut but not yet compressed
        self._queued = 0
        self._wrote_ident = False
        self._stage = None           # pinned staging of the framed output

    def get_ref(self):
        return self.w

    get_mut = get_ref

    def _emit_direct(self, buf):
        """The chunks of `buf` (cut like _inner_write cuts them) straight from
        the caller's memory through the pipelined host call; the framed bytes
        reach the writer as a view of pinned staging memory."""
        for lo in range(0, len(buf), self.DIRECT_MAX):
            part = buf[lo:lo + self.DIRECT_MAX]
            n = len(part)
            nch = (n + MAX_BLOCK_SIZE - 1) // MAX_BLOCK_SIZE
            lens = np.full(nch, MAX_BLOCK_SIZE, dtype=np.uint32)
            lens[-1] = n - (nch - 1) * MAX_BLOCK_SIZE
            cap = 10 + n + 8 * nch
            if self._stage is None or self._stage.nbytes < cap:
                if self._stage is not None:
                    self._stage.close()
                self._stage = HostBuffer(cap + cap // 8)
            k = encode_host_into(self.ctx, part, lens, self._stage,
                                 ident=not self._wrote_ident)
            self._wrote_ident = True
            self._write_all(self._stage.view[:k])

    def _write_all(self, buf):
        """io::Write::write_all on the inner writer: a writer may take fewer
        bytes than it is given (raw files, sockets).  `buf` is BORROWED, as a
        &[u8] is in the reference: it is only valid during the call (the
        direct path lends a view of pinned staging memory that the next
        write overwrites) - a writer that keeps bytes must copy them, which
        io.BytesIO, files and sockets do."""
        mv = memoryview(buf).cast("B")
        while len(mv):
            k = self.w.write(mv)
            if k is None or k >= len(mv):   # (None: everything, io.RawIOBase
                break                        # aside, which says so with 0)
            if k <= 0:
                raise OSError("FrameEncoder: the inner writer accepts no "
                              "more bytes (write returned 0)")
            mv = mv[k:]

    # reference Inner::write (src/write.rs:171-190): cut `buf` into chunks
    def _inner_write(self, buf):
        if len(buf) >= self.DIRECT_MIN and isinstance(buf, memoryview):
            self._emit()             # what is queued goes first
            self._emit_direct(buf)
            return len(buf)
        for o in range(0, len(buf), MAX_BLOCK_SIZE):
            self._queue.append(bytes(buf[o:o + MAX_BLOCK_SIZE]))
        self._queued += len(buf)
        if self._queued >= self.batch_bytes:
            self._emit()
        return len(buf)

    def _emit(self):
        if not self._queue:
            return
        lens = np.fromiter((len(c) for c in self._queue), dtype=np.uint32,
                           count=len(self._queue))
        host = bytearray().join(self._queue)
        self._queue, self._queued = [], 0
        framed = encode_host(self.ctx, host, lens,
                             ident=not self._wrote_ident)
        self._wrote_ident = True   # identifier only once (:167-170)
        self._write_all(framed)

    def write(self, buf):
        try:
            buf = memoryview(buf).cast("B")      # no copy
        except TypeError:
            buf = memoryview(bytes(buf))
        total = 0
        while True:  # src/write.rs:123-152
            free = MAX_BLOCK_SIZE - len(self._src)
            if len(buf) <= free:
                break
            if not self._src:
                n = self._inner_write(buf)
            else:
                self._src += buf[:free]
                self._flush_src()
                n = free
            buf = buf[n:]
            total += n
        self._src += buf
        return total + len(buf)

    def write_all(self, buf):
        self.write(buf)

    def _flush_src(self):
        if self._src:
            self._inner_write(self._src)
            self._src = bytearray()

    def flush(self):
        """src/write.rs:154-161: the partial block becomes a chunk; everything
        queued is compressed and written to the inner writer."""
        self._flush_src()
        self._emit()

    def into_inner(self):
        try:
            self.flush()
        except Exception as e:  # src/write.rs:91-97
            raise IntoInnerError(self, e) from e
        return self.w

    def _release(self):
        if self._stage is not None:   # the pinned staging of direct writes
            self._stage.close()
            self._stage = None

    def close(self):
        try:
            self.flush()
        finally:
            self._release()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        try:                # Drop ignores flush errors (src/write.rs:112-120)
            self.flush()
        except Exception:
            if a[0] is None:
                raise
        finally:
            self._release()


class _ReaderFailed(Exception):
    """The INNER reader raised (not the device): the caller may retry, what
    was read before stays staged."""

    def __init__(self, err):
        super().__init__(str(err))
        self.err = err


class HostReader:
    """A reader over pinned host memory (a HostBuffer's view, or any buffer)
    that LENDS its bytes - getbuffer() / tell() / seek(), like io.BytesIO -
    instead of copying them out: FrameDecoder then hands the device call the
    reader's own memory (BufRead::fill_buf / consume in the reference's
    language).  With a pinned buffer that is the whole host-to-device leg
    without a copy on the host."""

    def __init__(self, buf):
        self._v = memoryview(buf).cast("B")
        self._p = 0

    def getbuffer(self):
        return self._v

    def tell(self):
        return self._p

    def seek(self, pos, whence=0):
        self._p = max(0, min(len(self._v), pos if whence == 0 else
                             (self._p + pos if whence == 1
                              else len(self._v) + pos)))
        return self._p

    def read(self, n=-1):
        end = len(self._v) if n is None or n < 0 else min(len(self._v),
                                                           self._p + n)
        b = bytes(self._v[self._p:end])
        self._p = end
        return b

    def readinto(self, b):
        mv = memoryview(b).cast("B")
        k = min(len(mv), len(self._v) - self._p)
        mv[:k] = self._v[self._p:self._p + k]
        self._p += k
        return k


class FrameDecoder:
    """snap::read::FrameDecoder<R> (reference src/read.rs:47-239): `read`,
    `readinto`, `read_to_end`, `get_ref`, `get_mut`, `into_inner`.

    Takes at most `batch_bytes` from the reader at a time, cuts the batch at
    its last complete chunk (host scan of the chunk headers, which also sizes
    the output and serves as the side index) and decodes those chunks in one
    device call.  Bytes of the chunks in front of a bad chunk are returned
    first; the error is raised by the read that reaches it
    (src/read.rs:111-118).

    Where the bytes live: a reader that lends its buffer (io.BytesIO,
    HostReader: getbuffer / tell / seek) is decoded where it lies, nothing is
    copied on the host; any other reader fills pinned staging memory -
    through readinto / readinto1 when it has one (one copy, no allocation),
    through read otherwise.  Decoded bytes land in pinned room kept from
    batch to batch and are handed out from there (`read` copies what it
    returns once; `readinto` with room for a batch gets the bytes straight
    from the device call; `read_to_end` appends batch by batch to one
    bytearray)."""

    def __init__(self, rdr, ctx=None, batch_bytes=BATCH_BYTES):
        self.r = rdr
        self.ctx = ctx or raw.default_context()
        self.batch_bytes = max(int(batch_bytes), 1 << 17)
        self._lends = all(callable(getattr(rdr, a, None))
                          for a in ("getbuffer", "tell", "seek"))
        self._no_into = False
        self._in = None          # pinned staging of the input (grow-only)
        self._have = 0           # _in[:_have]: read, not decoded yet
        self._outbuf = None      # pinned room for a batch's output
        self._out = memoryview(b"")
        self._pos = 0
        self._err = None
        self._io_err = None      # the inner reader's error (retryable)
        self._eof = False
        self._seen_ident = False
        self._stale = bytearray(10)  # reference src[0..10): include/snapmi.h

    def get_ref(self):
        return self.r

    get_mut = get_ref

    def into_inner(self):
        return self.r

    def close(self):
        """Release the pinned buffers (they are also released when the
        decoder is collected)."""
        self._out = memoryview(b"")
        for b in (self._in, self._outbuf):
            if b is not None:
                b.close()
        self._in = self._outbuf = None

    def _stage(self, want):
        """Pinned input staging of at least `want` bytes, what is staged
        kept."""
        if self._in is None or self._in.nbytes < want:
            new = HostBuffer(want + want // 8)
            if self._have:
                new.view[:self._have] = self._in.view[:self._have]
            if self._in is not None:
                self._in.close()
            self._in = new
        return self._in.view

    def _pull(self, want):
        """Bytes for one batch: what is staged plus what the reader has NOW.
        The reference reads one chunk per call (src/read.rs:105-172);
        batching must not turn into waiting: one inner read at least, more
        only while the reader keeps filling what it is offered (a short read
        means it has no more at the moment - a pipe, a socket, a
        request/response peer that waits for our answer before it sends
        on).  A reader's exception leaves what was read before it staged."""
        view = self._stage(want)
        into = getattr(self.r, "readinto1", None) or getattr(
            self.r, "readinto", None)
        rd = getattr(self.r, "read1", None) or getattr(self.r, "read", None)
        try:
            while self._have < want and not self._eof:
                ask = want - self._have
                k = None
                if into is not None and not self._no_into:
                    try:
                        k = into(view[self._have:want]) or 0
                    except (NotImplementedError, io.UnsupportedOperation):
                        self._no_into = True   # declared, not implemented
                if k is None:
                    b = rd(ask)
                    k = len(b) if b else 0
                    view[self._have:self._have + k] = b[:k] if k else b""
                if not k:
                    self._eof = True
                    break
                self._have += k
                if k < ask:
                    break
        except Exception as e:  # noqa: BLE001 - the reader's, re-raised
            raise _ReaderFailed(e) from e
        return view[:self._have]

    def _room(self, nbytes):
        """Pinned memory for one batch's output, kept from batch to batch:
        the copy home is asynchronous into it (a fresh bytearray per batch
        cost a page fault per 4 KiB and a pageable copy)."""
        if self._outbuf is None or self._outbuf.nbytes < nbytes:
            if self._outbuf is not None:
                self._out = memoryview(b"")
                self._outbuf.close()
            self._outbuf = HostBuffer(nbytes + nbytes // 8)
        return self._outbuf.view[:nbytes]

    def _fill(self, direct=None):
        """Decode the next batch.  Into `direct` (a writable memoryview of
        the caller's, readinto) when that has room for every chunk of the
        batch - the bytes are then the caller's without another copy and
        their number is returned; into the decoder's own pinned room
        otherwise (returns 0, the bytes wait in _out)."""
        # (a lending reader costs no staging memory: four times the batch,
        # fewer device calls - each has a latency floor of a millisecond or
        # two - when the output goes straight to the caller)
        want = self.batch_bytes * (4 if self._lends and direct is not None
                                   else 1)
        lent = None
        try:
            while True:
                if self._lends:
                    # the reader's own memory: nothing is copied or staged
                    lent = self.r.getbuffer()
                    at = self.r.tell()
                    data = memoryview(lent)[at:at + want]
                    self._eof = at + want >= len(lent)
                else:
                    data = self._pull(want)
                if not len(data):
                    return 0
                # room for every data chunk of the batch (the scan counts
                # them): a chunk yields at most 65536 bytes
                nd, _ = count_chunks_host(data, self._seen_ident)
                room = max(nd, 1) * MAX_BLOCK_SIZE
                mine = direct is None or len(direct) < room
                out = self._room(room) if mine else direct[:room]
                try:
                    n, used, err = decode_host(self.ctx, data, out,
                                               self._seen_ident, self._eof,
                                               self._stale)
                except Error:
                    raise
                except Exception as e:  # noqa: BLE001 - a device failure is
                    self._err = Error(   # final: a retry would skip input
                        101, message=f"frame_decode_host: {e}")
                    raise self._err from e
                if err is None and used == 0:   # not one whole chunk yet
                    want = len(data) + self.batch_bytes
                    if self._eof:
                        raise Error(101, message="frame_decode_host made no "
                                                 "progress at end of input")
                    if lent is not None:
                        data.release()
                        lent = None
                    continue
                break
            if self._lends:
                self.r.seek(at + (used if err is None else len(data)))
            else:
                rest = self._have - used if err is None else 0
                if rest and used:
                    self._in.view[:rest] = self._in.view[used:self._have]
                self._have = rest
            self._seen_ident = self._seen_ident or used > 0
            self._err = err
            if mine:
                self._out, self._pos = out[:n], 0
                return 0
            self._out, self._pos = memoryview(b""), 0
            return n
        finally:
            if lent is not None:
                try:
                    data.release()
                except Exception:  # noqa: BLE001
                    pass

    def _drained(self):
        return self._eof and not self._have and (
            not self._lends or self.r.tell() >= len(self.r.getbuffer()))

    def readinto(self, b):
        """io::Read::read (reference src/read.rs:104-239): up to len(b)
        bytes into the writable buffer `b`; 0 at the end of the stream.  A
        buffer with room for a whole batch (pinned: HostBuffer.view) takes
        the decoded bytes straight from the device call."""
        mv = memoryview(b).cast("B")
        if len(mv) == 0:
            return 0
        while True:
            if self._pos < len(self._out):
                k = min(len(mv), len(self._out) - self._pos)
                mv[:k] = self._out[self._pos:self._pos + k]
                self._pos += k
                return k
            if self._err is not None:
                raise self._err
            if self._io_err is not None:
                e, self._io_err = self._io_err, None
                raise e
            if self._drained():
                return 0
            try:
                n = self._fill(mv)
            except _ReaderFailed as e:   # nothing is lost: what the reader
                raise e.err from None    # gave before stays staged
            if n:
                return n
            if not len(self._out) and self._err is None and self._drained():
                return 0

    def _gather(self, size, acc):
        """Up to `size` bytes (all that remain for None) appended to the
        bytearray `acc`; an error is raised once the bytes in front of it
        have been handed out."""
        need, got = size, 0
        while need is None or need > 0:
            if self._pos < len(self._out):
                end = len(self._out) if need is None else min(
                    len(self._out), self._pos + need)
                acc += self._out[self._pos:end]
                got += end - self._pos
                if need is not None:
                    need -= end - self._pos
                self._pos = end
                continue
            if self._err is not None:
                if got:
                    break          # hand out the good bytes first
                raise self._err
            if self._io_err is not None:
                if got:
                    break
                e, self._io_err = self._io_err, None
                raise e            # a caller may retry: nothing was lost
            if self._drained():
                break
            try:
                self._fill()
            except _ReaderFailed as e:
                self._io_err = e.err   # behind the bytes gathered so far
                continue
        return got

    def read(self, size=-1):
        """Up to `size` bytes (all remaining for size < 0), as bytes.  An
        error is raised once the bytes in front of it have been returned."""
        acc = bytearray()
        self._gather(None if size is None or size < 0 else size, acc)
        return bytes(acc)

    def read_to_end(self, buf=None):
        """io::Read::read_to_end (the reference appends to a Vec<u8>): all
        remaining bytes appended to the bytearray `buf` (a new one by
        default), which is returned - or the stream's error, with the bytes
        decoded in front of it in `.partial`."""
        acc = bytearray() if buf is None else buf
        self._gather(None, acc)
        if self._err is not None:
            e = self._err
            e.partial = acc
            raise e
        return acc


class ReadFrameEncoder:
    """snap::read::FrameEncoder<R> (reference src/read.rs:272-409): reading
    yields the framed stream.  Like the reference, every chunk is what ONE
    read of up to 65536 bytes from the inner reader returned (read.rs:378);
    up to `batch_bytes` of such reads are compressed in one device call."""

    def __init__(self, rdr, ctx=None, batch_bytes=BATCH_BYTES):
        self.r = rdr
        self.ctx = ctx or raw.default_context()
        self.batch_bytes = max(int(batch_bytes), MAX_BLOCK_SIZE)
        self._buf = b""
        self._pos = 0
        self._eof = False
        self._wrote_ident = False
        self._chunks = []        # read, not yet compressed
        self._pending = None     # a reader's error, raised behind _buf

    def get_ref(self):
        return self.r

    get_mut = get_ref

    def _fill(self):
        """Up to batch_bytes of inner reads -> one device call.  A read that
        fails does not lose the chunks gathered before it: they are
        compressed and handed out, the error is raised by the read that
        follows them (the reference does one inner read per outer read,
        src/read.rs:378, so an error there loses nothing either).  A short
        read ends the batch: the reader has no more right now."""
        total = sum(len(c) for c in self._chunks)
        while total < self.batch_bytes and self._pending is None:
            try:
                b = self.r.read(MAX_BLOCK_SIZE)
            except InterruptedError:
                continue                     # ErrorKind::Interrupted: retry
            except Exception as e:           # noqa: BLE001 - re-raised later
                self._pending = e
                break
            if not b:
                self._eof = True
                break
            self._chunks.append(bytes(b))
            total += len(b)
            if len(b) < MAX_BLOCK_SIZE:
                break
        self._buf, self._pos = b"", 0
        if not self._chunks:
            return
        chunks, self._chunks = self._chunks, []
        lens = np.fromiter((len(c) for c in chunks), dtype=np.uint32,
                           count=len(chunks))
        self._buf = encode_host(self.ctx, bytearray().join(chunks), lens,
                                ident=not self._wrote_ident)
        self._wrote_ident = True

    def read(self, size=-1):
        parts, need = [], (None if size is None or size < 0 else size)
        while need is None or need > 0:
            if self._pos < len(self._buf):
                end = len(self._buf) if need is None else min(
                    len(self._buf), self._pos + need)
                parts.append(self._buf[self._pos:end])
                if need is not None:
                    need -= end - self._pos
                self._pos = end
                continue
            if self._pending is not None:
                if parts:
                    break                    # the good bytes first
                e, self._pending = self._pending, None
                raise e
            if self._eof:
                break
            self._fill()
        return b"".join(parts)
