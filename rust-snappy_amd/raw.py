"""snap::raw -- host-side mirror of the reference's raw block API
(src/raw.rs:13-14) on top of the C ABI of libsnapmi.so.

  max_compress_len(n)            reference src/compress.rs:42-53
  decompress_len(buf)            reference src/decompress.rs:30-35
  Encoder().compress / compress_vec   reference src/compress.rs:99-169
  Decoder().decompress / decompress_vec  reference src/decompress.rs:75-110

plus the batched, device-resident form (`compress_batch`, `decompress_batch`)
that takes torch CUDA tensors -- torch is only the owner of device memory.
All compute happens in the HIP kernels; without a GPU these raise
`DeviceError`.
"""
import ctypes as C

from . import _lib
from .error import DeviceError, Error

MAX_INPUT_SIZE = 0xFFFFFFFF  # reference src/lib.rs:93
MAX_BLOCK_SIZE = 1 << 16     # reference src/lib.rs:97


def _raise(ctx, rc, err=None):
    if rc >= 100:
        msg = None
        if ctx is not None and ctx._h:
            msg = _lib.of(ctx).snapmi_last_error(ctx._h).decode()
        raise DeviceError(rc, message=msg or "no usable HIP device")
    if err is not None and err.kind == rc:
        raise Error(rc, err.a, err.b, err.c)
    raise Error(rc)


class TestOnlyOption(RuntimeError):
    """set_test_option on a context of the product library (the knobs of
    include/snapmi_test.h exist in libsnapmi_test.so only)."""
    __test__ = False


class Context:
    """snapmi_ctx: one HIP device + stream + device scratch.  `lib`: the
    loaded library that makes it (default: the process's, _lib.load())."""

    def __init__(self, device=0, stream=None, lib=None):
        self._h = None
        L = self._L = lib or _lib.load()
        h = C.c_void_p()
        rc = L.snapmi_ctx_create(int(device), stream, C.byref(h))
        if rc != 0:
            raise DeviceError(rc, message=f"snapmi_ctx_create({device}) failed"
                              ": no usable HIP device (no CPU fallback)")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            self._L.snapmi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, name, value):
        """snapmi_ctx_set_option: tuning knobs (never change results)."""
        rc = self._L.snapmi_ctx_set_option(self._h, name.encode(),
                                               int(value))
        if rc:
            _raise(self, rc)

    def set_test_option(self, name, value):
        """snapmi_ctx_set_test_option (include/snapmi_test.h): knobs of the
        test suite and the experiment drivers."""
        L = self._L
        if not hasattr(L, "snapmi_ctx_set_test_option"):
            raise TestOnlyOption(
                "set_test_option: this is the product library; the test "
                "knobs live in libsnapmi_test.so (SNAPMI_TESTING=1)")
        rc = L.snapmi_ctx_set_test_option(self._h, name.encode(),
                                          int(value))
        if rc:
            _raise(self, rc)

    def prepare(self, blocks, top_of_memory=False):
        """snapmi_ctx_prepare: the lane tables of a batch of `blocks` 64 KiB
        blocks now; top_of_memory: SNAPMI_PREPARE_TOP_OF_MEMORY (holds the
        whole device for a moment - see include/snapmi.h)."""
        rc = self._L.snapmi_ctx_prepare(self._h, int(blocks),
                                        1 if top_of_memory else 0)
        if rc:
            _raise(self, rc)

    def info(self, name):
        """snapmi_ctx_get_info: "scratch_bytes", "token_scratch_bytes",
        "token_pool_pages", "token_pool_pct_now", "token_pages_asked",
        "token_blocks_spilled" (include/snapmi.h)."""
        v = C.c_int64(0)
        rc = self._L.snapmi_ctx_get_info(self._h, name.encode(), C.byref(v))
        if rc:
            _raise(self, rc)
        return v.value

    def table_probe_log(self):
        """snapmi_table_probe_log: what the last placement of the lane tables
        probed, held, kept and took."""
        return self._L.snapmi_table_probe_log(self._h).decode()

    @property
    def stream(self):
        return self._L.snapmi_ctx_stream(self._h)

    def synchronize(self):
        rc = self._L.snapmi_ctx_synchronize(self._h)
        if rc:
            _raise(self, rc)

    def last_kernel(self):
        """snapmi_last_kernel: the dominant kernel of the last batch call."""
        return self._L.snapmi_last_kernel(self._h).decode()

    def last_timing(self):
        t = _lib.SnapmiTiming()
        rc = self._L.snapmi_last_timing(self._h, C.byref(t))
        if rc:
            _raise(self, rc)
        return {"plan_ms": t.plan_ms, "codec_ms": t.codec_ms,
                "compact_ms": t.compact_ms, "total_ms": t.total_ms,
                "codec_launches": t.codec_launches,
                "dominant_ms": t.dominant_ms}


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def max_compress_len(input_len):
    return _lib.load().snapmi_max_compress_len(int(input_len))


def decompress_len(data):
    data = bytes(data)
    n = C.c_size_t(0)
    err = _lib.SnapmiError()
    rc = _lib.load().snapmi_decompress_len(data, len(data), C.byref(n),
                                           C.byref(err))
    if rc:
        _raise(None, rc, err)
    return n.value


class Encoder:
    """snap::raw::Encoder.  Owns a context (device scratch + stream), like the
    reference's encoder owns its hash tables; reuse it."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def compress(self, data, output):
        """Compress into the writable buffer `output`; returns bytes written.
        `output` must hold max_compress_len(len(data)) bytes."""
        data = bytes(data)
        out = (C.c_char * len(output)).from_buffer(output)
        n = C.c_size_t(0)
        err = _lib.SnapmiError()
        rc = _lib.of(self.ctx).snapmi_raw_compress(self.ctx._h, data, len(data),
                                             out, len(output), C.byref(n),
                                             C.byref(err))
        if rc:
            _raise(self.ctx, rc, err)
        return n.value

    def compress_vec(self, data):
        cap = max_compress_len(len(data))
        if cap == 0:
            raise Error(1, len(data), MAX_INPUT_SIZE)
        buf = bytearray(cap)
        n = self.compress(data, buf)
        return bytes(buf[:n])


class Decoder:
    """snap::raw::Decoder (stateless in the reference)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def decompress(self, data, output):
        data = bytes(data)
        out = (C.c_char * max(len(output), 1)).from_buffer(
            output if len(output) else bytearray(1))
        n = C.c_size_t(0)
        err = _lib.SnapmiError()
        rc = _lib.of(self.ctx).snapmi_raw_decompress(self.ctx._h, data, len(data),
                                               out, len(output), C.byref(n),
                                               C.byref(err))
        if rc:
            _raise(self.ctx, rc, err)
        return n.value

    def decompress_vec(self, data):
        buf = bytearray(decompress_len(data))
        n = self.decompress(data, buf)
        return bytes(buf[:n])


# ---------------------------------------------------------------------
# batched, device-resident (torch tensors own the HBM)
# ---------------------------------------------------------------------
def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def compress_batch(ctx, in_ptrs, in_lens, out_ptrs, out_caps, out_lens,
                   errs=None, host_in_lens=None):
    """snapmi_compress_batch.  in_ptrs/out_ptrs: int64 CUDA tensors of device
    addresses; in_lens/out_caps/out_lens: uint64-as-int64 CUDA tensors;
    errs: optional uint8 CUDA tensor of 32*n bytes; host_in_lens: optional
    CPU int64 tensor (or None -> fetched from the device)."""
    n = in_ptrs.numel()
    h = None
    if host_in_lens is not None:
        h = C.c_void_p(host_in_lens.data_ptr())
    rc = _lib.of(ctx).snapmi_compress_batch(
        ctx._h, _ptr(in_ptrs), _ptr(in_lens), h, _ptr(out_ptrs),
        _ptr(out_caps), _ptr(out_lens), _ptr(errs), n)
    if rc:
        _raise(ctx, rc)


def decompress_batch(ctx, in_ptrs, in_lens, out_ptrs, out_caps, out_lens,
                     errs=None):
    n = in_ptrs.numel()
    rc = _lib.of(ctx).snapmi_decompress_batch(
        ctx._h, _ptr(in_ptrs), _ptr(in_lens), _ptr(out_ptrs), _ptr(out_caps),
        _ptr(out_lens), _ptr(errs), n)
    if rc:
        _raise(ctx, rc)


def decompress_stream(ctx, d_in, n_in, d_out, out_len, err):
    """ONE long raw stream (uint8 CUDA tensor d_in[:n_in]) decoded by many
    wavefronts into d_out; out_len: int64[1], err: uint8[32] CUDA tensors.
    Same results and errors as decompress_batch with one stream."""
    rc = _lib.of(ctx).snapmi_decompress_stream(
        ctx._h, _ptr(d_in), int(n_in), _ptr(d_out), d_out.numel(),
        _ptr(out_len), _ptr(err))
    if rc:
        _raise(ctx, rc)


def stream_decode_path(ctx):
    """0 = the last decompress_stream ran as pieces on many wavefronts, 1 =
    sequential path, -1 = none yet."""
    return _lib.of(ctx).snapmi_stream_decode_path(ctx._h)


def decompress_len_batch(ctx, in_ptrs, in_lens, out_lens, errs=None):
    n = in_ptrs.numel()
    rc = _lib.of(ctx).snapmi_decompress_len_batch(
        ctx._h, _ptr(in_ptrs), _ptr(in_lens), _ptr(out_lens), _ptr(errs), n)
    if rc:
        _raise(ctx, rc)
