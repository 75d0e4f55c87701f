"""Device-resident batches of raw streams (torch tensors own the HBM).

Plumbing only: lays streams out back to back in one uint8 tensor, builds the
pointer / length arrays the C ABI wants, and calls the batched entry points.
"""
import numpy as np
import torch

from . import raw

ALIGN = 16


def _align(x, a=ALIGN):
    return (x + a - 1) // a * a


class StreamBatch:
    """n byte streams in one device slab: data[offsets[i] : +lens[i]]."""

    def __init__(self, data, offsets, lens):
        self.data = data                      # uint8 CUDA tensor
        self.offsets = np.asarray(offsets, dtype=np.int64)
        self.lens = np.asarray(lens, dtype=np.int64)
        dev = data.device
        self.h_lens = torch.from_numpy(self.lens.copy())
        self.d_lens = self.h_lens.to(dev)
        self.d_ptrs = (torch.from_numpy(self.offsets).to(dev)
                       + data.data_ptr())

    @property
    def n(self):
        return len(self.lens)

    @classmethod
    def from_bytes(cls, streams, device="cuda:0"):
        lens = [len(s) for s in streams]
        offs, pos = [], 0
        for n in lens:
            offs.append(pos)
            pos += _align(max(n, 1))
        host = np.zeros(max(pos, ALIGN), dtype=np.uint8)
        for s, o in zip(streams, offs):
            host[o:o + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
        return cls(torch.from_numpy(host).to(device), offs, lens)

    @classmethod
    def empty(cls, caps, device="cuda:0"):
        """Output slab with capacity caps[i] for stream i."""
        offs, pos = [], 0
        for c in caps:
            offs.append(pos)
            pos += _align(max(int(c), 1))
        data = torch.empty(max(pos, ALIGN), dtype=torch.uint8, device=device)
        return cls(data, offs, [int(c) for c in caps])

    def stream_bytes(self, i, n=None):
        n = int(self.lens[i]) if n is None else int(n)
        o = int(self.offsets[i])
        return self.data[o:o + n].cpu().numpy().tobytes()


def read_errors(errs):
    """uint8 tensor [32*n] of snapmi_error -> list of (kind, a, b, c)."""
    raw_ = errs.cpu().numpy().tobytes()
    rec = np.frombuffer(raw_, dtype=np.dtype(
        [("kind", "<i4"), ("r", "<u4"), ("a", "<u8"), ("b", "<u8"),
         ("c", "<u8")]))
    return [(int(r["kind"]), int(r["a"]), int(r["b"]), int(r["c"]))
            for r in rec]


def compress(ctx, src: StreamBatch, check_caps=True):
    """Compress every stream of `src`; returns (dst batch, out_lens, errs)."""
    caps = [raw.max_compress_len(int(n)) or 32 for n in src.lens]
    dst = StreamBatch.empty(caps, src.data.device)
    dev = src.data.device
    out_lens = torch.zeros(src.n, dtype=torch.int64, device=dev)
    errs = torch.zeros(32 * src.n, dtype=torch.uint8, device=dev)
    raw.compress_batch(ctx, src.d_ptrs, src.d_lens, dst.d_ptrs,
                       dst.d_lens if check_caps else None, out_lens, errs,
                       host_in_lens=src.h_lens)
    ctx.synchronize()
    return dst, out_lens.cpu().numpy(), read_errors(errs)


def decompress(ctx, src: StreamBatch, caps=None):
    """Decompress every stream; caps default to the header lengths."""
    dev = src.data.device
    if caps is None:
        lens = torch.zeros(src.n, dtype=torch.int64, device=dev)
        raw.decompress_len_batch(ctx, src.d_ptrs, src.d_lens, lens)
        ctx.synchronize()
        caps = lens.cpu().numpy()
    dst = StreamBatch.empty(caps, dev)
    out_lens = torch.zeros(src.n, dtype=torch.int64, device=dev)
    errs = torch.zeros(32 * src.n, dtype=torch.uint8, device=dev)
    raw.decompress_batch(ctx, src.d_ptrs, src.d_lens, dst.d_ptrs, dst.d_lens,
                         out_lens, errs)
    ctx.synchronize()
    return dst, out_lens.cpu().numpy(), read_errors(errs)
