"""Multi-GPU sharding of a batch of independent streams.

Raw streams (and frame chunks) are independent, so the path shards with no
exchange during compute: every rank takes a contiguous range of streams,
balanced by uncompressed bytes.  The only collective is the optional gather
of the (variable-length) results, used to assemble one framed stream.

Works with any torch.distributed backend: `nccl` (= RCCL over xGMI) on GPUs,
`gloo` on CPU (tests).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def partition_by_bytes(lens, world):
    """Contiguous ranges [start, end) per rank, balanced by sum(lens)."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    if world <= 1:
        return [(0, n)]
    csum = np.concatenate([[0], np.cumsum(lens)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        # first index whose prefix reaches the target, kept monotone
        i = int(np.searchsorted(csum, target, side="left"))
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def exchange_sizes(local_bytes, device="cpu", group=None):
    """all_gather of one int64 per rank -> (sizes[world], offsets[world])."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(local_bytes)], dtype=torch.int64, device=device)
    out = [torch.zeros(1, dtype=torch.int64, device=device)
           for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    sizes = [int(t.item()) for t in out]
    offs = [0]
    for s in sizes[:-1]:
        offs.append(offs[-1] + s)
    return sizes, offs


def gatherv(local, dst=0, group=None):
    """Variable-length gather of 1-D uint8 tensors to rank `dst`.

    RCCL has no gatherv: sizes are exchanged first (one all_gather of a u64
    per rank), then every rank's slab goes straight into the root's buffer at
    its prefix offset as ONE group of point-to-point operations
    (batch_isend_irecv = ncclGroupStart ... ncclSend / ncclRecv ...
    ncclGroupEnd): on xGMI the root's 7 inbound links then receive
    in parallel, which a ring (per-link bound) would not.  Returns the
    concatenation on `dst`, None elsewhere."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # gloo moves host memory: stage (CPU tests, and bench.py's
        # --oversubscribe proof run of N ranks on one GPU)
        whole = gatherv(local.cpu(), dst, group)
        return None if whole is None else whole.to(local.device)
    sizes, offs = exchange_sizes(local.numel(), local.device, group)
    ops, out = [], None
    if rank == dst:
        out = torch.empty(sum(sizes), dtype=torch.uint8, device=local.device)
        out[offs[dst]:offs[dst] + sizes[dst]] = local
        for r in range(world):
            if r != dst and sizes[r]:
                ops.append(dist.P2POp(dist.irecv,
                                      out[offs[r]:offs[r] + sizes[r]], r,
                                      group))
    elif local.numel():
        ops.append(dist.P2POp(dist.isend, local.contiguous(), dst, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return out


class Comm:
    """snapmi_comm: an RCCL communicator behind the C ABI (include/snapmi.h
    section 5) - what a host without torch.distributed uses.  `ident` is the
    128-byte rendezvous id of snapmi_comm_unique_id(), made on one rank and
    handed to the others by any means."""

    def __init__(self, ctx, ident, rank, world):
        from . import _lib, raw
        self._L, self._raw, self.ctx = _lib.of(ctx), raw, ctx
        self.rank, self.world = rank, world
        h = C.c_void_p()
        rc = self._L.snapmi_comm_init(ctx._h, bytes(ident), rank, world,
                                      C.byref(h))
        if rc:
            raw._raise(ctx, rc)
        self._h = h

    @classmethod
    def wrap(cls, ctx, nccl_comm, rank, world):
        """snapmi_comm_wrap: around an ncclComm_t the host already has (its
        address as an int); the communicator stays the host's."""
        from . import _lib, raw
        self = cls.__new__(cls)
        self._L, self._raw, self.ctx = _lib.of(ctx), raw, ctx
        self.rank, self.world = rank, world
        h = C.c_void_p()
        rc = self._L.snapmi_comm_wrap(ctx._h, C.c_void_p(nccl_comm), rank,
                                      world, C.byref(h))
        if rc:
            raw._raise(ctx, rc)
        self._h = h
        return self

    @staticmethod
    def unique_id():
        from . import _lib
        buf = C.create_string_buffer(128)
        rc = _lib.load().snapmi_comm_unique_id(buf)
        if rc:
            raise RuntimeError(f"snapmi_comm_unique_id failed ({rc}): "
                               "is librccl reachable?")
        return buf.raw

    def gatherv(self, local, dst=0, cap=None):
        """snapmi_gatherv of a 1-D uint8 CUDA tensor; returns (the
        concatenation on `dst` / None elsewhere, sizes per rank)."""
        sizes = (C.c_uint64 * self.world)()
        total = C.c_uint64(0)
        n = local.numel()
        if cap is None:
            # sizes are not known before the call: the root passes a bound
            raise ValueError("gatherv needs the root's capacity (bytes)")
        out = (torch.empty(max(cap, 16), dtype=torch.uint8,
                           device=local.device)
               if self.rank == dst else None)
        rc = self._L.snapmi_gatherv(
            self.ctx._h, self._h, dst,
            C.c_void_p(local.data_ptr()) if n else None, n,
            C.c_void_p(out.data_ptr()) if out is not None else None,
            cap if out is not None else 0, sizes, C.byref(total))
        if rc:
            self._raw._raise(self.ctx, rc)
        return (out[:total.value] if out is not None else None), list(sizes)

    def close(self):
        if self._h:
            self._L.snapmi_comm_destroy(self._h)
            self._h = None
