"""One-shot all-reduce of small fp32 tensors through peer-mapped device memory.

The latency path next to RCCL: every rank pushes its message straight into an inbox inside each peer's
memory (all xGMI links of the GPU at once), raises a flag there, waits for the peers' flags on its own
inbox and sums the slots - one kernel launch of one workgroup per rank, no host round trip
(``csrc/kernels/p2p.hip``).  Meant for what the reference reduces with one NCCL call per scalar and a
``barrier()`` in front (/root/reference/2.distributed.py:71-75,219-223): metric vectors, SyncBatchNorm
statistics, BatchNorm buffers - messages of a few bytes to a few hundred KB.  Bandwidth-sized tensors
belong to :class:`~pytorch_distributed_amd.comm.rccl.RcclCommunicator`.

The regions are fine-grained device allocations shared with HIP IPC handles, so the ranks must be
processes on one node; the handles travel through a rendezvous store.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ..ops import _lib

_SIGS = {
    "pda_p2p_alloc": [ctypes.c_longlong, ctypes.POINTER(ctypes.c_void_p)],
    "pda_p2p_free": [ctypes.c_void_p],
    "pda_p2p_handle_bytes": [],
    "pda_p2p_get_handle": [ctypes.c_void_p, ctypes.c_char_p],
    "pda_p2p_open_handle": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)],
    "pda_p2p_close_handle": [ctypes.c_void_p],
    "pda_p2p_allreduce": [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_float, ctypes.c_void_p,
                          ctypes.c_int, ctypes.c_void_p],
}
MAX_WORLD = 16


def _bind(l):
    if not getattr(l, "_pda_p2p_bound", False):
        for name, sig in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        l._pda_p2p_bound = True
    return l


class P2PAllReduce:
    """``store``: anything with ``set(key, bytes)`` / ``get(key) -> bytes`` (blocking until the key exists);
    without one the handles are exchanged over ``group`` (a ``torch.distributed`` group, default WORLD).

    ``max_elems`` bounds the message length (fp32 elements); the region holds ``2 * world`` slots of
    that size.  ``timeout_ms`` bounds the in-kernel wait for the peers: past it the kernel gives up,
    raises the error flag (see :meth:`check`) and returns, so a dead peer cannot wedge the GPU.
    """

    def __init__(self, rank: int, world_size: int, store=None, device: Optional[torch.device] = None,
                 max_elems: int = 65536, key: str = "pda/p2p0", timeout_ms: int = 2000, group=None):
        if not 1 <= world_size <= MAX_WORLD:
            raise ValueError(f"P2PAllReduce supports 1..{MAX_WORLD} ranks on one node, got {world_size}")
        self._l = _bind(_lib.lib())
        self.rank, self.world_size = int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_elems = int(max_elems)
        self.timeout_ms = int(timeout_ms)
        self._seq = 0
        self._store, self._key, self._group = store, key, group
        self._opened = []
        self._own = ctypes.c_void_p()
        nbytes = 2 * self.world_size * self.max_elems * 4 + 2 * self.world_size * 4
        nbytes = (nbytes + 4095) // 4096 * 4096
        with torch.cuda.device(self.device):
            rc = self._l.pda_p2p_alloc(nbytes, ctypes.byref(self._own))
            if rc != 0:
                raise RuntimeError(f"P2PAllReduce: fine-grained allocation of {nbytes} bytes failed (hip error {rc})")
            hb = self._l.pda_p2p_handle_bytes()
            buf = ctypes.create_string_buffer(hb)
            handles = [None] * self.world_size
            if self.world_size > 1:
                rc = self._l.pda_p2p_get_handle(self._own, buf)
                if rc != 0:
                    raise RuntimeError(f"P2PAllReduce: hipIpcGetMemHandle failed (hip error {rc})")
                if store is not None:
                    store.set(f"{key}/h{self.rank}", bytes(buf.raw))
                else:
                    import torch.distributed as dist
                    dist.all_gather_object(handles, bytes(buf.raw), group=group)
            bases = (ctypes.c_void_p * MAX_WORLD)()
            for p in range(self.world_size):
                if p == self.rank:
                    bases[p] = self._own.value
                    continue
                raw = bytes(store.get(f"{key}/h{p}") if store is not None else handles[p])[:hb]
                ptr = ctypes.c_void_p()
                rc = self._l.pda_p2p_open_handle(raw, ctypes.byref(ptr))
                if rc != 0:
                    raise RuntimeError(f"P2PAllReduce: hipIpcOpenMemHandle for rank {p} failed (hip error {rc})")
                self._opened.append(ptr)
                bases[p] = ptr.value
            for p in range(self.world_size, MAX_WORLD):
                bases[p] = self._own.value
            self._bases = bases
            self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
            if self.world_size > 1:
                # nobody may push before every rank has mapped every region and nobody may free early:
                # a second round of keys is the barrier
                self._sync("ready")

    def all_reduce(self, t: torch.Tensor, op: str = "sum", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """In place (or into ``out``) on the current stream; ``op`` is ``"sum"`` or ``"avg"``."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("P2PAllReduce.all_reduce takes a contiguous fp32 device tensor")
        if t.numel() > self.max_elems:
            raise ValueError(f"message of {t.numel()} elements exceeds max_elems={self.max_elems}")
        if op not in ("sum", "avg"):
            raise ValueError(f"unsupported op {op!r}")
        out = t if out is None else out
        if out.dtype != torch.float32 or out.numel() != t.numel() or not out.is_contiguous():
            raise TypeError("out must be a contiguous fp32 tensor of the same length")
        self._seq += 1
        if self._seq >= 1 << 32:
            self._seq = 1 + (self._seq & 1)   # keep the parity sequence, skip 0 (the flags' initial value)
        scale = 1.0 / self.world_size if op == "avg" else 1.0
        rc = self._l.pda_p2p_allreduce(self._bases, t.data_ptr(), out.data_ptr(), t.numel(), self.rank,
                                       self.world_size, self.max_elems, self._seq, scale, self._err.data_ptr(),
                                       self.timeout_ms, _lib.stream())
        _lib.check(rc, "pda_p2p_allreduce")
        return out

    def _sync(self, tag: str) -> None:
        if self._store is not None:
            self._store.set(f"{self._key}/{tag}{self.rank}", b"1")
            for p in range(self.world_size):
                self._store.get(f"{self._key}/{tag}{p}")
        else:
            import torch.distributed as dist
            dist.barrier(group=self._group)

    def check(self) -> None:
        """Synchronises and raises if any earlier call gave up waiting for a peer."""
        if int(self._err.item()) != 0:
            raise RuntimeError("P2PAllReduce: a peer did not arrive within the timeout; results are invalid")

    def close(self) -> None:
        """Collective: waits (through the store) until every rank has finished using the regions, then
        unmaps and frees them.  Not called from ``__del__``: freeing a region a peer may still push into
        would fault that peer, so an object that is merely dropped keeps its memory until the process ends."""
        if getattr(self, "_own", None) is None:
            return
        torch.cuda.synchronize(self.device)
        if self.world_size > 1:
            self._sync("bye")
        for ptr in self._opened:
            self._l.pda_p2p_close_handle(ptr)
        self._opened = []
        if self._own:
            self._l.pda_p2p_free(self._own)
        self._own = None


class _Done:
    def wait(self):
        return True

    def is_completed(self):
        return True


class P2PCollectives:
    """The collectives interface ``DistributedDataParallel(comm=...)`` expects, on top of
    :class:`P2PAllReduce`: floating-point device tensors are reduced in chunks of ``max_elems`` by the
    peer-memory kernel (in stream order, staged through fp32 for 16-bit tensors); everything else (integer
    tensors, CPU tensors, min/max/prod) is staged through the host over ``group``.  One workgroup per rank
    moves the data, so this is a correctness backend for single-node jobs without RCCL (and what lets the
    multi-rank DDP path be tested with several processes on one GPU), not a bandwidth path."""

    def __init__(self, rank: int, world_size: int, store=None, group=None, device=None, max_elems: int = 1 << 20,
                 key: str = "pda/p2pc", timeout_ms: int = 5000):
        import torch.distributed as dist
        self._dist, self.group = dist, group
        self.rank, self.world_size = int(rank), int(world_size)
        self.backend = "p2p"
        self._ar = P2PAllReduce(rank, world_size, store=store, device=device, max_elems=max_elems, key=key,
                                timeout_ms=timeout_ms, group=group)

    def _host(self, t: torch.Tensor, fn) -> None:
        c = t.detach().cpu()
        fn(c)
        t.copy_(c)

    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        if t.is_cuda and t.is_floating_point() and op in ("sum", "avg") and t.is_contiguous():
            flat = t.view(-1)
            n, m = flat.numel(), self._ar.max_elems
            for off in range(0, n, m):
                piece = flat[off:off + m]
                if piece.dtype == torch.float32:
                    self._ar.all_reduce(piece, op)
                else:
                    f = piece.float()
                    self._ar.all_reduce(f, op)
                    piece.copy_(f)
        else:
            d = self._dist
            rop = {"sum": d.ReduceOp.SUM, "avg": d.ReduceOp.SUM, "prod": d.ReduceOp.PRODUCT, "min": d.ReduceOp.MIN,
                   "max": d.ReduceOp.MAX}[op]

            def run(c):
                d.all_reduce(c, op=rop, group=self.group)
                if op == "avg":
                    c /= self.world_size
            self._host(t, run)
        return _Done() if async_op else None

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if t.is_cuda and t.is_floating_point() and t.is_contiguous():
            if self.rank != src:
                t.zero_()
            self.all_reduce(t, "sum")   # everybody but src contributes zeros
        else:
            self._host(t, lambda c: self._dist.broadcast(c, src=src, group=self.group))
        return _Done() if async_op else None

    def barrier(self):
        self._dist.barrier(group=self.group)

    @property
    def device_collectives(self) -> bool:
        return True

    def check(self) -> None:
        self._ar.check()

    def close(self) -> None:
        self._ar.close()
