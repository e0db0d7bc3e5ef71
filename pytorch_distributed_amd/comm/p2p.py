"""All-reduce through peer-mapped device memory: a one-shot latency kernel and a two-shot bandwidth kernel.

The latency path next to RCCL: every rank pushes its message straight into an inbox inside each peer's
memory (all xGMI links of the GPU at once), raises a flag there, waits for the peers' flags on its own
inbox and sums the slots - one kernel launch of one workgroup per rank, no host round trip
(``csrc/kernels/p2p.hip``).  Meant for what the reference reduces with one NCCL call per scalar and a
``barrier()`` in front (/root/reference/2.distributed.py:71-75,219-223): metric vectors, SyncBatchNorm
statistics, BatchNorm buffers - messages of a few bytes to a few hundred KB.

:class:`P2PTwoShot` is the bandwidth path for gradient buckets (the NCCL all-reduce behind
``loss.backward()``, /root/reference/2.distributed.py:231): direct reduce-scatter + direct all-gather over
the mesh - every rank pushes shard j to rank j, reduces its own shard in fp32 and pushes the result to
everybody, 2 hops with all 7 xGMI links of a GPU busy instead of a ring's 14 hops through one link.
``nblk`` workgroups (default 16) so that it can run next to the backward kernels.
:class:`P2PCollectives` puts both behind the collectives interface of ``DistributedDataParallel``.

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
                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                          ctypes.c_int, ctypes.c_void_p],
    "pda_p2p_allreduce_twoshot": [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p,
                                  ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
}
MAX_WORLD = 16


def _bind(l):
    if not getattr(l, "_pda_p2p_bound", False):
        for name, sig in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        l.pda_p2p2_region_bytes.argtypes = [ctypes.c_longlong]
        l.pda_p2p2_region_bytes.restype = ctypes.c_longlong
        l._pda_p2p_bound = True
    return l


class _PeerRegion:
    """One fine-grained device allocation per rank, mapped into every peer with HIP IPC handles that travel
    through ``store`` (``set`` / blocking ``get``) or, without one, over a ``torch.distributed`` group."""

    def __init__(self, rank: int, world_size: int, nbytes: int, store, device, key: str, group, what: str):
        if not 1 <= world_size <= MAX_WORLD:
            raise ValueError(f"{what} supports 1..{MAX_WORLD} ranks on one node, got {world_size}")
        self._l = _bind(_lib.lib())
        self.rank, self.world_size = int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._store, self._key, self._group = store, key, group
        self._opened = []
        self._own = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._l.pda_p2p_alloc(nbytes, ctypes.byref(self._own))
            if rc != 0:
                raise RuntimeError(f"{what}: fine-grained allocation of {nbytes} bytes failed (hip error {rc})")
            hb = self._l.pda_p2p_handle_bytes()
            buf = ctypes.create_string_buffer(hb)
            handles = [None] * self.world_size
            if self.world_size > 1:
                rc = self._l.pda_p2p_get_handle(self._own, buf)
                if rc != 0:
                    raise RuntimeError(f"{what}: hipIpcGetMemHandle failed (hip error {rc})")
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
                    raise RuntimeError(f"{what}: hipIpcOpenMemHandle for rank {p} failed (hip error {rc})")
                self._opened.append(ptr)
                bases[p] = ptr.value
            for p in range(self.world_size, MAX_WORLD):
                bases[p] = self._own.value
            self._bases = bases
            self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
            # {sequence number of the last call, workgroups of the running call that are done}: device memory, read
            # and advanced by the kernels themselves, so a call recorded into a hipGraph stays in step on replay
            self._seq_state = torch.zeros(2, dtype=torch.int32, device=self.device)
            if self.world_size > 1:
                # nobody may push before every rank has mapped every region and nobody may free early:
                # a second round of keys is the barrier
                self._sync("ready")

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
            raise RuntimeError(f"{type(self).__name__}: a peer did not arrive within the timeout; results are invalid")

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


class P2PTwoShot(_PeerRegion):
    """Two-shot (direct reduce-scatter + all-gather) all-reduce of fp32 / fp16 / bf16 device tensors of any
    length: tensors larger than ``cap_bytes`` go through the region in pieces.  fp32 accumulation, result
    rounded once to the tensor's dtype; in stream order on the current stream."""

    def __init__(self, rank: int, world_size: int, store=None, device: Optional[torch.device] = None,
                 cap_bytes: int = 32 << 20, key: str = "pda/p2p2", timeout_ms: int = 5000, group=None, nblk: int = 16):
        self.cap_bytes = int(cap_bytes) // 4096 * 4096
        self.timeout_ms, self.nblk = int(timeout_ms), int(nblk)
        l = _bind(_lib.lib())
        super().__init__(rank, world_size, int(l.pda_p2p2_region_bytes(self.cap_bytes)), store, device, key, group,
                         "P2PTwoShot")

    def max_elems(self, dtype: torch.dtype) -> int:
        esz = torch.empty((), dtype=dtype).element_size()
        return self.cap_bytes // esz - self.world_size * (16 // esz)   # room for the rounded-up shards

    def all_reduce(self, t: torch.Tensor, op: str = "sum", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if t.dtype not in (torch.float32, torch.float16, torch.bfloat16) or not t.is_cuda:
            raise TypeError("P2PTwoShot.all_reduce takes an fp32 / fp16 / bf16 device tensor")
        if op not in ("sum", "avg"):
            raise ValueError(f"unsupported op {op!r}")
        out = t if out is None else out
        src, dst = torch.as_strided(t, (t.numel(),), (1,)), torch.as_strided(out, (out.numel(),), (1,))
        if dst.dtype != src.dtype or dst.numel() != src.numel():
            raise TypeError("out must have the dtype and length of the input")
        scale = 1.0 / self.world_size if op == "avg" else 1.0
        m = self.max_elems(t.dtype)
        vec = 16 // t.element_size()
        m -= m % vec                                        # pieces stay 16-byte aligned
        for off in range(0, src.numel(), m):
            a, b = src[off:off + m], dst[off:off + m]
            rc = self._l.pda_p2p_allreduce_twoshot(self._bases, _lib.dt(t), a.data_ptr(), b.data_ptr(), a.numel(),
                                                   self.rank, self.world_size, self.cap_bytes,
                                                   self._seq_state.data_ptr(), scale,
                                                   self._err.data_ptr(), self.timeout_ms, self.nblk, _lib.stream())
            _lib.check(rc, "pda_p2p_allreduce_twoshot")
        return out


class P2PAllReduce(_PeerRegion):
    """``store``: anything with ``set(key, bytes)`` / ``get(key) -> bytes`` (blocking until the key exists);
    without one the handles are exchanged over ``group`` (a ``torch.distributed`` group, default WORLD).

    ``max_elems`` bounds the message length (fp32 elements); the region holds ``2 * world`` slots of
    that size.  ``timeout_ms`` bounds the in-kernel wait for the peers: past it the kernel gives up,
    raises the error flag (see :meth:`check`) and returns, so a dead peer cannot wedge the GPU.
    """

    def __init__(self, rank: int, world_size: int, store=None, device: Optional[torch.device] = None,
                 max_elems: int = 65536, key: str = "pda/p2p0", timeout_ms: int = 2000, group=None):
        self.max_elems = int(max_elems)
        self.timeout_ms = int(timeout_ms)
        nbytes = 2 * int(world_size) * self.max_elems * 4 + 2 * int(world_size) * 4
        nbytes = (nbytes + 4095) // 4096 * 4096
        super().__init__(rank, world_size, nbytes, store, device, key, group, "P2PAllReduce")

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
        scale = 1.0 / self.world_size if op == "avg" else 1.0
        rc = self._l.pda_p2p_allreduce(self._bases, t.data_ptr(), out.data_ptr(), t.numel(), self.rank,
                                       self.world_size, self.max_elems, self._seq_state.data_ptr(), scale,
                                       self._err.data_ptr(),
                                       self.timeout_ms, _lib.stream())
        _lib.check(rc, "pda_p2p_allreduce")
        return out


class _Done:
    def wait(self):
        return True

    def is_completed(self):
        return True


class _StreamWork:
    """Completion of a collective that ran on the backend's own stream: ``wait()`` orders the current
    stream after it (device side)."""
    __slots__ = ("event", "keep")

    def __init__(self, event, keep):
        self.event, self.keep = event, keep

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self.event = self.keep = None
        return True

    def is_completed(self):
        return self.event is None or self.event.query()


class P2PCollectives:
    """The collectives interface ``DistributedDataParallel(comm=...)`` expects, on the peer-memory kernels:
    floating-point device tensors go through the two-shot kernel in their own dtype (fp32 accumulation),
    messages of at most ``small_elems`` fp32 elements through the one-shot latency kernel; everything else
    (integer tensors, CPU tensors, min/max/prod) is staged through the host over ``group``.
    ``overlap=True``: asynchronous all-reduces run on an own stream (same priority as the compute streams,
    see ``comm/rccl.py``) and overlap with the backward kernels enqueued after them.  Single node, one
    process per GPU; also what lets the multi-rank DDP path be tested with several processes on one GPU."""

    def __init__(self, rank: int, world_size: int, store=None, group=None, device=None, max_elems: int = 1 << 20,
                 key: str = "pda/p2pc", timeout_ms: int = 5000, cap_bytes: int = 32 << 20, nblk: int = 16,
                 small_elems: int = 16384, overlap: bool = False):
        import torch.distributed as dist
        self._dist, self.group = dist, group
        self.rank, self.world_size = int(rank), int(world_size)
        self.backend = "p2p-twoshot"
        self.small_elems = int(small_elems)
        self._ar = P2PAllReduce(rank, world_size, store=store, device=device, max_elems=self.small_elems,
                                key=key, timeout_ms=timeout_ms, group=group)
        self._ts = P2PTwoShot(rank, world_size, store=store, device=device, cap_bytes=cap_bytes, key=key + "2",
                              timeout_ms=timeout_ms, group=group, nblk=nblk)
        self.device = self._ts.device
        self.overlap = bool(overlap)
        self.stream = torch.cuda.Stream(device=self.device) if self.overlap else None

    def _host(self, t: torch.Tensor, fn) -> None:
        c = t.detach().cpu()
        fn(c)
        t.copy_(c)

    def _device_all_reduce(self, t: torch.Tensor, op: str) -> None:
        if t.dtype == torch.float32 and t.numel() <= self.small_elems and t.is_contiguous():
            self._ar.all_reduce(t, op)
        else:
            self._ts.all_reduce(t, op)

    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
        if t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16) and op in ("sum", "avg") and dense:
            if self.overlap and async_op:
                cur = torch.cuda.current_stream(self.device)
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    self._device_all_reduce(t, op)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                return _StreamWork(ev, t)
            self._device_all_reduce(t, op)
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
        if t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16) and t.is_contiguous():
            if self.rank != src:
                t.zero_()
            self._device_all_reduce(t, "sum")   # everybody but src contributes zeros
        elif t.is_cuda and not t.dtype.is_floating_point and not t.dtype.is_complex:
            # integer buffers (BatchNorm's num_batches_tracked) stay on the device too, so that a training step
            # can be recorded into a hipGraph: every byte becomes one fp32 value (exact), the all-reduce adds
            # zeros to it (exact), and the bytes are put back
            b = t.contiguous().reshape(-1).view(torch.uint8).to(torch.float32)
            if self.rank != src:
                b.zero_()
            self._device_all_reduce(b, "sum")
            t.copy_(b.to(torch.uint8).view(t.dtype).reshape(t.shape))
        else:
            self._host(t, lambda c: self._dist.broadcast(c, src=src, group=self.group))
        return _Done() if async_op else None

    def all_gather(self, out: torch.Tensor, t: torch.Tensor, async_op: bool = False):
        """Every rank's ``t`` into ``out`` (rank-major).  There is no dedicated gather kernel in this backend:
        floating-point tensors ride on the all-reduce (own slice + zeros), everything else goes through the host."""
        assert out.numel() == t.numel() * self.world_size and out.dtype == t.dtype
        if t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16) and out.is_contiguous():
            flat = out.view(-1)
            flat.zero_()
            flat[self.rank * t.numel():(self.rank + 1) * t.numel()].copy_(t.reshape(-1))
            self._device_all_reduce(flat, "sum")
        elif t.is_cuda and out.is_contiguous():
            # any other dtype (the fp64 dot products of the device Adasum, integer tensors): byte-exact on the
            # device - every byte travels as one fp32 value next to zeros - so that a caller on another thread (the
            # hvd engine) never issues a host collective on a group the main thread may be using as well
            nb = t.numel() * t.element_size()
            flat = torch.zeros(self.world_size * nb, device=t.device, dtype=torch.float32)
            flat[self.rank * nb:(self.rank + 1) * nb].copy_(t.contiguous().reshape(-1).view(torch.uint8))
            self._device_all_reduce(flat, "sum")
            out.copy_(flat.to(torch.uint8).view(out.dtype).reshape(out.shape))
        else:
            c = t.detach().cpu().contiguous()
            parts = [torch.empty_like(c) for _ in range(self.world_size)]
            self._dist.all_gather(parts, c, group=self.group)
            out.copy_(torch.cat([p.reshape(-1) for p in parts]).view_as(out))
        return _Done() if async_op else None

    def barrier(self):
        # drain the device, then meet on the host side of the group with a CPU tensor: a plain dist.barrier() on a
        # group that also has a device backend would make torch create an RCCL communicator of its own - which
        # never completes when the ranks share one GPU, the case this backend exists for
        torch.cuda.synchronize(self.device)
        self._dist.all_reduce(torch.zeros(1), group=self.group)

    @property
    def device_collectives(self) -> bool:
        return True

    def check(self) -> None:
        self._ar.check()
        self._ts.check()

    def close(self) -> None:
        self._ar.close()
        self._ts.close()
