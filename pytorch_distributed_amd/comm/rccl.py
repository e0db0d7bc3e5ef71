"""Collective back-ends behind DistributedDataParallel.

``TorchCollectives``  - ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on CPUs).
``RcclCommunicator``  - this package's own communicator (``csrc/kernels/rccl_comm.hip``): an RCCL
                        comm plus a dedicated high-priority HIP stream; the 128-byte unique id is
                        published by rank 0 through a rendezvous store (or any existing process
                        group), which is what `init_process_group('nccl')` does under the hood at
                        /root/reference/2.distributed.py:98.

Both expose the same small surface: ``rank``, ``world_size``, ``all_reduce(t, op, async_op)``,
``broadcast(t, src)``, ``all_gather(out, t)``, ``reduce_scatter(out, t, op)``, ``barrier()``;
asynchronous calls return a work object whose ``wait()`` orders the *current stream* after the
collective without blocking the host.  The communicator's stream discipline is the usual one: the
comm stream waits for the producing stream, runs the collective, records an event; consumers wait
for that event on the device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from ..ops import _lib

_OPS = {"sum": 0, "prod": 1, "min": 2, "max": 3, "avg": 4}
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.uint8: 3, torch.float64: 4, torch.int32: 5,
       torch.int64: 6}


class _DoneWork:
    """Work handle of a collective that had finished when the call returned."""

    def wait(self):
        return True

    def is_completed(self) -> bool:
        return True

    def synchronize(self, timeout_s: float = 0.0) -> bool:
        return True


class TorchCollectives:
    """Adapter over an initialised ``torch.distributed`` process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group if group is not None else dist.group.WORLD
        self.world_size = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.backend = dist.get_backend(self.group)

    def _global(self, r: int) -> int:
        d = self._dist
        return d.get_global_rank(self.group, r) if hasattr(d, "get_global_rank") else r

    def _host_staged(self, t: torch.Tensor) -> bool:
        """Device tensors on a group without device collectives (gloo) are staged through the host here, in
        the caller's stream order and synchronously: the group's own handling of device memory is not relied
        on (two ranks handing it device tensors from their engine threads never finished their first step)."""
        return t.is_cuda and not self.device_collectives

    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        d = self._dist
        if self._host_staged(t):
            h = t.detach().to("cpu", torch.float32 if t.dtype in (torch.float16, torch.bfloat16) else t.dtype)
            rop = {"sum": d.ReduceOp.SUM, "prod": d.ReduceOp.PRODUCT, "min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX,
                   "avg": d.ReduceOp.SUM}[op]
            d.all_reduce(h, op=rop, group=self.group)
            if op == "avg":
                h /= self.world_size
            t.copy_(h)
            return _DoneWork() if async_op else None
        rop = {"sum": d.ReduceOp.SUM, "prod": d.ReduceOp.PRODUCT, "min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX,
               "avg": d.ReduceOp.AVG}[op]
        return d.all_reduce(t, op=rop, group=self.group, async_op=async_op)

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if self._host_staged(t):
            h = t.detach().cpu()
            self._dist.broadcast(h, src=self._global(src), group=self.group)
            t.copy_(h)
            return _DoneWork() if async_op else None
        return self._dist.broadcast(t, src=self._global(src), group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, t: torch.Tensor, async_op: bool = False):
        if self._host_staged(t):
            ho = torch.empty(out.shape, dtype=out.dtype)
            self._dist.all_gather_into_tensor(ho, t.detach().cpu().contiguous(), group=self.group)
            out.copy_(ho)
            return _DoneWork() if async_op else None
        return self._dist.all_gather_into_tensor(out, t, group=self.group, async_op=async_op)

    def reduce_scatter(self, out: torch.Tensor, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        d = self._dist
        rop = {"sum": d.ReduceOp.SUM, "avg": d.ReduceOp.AVG, "min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX}[op]
        return d.reduce_scatter_tensor(out, t, op=rop, group=self.group, async_op=async_op)

    def barrier(self):
        self._dist.barrier(group=self.group)

    @property
    def device_collectives(self) -> bool:
        return "nccl" in str(self.backend)


class _RcclWork:
    """Completion of one collective = a ticket of the communicator's event ring (recorded on the comm
    stream right after the collective).  ``keep`` pins the operand tensors until the consumer stream has
    been ordered after the collective, so the caching allocator cannot hand their memory to another
    stream while RCCL still uses it."""
    __slots__ = ("comm", "ticket", "keep")

    def __init__(self, comm, ticket, keep):
        self.comm, self.ticket, self.keep = comm, ticket, keep

    def wait(self):
        """Device-side ordering of the current stream after the collective; the host does not block."""
        if self.ticket is not None:
            self.comm._wait_ticket(self.ticket)
            self.ticket = None
            self.keep = None
        return True

    def is_completed(self) -> bool:
        return self.ticket is None or self.comm._query_ticket(self.ticket)

    def synchronize(self, timeout_s: float = 0.0) -> bool:
        """Block the host until the collective has finished (bounded when ``timeout_s`` > 0);
        returns False on timeout."""
        if self.ticket is None:
            return True
        return self.comm._sync_ticket(self.ticket, timeout_s)


def _bind(l):
    if getattr(l, "_pda_rccl_bound", False):
        return l
    c_p, c_i, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    sigs = {
        "pda_rccl_available": ([], c_i),
        "pda_rccl_version": ([], c_i),
        "pda_rccl_unique_id": ([c_p], c_i),
        "pda_rccl_create": ([c_p, c_i, c_i, c_i, c_i], c_p),
        "pda_rccl_destroy": ([c_p], None),
        "pda_rccl_last_error": ([c_p], ctypes.c_char_p),
        "pda_rccl_stream": ([c_p], c_p),
        "pda_rccl_nranks": ([c_p], c_i),
        "pda_rccl_user_rank": ([c_p], c_i),
        "pda_rccl_wait": ([c_p, c_ll, c_p], c_i),
        "pda_rccl_query": ([c_p, c_ll], c_i),
        "pda_rccl_sync": ([c_p, c_ll, c_i], c_i),
        "pda_rccl_all_reduce": ([c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_i], c_ll),
        "pda_rccl_broadcast": ([c_p, c_p, c_ll, c_i, c_i, c_p, c_i], c_ll),
        "pda_rccl_all_gather": ([c_p, c_p, c_p, c_ll, c_i, c_p, c_i], c_ll),
        "pda_rccl_reduce_scatter": ([c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_i], c_ll),
        "pda_rccl_all_reduce_group": ([c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_i], c_ll),
        "pda_rccl_sendrecv": ([c_p, c_p, c_ll, c_p, c_ll, c_i, c_i, c_p, c_i], c_ll),
        "pda_rccl_debug_state": ([c_p, ctypes.c_char_p, c_i], c_i),
        "pda_rccl_handoff_mode": ([c_p], c_i),
        "pda_rccl_async_error": ([c_p], c_i),
        "pda_rccl_abort": ([c_p], c_i),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = res
    l._pda_rccl_bound = True
    return l


def exchange_unique_id(uid: bytes, rank: int, world_size: int, store=None, bootstrap_group=None,
                       key: str = "pda/rccl_uid", device=None) -> bytes:
    """Rank 0's 128-byte RCCL unique id -> every rank, through a key-value store (this package's
    TCPStore / FileStore / PrefixStore or a c10d store) or one broadcast over an existing process group."""
    if world_size <= 1:
        return uid
    if store is not None:
        if rank == 0:
            store.set(key, bytes(uid))
            return uid
        return bytes(store.get(key))[:128]
    import torch.distributed as dist
    if bootstrap_group is not None or dist.is_initialized():
        grp = bootstrap_group if bootstrap_group is not None else dist.group.WORLD
        on_dev = dist.get_backend(grp) == "nccl"
        t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).clone()
        t = t.to(device) if on_dev else t
        src = dist.get_global_rank(grp, 0) if hasattr(dist, "get_global_rank") else 0
        dist.broadcast(t, src=src, group=grp)
        return bytes(t.cpu().numpy().tobytes())
    raise RuntimeError("RcclCommunicator needs a store or a bootstrap process group to share the unique id")


class RcclCommunicator:
    """One RCCL communicator + dedicated comm stream per (process, device); the stream / event hand-off
    lives in ``csrc/kernels/rccl_comm.hip``.

    ``store``: anything with ``set(key, bytes)`` / ``get(key) -> bytes``; alternatively
    ``bootstrap_group`` = an existing ``torch.distributed`` group used once to broadcast the unique id.
    ``overlap`` (default on; ``PDA_RCCL_OVERLAP=0`` turns it off): collectives run on the comm stream and
    overlap with whatever the caller enqueues next on its own stream; off = the collective is simply the
    next operation of the caller's stream.

    The comm stream has the SAME priority as the compute streams, on purpose.  The GPU's command
    processor serves hardware queues in strict priority order and a queue whose head packet is a wait
    still counts as busy, so with streams of different priority that wait for each other (comm waits for
    the producer, the consumer waits for comm) the wait at the head of the higher-priority queue is never
    preempted in favour of the lower-priority queue holding the work it waits for: a device deadlock
    whenever the host has run ahead of the device (``benchmarks/overlap_probe.py`` variants ``native_prio``
    / ``native_low``; measured record in ``profiles/overlap_probe_summary.txt``).  ``priority`` is therefore a
    diagnostics knob (``"high"`` / ``"low"``), not a tuning knob.
    """

    def __init__(self, rank: int, world_size: int, device: Optional[torch.device] = None, store=None,
                 bootstrap_group=None, key: str = "pda/rccl_uid", overlap: Optional[bool] = None,
                 priority: str = "normal"):
        self._l = _bind(_lib.lib())
        if not self._l.pda_rccl_available():
            raise RuntimeError("RCCL could not be loaded (librccl.so)")
        self.rank, self.world_size = int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            if self._l.pda_rccl_unique_id(uid) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
        raw = exchange_unique_id(bytes(uid.raw), self.rank, self.world_size, store, bootstrap_group, key, self.device)
        uid = ctypes.create_string_buffer(raw[:128], 128)
        prio = {"normal": 0, "high": 1, "low": -1}[priority]
        torch.cuda.synchronize(self.device)
        self._h = self._l.pda_rccl_create(uid, self.rank, self.world_size, self.device.index, prio)
        if not self._h:
            raise RuntimeError("ncclCommInitRank failed")
        self.backend = "rccl-native"
        self.priority = priority
        # the comm stream belongs to the native communicator; this wrapper only serves synchronize() / tracing
        self.stream = torch.cuda.ExternalStream(self._l.pda_rccl_stream(self._h), device=self.device)
        self.overlap = os.environ.get("PDA_RCCL_OVERLAP", "1") == "1" if overlap is None else bool(overlap)

    # -- helpers ----------------------------------------------------------------------------
    def _err(self, what: str):
        err = self._l.pda_rccl_last_error(self._h) if self._h else b"communicator closed"
        return RuntimeError(f"RcclCommunicator.{what} failed: {err.decode() if err else 'unknown error'}")

    def _prep(self, t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda or t.device != self.device:
            raise ValueError(f"RcclCommunicator works on tensors of {self.device}, got {t.device}")
        if not t.is_contiguous() and not (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("collectives need dense tensors")
        if t.dtype not in _DT:
            raise TypeError(f"unsupported dtype {t.dtype}")
        return t

    def _cur(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _ov(self, stream_ordered: bool) -> int:
        return 0 if stream_ordered else int(self.overlap)

    def _done(self, ticket: int, what: str, async_op: bool, keep, stream_ordered: bool = False):
        if ticket is None or ticket < 0:
            raise self._err(what)
        if not self.overlap or stream_ordered:
            return _DoneWork() if async_op else None      # already in the caller's stream order
        w = _RcclWork(self, ticket, keep)
        if async_op:
            return w
        w.wait()
        return None

    def _wait_ticket(self, ticket: int) -> None:
        if self._h and self._l.pda_rccl_wait(self._h, ticket, self._cur()) != 0:
            raise self._err("wait")

    def _query_ticket(self, ticket: int) -> bool:
        return not self._h or self._l.pda_rccl_query(self._h, ticket) != 0

    def _sync_ticket(self, ticket: int, timeout_s: float) -> bool:
        if not self._h:
            return True
        rc = self._l.pda_rccl_sync(self._h, ticket, int(timeout_s * 1000))
        if rc < 0:
            raise self._err("synchronize")
        return rc == 0

    # -- collectives ------------------------------------------------------------------------
    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False, stream_ordered: bool = False):
        """``stream_ordered=True``: run as the next operation of the caller's current stream whatever the
        communicator's overlap setting (callers that already sit on a side stream of their own)."""
        t = self._prep(t)
        k = self._l.pda_rccl_all_reduce(self._h, t.data_ptr(), t.data_ptr(), t.numel(), _DT[t.dtype], _OPS[op],
                                        self._cur(), self._ov(stream_ordered))
        return self._done(k, "all_reduce", async_op, t, stream_ordered)

    def all_reduce_coalesced(self, tensors, op: str = "sum", async_op: bool = False):
        """Several same-dtype all-reduces as one RCCL group launch (one work object)."""
        ts = [self._prep(t) for t in tensors]
        if not ts:
            return _DoneWork() if async_op else None
        assert all(t.dtype == ts[0].dtype for t in ts)
        n = len(ts)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        cnts = (ctypes.c_longlong * n)(*[t.numel() for t in ts])
        k = self._l.pda_rccl_all_reduce_group(self._h, n, ptrs, cnts, _DT[ts[0].dtype], _OPS[op], self._cur(),
                                              int(self.overlap))
        return self._done(k, "all_reduce_coalesced", async_op, tuple(ts))

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False, stream_ordered: bool = False):
        t = self._prep(t)
        k = self._l.pda_rccl_broadcast(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], int(src), self._cur(),
                                       self._ov(stream_ordered))
        return self._done(k, "broadcast", async_op, t, stream_ordered)

    def all_gather(self, out: torch.Tensor, t: torch.Tensor, async_op: bool = False, stream_ordered: bool = False):
        t, out = self._prep(t), self._prep(out)
        assert out.numel() == t.numel() * self.world_size and out.dtype == t.dtype
        k = self._l.pda_rccl_all_gather(self._h, t.data_ptr(), out.data_ptr(), t.numel(), _DT[t.dtype], self._cur(),
                                        self._ov(stream_ordered))
        return self._done(k, "all_gather", async_op, (t, out), stream_ordered)

    def reduce_scatter(self, out: torch.Tensor, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        t, out = self._prep(t), self._prep(out)
        assert t.numel() == out.numel() * self.world_size and out.dtype == t.dtype
        k = self._l.pda_rccl_reduce_scatter(self._h, t.data_ptr(), out.data_ptr(), out.numel(), _DT[t.dtype],
                                            _OPS[op], self._cur(), int(self.overlap))
        return self._done(k, "reduce_scatter", async_op, (t, out))

    def sendrecv(self, send: torch.Tensor, recv: torch.Tensor, peer: int, async_op: bool = False,
                 stream_ordered: bool = False):
        """Send ``send`` to ``peer`` and receive ``recv`` from it (one RCCL group: ncclSend + ncclRecv)."""
        send, recv = self._prep(send), self._prep(recv)
        assert send.dtype == recv.dtype
        k = self._l.pda_rccl_sendrecv(self._h, send.data_ptr(), send.numel(), recv.data_ptr(), recv.numel(),
                                      _DT[send.dtype], int(peer), self._cur(), self._ov(stream_ordered))
        return self._done(k, "sendrecv", async_op, (send, recv), stream_ordered)

    def barrier(self, timeout_s: float = 0.0) -> None:
        """All ranks have reached this point AND their previously enqueued device work is done.  With
        ``timeout_s`` > 0 a peer that never arrives raises instead of blocking the host for ever."""
        flag = torch.zeros(1, device=self.device, dtype=torch.float32)
        k = self._l.pda_rccl_all_reduce(self._h, flag.data_ptr(), flag.data_ptr(), 1, _DT[flag.dtype], _OPS["sum"],
                                        self._cur(), int(self.overlap))
        if k < 0:
            raise self._err("barrier")
        if not self._sync_ticket(k, timeout_s):
            raise TimeoutError(f"RcclCommunicator.barrier: rank {self.rank} waited {timeout_s:.0f} s for its peers")
        torch.cuda.current_stream(self.device).synchronize()

    # -- introspection / health ---------------------------------------------------------------
    def nranks(self) -> int:
        """Communicator size as RCCL reports it (``ncclCommCount``)."""
        return int(self._l.pda_rccl_nranks(self._h))

    def rccl_version(self) -> int:
        return int(self._l.pda_rccl_version())

    def handoff(self) -> str:
        """How the comm stream and the compute stream are ordered against each other: ``"value"`` (stream
        memory operations on sequence counters) or ``"events"``."""
        return {2: "value", 1: "events", 0: "single-event"}[int(self._l.pda_rccl_handoff_mode(self._h))]

    def debug_state(self) -> str:
        """Completion state of the last tickets, the hand-off marker and the comm stream (for stall reports)."""
        buf = ctypes.create_string_buffer(1024)
        self._l.pda_rccl_debug_state(self._h, buf, 1024)
        return buf.value.decode()

    def async_error(self) -> int:
        """0 while the communicator is healthy, otherwise RCCL's asynchronous error code."""
        return int(self._l.pda_rccl_async_error(self._h))

    def abort(self) -> None:
        self._l.pda_rccl_abort(self._h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._l.pda_rccl_destroy(self._h)     # drains the comm stream first
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_collectives(self) -> bool:
        return True
