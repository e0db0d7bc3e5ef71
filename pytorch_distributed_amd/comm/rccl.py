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
    def wait(self):
        return True

    def is_completed(self):
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

    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        d = self._dist
        rop = {"sum": d.ReduceOp.SUM, "prod": d.ReduceOp.PRODUCT, "min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX,
               "avg": d.ReduceOp.AVG}[op]
        return d.all_reduce(t, op=rop, group=self.group, async_op=async_op)

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        return self._dist.broadcast(t, src=self._global(src), group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, t: torch.Tensor, async_op: bool = False):
        return self._dist.all_gather_into_tensor(out, t, group=self.group, async_op=async_op)

    def reduce_scatter(self, out: torch.Tensor, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        d = self._dist
        rop = {"sum": d.ReduceOp.SUM, "avg": d.ReduceOp.AVG, "min": d.ReduceOp.MIN, "max": d.ReduceOp.MAX}[op]
        return d.reduce_scatter_tensor(out, t, op=rop, group=self.group, async_op=async_op)

    def barrier(self):
        self._dist.barrier(group=self.group)

    @property
    def device_collectives(self) -> bool:
        return self.backend == "nccl"


class _RcclWork:
    """Completion of one collective = an event recorded on the comm stream after it."""
    __slots__ = ("event", "keep", "device")

    def __init__(self, event, keep, device=None):
        self.event, self.keep, self.device = event, keep, device

    def wait(self):
        if self.event is not None:
            # device-side ordering, the host does not block
            torch.cuda.current_stream(self.device).wait_event(self.event)
            self.event = None
            self.keep = None
        return True

    def is_completed(self) -> bool:
        return self.event is None or self.event.query()


def _bind(l):
    if getattr(l, "_pda_rccl_bound", False):
        return l
    c_p, c_i, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    sigs = {
        "pda_rccl_available": ([], c_i),
        "pda_rccl_unique_id": ([c_p], c_i),
        "pda_rccl_create": ([c_p, c_i, c_i, c_i], c_p),
        "pda_rccl_destroy": ([c_p], None),
        "pda_rccl_last_error": ([c_p], ctypes.c_char_p),
        "pda_rccl_all_reduce": ([c_p, c_p, c_p, c_ll, c_i, c_i, c_p], c_i),
        "pda_rccl_broadcast": ([c_p, c_p, c_ll, c_i, c_i, c_p], c_i),
        "pda_rccl_all_gather": ([c_p, c_p, c_p, c_ll, c_i, c_p], c_i),
        "pda_rccl_reduce_scatter": ([c_p, c_p, c_p, c_ll, c_i, c_i, c_p], c_i),
        "pda_rccl_async_error": ([c_p], c_i),
        "pda_rccl_abort": ([c_p], c_i),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = res
    l._pda_rccl_bound = True
    return l


class RcclCommunicator:
    """One RCCL communicator + comm stream per (process, device).

    ``store``: anything with ``set(key, bytes)`` / ``get(key) -> bytes`` (this package's TCPStore /
    FileStore / PrefixStore, or a c10d store); alternatively ``bootstrap_group`` = an existing
    ``torch.distributed`` group used once to broadcast the unique id.
    """

    def __init__(self, rank: int, world_size: int, device: Optional[torch.device] = None, store=None,
                 bootstrap_group=None, key: str = "pda/rccl_uid", overlap: Optional[bool] = None):
        self._l = _bind(_lib.lib())
        if not self._l.pda_rccl_available():
            raise RuntimeError("RCCL could not be loaded (librccl.so)")
        self.rank, self.world_size = int(rank), int(world_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            if self._l.pda_rccl_unique_id(uid) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
        if self.world_size > 1:
            if store is not None:
                if self.rank == 0:
                    store.set(key, bytes(uid.raw))
                else:
                    raw = store.get(key)
                    uid = ctypes.create_string_buffer(bytes(raw)[:128], 128)
            elif bootstrap_group is not None or torch.distributed.is_initialized():
                import torch.distributed as dist
                grp = bootstrap_group if bootstrap_group is not None else dist.group.WORLD
                on_dev = dist.get_backend(grp) == "nccl"
                t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
                t = t.to(self.device) if on_dev else t
                src = dist.get_global_rank(grp, 0) if hasattr(dist, "get_global_rank") else 0
                dist.broadcast(t, src=src, group=grp)
                uid = ctypes.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
            else:
                raise RuntimeError("RcclCommunicator needs a store or a bootstrap process group to share the unique id")
        torch.cuda.synchronize(self.device)
        self._h = self._l.pda_rccl_create(uid, self.rank, self.world_size, self.device.index or 0)
        if not self._h:
            raise RuntimeError("ncclCommInitRank failed")
        self.backend = "rccl-native"
        # dedicated high-priority stream: collectives overtake queued compute kernels where the
        # hardware allows and never serialise behind the backward pass they overlap with
        self.stream = torch.cuda.Stream(device=self.device, priority=-1)
        # overlap=False (default): a collective is the next operation of the caller's stream - no
        # cross-stream dependency at all.  overlap=True (or PDA_RCCL_OVERLAP=1) runs it on the comm
        # stream above so it overlaps with compute; with the HIP runtime of this image that mode was
        # seen to stall when collectives are enqueued from an autograd worker thread that existed
        # before the communicator (pure event chain, single rank), so it is opt-in.  The
        # torch.distributed back-end (TorchCollectives) always overlaps.
        self.overlap = os.environ.get("PDA_RCCL_OVERLAP", "0") == "1" if overlap is None else bool(overlap)

    # -- helpers ----------------------------------------------------------------------------
    def _check(self, rc, what: str) -> None:
        if rc is None or rc < 0:
            err = self._l.pda_rccl_last_error(self._h)
            raise RuntimeError(f"RcclCommunicator.{what} failed: {err.decode() if err else 'unknown error'}")

    def _prep(self, t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda or t.device != self.device:
            raise ValueError(f"RcclCommunicator works on tensors of {self.device}, got {t.device}")
        if not t.is_contiguous() and not (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("collectives need dense tensors")
        if t.dtype not in _DT:
            raise TypeError(f"unsupported dtype {t.dtype}")
        return t

    def _run(self, call, async_op: bool, keep):
        """comm stream waits for the caller's stream, runs ``call(stream_handle)``, records completion."""
        cur = torch.cuda.current_stream(self.device)
        if not self.overlap:
            # in-order mode: the collective is simply the next operation of the caller's stream
            self._check(call(cur.cuda_stream), "collective")
            return _DoneWork() if async_op else None
        self.stream.wait_stream(cur)
        self._check(call(self.stream.cuda_stream), "collective")
        ev = torch.cuda.Event()
        ev.record(self.stream)
        for t in (keep if isinstance(keep, tuple) else (keep,)):
            t.record_stream(self.stream)   # the allocator must not recycle the buffer under the collective
        w = _RcclWork(ev, keep, self.device)
        if async_op:
            return w
        w.wait()
        return None

    # -- collectives ------------------------------------------------------------------------
    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        t = self._prep(t)
        return self._run(lambda s: self._l.pda_rccl_all_reduce(self._h, t.data_ptr(), t.data_ptr(), t.numel(),
                                                               _DT[t.dtype], _OPS[op], s), async_op, t)

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        t = self._prep(t)
        return self._run(lambda s: self._l.pda_rccl_broadcast(self._h, t.data_ptr(), t.numel(), _DT[t.dtype],
                                                              int(src), s), async_op, t)

    def all_gather(self, out: torch.Tensor, t: torch.Tensor, async_op: bool = False):
        t, out = self._prep(t), self._prep(out)
        assert out.numel() == t.numel() * self.world_size and out.dtype == t.dtype
        return self._run(lambda s: self._l.pda_rccl_all_gather(self._h, t.data_ptr(), out.data_ptr(), t.numel(),
                                                               _DT[t.dtype], s), async_op, (t, out))

    def reduce_scatter(self, out: torch.Tensor, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        t, out = self._prep(t), self._prep(out)
        assert t.numel() == out.numel() * self.world_size and out.dtype == t.dtype
        return self._run(lambda s: self._l.pda_rccl_reduce_scatter(self._h, t.data_ptr(), out.data_ptr(), out.numel(),
                                                                   _DT[t.dtype], _OPS[op], s), async_op, (t, out))

    def barrier(self) -> None:
        """All ranks have reached this point AND their previously enqueued device work is done."""
        flag = torch.zeros(1, device=self.device, dtype=torch.float32)
        self.all_reduce(flag)
        self.stream.synchronize()
        torch.cuda.current_stream(self.device).synchronize()

    # -- health -----------------------------------------------------------------------------
    def async_error(self) -> int:
        """0 while the communicator is healthy, otherwise RCCL's asynchronous error code."""
        return int(self._l.pda_rccl_async_error(self._h))

    def abort(self) -> None:
        self._l.pda_rccl_abort(self._h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                self.stream.synchronize()
            except Exception:
                pass
            self._l.pda_rccl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_collectives(self) -> bool:
        return True
