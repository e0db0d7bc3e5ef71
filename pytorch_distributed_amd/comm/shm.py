"""Shared-memory CPU collective backend (python face of ``csrc/runtime/shm_backend.cc``).

``ShmCommunicator`` offers all_reduce / broadcast / all_gather / barrier (sync and async
with ``Work`` handles) on CPU tensors between the ranks of one host.  It is the CPU data
plane of the Horovod-style API (``--no-cuda``, /root/reference/5.2.horovod_pytorch_mnist.py:22,75-78)
and the control plane of its name negotiation.  Rendezvous (the segment name) goes through
a ``Store``.
"""
from __future__ import annotations

import os
import uuid
from typing import List, Optional

import torch

from . import _runtime

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.uint8: 3, torch.float64: 4, torch.int32: 5,
       torch.int64: 6}
SUM, PROD, MIN, MAX, AVG = 0, 1, 2, 3, 4


class Work:
    def __init__(self, comm: "ShmCommunicator", ticket: int, keep=None):
        self._comm, self._ticket, self._keep = comm, ticket, keep

    def wait(self) -> bool:
        rc = self._comm._l.pda_shm_wait(self._comm._h, self._ticket)
        self._keep = None
        if rc != 0:
            raise RuntimeError("shared-memory collective failed (peer timeout or abort)")
        return True

    def is_completed(self) -> bool:
        return self._comm._l.pda_shm_poll(self._comm._h, self._ticket) == 1


class ShmCommunicator:
    def __init__(self, store, rank: int, world_size: int, name: str = "default", slot_bytes: int = 4 << 20,
                 timeout: float = 300.0):
        self._l = _runtime.lib()
        self.rank, self.world_size = rank, world_size
        key = f"shm/{name}/segment"
        if rank == 0:
            seg = f"/pda_{os.getuid()}_{uuid.uuid4().hex[:12]}"
            self._h = self._l.pda_shm_create(seg.encode(), rank, world_size, slot_bytes, int(timeout * 1000))
            store.set(key, seg)
        else:
            seg = store.get(key, timeout).decode()
            self._h = self._l.pda_shm_create(seg.encode(), rank, world_size, slot_bytes, int(timeout * 1000))
        if not self._h:
            raise RuntimeError(f"rank {rank}: could not attach shared-memory segment {seg}")
        self.segment = seg
        self.barrier()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._l.pda_shm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check(t: torch.Tensor) -> None:
        if t.is_cuda or not t.is_contiguous() or t.dtype not in _DT:
            raise TypeError("shm backend needs contiguous CPU tensors of a supported dtype")

    def all_reduce(self, t: torch.Tensor, op: int = SUM, async_op: bool = False) -> Optional[Work]:
        self._check(t)
        w = Work(self, self._l.pda_shm_all_reduce_async(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], op), t)
        return w if async_op else w.wait() and None

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False) -> Optional[Work]:
        self._check(t)
        w = Work(self, self._l.pda_shm_broadcast_async(self._h, t.data_ptr(), t.numel() * t.element_size(), src), t)
        return w if async_op else w.wait() and None

    def all_gather(self, out: torch.Tensor, t: torch.Tensor, async_op: bool = False) -> Optional[Work]:
        """out: [world * t.numel()] (or [world, ...]) contiguous."""
        self._check(t)
        self._check(out)
        assert out.numel() == t.numel() * self.world_size
        w = Work(self, self._l.pda_shm_all_gather_async(self._h, t.data_ptr(), out.data_ptr(),
                                                        t.numel() * t.element_size()), (t, out))
        return w if async_op else w.wait() and None

    def all_gather_bytes(self, payload: bytes) -> List[bytes]:
        """Variable-length all-gather (two rounds: lengths, then padded payloads)."""
        n = torch.tensor([len(payload)], dtype=torch.int64)
        lens = torch.empty(self.world_size, dtype=torch.int64)
        self.all_gather(lens, n)
        m = int(lens.max())
        buf = torch.zeros(max(m, 1), dtype=torch.uint8)
        if payload:
            buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        out = torch.empty(self.world_size * max(m, 1), dtype=torch.uint8)
        self.all_gather(out, buf)
        out = out.view(self.world_size, -1)
        return [bytes(out[r, :int(lens[r])].tolist()) for r in range(self.world_size)]

    def reduce_scatter(self, out: torch.Tensor, t: torch.Tensor, op: int = SUM) -> None:
        tmp = t.clone()
        self.all_reduce(tmp, op)
        out.copy_(tmp.view(self.world_size, -1)[self.rank].view_as(out))

    def barrier(self, async_op: bool = False) -> Optional[Work]:
        w = Work(self, self._l.pda_shm_barrier_async(self._h))
        return w if async_op else w.wait() and None

    def abort(self) -> None:
        self._l.pda_shm_abort(self._h)
