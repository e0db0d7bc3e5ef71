"""Python face of the C++ reducer core (``csrc/runtime/reducer.cc``): bucket plan +
ready tracking.  See ``parallel/distributed.py`` for how it is driven."""
from __future__ import annotations

import ctypes
from typing import List, Sequence

from ..comm import _runtime


class BucketPlan:
    def __init__(self, sizes_bytes: Sequence[int], cap_bytes: int, first_cap_bytes: int, align: int = 16,
                 pad: int = 8 * 1024):
        self._l = _runtime.lib()
        n = len(sizes_bytes)
        arr = (ctypes.c_longlong * n)(*[int(s) for s in sizes_bytes])
        self._h = self._l.pda_reducer_create(n, arr, int(cap_bytes), int(first_cap_bytes), int(align), int(pad))
        self.n = n
        self._out = (ctypes.c_int * max(n, 1))()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._l.pda_reducer_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def handle(self) -> int:
        """Address of the native plan (for the C ready hooks)."""
        return int(self._h)

    def mark_ready_address(self) -> int:
        """Address of ``pda_reducer_mark_ready`` (called by the C ready hooks without going through ctypes)."""
        return ctypes.cast(self._l.pda_reducer_mark_ready, ctypes.c_void_p).value

    @property
    def num_buckets(self) -> int:
        return self._l.pda_reducer_num_buckets(self._h)

    def bucket_bytes(self, b: int) -> int:
        return self._l.pda_reducer_bucket_bytes(self._h, b)

    def bucket_params(self, b: int) -> List[int]:
        k = self._l.pda_reducer_bucket_params(self._h, b, self._out, self.n)
        return list(self._out[:k])

    def bucket_of(self, p: int) -> int:
        return self._l.pda_reducer_bucket_of(self._h, p)

    def offset_of(self, p: int) -> int:
        return self._l.pda_reducer_offset_of(self._h, p)

    def reset(self) -> None:
        self._l.pda_reducer_reset(self._h)

    def mark_ready(self, p: int) -> List[int]:
        k = self._l.pda_reducer_mark_ready(self._h, p, self._out, self.n)
        return list(self._out[:k]) if k else []

    def flush_unready(self) -> List[int]:
        k = self._l.pda_reducer_flush_unready(self._h, self._out, self.n)
        return list(self._out[:k]) if k else []

    def num_unready(self) -> int:
        return self._l.pda_reducer_num_unready(self._h)

    def record_arrival(self, on: bool) -> None:
        self._l.pda_reducer_record_arrival(self._h, int(on))

    def arrival_order(self) -> List[int]:
        k = self._l.pda_reducer_arrival(self._h, self._out, self.n)
        return list(self._out[:k])

    def rebuild(self, order: Sequence[int]) -> None:
        arr = (ctypes.c_int * len(order))(*[int(o) for o in order])
        self._l.pda_reducer_rebuild(self._h, arr, len(order))
