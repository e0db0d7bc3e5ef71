"""Rendezvous stores (python face of ``csrc/runtime/store.cc``).

``TCPStore(host, port, world_size, is_master)`` and ``FileStore(path, world_size)`` offer
``set / get / add / wait / check / delete_key / num_keys / compare_set`` like c10d's stores,
which is what ``init_process_group('tcp://127.0.0.1:23456', ...)`` and
``init_process_group('file://...', ...)`` build in the reference
(/root/reference/3.multiprocessing_distributed.py:102, /root/reference/6.distributed_slurm_main.py:101).
``PrefixStore`` namespaces keys; ``store_barrier`` is a store-only barrier with a timeout and
``Heartbeat`` a liveness beacon used by the launcher's failure detection.
"""
from __future__ import annotations

import ctypes
import threading
import time
from typing import Iterable, Optional, Union

from . import _runtime

_DEFAULT_TIMEOUT_S = 300.0


class StoreTimeout(TimeoutError):
    pass


def _b(v: Union[str, bytes]) -> bytes:
    return v.encode() if isinstance(v, str) else bytes(v)


class Store:
    def __init__(self, handle, timeout: float):
        if not handle:
            raise RuntimeError("could not create / connect the rendezvous store")
        self._l = _runtime.lib()
        self._h = handle
        self.timeout = timeout

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._l.pda_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, key: str, value: Union[str, bytes]) -> None:
        v = _b(value)
        if self._l.pda_store_set(self._h, key.encode(), v, len(v)) != 0:
            raise RuntimeError(f"store.set({key!r}) failed")

    def get(self, key: str, timeout: Optional[float] = None) -> bytes:
        """Blocks until ``key`` exists (``timeout`` seconds, default = the store's)."""
        t_ms = int(1000 * (self.timeout if timeout is None else timeout))
        cap = 4096
        while True:
            buf = ctypes.create_string_buffer(cap)
            n = self._l.pda_store_get(self._h, key.encode(), buf, cap, t_ms)
            if n == -1:
                raise StoreTimeout(f"store.get({key!r}) timed out after {t_ms / 1000:.1f}s")
            if n < 0:
                raise RuntimeError(f"store.get({key!r}) failed (connection lost?)")
            if n <= cap:
                return buf.raw[:n]
            cap = int(n)

    def add(self, key: str, amount: int = 1) -> int:
        out = ctypes.c_longlong(0)
        if self._l.pda_store_add(self._h, key.encode(), int(amount), ctypes.byref(out)) != 0:
            raise RuntimeError(f"store.add({key!r}) failed")
        return int(out.value)

    def check(self, keys: Union[str, Iterable[str]]) -> bool:
        keys = [keys] if isinstance(keys, str) else list(keys)
        ok = True
        for k in keys:
            rc = self._l.pda_store_check(self._h, k.encode())
            if rc < 0:
                raise RuntimeError(f"store.check({k!r}) failed: store unreachable")
            ok = ok and rc == 1
        return ok

    def wait(self, keys: Union[str, Iterable[str]], timeout: Optional[float] = None) -> None:
        keys = [keys] if isinstance(keys, str) else list(keys)
        deadline = time.monotonic() + (self.timeout if timeout is None else timeout)
        for k in keys:
            self.get(k, max(0.0, deadline - time.monotonic()))

    def delete_key(self, key: str) -> bool:
        return self._l.pda_store_delete(self._h, key.encode()) == 1

    def num_keys(self) -> int:
        return int(self._l.pda_store_num_keys(self._h))

    def append(self, key: str, value: Union[str, bytes]) -> int:
        """value(key) += value as ONE operation on the server (or under the file lock): concurrent appenders to
        one key lose nothing.  Returns the new length."""
        v = _b(value)
        n = self._l.pda_store_append(self._h, key.encode(), v, len(v))
        if n < 0:
            raise RuntimeError(f"store.append({key!r}) failed")
        return int(n)

    def compare_set(self, key: str, expected: Union[str, bytes], desired: Union[str, bytes]) -> bytes:
        e, d = _b(expected), _b(desired)
        cap = max(4096, len(d) + 16)
        buf = ctypes.create_string_buffer(cap)
        n = self._l.pda_store_compare_set(self._h, key.encode(), e, len(e), d, len(d), buf, cap)
        if n < 0:
            raise RuntimeError(f"store.compare_set({key!r}) failed")
        return buf.raw[:min(n, cap)]


class TCPStore(Store):
    def __init__(self, host: str, port: int, world_size: Optional[int] = None, is_master: bool = False,
                 timeout: float = _DEFAULT_TIMEOUT_S):
        l = _runtime.lib()
        h = l.pda_store_tcp_create(host.encode(), int(port), int(is_master), int(timeout * 1000))
        super().__init__(h, timeout)
        self.host, self.world_size, self.is_master = host, world_size, is_master
        self.port = l.pda_store_tcp_port(h)


class FileStore(Store):
    def __init__(self, path: str, world_size: Optional[int] = None, timeout: float = _DEFAULT_TIMEOUT_S):
        l = _runtime.lib()
        super().__init__(l.pda_store_file_create(path.encode(), int(timeout * 1000)), timeout)
        self.path, self.world_size = path, world_size


class PrefixStore:
    def __init__(self, prefix: str, store):
        self.prefix, self.store = prefix, store
        self.timeout = store.timeout

    def _k(self, key):
        return f"{self.prefix}/{key}"

    def set(self, key, value):
        return self.store.set(self._k(key), value)

    def get(self, key, timeout=None):
        return self.store.get(self._k(key), timeout)

    def add(self, key, amount=1):
        return self.store.add(self._k(key), amount)

    def check(self, keys):
        keys = [keys] if isinstance(keys, str) else list(keys)
        return self.store.check([self._k(k) for k in keys])

    def wait(self, keys, timeout=None):
        keys = [keys] if isinstance(keys, str) else list(keys)
        return self.store.wait([self._k(k) for k in keys], timeout)

    def delete_key(self, key):
        return self.store.delete_key(self._k(key))

    def compare_set(self, key, expected, desired):
        return self.store.compare_set(self._k(key), expected, desired)

    def append(self, key, value):
        return self.store.append(self._k(key), value)

    def num_keys(self):
        return self.store.num_keys()


def store_barrier(store, world_size: int, name: str = "barrier", timeout: Optional[float] = None) -> None:
    """All ``world_size`` participants arrive, the last one releases the rest."""
    n = store.add(f"{name}/count", 1)
    gen = (n - 1) // world_size
    if n % world_size == 0:
        store.set(f"{name}/release/{gen}", b"1")
    store.get(f"{name}/release/{gen}", timeout)


class Heartbeat:
    """Background thread that bumps ``hb/<rank>`` every ``interval`` seconds; ``dead_ranks`` lets a
    monitor spot ranks whose beacon stopped (failure detection for rendezvous and the launcher)."""

    def __init__(self, store, rank: int, interval: float = 2.0):
        self.store, self.rank, self.interval = store, rank, interval
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        while not self._stop.is_set():
            try:
                self.store.set(f"hb/{self.rank}", str(time.time()))
            except Exception:
                return
            self._stop.wait(self.interval)

    def stop(self, finished: bool = True):
        """``finished``: leave a beacon that never ages - a rank that ended its work in good order must not be
        reported dead by peers that are still running."""
        self._stop.set()
        if finished:
            try:
                self.store.set(f"hb/{self.rank}", str(time.time() + 1e9))
            except Exception:
                pass

    @staticmethod
    def dead_ranks(store, world_size: int, max_age: float = 10.0):
        """Ranks without a beacon younger than ``max_age`` seconds.  A store that cannot be reached raises (that is
        not the same finding as a silent peer)."""
        now, dead = time.time(), []
        for r in range(world_size):
            if not store.check(f"hb/{r}"):
                dead.append(r)
                continue
            if now - float(store.get(f"hb/{r}", 1.0)) > max_age:
                dead.append(r)
        return dead
