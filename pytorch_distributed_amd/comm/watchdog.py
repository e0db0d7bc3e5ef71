"""Failure detection for a running job.

The reference has none: a rank that dies leaves the others blocked inside a NCCL collective
(/root/reference/2.distributed.py:219-223 has a barrier and three all-reduces per step).  Here a small
monitor thread per process watches

* the communicator itself - ``RcclCommunicator.async_error()`` (``ncclCommGetAsyncError``): a non-zero
  code means a peer or a link went away under an in-flight collective; the communicator is aborted
  (``ncclCommAbort``) so the stuck kernel returns instead of hanging the GPU queue;
* the peers' heartbeats in the rendezvous store (``comm.store.Heartbeat``): a rank whose beacon is
  older than ``max_age`` seconds is reported dead.

On failure ``on_failure(reason)`` is called once (default: print the reason and terminate the process
with exit code 75, which makes the launcher stop the remaining ranks and report the first failure).
"""
from __future__ import annotations

import os
import sys
import threading
import time
from typing import Callable, Optional


def _default_on_failure(reason: str) -> None:
    print(f"[pda watchdog] {reason}; terminating this rank", file=sys.stderr, flush=True)
    os._exit(75)


class CommWatchdog:
    def __init__(self, comm=None, store=None, rank: int = 0, world_size: int = 1, interval: float = 1.0,
                 max_age: float = 15.0, on_failure: Optional[Callable[[str], None]] = None,
                 start_heartbeat: bool = True):
        self.comm, self.store, self.rank, self.world_size = comm, store, rank, world_size
        self.interval, self.max_age = float(interval), float(max_age)
        self.on_failure = on_failure or _default_on_failure
        self.failure: Optional[str] = None
        self._store_down_since: Optional[float] = None
        self._stop = threading.Event()
        self._hb = None
        if store is not None and start_heartbeat:
            from .store import Heartbeat
            self._hb = Heartbeat(store, rank, interval=max(self.interval, 0.2))  # starts beating immediately
        self._t0 = time.time()
        self._t = threading.Thread(target=self._run, name="pda-watchdog", daemon=True)
        self._t.start()

    def _check_once(self) -> Optional[str]:
        if self.comm is not None:
            try:
                code = int(self.comm.async_error())
            except Exception as e:  # communicator already gone
                return f"communicator health query failed: {e}"
            if code != 0:
                try:
                    self.comm.abort()
                except Exception:
                    pass
                return f"RCCL reported asynchronous error {code} on rank {self.rank}; communicator aborted"
        # peers get one max_age of grace after start-up to publish their first beat
        if self.store is not None and self.world_size > 1 and time.time() - self._t0 > self.max_age:
            from .store import Heartbeat
            try:
                dead = [r for r in Heartbeat.dead_ranks(self.store, self.world_size, self.max_age) if r != self.rank]
                self._store_down_since = None
            except Exception as e:
                # the store lives in rank 0's process: at the end of a job it may go away a moment before this
                # rank stops its own monitor, so only a store that STAYS unreachable counts as a failure
                now = time.time()
                if self._store_down_since is None:
                    self._store_down_since = now
                if now - self._store_down_since > self.max_age:
                    return f"rendezvous store unreachable for more than {self.max_age:.0f}s: {e}"
                return None
            if dead:
                return f"no heartbeat from rank(s) {dead} for more than {self.max_age:.0f}s"
        return None

    def _run(self) -> None:
        while not self._stop.wait(self.interval):
            reason = self._check_once()
            if reason is not None and not self._stop.is_set():
                self.failure = reason
                try:
                    self.on_failure(reason)
                finally:
                    return

    def stop(self, finished: bool = True) -> None:
        """``finished=False``: stop beating without the "ended in good order" marker (tests use it to play a rank
        that died)."""
        self._stop.set()
        if self._hb is not None:
            self._hb.stop(finished=finished)
        if self._t.is_alive() and threading.current_thread() is not self._t:
            self._t.join(timeout=2 * self.interval + 1)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


_ACTIVE = {"wd": None}


def start_watchdog(comm=None, interval: float = 1.0, max_age: Optional[float] = None,
                   on_failure: Optional[Callable[[str], None]] = None) -> Optional[CommWatchdog]:
    """Start (once per process) the monitor for the job the default process group describes: heartbeats through
    the package's rendezvous store and, when there is one, the asynchronous error state of the device
    communicator.  ``engine.fit``, the example scripts and the benchmark ranks call this right after the process
    group exists; ``PDA_WATCHDOG=0`` turns it off, ``PDA_WATCHDOG_MAX_AGE`` sets the heartbeat age (seconds) after
    which a silent peer counts as dead.  Returns None for single-process jobs or when no native store exists."""
    if _ACTIVE["wd"] is not None:
        return _ACTIVE["wd"]
    if os.environ.get("PDA_WATCHDOG", "1") == "0":
        return None
    from . import native_store, rank_info, device_communicator, PrefixStore
    info, st = rank_info(), native_store()
    if info is None or info.world_size <= 1 or st is None:
        return None
    if comm is None:
        try:
            comm = device_communicator(create=False)
        except Exception:  # noqa: BLE001
            comm = None
    if comm is not None and not hasattr(comm, "async_error"):
        comm = None
    if max_age is None:
        max_age = float(os.environ.get("PDA_WATCHDOG_MAX_AGE", "60"))
    _ACTIVE["wd"] = CommWatchdog(comm=comm, store=PrefixStore("watchdog", st), rank=info.rank,
                                 world_size=info.world_size, interval=interval, max_age=max_age,
                                 on_failure=on_failure)
    return _ACTIVE["wd"]


def stop_watchdog() -> None:
    wd = _ACTIVE["wd"]
    _ACTIVE["wd"] = None
    if wd is not None:
        wd.stop()
