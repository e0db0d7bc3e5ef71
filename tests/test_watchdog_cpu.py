"""Failure detection: communicator health polling and heartbeat monitoring (CPU only)."""
import threading
import time

from pytorch_distributed_amd.comm.store import TCPStore
from pytorch_distributed_amd.comm.watchdog import CommWatchdog


class _FakeComm:
    def __init__(self):
        self.code = 0
        self.aborted = False

    def async_error(self):
        return self.code

    def abort(self):
        self.aborted = True


def test_watchdog_aborts_communicator_on_async_error():
    comm, seen, fired = _FakeComm(), [], threading.Event()

    def on_failure(reason):
        seen.append(reason)
        fired.set()

    wd = CommWatchdog(comm=comm, interval=0.05, on_failure=on_failure)
    time.sleep(0.2)
    assert not seen and wd.failure is None            # healthy communicator: silent
    comm.code = 6
    assert fired.wait(3.0)
    assert comm.aborted and "asynchronous error 6" in seen[0] and wd.failure == seen[0]
    wd.stop()


def test_watchdog_reports_silent_peer():
    store = TCPStore("127.0.0.1", 0, 2, is_master=True, timeout=10)
    peer = TCPStore("127.0.0.1", store.port, 2, is_master=False, timeout=10)
    seen, fired = [], threading.Event()

    def on_failure(reason):
        seen.append(reason)
        fired.set()

    # rank 1 beats for a while, then goes silent
    wd1 = CommWatchdog(store=peer, rank=1, world_size=2, interval=0.1, max_age=1.0, on_failure=lambda r: None)
    wd0 = CommWatchdog(store=store, rank=0, world_size=2, interval=0.1, max_age=1.0, on_failure=on_failure)
    time.sleep(1.5)
    assert not seen                                    # both alive after the grace period
    wd1.stop()
    assert fired.wait(5.0)
    assert "rank(s) [1]" in seen[0]
    wd0.stop()
    peer.close()
    store.close()
