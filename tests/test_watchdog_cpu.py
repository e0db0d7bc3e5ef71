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
    wd1.stop(finished=False)
    assert fired.wait(5.0)
    assert "rank(s) [1]" in seen[0]
    wd0.stop()
    peer.close()
    store.close()


def test_finished_rank_is_not_reported_dead():
    store = TCPStore("127.0.0.1", 0, 2, is_master=True, timeout=10)
    peer = TCPStore("127.0.0.1", store.port, 2, is_master=False, timeout=10)
    seen = []
    wd1 = CommWatchdog(store=peer, rank=1, world_size=2, interval=0.1, max_age=0.5, on_failure=lambda r: None)
    wd0 = CommWatchdog(store=store, rank=0, world_size=2, interval=0.1, max_age=0.5, on_failure=seen.append)
    time.sleep(0.8)
    wd1.stop()                                         # rank 1 ends its work in good order
    time.sleep(1.5)
    assert not seen
    wd0.stop()
    peer.close()
    store.close()


def test_killed_rank_ends_the_job_instead_of_hanging_it(tmp_path):
    """Two real processes (no launcher that would clean up after them): rank 1 dies without a word while rank 0
    sits in what stands for a collective that will never complete; rank 0's watchdog must end it with the
    watchdog's exit code within a few heartbeat ages."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(root, "tests", "workers", "watchdog_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=root, PDA_WATCHDOG_MAX_AGE="3")
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    try:
        out1, _ = procs[1].communicate(timeout=120)
        out0, _ = procs[0].communicate(timeout=120)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert procs[1].returncode == 9, out1[-500:]
    assert procs[0].returncode == 75, (procs[0].returncode, out0[-1500:])
    assert "no heartbeat from rank(s) [1]" in out0 and "HUNG" not in out0
    assert time.time() - t0 < 90
