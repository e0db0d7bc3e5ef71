"""``spawn(fn, args, nprocs)``: start ``nprocs`` fresh interpreters that run ``fn(i, *args)``.

Behaviour of ``torch.multiprocessing.spawn`` as the reference uses it
(/root/reference/3.multiprocessing_distributed.py:84, /root/reference/6.distributed_slurm_main.py:95):
'spawn' start method, the parent blocks in ``join``; the first child that dies takes its
siblings down (SIGTERM, then SIGKILL) and its traceback is re-raised in the parent.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import signal
import sys
import tempfile
import time
import traceback
from typing import Callable, Optional, Sequence


class ProcessRaisedException(RuntimeError):
    def __init__(self, msg: str, error_index: int, pid: int):
        super().__init__(msg)
        self.error_index, self.pid = error_index, pid


class ProcessExitedException(RuntimeError):
    def __init__(self, msg: str, error_index: int, pid: int, exit_code: int, signal_name: Optional[str] = None):
        super().__init__(msg)
        self.error_index, self.pid, self.exit_code, self.signal_name = error_index, pid, exit_code, signal_name


def _wrap(fn, i, args, error_file, env):
    os.environ.update(env)
    try:
        fn(i, *args)
    except KeyboardInterrupt:
        pass
    except Exception:
        with open(error_file, "w") as f:
            f.write(traceback.format_exc())
        sys.exit(1)


class ProcessContext:
    def __init__(self, processes, error_files):
        self.processes, self.error_files = processes, error_files

    def pids(self):
        return [p.pid for p in self.processes]

    def _terminate_all(self, grace: float = 5.0) -> None:
        for p in self.processes:
            if p.is_alive():
                p.terminate()
        deadline = time.time() + grace
        for p in self.processes:
            p.join(max(0.0, deadline - time.time()))
        for p in self.processes:
            if p.is_alive():
                p.kill()
                p.join()

    def join(self, timeout: Optional[float] = None) -> bool:
        """True when every process has exited cleanly; raises on the first failure."""
        deadline = None if timeout is None else time.time() + timeout
        while True:
            alive = False
            for i, p in enumerate(self.processes):
                if p.is_alive():
                    alive = True
                    continue
                if p.exitcode not in (0, None):
                    self._terminate_all()
                    msg = ""
                    if os.path.exists(self.error_files[i]):
                        with open(self.error_files[i]) as f:
                            msg = f.read()
                    self._cleanup()
                    if msg:
                        raise ProcessRaisedException(f"\n\n-- Process {i} terminated with the following error:\n{msg}",
                                                     i, p.pid)
                    if p.exitcode < 0:
                        name = signal.Signals(-p.exitcode).name
                        raise ProcessExitedException(f"process {i} terminated with signal {name}", i, p.pid,
                                                     p.exitcode, name)
                    raise ProcessExitedException(f"process {i} terminated with exit code {p.exitcode}", i, p.pid,
                                                 p.exitcode)
            if not alive:
                self._cleanup()
                return True
            if deadline is not None and time.time() > deadline:
                return False
            time.sleep(0.05)

    def _cleanup(self):
        for f in self.error_files:
            try:
                os.unlink(f)
            except OSError:
                pass


def start_processes(fn: Callable, args: Sequence = (), nprocs: int = 1, join: bool = True, daemon: bool = False,
                    start_method: str = "spawn", env_per_rank: Optional[Callable[[int], dict]] = None):
    ctx = mp.get_context(start_method)
    procs, files = [], []
    for i in range(nprocs):
        fd, path = tempfile.mkstemp(prefix=f"pda_spawn_{i}_", suffix=".err")
        os.close(fd)
        os.unlink(path)
        env = {k: str(v) for k, v in (env_per_rank(i) if env_per_rank else {}).items()}
        p = ctx.Process(target=_wrap, args=(fn, i, tuple(args), path, env), daemon=daemon)
        p.start()
        procs.append(p)
        files.append(path)
    pc = ProcessContext(procs, files)
    if not join:
        return pc
    while not pc.join():
        pass
    return None


def spawn(fn: Callable, args: Sequence = (), nprocs: int = 1, join: bool = True, daemon: bool = False,
          start_method: str = "spawn"):
    """torch.multiprocessing.spawn-compatible signature."""
    return start_processes(fn, args, nprocs, join, daemon, start_method)
