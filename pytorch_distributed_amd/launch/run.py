"""Launcher CLI: ``python -m pytorch_distributed_amd.launch [opts] script.py [script args]``.

One tool for the reference's three external launch styles:

* ``--nproc_per_node N``            : torch.distributed.launch / torchrun style - N children with
  MASTER_ADDR, MASTER_PORT(29500), RANK, WORLD_SIZE, LOCAL_RANK, LOCAL_WORLD_SIZE in the
  environment; ``--use_argv_local_rank`` additionally appends ``--local_rank=i`` like the legacy
  launcher the reference scripts 2 and 4 expect (/root/reference/2.run.sh:5, /root/reference/4.run.sh:3).
* ``-np N [-H host:slots]``         : horovodrun style (/root/reference/5.run.sh:3) - same children plus
  HOROVOD_RANK / HOROVOD_SIZE / HOROVOD_LOCAL_RANK / HOROVOD_LOCAL_SIZE.
* under ``srun``                    : node rank and node count come from SLURM_NODEID / SLURM_NNODES,
  so the same command line works on every node (/root/reference/6.distributed_slurm_main.py:89-94).

Failure handling: the first child that exits non-zero gets its siblings terminated and its exit
code becomes the launcher's.  Every rank still sees all GPUs (RCCL's intra-node peer access and
``LOCAL_RANK``-based ``set_device`` both need that); unless ``--no_pin`` is given the launcher also
exports ``PDA_ASSIGNED_DEVICE`` = the entry of ``HIP_VISIBLE_DEVICES`` (or the ordinal) meant for
this local rank, for scripts that prefer an explicit assignment.
"""
from __future__ import annotations

import argparse
import os
import signal
import subprocess
import sys
import time


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m pytorch_distributed_amd.launch")
    ap.add_argument("--nproc_per_node", "--nproc-per-node", type=int, default=None)
    ap.add_argument("-np", "--num-proc", dest="np", type=int, default=None, help="horovodrun-style process count")
    ap.add_argument("-H", "--hosts", default=None, help="horovodrun-style host:slots list (single host supported)")
    ap.add_argument("--nnodes", type=int, default=None)
    ap.add_argument("--node_rank", "--node-rank", type=int, default=None)
    ap.add_argument("--master_addr", "--master-addr", default=None)
    ap.add_argument("--master_port", "--master-port", type=int, default=None)
    ap.add_argument("--use_argv_local_rank", "--use-argv-local-rank", action="store_true",
                    help="also pass --local_rank=<i> on the child's command line (legacy torch.distributed.launch)")
    ap.add_argument("--no_pin", "--no-pin", action="store_true", help="do not export PDA_ASSIGNED_DEVICE per rank")
    ap.add_argument("--no_python", "--no-python", action="store_true")
    ap.add_argument("-m", "--module", action="store_true", help="run the script as a python module")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    return ap.parse_args(argv)


def child_env(base: dict, rank: int, world: int, local_rank: int, local_size: int, node_rank: int, addr: str,
              port: int, pin: bool) -> dict:
    env = dict(base)
    env.update(MASTER_ADDR=addr, MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=str(local_rank), LOCAL_WORLD_SIZE=str(local_size), GROUP_RANK=str(node_rank),
               HOROVOD_RANK=str(rank), HOROVOD_SIZE=str(world), HOROVOD_LOCAL_RANK=str(local_rank),
               HOROVOD_LOCAL_SIZE=str(local_size), HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    if pin:
        vis = base.get("HIP_VISIBLE_DEVICES") or base.get("CUDA_VISIBLE_DEVICES")
        devs = [d for d in vis.split(",") if d] if vis else None
        env["PDA_ASSIGNED_DEVICE"] = devs[local_rank % len(devs)] if devs else str(local_rank)
    return env


def main(argv=None) -> int:
    a = parse(argv)
    e = os.environ
    nproc = a.nproc_per_node or a.np
    if a.hosts and a.np:
        slots = [int(h.split(":")[1]) if ":" in h else 1 for h in a.hosts.split(",")]
        if len(slots) > 1:
            print("launch: multi-host -H lists need one launcher per node (use srun or --nnodes/--node_rank)",
                  file=sys.stderr)
            return 2
        nproc = min(a.np, slots[0])
    if nproc is None:
        try:
            import torch
            nproc = max(1, torch.cuda.device_count())
        except Exception:
            nproc = 1
    nnodes = a.nnodes if a.nnodes is not None else int(e.get("SLURM_NNODES", 1))
    node_rank = a.node_rank if a.node_rank is not None else int(e.get("SLURM_NODEID", 0))
    addr = a.master_addr or e.get("MASTER_ADDR") or "127.0.0.1"
    port = a.master_port or int(e.get("MASTER_PORT", "29500"))
    world = nproc * nnodes
    procs = []
    for lr in range(nproc):
        rank = node_rank * nproc + lr
        cmd = [] if a.no_python else [sys.executable, "-u"]
        if a.module and not a.no_python:
            cmd.append("-m")
        cmd.append(a.script)
        cmd += a.script_args
        if a.use_argv_local_rank:
            cmd.append(f"--local_rank={lr}")
        env = child_env(dict(e), rank, world, lr, nproc, node_rank, addr, port, not a.no_pin)
        if a.verbose:
            print(f"[launch] rank {rank}/{world} local {lr}: {' '.join(cmd)}", flush=True)
        procs.append(subprocess.Popen(cmd, env=env))

    def kill_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    def on_signal(signum, frame):
        kill_all(signum)
    signal.signal(signal.SIGINT, on_signal)
    signal.signal(signal.SIGTERM, on_signal)
    rc = 0
    while procs:
        for p in list(procs):
            r = p.poll()
            if r is None:
                continue
            procs.remove(p)
            if r != 0 and rc == 0:
                rc = r
                print(f"[launch] a worker exited with code {r}; terminating the others", file=sys.stderr, flush=True)
                kill_all()
                t0 = time.time()
                while any(q.poll() is None for q in procs) and time.time() - t0 < 10:
                    time.sleep(0.1)
                kill_all(signal.SIGKILL)
        time.sleep(0.05)
    return rc if rc >= 0 else 128 - rc


if __name__ == "__main__":
    sys.exit(main())
