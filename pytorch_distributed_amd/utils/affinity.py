"""CPU placement of a rank next to its GPU.

One process per GPU, eight of them on a two-socket node: a rank whose host thread runs on the other socket pays
the inter-socket hop on every kernel launch, doorbell and pinned-memory copy, and a training step here is several
hundred launches.  ``bind_to_gpu`` restricts the calling thread (and every thread it creates afterwards: autograd
workers, data loaders) to the CPUs the kernel lists as local to the GPU's PCI device.  The reference leaves this to
the job scripts (``numactl`` / ``srun --cpu-bind`` around /root/reference/6.distributed_slurm_main.py); here the rank
does it itself, so it also holds under ``torchrun`` or any other launcher.
"""
from __future__ import annotations

import os
from typing import Optional, Set


def parse_cpulist(text: str) -> Set[int]:
    """``"0-3,8,10-11"`` -> {0, 1, 2, 3, 8, 10, 11}."""
    cpus: Set[int] = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_pci_address(index: int) -> Optional[str]:
    """``dddd:bb:dd.f`` of GPU ``index`` (None when torch does not expose it)."""
    import torch
    try:
        p = torch.cuda.get_device_properties(index)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except (AttributeError, RuntimeError, AssertionError):
        return None


def gpu_local_cpus(index: int, sysfs: str = "/sys/bus/pci/devices") -> Optional[Set[int]]:
    """CPUs of the NUMA node GPU ``index`` hangs off, or None when the platform does not say."""
    addr = gpu_pci_address(index)
    if addr is None:
        return None
    try:
        with open(os.path.join(sysfs, addr, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None
    return cpus or None


def bind_to_gpu(index: int, min_cpus: int = 4) -> Optional[Set[int]]:
    """Restrict the calling thread to the CPUs local to GPU ``index``; returns the set it was bound to, or None
    when nothing was changed: ``PDA_BIND_CPU=0``, no topology information, the local CPUs already cover (or barely
    intersect) what the process is allowed to use - a container's cpuset wins over the topology."""
    if os.environ.get("PDA_BIND_CPU", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    local = gpu_local_cpus(index)
    if not local:
        return None
    allowed = os.sched_getaffinity(0)
    target = allowed & local
    if len(target) < min_cpus or target == allowed:
        return None
    try:
        os.sched_setaffinity(0, target)
    except OSError:
        return None
    return target
