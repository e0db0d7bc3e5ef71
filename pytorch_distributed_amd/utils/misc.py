"""Small training-loop utilities: LR schedule, seeding, epoch CSV log, JSONL step log."""
from __future__ import annotations

import csv
import json
import os
import random
import time
import warnings
from typing import Optional

import torch


def adjust_learning_rate(optimizer, epoch: int, base_lr, decay_every: int = 30, factor: float = 0.1) -> float:
    """LR = base_lr * 0.1 ** (epoch // 30), set on every param group each epoch
    (/root/reference/2.distributed.py:337-341). ``base_lr`` may be an args namespace with ``.lr``."""
    if hasattr(base_lr, "lr"):
        base_lr = base_lr.lr
    lr = base_lr * (factor ** (epoch // decay_every))
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr
    return lr


def seed_everything(seed: Optional[int], deterministic: bool = True, warn: bool = True) -> None:
    """``--seed`` handling of the reference (/root/reference/2.distributed.py:82-90): seed python and
    torch RNGs, switch the kernels to their run-to-run reproducible forms (``backends.deterministic``: the
    wgrad split-K reduction stops using float atomics) and emit the same slowdown warning."""
    if seed is None:
        return
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        from .. import backends
        backends.deterministic = True   # the counterpart of cudnn.deterministic = True
    if warn:
        warnings.warn("You have chosen to seed training. This will turn on deterministic settings, "
                      "which can slow down your training considerably! You may see unexpected behavior "
                      "when restarting from checkpoints.")


class EpochCSV:
    """Append ``[start timestamp, epoch seconds]`` rows, as scripts 1 and 6 do
    (/root/reference/1.dataparallel.py:168,187-190)."""

    def __init__(self, path: str):
        self.path = path

    def write(self, start_ts: float, seconds: float) -> None:
        with open(self.path, "a+", newline="") as f:
            csv.writer(f).writerow([time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(start_ts)), seconds])


class JsonlLogger:
    def __init__(self, path: Optional[str]):
        self.path = path
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def log(self, **kw) -> None:
        if not self.path:
            return
        kw.setdefault("ts", time.time())
        with open(self.path, "a") as f:
            f.write(json.dumps(kw) + "\n")
