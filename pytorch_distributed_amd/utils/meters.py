"""Running meters with the reference's exact text format
(/root/reference/1.dataparallel.py:291-329): ``"{name} {val:fmt} ({avg:fmt})"`` joined by
tabs behind ``prefix[ i/N]`` - so logs of this framework diff cleanly against the
reference's - plus an images/sec meter (the benchmark metric) and device-side metric
accumulation helpers that avoid the reference's four host syncs per step."""
from __future__ import annotations

import time
from typing import List

import torch


class AverageMeter(object):
    """Computes and stores the average and current value."""

    def __init__(self, name: str, fmt: str = ":f"):
        self.name = name
        self.fmt = fmt
        self.reset()

    def reset(self) -> None:
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n: int = 1) -> None:
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self) -> str:
        fmtstr = "{name} {val" + self.fmt + "} ({avg" + self.fmt + "})"
        return fmtstr.format(**self.__dict__)


class ProgressMeter(object):
    def __init__(self, num_batches: int, meters: List[AverageMeter], prefix: str = "", rank0_only: bool = False):
        self.batch_fmtstr = self._get_batch_fmtstr(num_batches)
        self.meters = meters
        self.prefix = prefix
        self.rank0_only = rank0_only

    def line(self, batch: int) -> str:
        entries = [self.prefix + self.batch_fmtstr.format(batch)]
        entries += [str(meter) for meter in self.meters]
        return "\t".join(entries)

    def display(self, batch: int) -> None:
        if self.rank0_only:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
                return
        print(self.line(batch), flush=True)

    @staticmethod
    def _get_batch_fmtstr(num_batches: int) -> str:
        num_digits = len(str(num_batches // 1))
        fmt = "{:" + str(num_digits) + "d}"
        return "[" + fmt + "/" + fmt.format(num_batches) + "]"


class ThroughputMeter(AverageMeter):
    """images/sec: update(batch_images, seconds)."""

    def __init__(self, name: str = "Img/s", fmt: str = ":8.1f"):
        super().__init__(name, fmt)
        self.images = 0
        self.seconds = 0.0

    def reset(self) -> None:
        super().reset()
        self.images = 0
        self.seconds = 0.0

    def update(self, images: int, seconds: float) -> None:  # type: ignore[override]
        self.images += images
        self.seconds += seconds
        self.val = images / seconds if seconds > 0 else 0.0
        self.avg = self.images / self.seconds if self.seconds > 0 else 0.0
        self.sum, self.count = self.images, self.seconds


class DeviceMetrics:
    """{loss_sum, correct@1, correct@5, rows} accumulated on the device by the loss kernel.

    ``snapshot(group)`` all-reduces the 4 floats once (instead of barrier + three scalar
    all-reduces per step, /root/reference/2.distributed.py:219-227) and returns python
    floats: mean loss, acc@1, acc@5 (fractions in [0,1]) and the row count.
    """

    def __init__(self, device):
        self.buf = torch.zeros(4, device=device, dtype=torch.float32)

    def reset(self) -> None:
        self.buf.zero_()

    def snapshot(self, group=None, reset: bool = True):
        import torch.distributed as dist
        t = self.buf.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if t.is_cuda and "nccl" not in str(dist.get_backend(group)):
                t = t.cpu()     # a host-only group (gloo): four floats through its CPU path, not its device-tensor path
            dist.all_reduce(t, group=group)
        loss_sum, c1, c5, n = t.tolist()
        if reset:
            self.reset()
        n = max(n, 1.0)
        return loss_sum / n, c1 / n, c5 / n, int(n)


class StepTimer:
    """Wall-clock step timer that can fence the device, unlike the reference's
    un-synchronised ``time.time()`` deltas (/root/reference/2.distributed.py:204,235)."""

    def __init__(self, sync: bool = False):
        self.sync = sync
        self.t = time.perf_counter()

    def lap(self) -> float:
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        d = now - self.t
        self.t = now
        return d
