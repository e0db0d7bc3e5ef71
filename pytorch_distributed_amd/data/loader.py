"""Batch feeding.

* ``make_loader``: ``torch.utils.data.DataLoader`` with the reference's settings (pinned memory,
  sampler, worker count - and ``--workers`` is honoured, unlike /root/reference/1.dataparallel.py:143
  which hard-codes 2).
* ``DevicePrefetcher``: the reference's ``data_prefetcher`` (/root/reference/4.apex_distributed.py:80-133):
  a side HIP stream copies the next batch host->device (non_blocking from pinned memory) while the
  current one is being computed; ``next()`` makes the compute stream wait on the copy and
  ``record_stream``s the tensors for the caching allocator.  Normalisation is *not* applied here
  (the model's stem kernel fuses it: that is what removes the upstream double-normalisation bug);
  pass ``normalize=(mean, std)`` only for raw uint8 batches fed to models without a fused stem.
* ``SyntheticDeviceLoader``: a fixed number of identical on-device batches per epoch, for benchmarks.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch.utils.data import DataLoader


def make_loader(dataset, batch_size: int, sampler=None, workers: int = 2, pin_memory: bool = True,
                shuffle: bool = False, drop_last: bool = False) -> DataLoader:
    return DataLoader(dataset, batch_size=batch_size, shuffle=(shuffle and sampler is None), sampler=sampler,
                      num_workers=workers, pin_memory=pin_memory and torch.cuda.is_available(), drop_last=drop_last,
                      persistent_workers=workers > 0)


class DevicePrefetcher:
    def __init__(self, loader, device=None, normalize: Optional[Tuple] = None, channels_scale: float = 1.0):
        self.loader = loader
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.norm = None
        if normalize is not None:
            mean, std = normalize
            self.norm = (torch.tensor(mean, device=self.device).view(1, -1, 1, 1) * channels_scale,
                         torch.tensor(std, device=self.device).view(1, -1, 1, 1) * channels_scale)
        self._it = None
        self._next = None

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        self._it = iter(self.loader)
        self._preload()
        return self

    def _preload(self):
        try:
            images, target = next(self._it)
        except StopIteration:
            self._next = None
            return
        if self.cuda:
            with torch.cuda.stream(self.stream):
                images = images.to(self.device, non_blocking=True)
                target = target.to(self.device, non_blocking=True)
                if self.norm is not None:
                    images = images.float().sub_(self.norm[0]).div_(self.norm[1])
        elif self.norm is not None:
            images = images.float().sub_(self.norm[0]).div_(self.norm[1])
        self._next = (images, target)

    def __next__(self):
        if self._next is None:
            raise StopIteration
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        images, target = self._next
        if self.cuda:
            images.record_stream(torch.cuda.current_stream(self.device))
            target.record_stream(torch.cuda.current_stream(self.device))
        self._preload()
        return images, target

    next = __next__


class SyntheticDeviceLoader:
    def __init__(self, batch_size: int, shape, num_classes: int, steps: int, device, seed: int = 0,
                 dtype=torch.float32):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.images = torch.randn((batch_size, *shape), generator=g).to(device=device, dtype=dtype)
        self.target = torch.randint(0, num_classes, (batch_size,), generator=g).to(device)
        self.steps = steps

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.images, self.target
