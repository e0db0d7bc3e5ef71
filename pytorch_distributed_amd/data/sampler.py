"""DistributedSampler with torch's exact partitioning
(what ``torch.utils.data.distributed.DistributedSampler(train_dataset)`` gives the reference,
/root/reference/2.distributed.py:138,155,167-168, and with explicit ``num_replicas=hvd.size(),
rank=hvd.rank()``, /root/reference/5.horovod_distributed.py:135-137):
``num_samples = ceil(N / world)``, epoch-seeded ``randperm`` (``seed + epoch``), padding by
wrapping, rank r takes ``indices[r : total : world]``.  ``exact=True`` additionally reports which
of this rank's samples are padding duplicates so evaluation can mask them (the reference counts
them twice, /root/reference/2.distributed.py:155)."""
from __future__ import annotations

import math
from typing import Iterator, Optional

import torch
from torch.utils.data import Sampler


class DistributedSampler(Sampler):
    def __init__(self, dataset, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True,
                 seed: int = 0, drop_last: bool = False):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
                rank = dist.get_rank() if rank is None else rank
            else:
                num_replicas = 1 if num_replicas is None else num_replicas
                rank = 0 if rank is None else rank
        if rank >= num_replicas or rank < 0:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.epoch, self.drop_last, self.shuffle, self.seed = 0, drop_last, shuffle, seed
        n = len(dataset)
        if drop_last and n % num_replicas != 0:
            self.num_samples = math.ceil((n - num_replicas) / num_replicas)
        else:
            self.num_samples = math.ceil(n / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def _indices(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(n, generator=g).tolist()
        else:
            indices = list(range(n))
        if not self.drop_last:
            pad = self.total_size - len(indices)
            if pad > 0:
                if pad <= len(indices):
                    indices += indices[:pad]
                else:
                    indices += (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[:self.total_size]
        return indices

    def __iter__(self) -> Iterator[int]:
        indices = self._indices()[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)

    def padding_mask(self):
        """True for this rank's positions that are wrap-around duplicates."""
        n = len(self.dataset)
        pos = range(self.rank, self.total_size, self.num_replicas)
        return [p >= n for p in pos]

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
