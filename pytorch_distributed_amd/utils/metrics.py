"""Accuracy and cross-rank metric reduction with the reference's semantics.

* ``accuracy`` - the CIFAR scripts' variant: arg-max hit *fraction* in [0,1], returned
  twice as ``[acc, acc]`` ("top5 = top1"), as a 0-dim tensor on the outputs' device
  (/root/reference/2.distributed.py:344-364) - but without the ``.item()`` host sync.
* ``accuracy_topk`` - the true top-k percentage variant of the SLURM script
  (/root/reference/6.distributed_slurm_main.py:335-349).
* ``reduce_mean`` - clone, all-reduce SUM, divide by world size
  (/root/reference/2.distributed.py:71-75); ``reduce_mean_many`` fuses several scalars
  into one all-reduce.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def accuracy(output: torch.Tensor, target: torch.Tensor, topk: Sequence[int] = (1,)) -> List[torch.Tensor]:
    with torch.no_grad():
        pred = output.argmax(dim=1)
        acc = (pred == target).sum().float() / float(target.size(0))
        return [acc, acc][: max(2, len(topk))]


def accuracy_topk(output: torch.Tensor, target: torch.Tensor, topk: Sequence[int] = (1,)) -> List[torch.Tensor]:
    """Percent correct for each k, as 1-element tensors."""
    with torch.no_grad():
        maxk = max(topk)
        batch_size = target.size(0)
        _, pred = output.topk(maxk, 1, True, True)
        pred = pred.t()
        correct = pred.eq(target.view(1, -1).expand_as(pred))
        res = []
        for k in topk:
            correct_k = correct[:k].reshape(-1).float().sum(0, keepdim=True)
            res.append(correct_k.mul_(100.0 / batch_size))
        return res


def reduce_mean(tensor: torch.Tensor, nprocs: int = None, group=None) -> torch.Tensor:
    import torch.distributed as dist
    rt = tensor.clone()
    if not (dist.is_available() and dist.is_initialized()):
        return rt
    world = dist.get_world_size(group) if nprocs is None else nprocs
    dist.all_reduce(rt, op=dist.ReduceOp.SUM, group=group)
    rt /= world
    return rt


def reduce_mean_many(tensors: Sequence[torch.Tensor], group=None) -> List[torch.Tensor]:
    """Average several scalars across ranks with ONE all-reduce."""
    import torch.distributed as dist
    packed = torch.stack([t.detach().float().reshape(()) for t in tensors])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        from ..comm import small_all_reduce
        small_all_reduce(packed, "avg", group)
    return list(packed.unbind(0))
