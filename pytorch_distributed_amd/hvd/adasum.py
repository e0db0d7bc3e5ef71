"""Adasum reduction (``op=hvd.Adasum``, /root/reference/5.2.horovod_pytorch_mnist.py:30,184).

Pairwise rule: ``adasum(a, b) = (1 - a.b / (2|a|^2)) a + (1 - a.b / (2|b|^2)) b`` - orthogonal
gradients add, parallel gradients average - applied over a binary tree of ranks.  This
implementation gathers the ``size()`` tensors and evaluates the same tree on every rank
(deterministic, identical results everywhere); it favours correctness over wire efficiency, which
is enough for the small models the reference uses it with.  ``size()`` must be a power of two,
as in Horovod.
"""
from __future__ import annotations

import torch


def adasum_pair(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    af, bf = a.double().flatten(), b.double().flatten()
    dot = torch.dot(af, bf)
    na, nb = torch.dot(af, af), torch.dot(bf, bf)
    ca = 1.0 - (dot / (2.0 * na) if na > 0 else 0.0)
    cb = 1.0 - (dot / (2.0 * nb) if nb > 0 else 0.0)
    return (ca * a.double() + cb * b.double()).to(a.dtype)


def adasum_tree(tensors):
    level = list(tensors)
    while len(level) > 1:
        level = [adasum_pair(level[i], level[i + 1]) for i in range(0, len(level), 2)]
    return level[0]


def adasum_allreduce_(tensor: torch.Tensor) -> torch.Tensor:
    from . import mpi_ops as M
    world = M.size()
    if world & (world - 1):
        raise NotImplementedError("Adasum needs a power-of-two number of ranks")
    if world == 1:
        return tensor
    gathered = M.allgather(tensor.reshape(1, -1))
    res = adasum_tree([gathered[r].view_as(tensor) for r in range(world)])
    tensor.copy_(res)
    return tensor
