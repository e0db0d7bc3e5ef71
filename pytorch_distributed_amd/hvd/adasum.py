"""Adasum reduction (``op=hvd.Adasum``, /root/reference/5.2.horovod_pytorch_mnist.py:30,184).

Pairwise rule: ``adasum(a, b) = (1 - a.b / (2|a|^2)) a + (1 - a.b / (2|b|^2)) b`` - orthogonal
gradients add, parallel gradients average - applied over a binary tree of ranks (rank pairs at distance
1, then 2, 4, ...).  ``size()`` must be a power of two, as in Horovod.

Device tensors run vector-halving / distance-doubling (:func:`adasum_vhdd_`): at distance ``d`` a rank
trades half of its current segment with ``rank ^ d`` (one send/recv pair), so both hold the same piece of
the two operands; ``a.b``, ``|a|^2`` and ``|b|^2`` are taken by a HIP kernel on the piece
(``csrc/kernels/adasum.hip``, fp64 accumulation), the per-rank partials are all-gathered (3 doubles per
rank) and a second kernel sums those of the rank's block, forms the coefficients and combines the piece in
place.  After ``log2(size)`` levels every rank owns ``1/size`` of the result; one all-gather plus a
bit-reversal permutation rebuilds the vector.  Each rank moves about two vector lengths in total, where
gathering all ``size()`` vectors moves ``size()`` of them.

CPU tensors (and device tensors without device collectives) gather the ``size()`` tensors and evaluate the
same tree on every rank in fp64.
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


def _bitrev(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def _exchange(coll, send: torch.Tensor, recv: torch.Tensor, peer: int, rank: int, world: int) -> None:
    if hasattr(coll, "sendrecv"):
        coll.sendrecv(send, recv, peer)
        return
    # a backend without point-to-point calls: every rank sends the same count at a level, so an all-gather of the
    # pieces carries the partner's one too (correctness path for the peer-memory test backend)
    allp = torch.empty(world * send.numel(), device=send.device, dtype=send.dtype)
    coll.all_gather(allp, send.contiguous())
    recv.copy_(allp[peer * send.numel():(peer + 1) * send.numel()])


def adasum_vhdd_(tensor: torch.Tensor, coll, rank: int, world: int) -> torch.Tensor:
    """In-place Adasum of a dense device tensor over ``coll`` (``all_gather`` and, ideally, ``sendrecv``)."""
    from ..ops import _lib
    l = _lib.lib()
    dev, dt = tensor.device, tensor.dtype
    n = tensor.numel()
    flat = torch.as_strided(tensor, (n,), (1,))
    # work on a copy whose length divides by world and keeps every piece 16-byte aligned
    unit = world * (16 // tensor.element_size())
    npad = (n + unit - 1) // unit * unit
    work = torch.zeros(npad, device=dev, dtype=dt)
    work[:n].copy_(flat)
    levels = world.bit_length() - 1
    start, length = 0, npad
    parts = torch.empty(3, device=dev, dtype=torch.float64)
    gathered = torch.empty(world * 3, device=dev, dtype=torch.float64)
    recv = torch.empty(npad // 2, device=dev, dtype=dt)
    st = _lib.stream
    for lv in range(levels):
        d = 1 << lv
        peer = rank ^ d
        half = length // 2
        upper = (rank & d) != 0
        keep_off = start + (half if upper else 0)
        send_off = start + (0 if upper else half)
        mine = work[keep_off:keep_off + half]
        theirs = recv[:half]
        _exchange(coll, work[send_off:send_off + half], theirs, peer, rank, world)
        a, b = (theirs, mine) if upper else (mine, theirs)     # a: the operand of the lower sub-block
        parts.zero_()
        _lib.check(l.pda_adasum_dots(_lib.dt(dt), a.data_ptr(), b.data_ptr(), half, parts.data_ptr(), st()),
                   "pda_adasum_dots")
        coll.all_gather(gathered, parts)
        blk = rank & ~(2 * d - 1)
        _lib.check(l.pda_adasum_combine(_lib.dt(dt), a.data_ptr(), b.data_ptr(), mine.data_ptr(), half,
                                        gathered.data_ptr(), blk, 2 * d, st()), "pda_adasum_combine")
        start, length = keep_off, half
    piece = work[start:start + length]
    full = torch.empty(npad, device=dev, dtype=dt)
    coll.all_gather(full, piece.contiguous())
    # rank r ended up with piece number bitrev(r): put the pieces back in vector order
    perm = torch.tensor([_bitrev(p, levels) for p in range(world)], device=dev)
    ordered = full.view(world, length).index_select(0, perm).reshape(-1)
    flat.copy_(ordered[:n])
    return tensor


def adasum_allreduce_(tensor: torch.Tensor) -> torch.Tensor:
    """Host path (CPU tensors, or no engine): gather the ``size()`` tensors and evaluate the tree in fp64.
    Device tensors are routed to :func:`adasum_vhdd_` by the engine (``hvd/mpi_ops.py``)."""
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
