"""``broadcast_parameters`` / ``broadcast_optimizer_state`` / ``broadcast_object``
(/root/reference/5.horovod_distributed.py:116,122, /root/reference/5.2.horovod_pytorch_mnist.py:174-175).
"""
from __future__ import annotations

import collections
import io
import pickle

import torch

from . import mpi_ops as M


def broadcast_parameters(params, root_rank: int = 0) -> None:
    """Broadcast a ``state_dict()`` / ``named_parameters()`` / list of (name, tensor) from ``root_rank``:
    one named async broadcast per entry (parameters *and* buffers), then a single synchronize pass."""
    if isinstance(params, dict):
        items = sorted(params.items())
    elif isinstance(params, (list, tuple)) or hasattr(params, "__iter__"):
        items = list(params)
        if items and not isinstance(items[0], tuple):
            raise ValueError("invalid params: expected a dict or an iterable of (name, tensor)")
    else:
        raise ValueError(f"invalid params of type {type(params)}")
    handles = []
    for name, p in items:
        if p is None or not torch.is_tensor(p):
            continue
        t = p.data if isinstance(p, torch.nn.Parameter) else p
        if t.is_contiguous():
            handles.append((None, None, M.broadcast_async_(t, root_rank, name=f"{name}")))
        elif t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
            # dense but permuted storage (channels-last conv weights): broadcast the physical buffer
            flat = torch.as_strided(t, (t.numel(),), (1,))
            handles.append((None, None, M.broadcast_async_(flat, root_rank, name=f"{name}")))
        else:
            # a strided view (transposed / sliced tensor): the collective runs on a dense temporary that is
            # copied back into the view once it has arrived
            tmp = t.contiguous()
            handles.append((t, tmp, M.broadcast_async_(tmp, root_rank, name=f"{name}")))
    for dst, tmp, h in handles:
        M.synchronize(h)
        if dst is not None:
            dst.copy_(tmp)
    from ..ops.conv import bump_param_generation
    bump_param_generation()


def broadcast_object(obj, root_rank: int = 0, name: str = None):
    """Pickle on the root, broadcast length + bytes, unpickle elsewhere (objects of this job only)."""
    name = name or type(obj).__name__
    if M.size() == 1:
        return obj
    if M.rank() == root_rank:
        b = io.BytesIO()
        pickle.dump(obj, b)
        data = torch.frombuffer(bytearray(b.getvalue()), dtype=torch.uint8).clone()
        n = torch.tensor([data.numel()], dtype=torch.int64)
    else:
        n = torch.zeros(1, dtype=torch.int64)
        data = None
    M.broadcast_(n, root_rank, name=f"{name}.sz")
    if data is None:
        data = torch.zeros(int(n.item()), dtype=torch.uint8)
    M.broadcast_(data, root_rank, name=f"{name}.data")
    return pickle.loads(bytes(data.tolist())) if M.rank() != root_rank else obj


def broadcast_optimizer_state(optimizer, root_rank: int = 0) -> None:
    """Make every rank's optimizer state equal to ``root_rank``'s: state tensors (momentum buffers)
    are broadcast in place, scalar hyper-parameters travel wrapped in tensors.  If the state is
    still empty, one step with zero gradients materialises it first (as Horovod does)."""
    if isinstance(optimizer, torch.optim.LBFGS):
        raise ValueError("cannot broadcast torch.optim.LBFGS state")
    state_dict = optimizer.state_dict()
    if len(state_dict["state"]) == 0:
        saved = []
        for group in optimizer.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    saved.append((p, p.grad, p.data.clone()))
                    p.grad = torch.zeros_like(p)
        # a zero-gradient step may still move weights (weight decay): restore them afterwards
        step = getattr(optimizer, "_hvd_super_step", None) or optimizer.step
        step()
        for p, g, d in saved:
            p.grad = g
            p.data.copy_(d)
        state_dict = optimizer.state_dict()
    if len(state_dict["state"]) == 0:
        return  # stateless optimizer (plain SGD without momentum)
    handles, scalars = [], collections.OrderedDict()
    for gi, group in enumerate(state_dict["param_groups"]):
        for k, v in group.items():
            if k == "params":
                continue
            if isinstance(v, (int, float, bool)) and not isinstance(v, bool) or isinstance(v, bool):
                scalars[(gi, k)] = v
        for pid in group["params"]:
            if pid not in state_dict["state"]:
                continue
            for k, v in sorted(state_dict["state"][pid].items()):
                if torch.is_tensor(v):
                    flat = v if v.is_contiguous() else torch.as_strided(v, (v.numel(),), (1,))
                    handles.append(M.broadcast_async_(flat, root_rank, name=f"opt.{gi}.{pid}.{k}"))
                elif isinstance(v, (int, float)):
                    scalars[("state", pid, k)] = v
    for h in handles:
        M.synchronize(h)
    if scalars:
        keys = list(scalars.keys())
        t = torch.tensor([float(scalars[k]) for k in keys], dtype=torch.float64)
        M.broadcast_(t, root_rank, name="opt.scalars")
        for k, v in zip(keys, t.tolist()):
            orig = scalars[k]
            val = type(orig)(v) if not isinstance(orig, bool) else bool(v)
            if k[0] == "state":
                state_dict["state"][k[1]][k[2]] = val
            else:
                state_dict["param_groups"][k[0]][k[1]] = val
        optimizer.load_state_dict(state_dict)
