"""Single-process multi-GPU ``DataParallel``: scatter / replicate / parallel_apply / gather.

The reference's first launch style, ``nn.DataParallel(model, device_ids=gpus, output_device=gpus[0])``
(/root/reference/1.dataparallel.py:109,222-233).  Per forward:

1. *scatter*   - the batch is chunked along ``dim`` and each chunk is copied to its device over
   xGMI peer-to-peer on a per-device copy stream (``wait_stream`` + ``record_stream`` hand-off);
2. *replicate* - parameters and buffers of the master copy (on ``device_ids[0]``) are packed into
   one flat buffer per dtype (one multi-tensor launch); ONE kernel on the master device then stores
   the buffer into every replica's device over xGMI peer access (``csrc/kernels/peer.hip``: the source
   is read once, all outbound links carry data at the same time), and every replica's tensors are
   *views* of its device's buffer (no unflatten copies);
3. *parallel_apply* - one python thread per replica, each bound to its device;
4. *gather*    - outputs are copied to ``output_device`` and concatenated.

Backward is autograd through the same graph: the replicate node's backward is the
reduce-add of every replica's flat gradient buffer onto the master device - one N-way ``add_n``
launch on the master that reads the replicas' buffers directly through their peer mappings (no
staging copies) - i.e. torch's ``ReduceAddCoalesced``.  Device pairs without a peer path (and CPU
"devices") take runtime copies instead.
As upstream, BatchNorm statistics are per replica and only replica 0 (the master module itself)
keeps its running statistics.  Devices may be repeated or be ``"cpu"``, which is how the CPU
test-suite exercises the whole path without GPUs.
"""
from __future__ import annotations

import threading
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from ..ops import multi_tensor as mt


def _dev(d) -> torch.device:
    if isinstance(d, torch.device):
        return d
    if isinstance(d, int):
        return torch.device("cuda", d)
    return torch.device(d)


def _physical_flat(t: torch.Tensor) -> torch.Tensor:
    return torch.as_strided(t, (t.numel(),), (1,))


def _is_dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


# ----------------------------------------------------------------------------------- comm primitives
_COPY_STREAMS: Dict[torch.device, "torch.cuda.Stream"] = {}


def _copy_stream(dev: torch.device):
    if dev not in _COPY_STREAMS:
        _COPY_STREAMS[dev] = torch.cuda.Stream(dev)
    return _COPY_STREAMS[dev]


_PEER_OK: Dict[tuple, bool] = {}


def _peer_ok(a: torch.device, b: torch.device) -> bool:
    """Kernels running on ``a`` may dereference pointers into ``b``'s memory (peer access enabled on demand)."""
    if a.type != "cuda" or b.type != "cuda":
        return False
    ia = a.index if a.index is not None else torch.cuda.current_device()
    ib = b.index if b.index is not None else torch.cuda.current_device()
    key = (ia, ib)
    if key not in _PEER_OK:
        from ..ops import _lib
        torch.cuda.init()
        _PEER_OK[key] = _lib.lib().pda_peer_enable(ia, ib) == 0
    return _PEER_OK[key]


def _peer_broadcast(flat: torch.Tensor, devices: Sequence[torch.device]) -> Optional[List[torch.Tensor]]:
    """``flat`` (on its own device) -> one copy per entry of ``devices`` with a single direct-store kernel;
    None when some pair has no peer path (the caller copies through the runtime then)."""
    import ctypes
    from ..ops import _lib
    src = flat.device
    if src.type != "cuda" or not all(_peer_ok(src, d) for d in devices):
        return None
    outs, ptrs = [], []
    for d in devices:
        if d == src:
            outs.append(flat)
        else:
            o = torch.empty_like(flat, device=d)
            outs.append(o)
            ptrs.append(o.data_ptr())
    for k in range(0, len(ptrs), 8):
        part = ptrs[k:k + 8]
        arr = (ctypes.c_void_p * len(part))(*part)
        with torch.cuda.device(src):
            _lib.check(_lib.lib().pda_peer_bcast(flat.data_ptr(), arr, len(part), flat.numel() * flat.element_size(),
                                                 torch.cuda.current_stream(src).cuda_stream), "pda_peer_bcast")
    if ptrs:
        # the replicas' streams must not read their copy before the store kernel has finished
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(src))
        for d in set(devices):
            if d != src:
                torch.cuda.current_stream(d).wait_event(ev)
    return outs


def scatter(tensor: torch.Tensor, devices: Sequence, dim: int = 0) -> List[torch.Tensor]:
    """Chunk ``tensor`` along ``dim`` and copy chunk i to ``devices[i]`` (differentiable)."""
    devices = [_dev(d) for d in devices]
    chunks = tensor.chunk(len(devices), dim)
    out = []
    for c, d in zip(chunks, devices):
        if d.type == "cuda" and c.device != d and not c.requires_grad:
            s = _copy_stream(d)
            s.wait_stream(torch.cuda.current_stream(c.device) if c.is_cuda else torch.cuda.current_stream(d))
            with torch.cuda.stream(s):
                moved = c.to(d, non_blocking=True)
            torch.cuda.current_stream(d).wait_stream(s)
            moved.record_stream(torch.cuda.current_stream(d))
            out.append(moved)
        else:
            out.append(c.to(d, non_blocking=True) if c.device != d else c)
    return out


def scatter_kwargs(inputs, kwargs, devices, dim: int = 0):
    n = len(devices)

    def scat(obj):
        if torch.is_tensor(obj):
            parts = scatter(obj, devices, dim)
            return parts + [parts[-1][:0]] * (n - len(parts)) if len(parts) < n else parts
        if isinstance(obj, (list, tuple)) and obj:
            cols = list(zip(*[scat(o) for o in obj]))
            return [type(obj)(c) for c in cols]
        if isinstance(obj, dict) and obj:
            cols = list(zip(*[scat(v) for v in obj.values()]))
            return [dict(zip(obj.keys(), c)) for c in cols]
        return [obj] * n
    ins = scat(tuple(inputs)) if inputs else [()] * n
    kws = scat(kwargs) if kwargs else [{}] * n
    return ins, kws


def gather(outputs: Sequence, target, dim: int = 0):
    target = _dev(target)

    def g(outs):
        o0 = outs[0]
        if torch.is_tensor(o0):
            moved = [o.to(target, non_blocking=True) for o in outs]
            if o0.dim() == 0:
                return torch.stack(moved)
            return torch.cat(moved, dim)
        if isinstance(o0, dict):
            return type(o0)((k, g([o[k] for o in outs])) for k in o0)
        if isinstance(o0, (list, tuple)):
            return type(o0)(g(list(c)) for c in zip(*outs))
        return o0
    return g(list(outputs))


def broadcast_coalesced(tensors: Sequence[torch.Tensor], devices: Sequence) -> List[List[torch.Tensor]]:
    """Copies of ``tensors`` on every device, through one flat buffer per dtype (no autograd)."""
    devices = [_dev(d) for d in devices]
    out = [[None] * len(tensors) for _ in devices]
    by_dtype: Dict[torch.dtype, List[int]] = {}
    for i, t in enumerate(tensors):
        by_dtype.setdefault(t.dtype, []).append(i)
    for dt, idxs in by_dtype.items():
        srcs = [_physical_flat(tensors[i].detach()) if _is_dense(tensors[i]) else tensors[i].detach().reshape(-1)
                for i in idxs]
        flat = mt.flatten(srcs) if dt in (torch.float32, torch.float16, torch.bfloat16) else torch.cat(srcs)
        direct = _peer_broadcast(flat, devices) if flat.is_cuda and flat.data_ptr() % 16 == 0 else None
        for di, d in enumerate(devices):
            if direct is not None:
                fd = direct[di]
            else:
                fd = flat if di == 0 and flat.device == d else flat.to(d, non_blocking=True)
            off = 0
            for i in idxs:
                t = tensors[i]
                if _is_dense(t):
                    out[di][i] = torch.as_strided(fd, t.shape, t.stride(), off)
                else:
                    out[di][i] = fd[off:off + t.numel()].view(t.shape)
                off += t.numel()
    return out


def reduce_add_coalesced(per_device: Sequence[Sequence[torch.Tensor]], target) -> List[torch.Tensor]:
    """Sum tensor lists living on several devices onto ``target``: one flat peer copy per device and
    one fused accumulation per copy."""
    target = _dev(target)
    n = len(per_device[0])
    first = [t if t.device == target else t.to(target) for t in per_device[0]]
    # a gradient that is a strided slice of something larger (the stem's dW is cut out of a padded buffer)
    # has no physical order to add in: bring it to plain contiguous layout first
    first = [t if _is_dense(t) else t.contiguous() for t in first]

    def flat_like(t, ref):
        """1-D view/copy of t whose element order is ref's PHYSICAL order (so flats can be added blindly)."""
        if _is_dense(t) and t.stride() == ref.stride():
            return t.reshape(-1) if t.is_contiguous() else _physical_flat(t)
        same = torch.empty_strided(ref.shape, ref.stride(), dtype=t.dtype, device=t.device)
        same.copy_(t)
        return _physical_flat(same)

    acc = mt.flatten([flat_like(t, t) for t in first])
    flats = []
    for lst in per_device[1:]:
        flat = mt.flatten([flat_like(t, r) for t, r in zip(lst, first)])
        if flat.device != target and _peer_ok(target, flat.device) and flat.data_ptr() % 16 == 0:
            # leave it where it is: the reduce kernel on the target reads it through the peer mapping, once the
            # replica's stream has produced it
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            torch.cuda.current_stream(target).wait_event(ev)
            flat.record_stream(torch.cuda.current_stream(target))
            flats.append(flat)
        else:
            flats.append(flat.to(target, non_blocking=True))   # no peer path: one runtime copy per replica
    if flats and acc.is_cuda and acc.dtype in (torch.float32, torch.float16, torch.bfloat16):
        from ..ops.elementwise import add_n
        add_n([acc] + flats, out=acc)                        # N-way reduce-add: every buffer is read once
    else:
        for flat in flats:
            mt.axpby(mt.TensorLists([[acc], [flat], [acc]]), 1.0, 1.0)
    res, off = [], 0
    for t in first:
        if _is_dense(t):
            res.append(torch.as_strided(acc, t.shape, t.stride(), off))
        else:
            res.append(acc[off:off + t.numel()].view(t.shape))
        off += t.numel()
    assert len(res) == n
    return res


class _Broadcast(torch.autograd.Function):
    """Differentiable replicate: forward = broadcast_coalesced, backward = reduce_add_coalesced."""

    @staticmethod
    def forward(ctx, devices, *params):
        ctx.devices = devices
        ctx.n = len(params)
        ctx.src = params[0].device if params else None
        copies = broadcast_coalesced(params, devices)
        flat = tuple(t for dev_list in copies for t in dev_list)
        return flat

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        per_dev = []
        for di in range(len(ctx.devices)):
            gl = list(grads[di * n:(di + 1) * n])
            per_dev.append(gl)
        # a replica may not have touched a parameter: treat its gradient as zero
        for gl in per_dev:
            for i, g in enumerate(gl):
                if g is None:
                    ref = next(x[i] for x in per_dev if x[i] is not None)
                    gl[i] = torch.zeros_like(ref, device=gl[0].device if gl[0] is not None else ref.device)
        summed = reduce_add_coalesced(per_dev, ctx.src)
        return (None,) + tuple(summed)


# ----------------------------------------------------------------------------------- replicate / apply
def replicate(module: nn.Module, devices: Sequence) -> List[nn.Module]:
    """Replica 0 is ``module`` itself; replicas 1.. are shallow module copies whose parameters are
    autograd-connected broadcast copies and whose buffers are plain copies."""
    devices = [_dev(d) for d in devices]
    if len(devices) == 1:
        return [module]
    peers = devices[1:]
    params = [p for p in module.parameters()]
    buffers = [b for b in module.buffers()]
    pcopies = _Broadcast.apply(peers, *params) if params else ()
    bcopies = broadcast_coalesced(buffers, peers) if buffers else [[] for _ in peers]
    pidx = {id(p): i for i, p in enumerate(params)}
    bidx = {id(b): i for i, b in enumerate(buffers)}
    mods = list(module.modules())
    midx = {id(m): i for i, m in enumerate(mods)}
    replicas = [module]
    for di, _ in enumerate(peers):
        copies = [m._replicate_for_data_parallel() for m in mods]
        for m, c in zip(mods, copies):
            for k, child in m._modules.items():
                c._modules[k] = None if child is None else copies[midx[id(child)]]
            for k, p in m._parameters.items():
                if p is None:
                    c._parameters[k] = None
                else:
                    setattr(c, k, pcopies[di * len(params) + pidx[id(p)]])  # non-leaf tensor attribute
            for k, b in m._buffers.items():
                c._buffers[k] = None if b is None else bcopies[di][bidx[id(b)]]
            if hasattr(m, "_cache"):  # per-replica low-precision weight caches
                c._cache = type(m._cache)()
            if "_pda_cache" in c.__dict__:  # (Linear: same)
                c.__dict__["_pda_cache"] = type(m.__dict__["_pda_cache"])()
        replicas.append(copies[0])
    return replicas


def parallel_apply(replicas: Sequence[nn.Module], inputs, kwargs_tup, devices) -> list:
    devices = [_dev(d) for d in devices]
    results = [None] * len(replicas)
    grad_enabled, autocast = torch.is_grad_enabled(), torch.is_autocast_enabled()

    def worker(i):
        try:
            torch.set_grad_enabled(grad_enabled)
            d = devices[i]
            if d.type == "cuda":
                with torch.cuda.device(d), torch.autocast("cuda", enabled=autocast):
                    results[i] = replicas[i](*inputs[i], **kwargs_tup[i])
            else:
                results[i] = replicas[i](*inputs[i], **kwargs_tup[i])
        except Exception as e:  # noqa: BLE001 - re-raised in the caller's thread
            results[i] = e
    if len(replicas) == 1:
        worker(0)
    else:
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(replicas))]
        [t.start() for t in ths]
        [t.join() for t in ths]
    for r in results:
        if isinstance(r, Exception):
            raise r
    return results


class DataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids: Optional[Sequence] = None, output_device=None, dim: int = 0):
        super().__init__()
        if device_ids is None:
            device_ids = list(range(torch.cuda.device_count())) if torch.cuda.is_available() else ["cpu"]
        self.module = module
        self.device_ids = [_dev(d) for d in device_ids]
        self.output_device = _dev(output_device) if output_device is not None else self.device_ids[0]
        self.dim = dim
        self.src_device = self.device_ids[0]

    def forward(self, *inputs, **kwargs):
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            if t.device != self.src_device:
                raise RuntimeError(f"module must have its parameters and buffers on device {self.src_device} "
                                   f"(device_ids[0]) but found one of them on device: {t.device}")
        if len(self.device_ids) == 1:
            return self.module(*inputs, **kwargs)
        ins, kws = scatter_kwargs(inputs, kwargs, self.device_ids, self.dim)
        used = [i for i in range(len(self.device_ids))
                if not (ins[i] and torch.is_tensor(ins[i][0]) and ins[i][0].shape[self.dim] == 0)]
        devs = [self.device_ids[i] for i in used]
        replicas = replicate(self.module, devs)
        outs = parallel_apply(replicas, [ins[i] for i in used], [kws[i] for i in used], devs)
        return gather(outs, self.output_device, self.dim)
