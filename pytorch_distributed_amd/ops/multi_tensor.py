"""Multi-tensor apply: one kernel launch over a whole list of tensors.

``TensorLists`` caches the device-side pointer/chunk tables for a fixed set of
tensor lists (parameters, gradients, momentum buffers ...), so the steady-state
cost of a fused optimizer step or a gradient unscale is a single launch with no
host-side table building.  CPU tensors take an equivalent ``torch._foreach``/loop
path so the same call sites work in the CPU test-suite.

Reference: apex ``amp_C.multi_tensor_scale/axpby/l2norm`` reached through
``amp.scale_loss`` (/root/reference/4.apex_distributed2.py:289-291), apex_C
flatten/unflatten inside apex DDP (:178), and ``torch.optim.SGD.step``
(/root/reference/2.distributed.py:232).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib

_CHUNK = 32768


class TensorLists:
    """Device tables for L parallel lists of equally-shaped tensors."""

    def __init__(self, lists: Sequence[Sequence[torch.Tensor]]):
        self.lists = [list(l) for l in lists]
        n = len(self.lists[0])
        assert all(len(l) == n for l in self.lists), "tensor lists must have equal length"
        for i in range(n):
            ne = self.lists[0][i].numel()
            for l in self.lists:
                assert l[i].numel() == ne, "corresponding tensors must have equal numel"
                assert l[i].is_contiguous() or _dense(l[i]), "multi-tensor ops need dense tensors"
        self.device = self.lists[0][0].device if n else torch.device("cpu")
        self.is_cuda = self.device.type == "cuda"
        self._meta = None
        self._blk = None
        self._ptrs = None
        self.nblocks = 0

    def dtypes(self):
        return [l[0].dtype if l else torch.float32 for l in self.lists]

    def freeze(self) -> "TensorLists":
        """Build the device tables now and drop the references to the tensors (only one empty tensor per list
        stays, for its dtype).  A cache of frozen tables keyed by the tensors' addresses does not keep gradient
        storage alive; the caller guarantees the addresses are live whenever the tables are used."""
        if self.is_cuda and self.lists[0]:
            self.tables()
            self.lists = [[l[0].new_empty(0)] for l in self.lists]
            self._frozen = True
        return self

    def tables(self):
        if getattr(self, "_frozen", False):
            return self._meta, self._blk, self.nblocks
        ptrs = tuple(t.data_ptr() for l in self.lists for t in l)
        if self._meta is None or ptrs != self._ptrs:
            L = len(self.lists)
            meta, blk = [], []
            for i, t0 in enumerate(self.lists[0]):
                ne = t0.numel()
                meta.extend(l[i].data_ptr() for l in self.lists)
                meta.append(ne)
                for c in range((ne + _CHUNK - 1) // _CHUNK):
                    blk.extend((i, c))
            self._meta = torch.tensor(meta, dtype=torch.int64).view(-1, L + 1).to(self.device)
            self._blk = torch.tensor(blk, dtype=torch.int32).view(-1, 2).to(self.device)
            self.nblocks = len(blk) // 2
            self._ptrs = ptrs
        return self._meta, self._blk, self.nblocks


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.is_contiguous()


def _flat(t: torch.Tensor) -> torch.Tensor:
    """1-D alias over a dense tensor's memory, whatever its (dense) stride order."""
    return torch.as_strided(t, (t.numel(),), (1,))


def scale(tl: TensorLists, scale_value: float = 1.0, dev_scale: Optional[torch.Tensor] = None,
          invert: bool = False, noop_flag: Optional[torch.Tensor] = None, check_only: bool = False) -> None:
    """lists[1][i] = lists[0][i] * scale (* dev_scale or / dev_scale); flags non-finite inputs.

    ``check_only`` reads and flags but stores nothing (the overflow check in front of a fused optimizer)."""
    src, dst = tl.lists[0], tl.lists[1]
    if not src:
        return
    if tl.is_cuda:
        meta, blk, nb = tl.tables()
        l = _lib.lib()
        _lib.check(l.pda_mt_scale(_lib.dt(src[0]), _lib.dt(dst[0]), meta.data_ptr(), blk.data_ptr(), nb,
                                  float(scale_value), _lib.ptr(dev_scale), int(bool(invert)) | (2 if check_only else 0),
                                  _lib.ptr(noop_flag), _lib.stream()), "pda_mt_scale")
        return
    s = float(scale_value)
    if dev_scale is not None:
        s = s / float(dev_scale) if invert else s * float(dev_scale)
    bad = False
    for a, b in zip(src, dst):
        af = _flat(a).float()
        bad = bad or not bool(torch.isfinite(af).all())
        if check_only:
            continue
        _flat(b).copy_(af * s)
        if b.dtype != a.dtype and b.element_size() < a.element_size():
            bad = bad or not bool(torch.isfinite(_flat(b).float()).all())
    if bad and noop_flag is not None:
        noop_flag.fill_(1)


def axpby(tl: TensorLists, a: float, b: float, noop_flag: Optional[torch.Tensor] = None) -> None:
    """lists[2][i] = a * lists[0][i] + b * lists[1][i]."""
    xs, ys, outs = tl.lists
    if not xs:
        return
    if tl.is_cuda:
        meta, blk, nb = tl.tables()
        l = _lib.lib()
        _lib.check(l.pda_mt_axpby(_lib.dt(xs[0]), _lib.dt(ys[0]), _lib.dt(outs[0]), meta.data_ptr(), blk.data_ptr(), nb,
                                  float(a), float(b), _lib.ptr(noop_flag), _lib.stream()), "pda_mt_axpby")
        return
    bad = False
    for x, y, o in zip(xs, ys, outs):
        xf, yf = _flat(x).float(), _flat(y).float()
        bad = bad or not bool(torch.isfinite(xf).all() and torch.isfinite(yf).all())
        _flat(o).copy_(a * xf + b * yf)
    if bad and noop_flag is not None:
        noop_flag.fill_(1)


def l2norm(tl: TensorLists, per_tensor: bool = False):
    """-> (global L2 norm [0-dim tensor], per-tensor norms or None)."""
    xs = tl.lists[0]
    n = len(xs)
    if tl.is_cuda and n:
        meta, blk, nb = tl.tables()
        l = _lib.lib()
        out = torch.zeros(1 + n, device=tl.device, dtype=torch.float32)
        _lib.check(l.pda_mt_l2norm(_lib.dt(xs[0]), meta.data_ptr(), blk.data_ptr(), nb, out.data_ptr(),
                                   int(per_tensor), _lib.stream()), "pda_mt_l2norm")
        _lib.check(l.pda_sqrt_inplace(out.data_ptr(), 1 + n, _lib.stream()), "pda_sqrt_inplace")
        return out[0], (out[1:] if per_tensor else None)
    sq = [(_flat(x).float() ** 2).sum() for x in xs]
    tot = torch.sqrt(torch.stack(sq).sum()) if sq else torch.zeros(())
    return tot, (torch.sqrt(torch.stack(sq)) if per_tensor and sq else None)


def sgd(tl: TensorLists, hp: torch.Tensor, first_run: bool, dev_scale: Optional[torch.Tensor] = None,
        noop_flag: Optional[torch.Tensor] = None, grad_scale: float = 1.0) -> None:
    """Fused momentum-SGD over lists (grads, fp32 params, momentum buffers[, model copies]).

    ``hp`` = float tensor {lr, momentum, dampening, weight_decay, nesterov} on the
    parameters' device; ``dev_scale`` the loss scale the grads still carry;
    ``noop_flag`` non-zero skips the whole update (gradient overflow); ``grad_scale`` is an extra
    factor on the gradient (the 1/world of a SUM all-reduce consumed straight from its wire buffer).
    """
    grads, params, bufs = tl.lists[0], tl.lists[1], tl.lists[2]
    models = tl.lists[3] if len(tl.lists) > 3 else None
    if not grads:
        return
    if tl.is_cuda:
        meta, blk, nb = tl.tables()
        l = _lib.lib()
        _lib.check(l.pda_mt_sgd(_lib.dt(grads[0]), _lib.dt(models[0]) if models else -1, meta.data_ptr(),
                                blk.data_ptr(), nb, hp.data_ptr(), _lib.ptr(dev_scale), _lib.ptr(noop_flag),
                                int(first_run), float(grad_scale), _lib.stream()), "pda_mt_sgd")
        return
    if noop_flag is not None and int(noop_flag) != 0:
        return
    lr, mom, damp, wd, nest = [float(v) for v in hp.tolist()]
    inv = (1.0 / float(dev_scale) if dev_scale is not None else 1.0) * float(grad_scale)
    for i, (g, p, m) in enumerate(zip(grads, params, bufs)):
        gf = _flat(g).float() * inv + wd * _flat(p)
        if mom != 0.0:
            mf = _flat(m)
            if first_run:
                mf.copy_(gf)
            else:
                mf.mul_(mom).add_(gf, alpha=1.0 - damp)
            gf = gf + mom * mf if nest else mf
        _flat(p).add_(gf, alpha=-lr)
        if models:
            _flat(models[i]).copy_(_flat(p))


def scaler_update(state: torch.Tensor, noop_flag: torch.Tensor, growth: float, backoff: float, interval: int,
                  min_scale: float, max_scale: float) -> None:
    """Dynamic loss-scale state machine; state = {scale, growth_tracker, skipped, last_overflow}."""
    if state.is_cuda:
        l = _lib.lib()
        _lib.check(l.pda_scaler_update(state.data_ptr(), noop_flag.data_ptr(), float(growth), float(backoff),
                                       int(interval), float(min_scale), float(max_scale), _lib.stream()),
                   "pda_scaler_update")
        return
    overflow = int(noop_flag) != 0
    sc, tr = float(state[0]), float(state[1])
    if overflow:
        sc = max(sc * backoff, min_scale)
        tr = 0.0
        state[2] += 1
    else:
        tr += 1
        if tr >= interval:
            sc = min(sc * growth, max_scale)
            tr = 0.0
    state[0], state[1], state[3] = sc, tr, float(overflow)
    noop_flag.zero_()


def flatten(tensors: List[torch.Tensor], out: Optional[torch.Tensor] = None, dtype=None) -> torch.Tensor:
    """Pack tensors into one contiguous 1-D buffer (with cast) in a single launch."""
    total = sum(t.numel() for t in tensors)
    dtype = dtype or tensors[0].dtype
    if out is None:
        out = torch.empty(total, device=tensors[0].device, dtype=dtype)
    views, off = [], 0
    for t in tensors:
        views.append(out[off:off + t.numel()])
        off += t.numel()
    scale(TensorLists([tensors, views]), 1.0)
    return out


def unflatten(flat: torch.Tensor, like: List[torch.Tensor]) -> None:
    """Scatter a flat buffer back into ``like`` (with cast), one launch."""
    views, off = [], 0
    for t in like:
        views.append(flat[off:off + t.numel()])
        off += t.numel()
    scale(TensorLists([views, like]), 1.0)
