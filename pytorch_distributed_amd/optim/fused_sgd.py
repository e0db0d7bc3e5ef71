"""Momentum SGD as one multi-tensor HIP launch per parameter group.

Same update rule and ``state_dict`` layout as ``torch.optim.SGD`` (per-parameter
``momentum_buffer``; first step initialises the buffer with the gradient), i.e. the
optimizer every reference script builds at /root/reference/2.distributed.py:120
(lr 0.1, momentum 0.9, weight_decay 1e-4).  On top of that the kernel

* divides the gradients by the mixed-precision loss scale it reads from device
  memory (so no separate unscale pass is needed),
* skips the whole update when the overflow flag written by the gradient check is
  set (no host round trip: the decision is taken on the device),
* optionally writes the half-precision model copy of every fp32 master weight in
  the same pass (apex "O2" master weights).

Hyper-parameters live in a small device tensor per group so that changing the
learning rate (``adjust_learning_rate``) does not need a new kernel argument and
stays valid under hipGraph replay.
"""
from __future__ import annotations

from typing import Dict, List

import torch
from torch.optim import Optimizer

from ..ops import multi_tensor as mt, streams
from ..ops.conv import bump_param_generation


def effective_grad(p: torch.Tensor):
    """``(gradient tensor, factor still owed to it)`` the optimizer consumes for ``p``: normally
    ``(p.grad, 1.0)``; a DistributedDataParallel wrapper in fused-wire mode hands over a view of the
    all-reduced low-precision wire bucket instead (``p._pda_wire_grad``), so the reduced gradient is never
    cast back into ``p.grad``."""
    w = getattr(p, "_pda_wire_grad", None)
    if w is not None:
        return w
    return p.grad, 1.0


def _dense_like(p: torch.Tensor) -> bool:
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


class FusedSGD(Optimizer):
    def __init__(self, params, lr: float = 0.1, momentum: float = 0.0, dampening: float = 0.0,
                 weight_decay: float = 0.0, nesterov: bool = False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self._hp: Dict[int, torch.Tensor] = {}
        self._hp_host: Dict[int, tuple] = {}
        self._lists: Dict[int, tuple] = {}
        self._amp_scaler = None          # set by amp.initialize
        self._model_copies: Dict[int, torch.Tensor] = {}  # id(param) -> half copy updated in-kernel
        for g in self.param_groups:
            for p in g["params"]:
                # let this package's conv ops accumulate weight gradients straight into an existing .grad
                p._pda_direct_grad = True
                # a DistributedDataParallel wrapper may keep reduced gradients in its wire buckets for us
                p._pda_fused_opt = True
                ddp = getattr(p, "_pda_ddp", None)
                ddp = ddp() if ddp is not None else None
                if ddp is not None:
                    ddp.enable_fused_wire()

    # -- mixed precision hooks -------------------------------------------------------------
    def attach_scaler(self, scaler) -> None:
        self._amp_scaler = scaler

    def register_model_copy(self, param: torch.Tensor, copy: torch.Tensor) -> None:
        """Have ``step`` also write ``copy`` (same numel, fp16/bf16) = cast(param) in the update kernel."""
        assert copy.numel() == param.numel()
        self._model_copies[id(param)] = copy
        self._lists.clear()

    # -- gradient arena ---------------------------------------------------------------------
    def _build_arena(self) -> None:
        """Re-home every existing fp32 device gradient into one flat buffer per device, so that
        clearing the gradients is one memset instead of one fill kernel per parameter."""
        self._arena = {}
        self._arena_ptrs = []
        per_dev: Dict[torch.device, List[torch.Tensor]] = {}
        for g in self.param_groups:
            for p in g["params"]:
                gr = p.grad
                if gr is not None and gr.is_cuda and gr.dtype == torch.float32 and gr.shape == p.shape:
                    per_dev.setdefault(gr.device, []).append(p)
        for dev, ps in per_dev.items():
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64  # 256-byte aligned slots keep 16-byte vector access legal
            flat = torch.zeros(total, device=dev, dtype=torch.float32)
            for p, o in zip(ps, offs):
                # same logical shape AND memory format as the parameter (conv weights are channels-last)
                view = torch.as_strided(flat, p.shape, p.stride(), o) if _dense_like(p) else None
                if view is None:
                    continue
                p.grad = view
                self._arena_ptrs.append((p, view.data_ptr()))
            self._arena[dev] = flat
        self._lists.clear()

    def zero_grad(self, set_to_none: bool = False) -> None:
        """``set_to_none=False`` (the default here, unlike torch): gradients persist, which is what lets
        the conv / BatchNorm backward kernels accumulate into them directly."""
        if set_to_none:
            self._arena = None
            return super().zero_grad(set_to_none=True)
        arena = getattr(self, "_arena", None)
        if arena is not None and all(p.grad is not None and p.grad.data_ptr() == ptr for p, ptr in self._arena_ptrs):
            covered = {id(p) for p, _ in self._arena_ptrs}
            for flat in arena.values():
                flat.zero_()
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None and id(p) not in covered:
                        p.grad.zero_()
            return
        params = [p for g in self.param_groups for p in g["params"]]
        managed = any(getattr(p, "_pda_ddp", None) is not None for p in params)  # a DDP wrapper owns the gradient storage
        if not managed and any(p.grad is not None and p.grad.is_cuda for p in params):
            self._build_arena()  # fresh zeros
            covered = {id(p) for p, _ in self._arena_ptrs}
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None and id(p) not in covered:
                        p.grad.zero_()
            return
        super().zero_grad(set_to_none=False)

    # -- helpers ----------------------------------------------------------------------------
    def _group_hp(self, gi: int, group, device) -> torch.Tensor:
        host = (float(group["lr"]), float(group["momentum"]), float(group["dampening"]),
                float(group["weight_decay"]), 1.0 if group["nesterov"] else 0.0)
        if self._hp_host.get(gi) != host or gi not in self._hp or self._hp[gi].device != device:
            t = torch.tensor(host, dtype=torch.float32)
            if gi in self._hp and self._hp[gi].device == device:
                self._hp[gi].copy_(t, non_blocking=True)
            else:
                self._hp[gi] = t.to(device)
            self._hp_host[gi] = host
        return self._hp[gi]

    def sync_hyperparameters(self) -> None:
        """Push lr / momentum / weight decay of every group to their device tensors now.  ``step`` does
        this itself; a step replayed from a hipGraph does not run python, so schedulers call this after
        changing ``param_groups`` (the kernel reads the values from device memory)."""
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.is_cuda]
            if ps and gi in self._hp:
                self._group_hp(gi, group, ps[0].device)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        streams.join()   # weight gradients still in flight on the backward pass' side stream (no-op normally)
        scaler = self._amp_scaler
        dev_scale = noop = None
        if scaler is not None and scaler.enabled:
            dev_scale = scaler.scale_tensor if scaler.grads_scaled else None
            noop = scaler.overflow_flag
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            first = any("momentum_buffer" not in self.state[p] for p in ps) and group["momentum"] != 0
            # partition by (grad dtype, has model copy): one launch each
            buckets: Dict[tuple, List[torch.Tensor]] = {}
            eff = {}
            for p in ps:
                mc = self._model_copies.get(id(p))
                g, gs = effective_grad(p)
                eff[id(p)] = g
                key = (g.dtype, None if mc is None else mc.dtype, "momentum_buffer" in self.state[p], float(gs))
                buckets.setdefault(key, []).append(p)
            for key, plist in buckets.items():
                fresh = not key[2] and group["momentum"] != 0
                for p in plist:
                    if "momentum_buffer" not in self.state[p]:
                        self.state[p]["momentum_buffer"] = torch.zeros_like(p, dtype=torch.float32)
                ck = (gi, key, tuple(id(p) for p in plist), tuple(eff[id(p)].data_ptr() for p in plist))
                tl = self._lists.get(ck)
                if tl is None:
                    lists = [[eff[id(p)] for p in plist], list(plist),
                             [self.state[p]["momentum_buffer"] for p in plist]]
                    if key[1] is not None:
                        lists.append([self._model_copies[id(p)] for p in plist])
                    tl = mt.TensorLists(lists)
                    if len(self._lists) > 64:
                        self._lists.clear()
                    self._lists[ck] = tl
                mt.sgd(tl, self._group_hp(gi, group, plist[0].device), first_run=fresh, dev_scale=dev_scale,
                       noop_flag=noop, grad_scale=key[3])
        for group in self.param_groups:   # a wire-bucket hand-over is good for one update
            for p in group["params"]:
                if getattr(p, "_pda_wire_grad", None) is not None:
                    p._pda_wire_grad = None
        bump_param_generation(self._model_copies.values())
        if scaler is not None and scaler.enabled:
            scaler.after_step()
        return loss
