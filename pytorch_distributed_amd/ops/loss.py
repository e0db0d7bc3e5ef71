"""Softmax cross-entropy with device-resident metrics.

``cross_entropy(logits, target)`` returns the mean loss as a 0-dim tensor; the GPU
kernel writes d(logits) in the same launch and accumulates
``{loss_sum, correct@1, correct@5, rows}`` into a 4-float ``metrics`` tensor, so a
training loop can all-reduce one small vector per step (or per epoch) instead of
the reference's barrier + three scalar all-reduces + four ``.item()`` syncs
between forward and backward (/root/reference/2.distributed.py:217-227).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib


class _SoftmaxCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, metrics, reduction):
        l = _lib.lib()
        logits = logits.contiguous()
        B, NC = logits.shape
        dlogits = torch.empty_like(logits)
        row_loss = torch.empty((B,), device=logits.device, dtype=torch.float32)
        gs = 1.0 / B if reduction == "mean" else 1.0
        _lib.check(l.pda_softmax_ce(_lib.dt(logits), logits.data_ptr(), target.data_ptr(), B, NC, gs,
                                    dlogits.data_ptr(), row_loss.data_ptr(), _lib.ptr(metrics), 0, _lib.stream()),
                   "pda_softmax_ce")
        ctx.save_for_backward(dlogits)
        s = row_loss.sum()
        return s / B if reduction == "mean" else s

    @staticmethod
    def backward(ctx, gout):
        l = _lib.lib()
        (saved,) = ctx.saved_tensors
        dlogits = saved.clone()  # keep the saved unit gradient intact for a second backward
        g = gout.reshape(1).float().contiguous()  # stays on device: no host sync for the loss scale
        _lib.check(l.pda_scale_by_devscalar(_lib.dt(dlogits), dlogits.data_ptr(), g.data_ptr(), dlogits.numel(),
                                            _lib.stream()), "pda_scale_by_devscalar")
        return dlogits, None, None, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, metrics: Optional[torch.Tensor] = None,
                  reduction: str = "mean") -> torch.Tensor:
    """Mean (or summed) softmax cross-entropy; math in fp32 whatever the logits dtype."""
    if not logits.is_cuda:
        lf = logits.float()
        if metrics is not None:
            update_metrics_ref(lf.detach(), target, metrics)
        return F.cross_entropy(lf, target, reduction=reduction)
    return _SoftmaxCEFn.apply(logits, target, metrics, reduction)


def update_metrics_ref(logits: torch.Tensor, target: torch.Tensor, metrics: torch.Tensor) -> None:
    with torch.no_grad():
        loss_sum = F.cross_entropy(logits, target, reduction="sum")
        zt = logits.gather(1, target.view(-1, 1))
        idx = torch.arange(logits.shape[1], device=logits.device).view(1, -1)
        greater = ((logits > zt) | ((logits == zt) & (idx < target.view(-1, 1)))).sum(1)
        metrics += torch.stack([loss_sum, (greater < 1).sum().float(), (greater < 5).sum().float(),
                                torch.tensor(float(logits.shape[0]), device=logits.device)])


class _LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        l = _lib.lib()
        logits = logits.contiguous()
        B, NC = logits.shape
        out = torch.empty_like(logits)
        dummy_t = torch.zeros((B,), device=logits.device, dtype=torch.int64)
        _lib.check(l.pda_softmax_ce(_lib.dt(logits), logits.data_ptr(), dummy_t.data_ptr(), B, NC, 1.0, 0, 0, 0,
                                    out.data_ptr(), _lib.stream()), "pda_softmax_ce(logprob)")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (lp,) = ctx.saved_tensors
        gf = g.float()
        return (gf - lp.float().exp() * gf.sum(dim=1, keepdim=True)).to(g.dtype)


def log_softmax(logits: torch.Tensor) -> torch.Tensor:
    """Row-wise log-softmax of a 2-D tensor (the MNIST net's output,
    /root/reference/5.2.horovod_pytorch_mnist.py:52)."""
    if not logits.is_cuda:
        return F.log_softmax(logits.float(), dim=1).to(logits.dtype)
    return _LogSoftmaxFn.apply(logits)


def nll_loss(logprob: torch.Tensor, target: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """Negative log-likelihood; the gather + mean is left to autograd-aware torch ops."""
    return F.nll_loss(logprob.float(), target, reduction=reduction)


def new_metrics(device) -> torch.Tensor:
    return torch.zeros(4, device=device, dtype=torch.float32)
