"""Stand-alone element-wise ops (ReLU, dropout, channel dropout) on the HIP kernels
of ``csrc/kernels/elementwise.hip``; stock PyTorch on the CPU.

Dropout draws its mask from a counter-based Philox stream keyed by
``(seed, offset)``, so backward regenerates the mask instead of storing it
(reference: nn.Dropout2d / F.dropout in
/root/reference/5.2.horovod_pytorch_mnist.py:41,47,50).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib

_DROPOUT_CALLS = 0


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        l = _lib.lib()
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(l.pda_relu_fwd(_lib.dt(x), x.data_ptr(), y.data_ptr(), x.numel(), _lib.stream()), "pda_relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        l = _lib.lib()
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(l.pda_relu_bwd(_lib.dt(dy), dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _lib.stream()),
                   "pda_relu_bwd")
        return dx


def relu(x: torch.Tensor) -> torch.Tensor:
    return torch.relu(x) if not x.is_cuda else _ReluFn.apply(x)


def _launch_dropout(x, p, seed, offset, channel_mode, per_image, Cc):
    l = _lib.lib()
    y = torch.empty_like(x)
    _lib.check(l.pda_dropout(_lib.dt(x), x.data_ptr(), y.data_ptr(), x.numel(), float(p), seed, offset,
                             int(channel_mode), per_image, Cc, _lib.stream()), "pda_dropout")
    return y


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, channel_mode):
        global _DROPOUT_CALLS
        x = x.contiguous()
        seed = int(torch.initial_seed()) & ((1 << 63) - 1)
        _DROPOUT_CALLS += 1
        offset = _DROPOUT_CALLS
        if channel_mode:
            Cc = x.shape[-1]
            per_image = x.numel() // x.shape[0]
        else:
            Cc, per_image = 1, 1
        ctx.cfg = (p, seed, offset, channel_mode, per_image, Cc)
        return _launch_dropout(x, p, seed, offset, channel_mode, per_image, Cc)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset, channel_mode, per_image, Cc = ctx.cfg
        return _launch_dropout(dy.contiguous(), p, seed, offset, channel_mode, per_image, Cc), None, None


def dropout(x: torch.Tensor, p: float = 0.5, training: bool = True) -> torch.Tensor:
    if not training or p == 0.0:
        return x
    if not x.is_cuda:
        return F.dropout(x, p, True)
    return _DropoutFn.apply(x, p, False)


def dropout2d(x: torch.Tensor, p: float = 0.5, training: bool = True) -> torch.Tensor:
    """Channel dropout on an NHWC tensor: a dropped channel is zero over the whole image."""
    if not training or p == 0.0:
        return x
    if not x.is_cuda:
        return F.dropout2d(x.permute(0, 3, 1, 2), p, True).permute(0, 2, 3, 1).contiguous()
    return _DropoutFn.apply(x, p, True)
