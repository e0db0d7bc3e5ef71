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


class _ReluCapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cap):
        l = _lib.lib()
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(l.pda_relu_cap_fwd(_lib.dt(x), x.data_ptr(), y.data_ptr(), x.numel(), cap, _lib.stream()),
                   "pda_relu_cap_fwd")
        ctx.save_for_backward(y)
        ctx.cap = cap
        return y

    @staticmethod
    def backward(ctx, dy):
        l = _lib.lib()
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(l.pda_relu_cap_bwd(_lib.dt(dy), dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), ctx.cap,
                                      _lib.stream()), "pda_relu_cap_bwd")
        return dx, None


def relu6(x: torch.Tensor) -> torch.Tensor:
    """``min(max(x, 0), 6)`` (MobileNet's activation)."""
    return F.hardtanh(x, 0.0, 6.0) if not x.is_cuda else _ReluCapFn.apply(x, 6.0)


_REPLAY: dict = {}          # per device: int64 counter added to every dropout offset on the device
_REPLAY_STRIDE = 1 << 24    # (far above the number of dropout calls of one step)


def _replay_counter(device: torch.device) -> torch.Tensor:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _REPLAY.get(idx)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=torch.device("cuda", idx))
        _REPLAY[idx] = t
    return t


def advance_replay_counter(device) -> None:
    """Called by ``utils.graph.GraphedStep`` as the last operation of a recorded step: the host-side dropout offsets
    are constants of the recording, this device-side counter moves on with every replay - so the replays draw fresh
    masks (a recorded VGG / MobileNet / MNIST-net step would otherwise drop the same units every iteration)."""
    device = torch.device(device)
    if device.type == "cuda" and (device.index if device.index is not None else torch.cuda.current_device()) in _REPLAY:
        _replay_counter(device).add_(_REPLAY_STRIDE)


def _launch_dropout(x, p, seed, offset, channel_mode, per_image, Cc):
    l = _lib.lib()
    y = torch.empty_like(x)
    _lib.check(l.pda_dropout(_lib.dt(x), x.data_ptr(), y.data_ptr(), x.numel(), float(p), seed, offset,
                             int(channel_mode), per_image, Cc, _replay_counter(x.device).data_ptr(), _lib.stream()),
               "pda_dropout")
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


def add_n(tensors, out: torch.Tensor = None) -> torch.Tensor:
    """Sum of up to 8 same-shaped dense tensors in one pass (``out`` may be ``tensors[0]``).
    GPU: one launch of ``add_n_kernel``; CPU: plain adds.  No autograd."""
    import ctypes
    ts = list(tensors)
    # (sources may live on other GPUs: the kernel runs on out's device and reads them through peer mappings)
    assert ts and all(t.shape == ts[0].shape and t.dtype == ts[0].dtype and t.device.type == ts[0].device.type
                      for t in ts)
    if out is None:
        out = torch.empty_like(ts[0])
    if not ts[0].is_cuda:
        acc = ts[0].clone() if out is not ts[0] else ts[0]
        for t in ts[1:]:
            acc = acc.add_(t) if acc is ts[0] and out is ts[0] else acc + t
        if out is not acc:
            out.copy_(acc)
        return out
    while len(ts) > 8:  # fold in groups of 8
        head = add_n(ts[:8])
        ts = [head] + ts[8:]
    ts = [t if t.is_contiguous() else t.contiguous() for t in ts]
    l = _lib.lib()
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    with torch.cuda.device(out.device):       # the kernel runs where the result lives
        _lib.check(l.pda_add_n(_lib.dt(ts[0]), arr, len(ts), out.data_ptr(), ts[0].numel(),
                               torch.cuda.current_stream(out.device).cuda_stream), "pda_add_n")
    return out
