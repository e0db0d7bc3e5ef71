"""Fully connected layer on the convolution GEMM kernels.

``y = x @ W^T + b`` is the 1x1 convolution of a ``[B, 1, 1, K]`` "image" with ``N`` filters, so the forward
runs on the implicit-GEMM fprop kernel, the input gradient on the dgrad kernel (the transposed weight is
zero-padded along the reduction dimension to a multiple of the kernel's K step: 1000 classes -> 1024), the
weight gradient on the split-K wgrad kernel (accumulating straight into ``weight.grad`` when the parameter
allows it) and the bias gradient on the column-sum kernel.  A small batch under a deep reduction (the
classifier layers of AlexNet / VGG: 128 x 25088 x 4096) has only a few dozen output tiles for 256 CUs: there
the forward splits K over workgroups (``conv.gemm_splitk``) and the input gradient runs as a split-K
"weight gradient" whose pixels are the N outputs, which also reads the weight as it is stored.  What the reference reaches through
``nn.Linear`` inside torchvision's ResNets (/root/reference/2.distributed.py:213) and in the MNIST net
(/root/reference/5.2.horovod_pytorch_mnist.py:42-43).  Shapes the tiled kernels do not take (a reduction
dimension that is not a multiple of 32 half / 16 float elements, e.g. the 50 -> 10 layer) run on the any-shape
MFMA GEMM (``csrc/kernels/gemm_anyk.hip``); CPU tensors use stock PyTorch.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from . import conv as C
from .. import backends


class LinearCache:
    """Low-precision weight and (built on first use) its zero-padded transpose, rebuilt only when the parameter
    changed."""
    __slots__ = ("key", "w_lp", "w_t", "pinned_lp")

    def __init__(self):
        self.key = None
        self.w_lp = None
        self.w_t = None
        self.pinned_lp = None   # (dtype, tensor, torch version): 16-bit copy the optimizer kernel keeps current (amp O2)

    def get(self, weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        key = (weight.data_ptr(), weight._version, C.param_generation(), dtype)
        if self.key != key:
            w = weight.detach()
            pin = self.pinned_lp
            if w.dtype == dtype:
                lp = w
            elif pin is not None and pin[0] == dtype and pin[1].device == w.device and pin[1].shape == w.shape:
                lp = pin[1]
                # stale when the master changed through torch (version counter) or through a raw kernel other than
                # the optimizer that owns this copy (parameter broadcast: generation counter)
                if pin[2] != weight._version or C._PINNED_SYNCED.get(lp.data_ptr()) != C.param_generation():
                    lp.copy_(w)
                    self.pinned_lp = (dtype, lp, weight._version)
                    C._PINNED_SYNCED[lp.data_ptr()] = C.param_generation()
            else:
                lp = w.to(dtype)
            self.w_lp = lp
            self.w_t = None
            self.key = key
        return self.w_lp

    def transposed(self, step: int) -> torch.Tensor:
        """``[K][Np]`` copy of the weight last returned by :meth:`get` (dgrad's "CRSK" weight with R = S = 1),
        ``Np`` = N rounded up to the kernel's reduction step."""
        if self.w_t is None:
            N, K = self.w_lp.shape
            Np = (N + step - 1) // step * step
            wt = torch.zeros((K, Np), device=self.w_lp.device, dtype=self.w_lp.dtype)
            wt[:, :N].copy_(self.w_lp.t())
            self.w_t = wt
        return self.w_t


# input gradients of batches up to this many rows go through the split-K kernel (see _LinearFn.backward)
_SPLITK_MAX_BATCH = int(os.environ.get("PDA_LINEAR_SPLITK_MAX_BATCH", "512"))


# forward: workgroups a split-K launch aims for (0 = never split)
_SPLITK_WORKGROUPS = int(os.environ.get("PDA_LINEAR_SPLITK_WORKGROUPS", "256"))


def _kstep(dtype: torch.dtype) -> int:
    return 64 // torch.empty((), dtype=dtype).element_size()


def supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not x.is_cuda or x.dtype not in (torch.float32, torch.float16, torch.bfloat16) or x.dim() != 2:
        return False
    N, K = weight.shape
    vec = 16 // x.element_size()
    return K % _kstep(x.dtype) == 0 and N % vec == 0 and x.shape[0] > 0


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cache: LinearCache):
        x = x.contiguous()
        B, K = x.shape
        N = weight.shape[0]
        w_lp = cache.get(weight, x.dtype)
        y = None
        tiles = ((B + 127) // 128) * ((N + 127) // 128)
        if tiles * 2 <= _SPLITK_WORKGROUPS and K >= 2048 and _SPLITK_WORKGROUPS > 0 and not backends.deterministic:
            # few output tiles under a deep reduction: split K over enough workgroups to fill the chip (float atomics
            # into the fp32 accumulator: not when run-to-run reproducibility is asked for)
            ksplit = min(_SPLITK_WORKGROUPS // tiles, K // 256)
            acc = C.gemm_splitk(x, w_lp, ksplit)
            if acc is not None:
                y = acc.to(x.dtype)
        if y is None:
            y = C.fprop_gpu(x.view(B, 1, 1, K), w_lp.view(N, 1, 1, K), (1, 1), (0, 0))
            if y is None:
                raise RuntimeError("linear: the GEMM kernel refused a shape that supported() accepted")
            y = y.view(B, N)
        if bias is not None:
            l = _lib.lib()
            _lib.check(l.pda_bias_add(_lib.dt(y), y.data_ptr(), bias.float().data_ptr(), y.data_ptr(), y.numel(), N, 0,
                                      _lib.stream()), "pda_bias_add")
        ctx.save_for_backward(x, weight)
        ctx.w_lp = w_lp
        ctx.cache = cache
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, K = x.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            vec = 16 // x.element_size()
            if B <= _SPLITK_MAX_BATCH and B % vec == 0:
                # few rows: dX[B][K] = sum over n of dY^T[n][B] (x) W[n][K] has the shape of a weight gradient with
                # the N outputs as its "pixels", so the split-K kernel fills the chip where the dgrad kernel would
                # launch ceil(B/128) x ceil(K/128) tiles - and it reads the weight as it is stored: no transposed
                # copy to rebuild after every optimizer step
                acc = C.wgrad_gpu(dy.t().contiguous().view(N, 1, 1, B), ctx.w_lp.view(N, 1, 1, K), 1, 1, (1, 1), (0, 0))
                if acc is not None:
                    dx = acc.view(B, K).to(x.dtype)
            if dx is None:
                if ctx.cache.w_lp is not ctx.w_lp:           # the parameter changed between forward and backward
                    ctx.cache = LinearCache()
                    ctx.cache.w_lp = ctx.w_lp
                w_t = ctx.cache.transposed(_kstep(x.dtype))
                Np = w_t.shape[1]
                dyp = dy if Np == N else F.pad(dy, (0, Np - N))
                dx = C.dgrad_gpu(dyp.view(B, 1, 1, Np), w_t.view(K, 1, 1, Np), (1, 1), (1, 1), (0, 0))
                dx = dx.view(B, K) if dx is not None else dy.float().matmul(weight.float()).to(x.dtype)
        if ctx.needs_input_grad[1]:
            sink = None
            g = weight.grad
            if (getattr(weight, "_pda_direct_grad", False) and g is not None and g.dtype == torch.float32
                    and g.is_cuda and g.is_contiguous() and g.shape == weight.shape):
                sink = g.view(N, 1, 1, K)
            dwk = C.wgrad_gpu(dy.view(B, 1, 1, N), x.view(B, 1, 1, K), 1, 1, (1, 1), (0, 0), out=sink)
            if dwk is None:
                dw = dy.float().t().matmul(x.float()).to(weight.dtype)
            elif sink is not None:
                C.notify_grad_ready(weight)
            else:
                dw = dwk.view(N, K).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(N, device=dy.device, dtype=torch.float32)
            _lib.check(_lib.lib().pda_colsum(_lib.dt(dy), dy.data_ptr(), db.data_ptr(), B, N,
                                             int(backends.deterministic), _lib.stream()), "pda_colsum")
        return dx, dw, db, None


class _AnyShapeLinearFn(torch.autograd.Function):
    """Linear layer whose shape the tiled kernels do not take (an in-feature count that is not a whole number of
    K steps - the MNIST net's 320 -> 50 -> 10 head in fp32 has such - or an odd out-feature count): all three
    products on the any-shape MFMA GEMM (``csrc/kernels/gemm_anyk.hip``)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .conv import gemm_nt
        w = weight.detach() if weight.dtype == x.dtype else weight.detach().to(x.dtype)
        ctx.save_for_backward(x, w)
        ctx.meta = (weight.dtype, None if bias is None else bias.dtype)
        return gemm_nt(x, w, None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, dy):
        from .conv import gemm_nt
        x, w = ctx.saved_tensors
        wdtype, bdtype = ctx.meta
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nt(dy, w.t().contiguous())
        if ctx.needs_input_grad[1]:
            dw = gemm_nt(dy.t().contiguous(), x.t().contiguous()).to(wdtype)
        if bdtype is not None and ctx.needs_input_grad[2]:
            db = dy.float().sum(0).to(bdtype)
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           cache: Optional[LinearCache] = None) -> torch.Tensor:
    """``x [.., K] @ weight[N, K]^T + bias`` in ``x``'s dtype (fp32 accumulation on the matrix cores)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if supported(x2, weight):
        y = _LinearFn.apply(x2, weight, bias, cache if cache is not None else LinearCache())
        return y.view(*lead, weight.shape[0])
    if x2.is_cuda and x2.dtype in (torch.float32, torch.float16, torch.bfloat16) and x2.shape[0] > 0:
        return _AnyShapeLinearFn.apply(x2, weight, bias).view(*lead, weight.shape[0])
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if bias is None or bias.dtype == x.dtype else bias.to(x.dtype)
    return F.linear(x, w, b)   # CPU tensors
