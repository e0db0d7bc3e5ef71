"""Fully connected layer on the convolution GEMM kernels.

``y = x @ W^T + b`` is the 1x1 convolution of a ``[B, 1, 1, K]`` "image" with ``N`` filters, so the forward
runs on the implicit-GEMM fprop kernel, the input gradient on the dgrad kernel (the transposed weight is
zero-padded along the reduction dimension to a multiple of the kernel's K step: 1000 classes -> 1024), the
weight gradient on the split-K wgrad kernel (accumulating straight into ``weight.grad`` when the parameter
allows it) and the bias gradient on the column-sum kernel.  What the reference reaches through
``nn.Linear`` inside torchvision's ResNets (/root/reference/2.distributed.py:213) and in the MNIST net
(/root/reference/5.2.horovod_pytorch_mnist.py:42-43).  Shapes the MFMA kernels do not take (a reduction
dimension that is not a multiple of 32 half / 16 float elements, e.g. the 50 -> 10 layer) and CPU tensors use
the library GEMM.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from . import conv as C
from .. import backends


class LinearCache:
    """Low-precision weight and its zero-padded transpose, rebuilt only when the parameter changed."""
    __slots__ = ("key", "w_lp", "w_t")

    def __init__(self):
        self.key = None
        self.w_lp = None
        self.w_t = None

    def get(self, weight: torch.Tensor, dtype: torch.dtype, step: int):
        key = (weight.data_ptr(), weight._version, C.param_generation(), dtype)
        if self.key != key:
            w = weight.detach()
            N, K = w.shape
            self.w_lp = w if w.dtype == dtype else w.to(dtype)
            Np = (N + step - 1) // step * step
            wt = torch.zeros((K, Np), device=w.device, dtype=dtype)       # [K][Np]: dgrad's "CRSK" weight, R=S=1
            wt[:, :N].copy_(self.w_lp.t())
            self.w_t = wt
            self.key = key
        return self.w_lp, self.w_t


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
        step = _kstep(x.dtype)
        w_lp, w_t = cache.get(weight, x.dtype, step)
        y = C.fprop_gpu(x.view(B, 1, 1, K), w_lp.view(N, 1, 1, K), (1, 1), (0, 0))
        if y is None:
            raise RuntimeError("linear: the GEMM kernel refused a shape that supported() accepted")
        y = y.view(B, N)
        if bias is not None:
            l = _lib.lib()
            _lib.check(l.pda_bias_add(_lib.dt(y), y.data_ptr(), bias.float().data_ptr(), y.data_ptr(), y.numel(), N, 0,
                                      _lib.stream()), "pda_bias_add")
        ctx.save_for_backward(x, weight)
        ctx.w_t = w_t
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
            Np = ctx.w_t.shape[1]
            dyp = dy if Np == N else F.pad(dy, (0, Np - N))
            dx = C.dgrad_gpu(dyp.view(B, 1, 1, Np), ctx.w_t.view(K, 1, 1, Np), (1, 1), (1, 1), (0, 0))
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


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           cache: Optional[LinearCache] = None) -> torch.Tensor:
    """``x [.., K] @ weight[N, K]^T + bias`` in ``x``'s dtype (fp32 accumulation on the matrix cores)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if supported(x2, weight):
        y = _LinearFn.apply(x2, weight, bias, cache if cache is not None else LinearCache())
        return y.view(*lead, weight.shape[0])
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if bias is None or bias.dtype == x.dtype else bias.to(x.dtype)
    return F.linear(x, w, b)
