"""BatchNorm building blocks (NHWC, viewed as [M, C]) and the stand-alone
BatchNorm(+ReLU, +residual) autograd op.

GPU path = the kernels of ``csrc/kernels/bn.hip``; CPU path = ``F.batch_norm``.
``group`` enables SyncBatchNorm: per-channel (sum, sum of squares, count) are
all-reduced between the statistics pass and the normalisation pass, and
(sum dy, sum dy*xhat) in backward - two small all-reduces per layer per direction.
Reference: torchvision BatchNorm2d under /root/reference/2.distributed.py:213;
SyncBN itself is absent upstream; it is provided for small per-GPU batches.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib


class BNState:
    """Mutable, non-tensor configuration handed to the autograd ops."""

    __slots__ = ("eps", "momentum", "training", "running_mean", "running_var", "num_batches_tracked", "group")

    def __init__(self, eps, momentum, training, running_mean, running_var, num_batches_tracked, group=None):
        self.eps = eps
        self.momentum = momentum
        self.training = training
        self.running_mean = running_mean
        self.running_var = running_var
        self.num_batches_tracked = num_batches_tracked
        self.group = group


def _world(group) -> int:
    if group is None:
        return 1
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_initialized() else 1


def stats_partials(y2d: torch.Tensor):
    """Per-row-block column sums of y[M][C] -> (psum, psq) each [nparts][C] fp32."""
    l = _lib.lib()
    M, C = y2d.shape
    n = l.pda_bn_num_partials(M, C, _lib.dt(y2d))
    part = torch.empty((2, n, C), device=y2d.device, dtype=torch.float32)
    _lib.check(l.pda_bn_stats_partial(_lib.dt(y2d), y2d.data_ptr(), M, C, part[0].data_ptr(), part[1].data_ptr(),
                                      _lib.stream()), "pda_bn_stats_partial")
    return part[0], part[1]


def compress_partials(pa: torch.Tensor, pb: torch.Tensor):
    """Thousands of per-tile partial rows ``[n, C]`` (x2) -> at most 128 rows, with a wide launch; the
    finalize-type kernels reduce whatever they are given with one workgroup per channel group, which is the
    right shape for tens of rows and a slow one for ten thousand.  Returns the inputs when n is small."""
    l = _lib.lib()
    n, C = pa.shape
    rs = l.pda_bn_compress_rows(n, C)
    if rs <= 0:
        return pa, pb
    out = torch.empty((2, rs, C), device=pa.device, dtype=torch.float32)
    _lib.check(l.pda_bn_compress_partials(pa.data_ptr(), pb.data_ptr(), n, C, out[0].data_ptr(), out[1].data_ptr(),
                                          rs, _lib.stream()), "pda_bn_compress_partials")
    return out[0], out[1]


_STATS_GENERATION = 0   # bumped whenever a kernel rewrites running statistics (their torch version counters do not move)


def stats_generation() -> int:
    return _STATS_GENERATION


def finalize(psum, psq, count: float, gamma, beta, st: BNState, clear: bool = False, lp: Optional[torch.dtype] = None):
    """partials -> (mean, invstd, scale, shift, count); updates running stats in place.

    ``clear`` zeroes the partial rows once they have been read (persistent accumulation rows of
    ``conv.WeightCache.stats_buffer``).  ``lp`` (fp16 / bf16): the same kernel also writes the coefficient tables
    of a consumer that normalises on load, and the result grows by what :func:`lp_tables` returns:
    ``(..., count, scale_lp, shift_lp, scale_rt, shift_rt)``."""
    l = _lib.lib()
    C = psum.shape[-1]
    if not clear:
        psum, psq = compress_partials(psum, psq)
    nparts = psum.shape[0]
    dev = psum.device
    world = _world(st.group)
    if world > 1:
        import torch.distributed as dist
        sums = torch.empty((2 * C,), device=dev, dtype=torch.float32)
        _lib.check(l.pda_bn_reduce_partials(psum.data_ptr(), psq.data_ptr(), nparts, C, sums.data_ptr(),
                                            int(clear), _lib.stream()), "pda_bn_reduce_partials")
        clear = False
        from ..comm import small_all_reduce   # [2, C] floats: the latency path when it is enabled
        small_all_reduce(sums, "sum", st.group)
        # every rank of a data-parallel job runs the same per-rank batch, so the global
        # element count is known on the host and no device->host read is needed
        count = count * world
        psum, psq, nparts = sums[:C], sums[C:], 1
    out = torch.empty((4, C), device=dev, dtype=torch.float32)
    rm = st.running_mean if (st.training and st.running_mean is not None) else None
    rv = st.running_var if (st.training and st.running_var is not None) else None
    if rm is not None:
        global _STATS_GENERATION
        _STATS_GENERATION += 1
    mom = st.momentum
    nbt = None
    if st.num_batches_tracked is not None and st.training:
        if mom is None:  # cumulative moving average, as torch does for momentum=None (needs the count on the host)
            st.num_batches_tracked += 1
            mom = 1.0 / float(st.num_batches_tracked.item())
        else:
            nbt = st.num_batches_tracked  # incremented by the finalize kernel itself
    if lp is not None and C % 8 == 0:
        lpt = torch.empty((2, C), device=dev, dtype=lp)
        rt = torch.empty((2, C), device=dev, dtype=torch.float32)
        _lib.check(l.pda_bn_finalize_lp(psum.data_ptr(), psq.data_ptr(), nparts, C, float(count), float(st.eps),
                                        float(mom if mom is not None else 0.1), _lib.ptr(gamma), _lib.ptr(beta),
                                        _lib.ptr(rm), _lib.ptr(rv), out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), out[3].data_ptr(), _lib.ptr(nbt), int(clear),
                                        _lib.dt(lp), lpt.data_ptr(), rt.data_ptr(), _lib.stream()),
                   "pda_bn_finalize_lp")
        return out[0], out[1], out[2], out[3], count, lpt[0], lpt[1], rt[0], rt[1]
    _lib.check(l.pda_bn_finalize(psum.data_ptr(), psq.data_ptr(), nparts, C, float(count), float(st.eps),
                                 float(mom if mom is not None else 0.1), _lib.ptr(gamma), _lib.ptr(beta),
                                 _lib.ptr(rm), _lib.ptr(rv), out[0].data_ptr(), out[1].data_ptr(),
                                 out[2].data_ptr(), out[3].data_ptr(), _lib.ptr(nbt), int(clear), _lib.stream()),
               "pda_bn_finalize")
    if lp is not None:
        return (out[0], out[1], out[2], out[3], count) + lp_tables(out[2], out[3], lp)
    return out[0], out[1], out[2], out[3], count


def lp_tables(scale: torch.Tensor, shift: torch.Tensor, dtype: torch.dtype):
    """``(scale_lp, shift_lp, scale_rt, shift_rt)``: the coefficients rounded to the 16-bit activation type (what a
    consumer that normalises on load multiplies with) and those rounded values back in fp32 (what the ReLU-mask
    recomputation of the backward pass has to use to agree with that forward element by element)."""
    l = _lib.lib()
    C = scale.numel()
    C8 = (C + 7) // 8 * 8    # every table starts 16-byte aligned
    lp = torch.empty((2, C8), device=scale.device, dtype=dtype)
    rt = torch.empty((2, C8), device=scale.device, dtype=torch.float32)
    if C8 != C:
        raise ValueError("lp_tables: channel count must be a multiple of 8")
    _lib.check(l.pda_bn_lp_tables(_lib.dt(dtype), C, scale.data_ptr(), shift.data_ptr(), lp.data_ptr(), rt.data_ptr(),
                                  _lib.stream()), "pda_bn_lp_tables")
    return lp[0], lp[1], rt[0], rt[1]


def eval_coeffs(gamma, beta, st: BNState):
    l = _lib.lib()
    C = st.running_mean.numel()
    out = torch.empty((2, C), device=st.running_mean.device, dtype=torch.float32)
    _lib.check(l.pda_bn_eval_coeffs(C, float(st.eps), _lib.ptr(gamma), _lib.ptr(beta), st.running_mean.data_ptr(),
                                    st.running_var.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), _lib.stream()),
               "pda_bn_eval_coeffs")
    return out[0], out[1]


def eval_moments(st: BNState):
    """(mean, invstd) of eval-mode normalisation: the running statistics, as fp32 vectors."""
    return st.running_mean.float(), torch.rsqrt(st.running_var.float() + st.eps)


def eval_affine_grads(g: torch.Tensor, y: torch.Tensor, mean: torch.Tensor, invstd: torch.Tensor):
    """dgamma, dbeta of eval-mode BatchNorm (fixed statistics: dgamma = sum g*xhat, dbeta = sum g).
    Fine-tuning with frozen statistics is rare enough that plain tensor ops do."""
    dims = tuple(range(g.dim() - 1))
    gf = g.float()
    return (gf * ((y.float() - mean) * invstd)).sum(dims), gf.sum(dims)


def relu_cap(relu) -> Optional[float]:
    """The activation argument of this module is ``False`` (none), ``True`` (ReLU) or a positive number ``c``
    (ReLU clipped at ``c``: ``min(max(v, 0), c)``, ReLU6 for ``c = 6``); returns ``c`` for the last form."""
    if relu is True or relu is False or relu is None or not relu:
        return None
    return float(relu)


def relu_mask(out: torch.Tensor, relu) -> torch.Tensor:
    """Where the activation ``relu`` (see :func:`relu_cap`) that produced ``out`` passes a gradient."""
    cap = relu_cap(relu)
    return (out > 0) if cap is None else ((out > 0) & (out < cap))


def apply(y: torch.Tensor, scale, shift, res: Optional[torch.Tensor], relu, want_mask: bool = False,
          res_affine: Optional[tuple] = None):
    """``relu(y * scale + shift + res)``.  ``want_mask``: also return the sign bits of the result, one byte per
    16-byte vector (bit e = element e is positive): what the ReLU backward needs of this tensor, at 1/16 of its
    size (``conv.dgrad_gpu(mask=..., mask_bits=True)``).

    ``res_affine = (scale_r, shift_r)`` (fp32 ``[C]``): ``res`` is the raw output of the shortcut convolution and
    ``res * scale_r + shift_r`` is added - its BatchNorm applied here instead of in a pass of its own."""
    l = _lib.lib()
    C = y.shape[-1]
    M = y.numel() // C
    out = torch.empty_like(y)
    cap = relu_cap(relu)
    if res_affine is not None:
        if cap is not None or res is None:
            raise ValueError("norm.apply: res_affine needs a residual and a plain (or no) ReLU")
        bits = (torch.empty(y.numel() * y.element_size() // 16, device=y.device, dtype=torch.uint8) if want_mask
                else None)
        _lib.check(l.pda_bn_apply_res_affine(_lib.dt(y), y.data_ptr(), res.data_ptr(), out.data_ptr(), scale.data_ptr(),
                                             shift.data_ptr(), res_affine[0].data_ptr(), res_affine[1].data_ptr(), M, C,
                                             int(relu), _lib.ptr(bits), _lib.stream()), "pda_bn_apply_res_affine")
        return (out, bits) if want_mask else out
    if cap is not None:
        if want_mask:
            raise ValueError("norm.apply: sign-bit masks describe a plain ReLU, not a clipped one")
        _lib.check(l.pda_bn_apply_cap(_lib.dt(y), y.data_ptr(), _lib.ptr(res), out.data_ptr(), scale.data_ptr(),
                                      shift.data_ptr(), M, C, cap, _lib.stream()), "pda_bn_apply_cap")
        return out
    if want_mask:
        bits = torch.empty(y.numel() * y.element_size() // 16, device=y.device, dtype=torch.uint8)
        _lib.check(l.pda_bn_apply_mask(_lib.dt(y), y.data_ptr(), _lib.ptr(res), out.data_ptr(), scale.data_ptr(),
                                       shift.data_ptr(), M, C, int(relu), bits.data_ptr(), _lib.stream()),
                   "pda_bn_apply_mask")
        return out, bits
    _lib.check(l.pda_bn_apply(_lib.dt(y), y.data_ptr(), _lib.ptr(res), out.data_ptr(), scale.data_ptr(),
                              shift.data_ptr(), M, C, int(relu), _lib.stream()), "pda_bn_apply")
    return out


def _vec_sink(param):
    """``param.grad`` if this package may accumulate into it directly (see ``conv.grad_sink``)."""
    if param is None or not getattr(param, "_pda_direct_grad", False):
        return None
    g = param.grad
    if g is None or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous() or g.shape != param.shape:
        return None
    return g


def backward(dout, y, out_for_mask, mean, invstd, scale, shift, gamma, count: float, relu, want_dres: bool,
             group=None, beta=None, scratch=None, presums=None, presums_clear: bool = False, coef_only: bool = False):
    """-> dy, dres (or None), dgamma, dbeta.

    ``coef_only``: no apply pass (with ``presums`` no pass over the tensors at all) - the first result is the fp32 ``[5, C]``
    coefficient table (rows A, B, K of ``dy = A*dout + B*y + K``, then scale, shift) for a kernel that applies it on
    the fly (``conv.bottleneck_tail_bwd``).

    ``presums`` (fp32 ``[2, rows, C]``: partial sums of ``dout`` and of ``dout*y``, as returned by
    ``conv.dgrad_gpu(..., bn_sums=True)``) replaces this function's own reduction pass over dout and y;
    ``presums_clear``: they sit in a persistent accumulation buffer that has to be all zeros again afterwards.

    When both affine parameters opted into direct gradient accumulation, their gradients are added
    to ``.grad`` by the finalize kernel and ``None`` is returned for them.  ``scratch`` (all-zero fp32
    ``[2, slots, C]``, zero again on return) replaces the per-workgroup partial rows by ``slots``
    atomically accumulated ones."""
    l = _lib.lib()
    C = y.shape[-1]
    M = y.numel() // C
    dev = y.device
    n = l.pda_bn_num_partials(M, C, _lib.dt(y))
    slots = 0
    if presums is not None:
        if presums.shape[0] != 2 or presums.shape[2] != C or presums.dtype != torch.float32 or relu:
            raise ValueError("norm.backward: presums must be fp32 [2, rows, C] sums of an already masked gradient")
        pdg, pdb = compress_partials(presums[1], presums[0])
        n = pdg.shape[0]
    elif scratch is not None and n > scratch.shape[1]:
        part, slots, n = scratch, scratch.shape[1], scratch.shape[1]
        pdg, pdb = part[0], part[1]
    else:
        part = torch.empty((2, n, C), device=dev, dtype=torch.float32)
        pdg, pdb = part[0], part[1]
    cap = relu_cap(relu)
    if presums is None and cap is not None:
        _lib.check(l.pda_bn_bwd_reduce_cap(_lib.dt(y), dout.data_ptr(), y.data_ptr(), _lib.ptr(out_for_mask),
                                           mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), M,
                                           C, cap, pdg.data_ptr(), pdb.data_ptr(), slots, _lib.stream()),
                   "pda_bn_bwd_reduce_cap")
    elif presums is None:
        _lib.check(l.pda_bn_bwd_reduce(_lib.dt(y), dout.data_ptr(), y.data_ptr(), _lib.ptr(out_for_mask),
                                       mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), M, C,
                                       int(relu), pdg.data_ptr(), pdb.data_ptr(), slots, _lib.stream()),
                   "pda_bn_bwd_reduce")
    dgb = torch.empty((2, C), device=dev, dtype=torch.float32)
    coef = torch.empty((5, C), device=dev, dtype=torch.float32)
    sync = _world(group) > 1
    sink_g, sink_b = _vec_sink(gamma), _vec_sink(beta)
    direct = sink_g is not None and sink_b is not None
    _lib.check(l.pda_bn_bwd_finalize(pdg.data_ptr(), pdb.data_ptr(), n, C, dgb[0].data_ptr(),
                                     dgb[1].data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                     shift.data_ptr(), _lib.ptr(gamma), float(count),
                                     0 if sync else coef.data_ptr(), sink_g.data_ptr() if direct else 0,
                                     sink_b.data_ptr() if direct else 0,
                                     int(slots > 0 or (presums is not None and presums_clear)), int(presums is not None),
                                     _lib.stream()),
               "pda_bn_bwd_finalize")
    if sync:
        import torch.distributed as dist
        tot = dgb.clone()
        from ..comm import small_all_reduce
        small_all_reduce(tot, "sum", group)  # dx needs the global sums; the parameter grads stay local
        _lib.check(l.pda_bn_bwd_coefs(tot[0].data_ptr(), tot[1].data_ptr(), C, mean.data_ptr(), invstd.data_ptr(),
                                      scale.data_ptr(), shift.data_ptr(), _lib.ptr(gamma), float(count),
                                      coef.data_ptr(), _lib.stream()), "pda_bn_bwd_coefs")
    if coef_only:
        if want_dres or cap is not None:
            raise ValueError("norm.backward: coef_only has no residual gradient to write and knows no clipped ReLU")
        if direct:
            for p in (gamma, beta):
                cb = getattr(p, "_pda_grad_ready", None)
                if cb is not None:
                    cb()
            return coef, None, None, None
        return coef, None, dgb[0], dgb[1]
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    if cap is not None:
        _lib.check(l.pda_bn_bwd_apply_cap(_lib.dt(y), dout.data_ptr(), y.data_ptr(), _lib.ptr(out_for_mask),
                                          coef.data_ptr(), M, C, cap, dy.data_ptr(), _lib.ptr(dres), _lib.stream()),
                   "pda_bn_bwd_apply_cap")
    else:
        _lib.check(l.pda_bn_bwd_apply(_lib.dt(y), dout.data_ptr(), y.data_ptr(), _lib.ptr(out_for_mask),
                                      coef.data_ptr(), M, C, int(relu), dy.data_ptr(), _lib.ptr(dres), _lib.stream()),
                   "pda_bn_bwd_apply")
    if direct:
        for p in (gamma, beta):
            cb = getattr(p, "_pda_grad_ready", None)
            if cb is not None:
                cb()
        return dy, dres, None, None
    return dy, dres, dgb[0], dgb[1]


def apply_backward_coefs(dout, y, coef, relu: bool = True):
    """``dy = coef[0]*g + coef[1]*y + coef[2]`` (``g`` = ``dout`` masked by the ReLU recomputed from ``y`` when
    ``relu``) for a coefficient table obtained with ``backward(..., coef_only=True)``: the apply pass on its own."""
    l = _lib.lib()
    C = y.shape[-1]
    M = y.numel() // C
    dy = torch.empty_like(y)
    _lib.check(l.pda_bn_bwd_apply(_lib.dt(y), dout.data_ptr(), y.data_ptr(), None, coef.data_ptr(), M, C, int(relu),
                                  dy.data_ptr(), None, _lib.stream()), "pda_bn_bwd_apply")
    return dy


class _BatchNormActFn(torch.autograd.Function):
    """out = relu(bn(y) + residual) on the GPU kernels (stand-alone, not fused with a conv)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, relu, st: BNState):
        y = y.contiguous()
        C = y.shape[-1]
        M = y.numel() // C
        if residual is not None:
            residual = residual.contiguous()
        if st.training or st.running_mean is None:
            psum, psq = stats_partials(y.view(M, C))
            mean, invstd, scale, shift, count = finalize(psum, psq, float(M), gamma, beta, st)
        else:
            scale, shift = eval_coeffs(gamma, beta, st)
            mean = invstd = None
            count = float(M)
        out = apply(y, scale, shift, residual, relu)
        ctx.relu, ctx.has_res, ctx.count, ctx.group = relu, residual is not None, count, st.group
        ctx.train = mean is not None
        ctx.beta = beta
        if ctx.train:
            ctx.save_for_backward(y, out if (relu and residual is not None) else None, mean, invstd, scale, shift, gamma)
        elif gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            ctx.save_for_backward(out if relu else None, scale, y, *eval_moments(st))
        else:
            ctx.save_for_backward(out if relu else None, scale, None, None, None)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        if not ctx.train:  # eval-mode BN is an affine map
            out, scale, y, mean, invstd = ctx.saved_tensors
            g = dout if out is None else dout * relu_mask(out, ctx.relu).to(dout.dtype)
            dg = db = None
            if y is not None:
                dg, db = eval_affine_grads(g, y, mean, invstd)
            return (g.float() * scale).to(dout.dtype), dg, db, (g if ctx.has_res else None), None, None
        y, out, mean, invstd, scale, shift, gamma = ctx.saved_tensors
        dy, dres, dg, db = backward(dout, y, out, mean, invstd, scale, shift, gamma, ctx.count, ctx.relu,
                                    ctx.has_res, ctx.group, ctx.beta)
        if dg is not None and gamma is not None:
            dg, db = dg.to(gamma.dtype), db.to(gamma.dtype)
        else:
            dg = db = None
        return dy, dg, db, dres, None, None


def batch_norm_act_ref(y, gamma, beta, residual, relu, st: BNState):
    """CPU path; NHWC in/out."""
    x = y.permute(0, 3, 1, 2) if y.dim() == 4 else y
    use_batch = st.training or st.running_mean is None
    if st.training and st.num_batches_tracked is not None:
        st.num_batches_tracked += 1
    mom = st.momentum
    if mom is None:
        mom = 1.0 / float(st.num_batches_tracked) if st.num_batches_tracked is not None and use_batch else 0.0
    if use_batch and _world(st.group) > 1:
        o = _sync_bn_ref(x, gamma, beta, st, mom)
    else:
        o = F.batch_norm(x.float(), st.running_mean, st.running_var, None if gamma is None else gamma.float(),
                         None if beta is None else beta.float(), use_batch, mom, st.eps).to(y.dtype)
    if y.dim() == 4:
        o = o.permute(0, 2, 3, 1)
    if residual is not None:
        o = o + residual
    if relu:
        cap = relu_cap(relu)
        o = torch.relu(o) if cap is None else torch.clamp(o, 0.0, cap)
    return o.contiguous()


def _sync_bn_ref(x, gamma, beta, st: BNState, mom):
    """SyncBN on the CPU/gloo path: differentiable all-reduce of (sum, sumsq, count)."""
    import torch.distributed as dist
    from torch.distributed.nn.functional import all_reduce as ar
    xf = x.float()
    dims = [0] + list(range(2, xf.dim()))
    C = xf.shape[1]
    n_local = xf.numel() // C
    packed = torch.cat([xf.sum(dims), (xf * xf).sum(dims), torch.tensor([float(n_local)])])
    packed = ar(packed, group=st.group)
    cnt = packed[2 * C]
    mean = packed[:C] / cnt
    var = (packed[C:2 * C] / cnt - mean * mean).clamp_min(0.0)
    if st.training and st.running_mean is not None:
        with torch.no_grad():
            st.running_mean.mul_(1 - mom).add_(mean.detach() * mom)
            st.running_var.mul_(1 - mom).add_(var.detach() * (cnt / (cnt - 1).clamp_min(1.0)) * mom)
    shape = [1, C] + [1] * (xf.dim() - 2)
    o = (xf - mean.view(shape)) * torch.rsqrt(var.view(shape) + st.eps)
    if gamma is not None:
        o = o * gamma.float().view(shape) + beta.float().view(shape)
    return o.to(x.dtype)


def batch_norm_act(y, gamma, beta, residual, relu, st: BNState):
    if not y.is_cuda or (y.shape[-1] * y.element_size()) % 4 != 0 or (y.numel() * y.element_size()) % 16 != 0:
        # CPU, or an odd 16-bit channel count: stock math.  (Channel counts that do not fill 16-byte vectors - the
        # 58 / 116 channels per branch of ShuffleNetV2 - run on the 4-byte-per-lane instantiations of the kernels.)
        return batch_norm_act_ref(y, gamma, beta, residual, relu, st)
    return _BatchNormActFn.apply(y, gamma, beta, residual, relu, st)
