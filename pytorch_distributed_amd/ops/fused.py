"""Fused ResNet building blocks.

``conv_bn_act``  : conv (implicit GEMM, BatchNorm statistics accumulated in the GEMM
                   epilogue) -> finalize -> one pass doing normalise + residual add + ReLU.
                   Backward: BN reduce -> BN apply (+ReLU mask, + residual gradient) ->
                   dgrad / wgrad on MFMA.
``stem_conv_bn_relu_pool`` : the 7x7/s2 stem on a 3-channel NCHW image: the loader's
                   layout change + normalisation + cast + zero border run as one kernel,
                   the conv runs as an implicit GEMM over (row tap x 8 contiguous pixels x 4
                   padded channels), followed by BN+ReLU and the 3x3/s2 max-pool.

One autograd node per conv+BN+activation keeps the Python/autograd overhead of a
ResNet-50 step at ~55 nodes instead of ~175 (reference: the torchvision Bottleneck/BasicBlock forward under
/root/reference/2.distributed.py:213).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, conv as C, norm as Nm, pool as P, streams
from .. import backends


# Smallest GEMM depth (Cout*R*S of the consuming conv's dgrad) for which the dgrad epilogue also takes the
# producer's BatchNorm-backward sums.  Kernel by kernel the fusion only pays on deep GEMMs (a shallow 1x1
# dgrad is memory-bound and its epilogue streams slower than the stand-alone reduction), but the whole
# ResNet-50 step is fastest with every layer fused (threshold sweep in profiles/bench_1gpu_summary.txt),
# hence the default 0.
_SUMS_MIN_K = int(os.environ.get("PDA_FUSE_BN_SUMS_MIN_K", "0"))
# opt-in: the consumer's dgrad epilogue accumulates the producer's BatchNorm-backward sums into 64 shared rows (float
# atomics) instead of writing one row per tile for a compression launch to reduce afterwards.  47 launches fewer per
# ResNet-50 step, but no faster (33.37 / 33.17 ms against 33.35 / 33.07: the small launches already hide under the
# side stream's weight gradients) and the atomics make the sums' rounding vary from run to run, so it stays off.
_SUMS_SLOTS = os.environ.get("PDA_BN_SUMS_SLOTS", "0") == "1"


# Inference (no_grad + eval): BatchNorm's running-statistics affine is folded into the cached low-precision
# weights and the GEMM epilogue adds the bias, the residual and the ReLU: one kernel per conv+BN(+add)(+ReLU),
# no normalisation pass.  PDA_FOLD_EVAL_BN=0 keeps the two-pass form.
_FOLD_EVAL_BN = os.environ.get("PDA_FOLD_EVAL_BN", "1") == "1"


def _conv_folded_bn(x, weight, gamma, beta, residual, stride, pad, relu, st: Nm.BNState, cache: C.WeightCache):
    x = x.contiguous()
    K = weight.shape[0]
    if K % (16 // x.element_size()) != 0:
        return None                                  # the bias lives in the vector write-out
    vers = tuple(t._version if t is not None else -1 for t in (gamma, beta, st.running_mean, st.running_var))
    vers += (Nm.stats_generation(), st.running_mean.data_ptr(), float(st.eps))
    w_f, b_f = cache.folded(weight, x.dtype, vers, lambda: Nm.eval_coeffs(gamma, beta, st))
    res = residual.contiguous() if residual is not None else None
    o = C.fprop_gpu(x, w_f, stride, pad, res=res, relu=bool(relu), bias=b_f)
    if o is not None and Nm.relu_cap(relu) is not None:
        o.clamp_(max=Nm.relu_cap(relu))
    return o


# Smallest weight-gradient GEMM (FLOP) that is launched on the backward side stream (ops/streams.py).
_SIDE_STREAM_MIN_FLOP = float(os.environ.get("PDA_WGRAD_STREAM_MIN_GFLOP", "8")) * 1e9

# Block-output ReLU masks travel to the consuming conv's dgrad epilogue as sign bits (BNBackwardLink.maskbits).
_MASK_BITS = os.environ.get("PDA_RELU_MASK_BITS", "1") == "1"

# Normalise-on-load between a conv+BatchNorm+ReLU and its single consumer (see BNBackwardLink.defer): "1" / "all" =
# both 16-bit types, "fp16" = fp16 only, "0" = off.  bf16 has no packed FMA, so its on-load normalisation runs in
# fp32, and with the tiled kernels alone that cost more than the pass it saved (ResNet-50, batch 400: 34.1 vs
# 33.5 ms per step) - but the A-stationary pointwise kernel and the fused bottleneck-tail backward only exist
# behind this hand-off: 30.25 -> 28.7 ms per step in bf16 with it on.
_DEFER_APPLY = os.environ.get("PDA_DEFER_BN_APPLY", "1") in ("1", "all", "fp16")
_DEFER_ALL_16BIT = os.environ.get("PDA_DEFER_BN_APPLY", "1") in ("1", "all")


# Backward of a bottleneck's last 1x1 conv on the 64 -> 256 geometry: BatchNorm backward apply + dgrad + wgrad in one
# pass over the 256-channel tensors (csrc/kernels/bottleneck_tail_bwd.hip) instead of three kernels and five passes.
# (0: off, 1: the 64 -> 256 geometry only, 2: also 128 -> 512, where the weight gradient stays a kernel of its own)
_FUSE_TAIL_BWD = int(os.environ.get("PDA_FUSE_TAIL_BWD", "3"))   # 1: 64->256 tails, 2: + 128->512 tails, 3: + 64->256 shortcut
# Backward of the stem: BatchNorm backward apply + weight gradient in one pass (csrc/kernels/stem_bwd.hip).
_FUSE_STEM_BWD = os.environ.get("PDA_FUSE_STEM_BWD", "1") == "1"
_DEFER_SHORTCUT = os.environ.get("PDA_DEFER_SHORTCUT", "1") == "1"
_COMPACT_SHORTCUT_GRAD = os.environ.get("PDA_COMPACT_SHORTCUT_GRAD", "1") == "1"
_DEFER_BLOCK_OUT = os.environ.get("PDA_DEFER_BLOCK_OUT", "1") == "1"


class BNBackwardLink:
    """Hand-off between a fused op (the producer) and the ONE fused op that consumes its output.

    The producer, called with ``dout_premasked=True, out_link=link``, records here what the backward of
    its BatchNorm needs: ``y`` (its conv output before normalisation, ``None`` outside training mode),
    ``scale``/``shift`` and whether its ReLU mask can be recomputed from ``y`` (``remask``: no residual
    was added before the ReLU).  The consumer, called with ``in_link=link``, applies the producer's ReLU
    mask in its own dgrad epilogue and, while the gradient tile is still on chip, also takes the two
    per-channel sums (``sum g``, ``sum g*y``) that the producer's BatchNorm backward would otherwise get
    from a separate pass over ``g`` and ``y``; it leaves them in ``sums`` for the producer's backward,
    which runs next.  ``sums is None`` there (consumer in a fallback path) just means the producer reduces
    the - already masked - gradient itself.

    With ``defer=True`` the pair also skips the producer's normalisation pass in the FORWARD direction: the
    producer hands its raw conv output on (``deferred`` set, ``xaffine`` = scale / shift in the activation
    dtype) and the consumer's forward and weight-gradient kernels apply ``relu(y * scale + shift)`` to their
    operand tiles on load (``conv.fprop_gpu(in_affine=...)``), so the normalised activation is neither written
    nor read back.  A producer whose ReLU follows a residual add (its mask cannot be recomputed from ``y``) leaves
    the sign bits of its output in ``maskbits`` (one byte per 16-byte vector, written by the normalisation pass):
    the consumer's epilogue reads those instead of the activation itself.  Only a producer with a ReLU, no residual and batch statistics defers, and only for 16-bit
    activations; a consumer that cannot fuse materialises the activation itself (``materialize_deferred``).

    With ``defer_out=True`` (the link between the LAST op of a residual block and the first 1x1 conv of the next
    block, which also owns the next block's skip branch through ``passthrough``) the producer's whole last pass -
    BatchNorm, shortcut add, ReLU, sign bits - moves into the consumer's forward kernel: the producer hands its raw
    conv output on and leaves ``(scale, shift, residual, residual affine)`` in ``mat``; the consumer's pointwise
    kernel forms the block output on its operand rows, stores it (it is the next skip branch, and what the
    backward needs) with its sign bits, and multiplies it in (``conv.pw_fprop_materialize``).  The block output is
    written once and not read back by that conv.  A consumer that cannot do this runs the producer's pass itself
    (``materialize_block_output``); whoever takes the link must do one or the other."""

    __slots__ = ("y", "scale", "shift", "relu", "remask", "sums", "defer", "deferred", "xaffine", "maskbits", "sums_buf",
                 "defer_out", "mat")

    def __init__(self, defer: bool = False, defer_out: bool = False):
        self.y = self.scale = self.shift = self.sums = self.xaffine = self.maskbits = self.sums_buf = None
        self.relu = True
        self.remask = False
        self.defer = defer
        self.deferred = False
        self.defer_out = defer_out
        self.mat = None


class ShortcutLink:
    """Hand-off between the conv + BatchNorm of a residual block's shortcut (the producer, called with
    ``res_out=link``) and the block's last fused op, which adds the shortcut as its ``residual`` (called with
    ``res_in=link``).  In training the producer then skips its normalisation pass: it returns its raw conv output
    and leaves the BatchNorm's ``(scale, shift)`` here, and the consumer's apply kernel adds
    ``residual * scale + shift`` - one write and one read of the block-wide tensor less.  Towards autograd the raw
    tensor stands for the normalised one (the gradient of an affine map's output is what the producer's BatchNorm
    backward expects), so ``residual`` must have no other use.  ``affine`` stays ``None`` whenever the producer did
    normalise (evaluation, CPU, unfused layers).

    The same object carries the shortcut's input gradient the other way when the shortcut is a stride-2 1x1 conv and
    the block's first op (called with ``alias_link=link``; it owns the skip branch through ``passthrough``) is a
    stride-1 conv on the fused path: three quarters of that gradient are zeros, so the shortcut's backward computes
    only the compact ``[N, ceil(H/2), ceil(W/2), C]`` part, leaves it in ``dx_compact`` and hands autograd a
    zero-stride placeholder of the full shape; the first op's dgrad epilogue adds the compact tensor at the even
    pixels instead of reading a full-size one (``conv.dgrad_gpu(res_s2=True)``)."""

    __slots__ = ("affine", "compact_ok", "dx_compact")

    def __init__(self):
        self.affine = None
        self.compact_ok = False
        self.dx_compact = None


class _AffineFn(torch.autograd.Function):
    """``y * scale + shift`` of a shortcut whose producer left the normalisation to a consumer that cannot do it."""

    @staticmethod
    def forward(ctx, y, scale, shift):
        return Nm.apply(y.contiguous(), scale, shift, None, False)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class _BlockOutputFn(torch.autograd.Function):
    """The block output ``relu(y * scale + shift + residual)`` of a producer that left its last pass to a consumer
    which cannot run it inside its own kernel.  Towards autograd the producer's raw output already stands for the
    block output, so the gradient passes through unchanged (the consumer masks it, as ``in_link`` demands)."""

    @staticmethod
    def forward(ctx, y, link):
        scale, shift, res, raff = link.mat
        link.mat = None
        out, link.maskbits = Nm.apply(y.contiguous(), scale, shift, res, True, want_mask=True, res_affine=raff)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def materialize_block_output(x: torch.Tensor, in_link) -> torch.Tensor:
    """``x`` itself, or the block output if ``in_link`` says the producing block left its last pass undone."""
    if in_link is None or in_link.mat is None:
        return x
    return _BlockOutputFn.apply(x, in_link)


def materialize_shortcut(residual, res_in):
    """``residual`` itself, or its normalised form if ``res_in`` says the producer skipped that pass."""
    if res_in is None or res_in.affine is None:
        return residual
    return _AffineFn.apply(residual, *res_in.affine)


class _ConvBnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, stride, pad, relu: bool, st: Nm.BNState,
                cache: C.WeightCache, passthrough: bool = False, mask_dx: bool = False,
                dout_premasked: bool = False, in_link: Optional[BNBackwardLink] = None,
                out_link: Optional[BNBackwardLink] = None, res_in: Optional[ShortcutLink] = None,
                res_out: Optional[ShortcutLink] = None, alias_link: Optional[ShortcutLink] = None):
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        dtype = x.dtype
        K, Cin, R, S = weight.shape
        w_lp = cache.low_precision(weight, dtype)
        train_stats = st.training or st.running_mean is None
        sbuf = cache.stats_buffer(K, x.device) if train_stats else None
        in_aff = None
        r = None
        if in_link is not None and in_link.mat is not None:
            # x is the previous block's raw last conv output: finish that block on the way in (see BNBackwardLink)
            m_scale, m_shift, m_res, m_raff = in_link.mat
            in_link.mat = None
            rm = None
            if passthrough and (R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(pad) == (0, 0):
                rm = C.pw_fprop_materialize(x, m_res, m_scale, m_shift if m_raff is None else m_shift + m_raff[1],
                                            None if m_raff is None else m_raff[0], w_lp, stats=train_stats,
                                            stats_buf=sbuf)
            if rm is not None:
                x, in_link.maskbits = rm[0], rm[1]
                r = (rm[2], rm[3], rm[4]) if train_stats else rm[2]
            else:
                x, in_link.maskbits = Nm.apply(x, m_scale, m_shift, m_res, True, want_mask=True, res_affine=m_raff)
        if in_link is not None and in_link.deferred:
            # x is the producer's raw conv output: normalise it inside this conv's operand loads
            assert not passthrough, "a deferred activation has exactly one consumer"
            in_aff = in_link.xaffine
            r = C.fprop_gpu(x, w_lp, stride, pad, stats=train_stats, stats_buf=sbuf, in_affine=in_aff)
            if r is None:
                x = Nm.apply(x, in_link.scale, in_link.shift, None, True)
                in_aff = None
        if in_aff is None and r is None:
            r = C.fprop_gpu(x, w_lp, stride, pad, stats=train_stats, stats_buf=sbuf)
        if r is None:
            raise RuntimeError(
                f"conv_bn_act: shape not supported by the MFMA kernels (Cin={Cin}, dtype={dtype}); "
                "use nn.Conv2d + nn.BatchNorm2d modules for this layer")
        ctx.premasked = bool(dout_premasked and relu)
        # the consumer normalises on load (see BNBackwardLink.defer): decided here so that the finalize kernel
        # can write the 16-bit coefficient tables in the same launch
        defer = (out_link is not None and out_link.defer and ctx.premasked and train_stats and residual is None
                 and (x.dtype == torch.float16 or (_DEFER_ALL_16BIT and x.element_size() == 2)) and K % 8 == 0
                 and _DEFER_APPLY)
        tables = None
        if train_stats:
            y, psum, psq = r
            M = y.numel() // K
            fin = Nm.finalize(psum, psq, float(M), gamma, beta, st,
                              clear=sbuf is not None and psum.data_ptr() == sbuf.data_ptr(),
                              lp=dtype if defer else None)
            mean, invstd, scale, shift, count = fin[:5]
            tables = fin[5:] if defer else None
        else:
            y = r
            scale, shift = Nm.eval_coeffs(gamma, beta, st)
            mean = invstd = None
            if gamma is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
                mean, invstd = Nm.eval_moments(st)   # frozen statistics, trainable affine
            count = float(y.numel() // K)
        res_aff = res_in.affine if (res_in is not None and residual is not None) else None
        if (res_out is not None and _DEFER_SHORTCUT and train_stats and not relu and residual is None and not passthrough
                and out_link is None and not defer):
            # a shortcut branch: the block's last op adds y * scale + shift itself (see ShortcutLink)
            out = y
            res_out.affine = (scale, shift)
        elif (out_link is not None and out_link.defer_out and _DEFER_BLOCK_OUT and ctx.premasked and train_stats
              and residual is not None and relu is True and _MASK_BITS and y.element_size() == 2 and K in (256, 512)
              and not passthrough and not defer):
            # the last op of a residual block whose successor finishes it inside its first conv (see BNBackwardLink)
            out = y
            out_link.mat = (scale, shift, residual, res_aff)
        elif defer:
            # the consumer normalises on load; autograd still sees this tensor as "the activation"
            out = y
            out_link.deferred = True
            xs, xf, rs, rf = tables
            out_link.xaffine = (xs, xf)
            # the mask recomputation in the consumer's dgrad epilogue decides with the same rounded coefficients
            # as its forward did (an fp32 product and a 16-bit FMA of the same operands have the same sign)
            out_link.scale, out_link.shift = rs, rf
        elif (out_link is not None and ctx.premasked and residual is not None and _MASK_BITS
              and K % (16 // y.element_size()) == 0):
            # the consumer masks this op's gradient: hand it the sign bits, 1/16 of the activation's bytes
            out, out_link.maskbits = Nm.apply(y, scale, shift, residual, relu, want_mask=True, res_affine=res_aff)
        else:
            out = Nm.apply(y, scale, shift, residual, relu, res_affine=res_aff)
        ctx.cfg = (stride, pad, relu, residual is not None, count, st.group, cache, train_stats)
        ctx.beta = beta
        ctx.mask_dx = mask_dx
        ctx.in_aff = in_aff
        ctx.in_link = in_link
        ctx.res_out = res_out
        ctx.alias_link = alias_link if passthrough else None
        if ctx.alias_link is not None:
            alias_link.dx_compact = None
            alias_link.compact_ok = (_COMPACT_SHORTCUT_GRAD and tuple(stride) == (1, 1) and x.element_size() == 2
                                     and Cin % 8 == 0 and ctx.needs_input_grad[0])
        ctx.out_link = out_link if dout_premasked else None
        if ctx.out_link is not None:
            out_link.relu = relu
            out_link.remask = relu and residual is None
            if not defer:
                out_link.scale, out_link.shift = scale, shift
            out_link.y = y if train_stats else None
            out_link.sums = None
            # where the consumer's dgrad epilogue may accumulate this layer's backward reductions: this layer's own
            # zero-on-entry statistics rows (its backward reads and clears them)
            out_link.sums_buf = cache.stats_buffer(K, x.device) if train_stats else None
        mask_src = out if (relu and not ctx.premasked and (residual is not None or not train_stats)) else None
        ctx.save_for_backward(x, weight, y if (train_stats or mean is not None) else None, mask_src, mean, invstd,
                              scale, shift, gamma)
        if passthrough:
            # second output = alias of the input (or the block output this op materialised from the previous block's
            # raw one): whatever consumes it (the block's identity branch) sends its gradient back through THIS node,
            # where the dgrad epilogue adds it for free
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, dout, dalias=None):
        x, weight, y, mask_src, mean, invstd, scale, shift, gamma = ctx.saved_tensors
        stride, pad, relu, has_res, count, group, cache, train_stats = ctx.cfg
        dout = dout.contiguous()
        K, Cin, R, S = weight.shape
        presums = None
        presums_clear = False
        if ctx.out_link is not None:
            presums, ctx.out_link.sums = ctx.out_link.sums, None
            presums_clear = presums is not None and presums is ctx.out_link.sums_buf
            ctx.out_link.y = None
        link = ctx.in_link
        # the skip branch's gradient may have come the compact way (see ShortcutLink): `dalias` is a placeholder then
        res_c = None
        if ctx.alias_link is not None and ctx.alias_link.dx_compact is not None:
            res_c, ctx.alias_link.dx_compact = ctx.alias_link.dx_compact, None
            dalias = None
        tail_dx = None
        geom_small = (K, Cin, R, S) == (256, 64, 1, 1)
        if (_FUSE_TAIL_BWD and train_stats and ctx.premasked and presums is not None and ctx.in_aff is not None
                and (geom_small or ((K, Cin, R, S) == (512, 128, 1, 1) and _FUSE_TAIL_BWD >= 2)) and tuple(stride) == (1, 1)
                and tuple(pad) == (0, 0) and dout.element_size() == 2 and dout.dtype == x.dtype and y is not None
                and dalias is None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                and link is not None and link.y is not None and link.y.data_ptr() == x.data_ptr() and link.relu
                and link.remask
                and (not geom_small or (C.grad_sink(weight) is not None and not backends.deterministic))):
            # BatchNorm backward apply and dgrad (with the producer's ReLU mask and BatchNorm sums) in one pass over dout
            # and y; on the 64 -> 256 geometry the wgrad too (dy is formed per tile on chip and never stored), on
            # 128 -> 512 dy is also written for the weight-gradient kernel below
            coef, _, dg, db = Nm.backward(dout, y, None, mean, invstd, scale, shift, gamma, count, False, False, group,
                                          ctx.beta, cache.stats_buffer(K, dout.device), presums=presums,
                                          presums_clear=presums_clear, coef_only=True)
            r = C.bottleneck_tail_bwd(dout.view(-1, K), y.view(-1, K), coef, x.view(-1, Cin), ctx.in_aff,
                                      cache.transposed(weight, dout.dtype), C.grad_sink(weight) if geom_small else None)
            if r is None:
                raise RuntimeError("conv_bn_act: the fused bottleneck-tail backward refused a shape its caller checked")
            tail_dx, link.sums = r[0].view(x.shape), r[1]
            dres = dout if has_res else None
            if geom_small:
                C.notify_grad_ready(weight)
                if dg is not None and gamma is not None and gamma.dtype != dg.dtype:
                    dg, db = dg.to(gamma.dtype), db.to(gamma.dtype)
                return (tail_dx, None, dg, db, dres, None, None, None, None, None, None, None, None, None, None, None, None, None)
            dy = r[2].view(y.shape)
        elif (_FUSE_TAIL_BWD >= 3 and train_stats and geom_small and not relu and not has_res and not ctx.premasked
              and ctx.out_link is None and link is None and ctx.in_aff is None and not ctx.mask_dx and dalias is None
              and tuple(stride) == (1, 1) and tuple(pad) == (0, 0) and dout.element_size() == 2 and dout.dtype == x.dtype
              and y is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
              and C.grad_sink(weight) is not None and not backends.deterministic):
            # a 64 -> 256 shortcut (1x1 conv + BatchNorm, no activation, nothing owed to the producer of x): the same
            # kernel without input transform, mask and sums - BatchNorm backward apply, dgrad and wgrad in one pass
            coef, _, dg, db = Nm.backward(dout, y, None, mean, invstd, scale, shift, gamma, count, False, False, group,
                                          ctx.beta, cache.stats_buffer(K, dout.device), coef_only=True)
            r = C.bottleneck_tail_bwd(dout.view(-1, K), y.view(-1, K), coef, x.view(-1, Cin), None,
                                      cache.transposed(weight, dout.dtype), C.grad_sink(weight))
            if r is None:
                raise RuntimeError("conv_bn_act: the fused shortcut backward refused a shape its caller checked")
            C.notify_grad_ready(weight)
            if dg is not None and gamma is not None and gamma.dtype != dg.dtype:
                dg, db = dg.to(gamma.dtype), db.to(gamma.dtype)
            return (r[0].view(x.shape), None, dg, db, None, None, None, None, None, None, None, None, None, None, None, None, None, None)
        elif train_stats and (ctx.premasked or (ctx.out_link is not None and not relu)):
            # the consumer of `out` already zeroed the gradient where out <= 0: no mask to read, and the
            # residual branch's gradient IS dout; with presums it also took the reduction over (dout, y)
            dy, _, dg, db = Nm.backward(dout, y, None, mean, invstd, scale, shift, gamma, count, False,
                                        False, group, ctx.beta, cache.stats_buffer(K, dout.device),
                                        presums=presums, presums_clear=presums_clear)
            dres = dout if has_res else None
        elif train_stats:
            dy, dres, dg, db = Nm.backward(dout, y, mask_src, mean, invstd, scale, shift, gamma, count, relu,
                                           has_res, group, ctx.beta, cache.stats_buffer(K, dout.device))
        else:
            g = dout if (not relu or ctx.premasked) else dout * Nm.relu_mask(mask_src, relu).to(dout.dtype)
            dy = (g.float() * scale).to(dout.dtype)
            dres = g if has_res else None
            dg = db = None
            if mean is not None:
                dg, db = Nm.eval_affine_grads(g, y, mean, invstd)
        dx = dw = None
        # the weight gradient goes to the side stream when it accumulates in place (nothing in this pass
        # reads it): it is launched first, so it runs beside this layer's dgrad and the layers below
        wgrad_done = False
        if ctx.needs_input_grad[1]:
            sink = C.grad_sink(weight)
            # ... when the kernel is long enough to be worth a fork / join (two events and a stream switch cost the
            # host ~40 us): launch-bound steps - ResNet-18 on 32x32 images - ran 1.5x slower with every wgrad forked
            big = 2.0 * dy.numel() * R * S * Cin >= _SIDE_STREAM_MIN_FLOP
            side = streams.fork(dy.device) if (sink is not None and big) else None
            if side is not None:
                with torch.cuda.stream(side):
                    dwk = C.wgrad_gpu(dy, x, R, S, stride, pad, out=sink, in_affine=ctx.in_aff)
                if dwk is None:
                    raise RuntimeError("conv_bn_act: wgrad shape not supported")
                streams.keep_alive(side, dy, x, *(ctx.in_aff or ()))
                streams.mark(dy.device)
                C.notify_grad_ready(weight)
                wgrad_done = True
        sl = ctx.res_out
        if tail_dx is not None:
            dx = tail_dx
        elif (ctx.needs_input_grad[0] and sl is not None and sl.compact_ok and (R, S) == (1, 1) and tuple(stride) == (2, 2)
              and tuple(pad) == (0, 0) and dalias is None and link is None and not ctx.mask_dx and dy.element_size() == 2
              and K % 8 == 0 and Cin % 8 == 0):
            # a stride-2 1x1 shortcut: only the even pixels of x have a gradient - a plain GEMM on the output grid;
            # the op that owns x adds it from there (autograd gets a zero-stride stand-in of the full shape)
            dxc = C.dgrad_gpu(dy, cache.transposed(weight, dy.dtype), (dy.shape[1], dy.shape[2]), (1, 1), (0, 0))
            if dxc is None:
                raise RuntimeError("conv_bn_act: no pointwise dgrad kernel for the compact shortcut gradient")
            sl.dx_compact = dxc
            dx = dxc.as_strided(x.shape, (0, 0, 0, 0))
        elif ctx.needs_input_grad[0]:
            if res_c is not None:
                dalias = None
            elif dalias is not None:
                dalias = dalias.contiguous()
            # what this op does on behalf of the op that produced x (see BNBackwardLink)
            link = ctx.in_link
            mask_x = ctx.mask_dx or (link is not None and link.relu)
            bn_in = sc = sf = None
            if (link is not None and link.y is not None and link.y.shape == x.shape and link.y.dtype == dy.dtype
                    and (K * R * S >= _SUMS_MIN_K or ctx.in_aff is not None)):
                bn_in = link.y
                if link.relu and link.remask:
                    sc, sf, mask_x = link.scale, link.shift, False
            mk, mkbits = (x if mask_x else None), False
            if (mask_x and link is not None and link.maskbits is not None
                    and link.maskbits.numel() * 16 == x.numel() * x.element_size()):
                mk, mkbits = link.maskbits, True
            r = C.dgrad_gpu(dy, cache.transposed(weight, dy.dtype), (x.shape[1], x.shape[2]), stride, pad,
                            res=dalias if res_c is None else res_c, mask=mk, bn_input=bn_in, bn_scale=sc, bn_shift=sf,
                            bn_sums=bn_in is not None, mask_bits=mkbits,
                            sums_buf=link.sums_buf if (bn_in is not None and _SUMS_SLOTS) else None,
                            res_s2=res_c is not None)
            if r is None and res_c is not None:
                # no kernel adds the compact form for this shape: expand it and go on as with a full-size gradient
                dalias = torch.zeros_like(x)
                dalias[:, ::2, ::2] = res_c
                res_c = None
                r = C.dgrad_gpu(dy, cache.transposed(weight, dy.dtype), (x.shape[1], x.shape[2]), stride, pad,
                                res=dalias, mask=mk, bn_input=bn_in, bn_scale=sc, bn_shift=sf,
                                bn_sums=bn_in is not None, mask_bits=mkbits,
                                sums_buf=link.sums_buf if (bn_in is not None and _SUMS_SLOTS) else None)
            if r is None:
                if (R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(pad) == (0, 0):
                    # pointwise conv with a filter count the dgrad kernel does not step through (16, 24 in
                    # MobileNet): dX = dY [pixels, K] x W [K, Cin] on the any-shape MFMA GEMM
                    dx = C.gemm_nt(dy.reshape(-1, K),
                                   cache.low_precision(weight, dy.dtype).reshape(K, Cin).t().contiguous())
                    dx = dx.view(x.shape)
                else:
                    dx = torch.empty_like(x)
                    C.direct_conv(1, dy, cache.low_precision(weight, dy.dtype), None, dx,
                                  (x.shape[0], x.shape[1], x.shape[2], Cin, dy.shape[1], dy.shape[2], K, R, S),
                                  stride, pad)
                if dalias is not None:
                    dx = dx + dalias
                if ctx.in_aff is not None:   # x is the raw conv output: the mask is that of the normalised value
                    dx = dx * ((x.float() * link.scale + link.shift) > 0).to(dx.dtype)
                elif mask_x or sc is not None:
                    dx = dx * (x > 0).to(dx.dtype)
            elif bn_in is not None:
                dx, link.sums = r
            else:
                dx = r
        elif dalias is not None:
            masked = ctx.mask_dx or (ctx.in_link is not None and ctx.in_link.relu)
            dx = dalias * (x > 0).to(dalias.dtype) if masked else dalias
        if ctx.needs_input_grad[1] and not wgrad_done:
            sink = C.grad_sink(weight)
            dwk = C.wgrad_gpu(dy, x, R, S, stride, pad, out=sink, in_affine=ctx.in_aff)
            if dwk is None:
                raise RuntimeError("conv_bn_act: wgrad shape not supported")
            if sink is not None:
                C.notify_grad_ready(weight)   # accumulated in place: nothing for autograd to add
            else:
                dw = dwk.permute(0, 3, 1, 2)
                if dw.dtype != weight.dtype:
                    dw = dw.to(weight.dtype)
        if dg is not None and gamma is not None and gamma.dtype != dg.dtype:
            dg, db = dg.to(gamma.dtype), db.to(gamma.dtype)
        return dx, dw, dg, db, dres, None, None, None, None, None, None, None, None, None, None, None, None, None


def conv_bn_act(x, weight, gamma, beta, residual, stride, pad, relu, st: Nm.BNState,
                cache: Optional[C.WeightCache] = None, passthrough: bool = False, mask_dx: bool = False,
                dout_premasked: bool = False, in_link: Optional[BNBackwardLink] = None,
                out_link: Optional[BNBackwardLink] = None, res_in: Optional[ShortcutLink] = None,
                res_out: Optional[ShortcutLink] = None, alias_link: Optional[ShortcutLink] = None):
    """NHWC fused conv + BatchNorm (+residual) (+ReLU).

    ``relu``: ``False``, ``True``, or a positive number ``c`` for a ReLU clipped at ``c`` (ReLU6: ``relu=6.0``),
    applied and differentiated inside the BatchNorm passes like the plain one; a clipped activation takes no
    part in the mask / link hand-offs below, which describe ``x > 0`` only.

    ``passthrough=True`` returns ``(out, x_alias)``: feed ``x_alias`` to the other consumer of ``x``
    (the residual branch of a ResNet block) and its gradient is summed inside this op's dgrad
    epilogue instead of by a separate element-wise add.

    ``mask_dx`` / ``dout_premasked`` move a ReLU backward from the op that owns the ReLU into the op that
    consumes its output.  A consumer called with ``mask_dx=True`` declares that ``x`` is a ReLU output
    whose every use goes through this op (``passthrough`` covers the skip branch) and writes its input
    gradient already multiplied by ``x > 0``; the producer, called with ``dout_premasked=True``, then
    skips its own mask: it neither re-reads its output nor writes a separate masked copy for its
    residual branch.  ``dout_premasked`` without a masking consumer gives wrong gradients; ``mask_dx``
    alone is always correct (masking twice is harmless).

    ``out_link`` (with ``dout_premasked=True``) / ``in_link``: a :class:`BNBackwardLink` shared by this op
    and the single consumer of its output, through which the consumer's dgrad epilogue also delivers the
    reductions of this op's BatchNorm backward.  A consumer given ``in_link`` always applies the
    producer's ReLU mask, so the pair stays correct when the sums cannot be produced.

    ``res_out`` / ``res_in`` / ``alias_link``: a :class:`ShortcutLink` shared by a shortcut branch's op (no
    activation), the op that takes its output as ``residual`` and the ``passthrough`` op whose alias output is the
    shortcut's input."""
    if in_link is not None and in_link.mat is not None and (
            not x.is_cuda or (not torch.is_grad_enabled() and not st.training and st.running_mean is not None)):
        x = materialize_block_output(x, in_link)
    if res_in is not None and res_in.affine is not None and (not x.is_cuda or Nm.relu_cap(relu) is not None):
        residual = materialize_shortcut(residual, res_in)
        res_in = None
    if Nm.relu_cap(relu) is not None and (dout_premasked or out_link is not None):
        raise ValueError("conv_bn_act: a clipped ReLU cannot hand its mask to the consumer (dout_premasked / out_link)")
    if not x.is_cuda:
        # reference path: the ReLU masks its own gradient, so both hints can be ignored together
        y = C.conv2d_ref(x, weight, None, stride, pad)
        o = Nm.batch_norm_act_ref(y, gamma, beta, residual, relu, st)
        return (o, x) if passthrough else o
    if not torch.is_grad_enabled() and not st.training and st.running_mean is not None and _FOLD_EVAL_BN:
        o = _conv_folded_bn(x, weight, gamma, beta, materialize_shortcut(residual, res_in), tuple(stride), tuple(pad),
                            relu, st, cache or C.WeightCache())
        if o is not None:
            return (o, x) if passthrough else o
    return _ConvBnActFn.apply(x, weight, gamma, beta, residual, tuple(stride), tuple(pad), relu, st,
                              cache or C.WeightCache(), passthrough, mask_dx, dout_premasked, in_link, out_link,
                              res_in, res_out, alias_link)


class _MaterializeFn(torch.autograd.Function):
    """``relu(y * scale + shift)`` of a deferred activation as a tensor, for a consumer that cannot normalise
    on load.  Towards autograd the producer's output already *is* the activation, so the gradient passes
    through unchanged (the consumer masks it, as ``in_link`` demands)."""

    @staticmethod
    def forward(ctx, y, scale, shift):
        return Nm.apply(y.contiguous(), scale, shift, None, True)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def materialize_deferred(x: torch.Tensor, in_link) -> torch.Tensor:
    """``x`` itself, or the normalised activation if ``in_link`` says the producer deferred it."""
    if in_link is None or not in_link.deferred:
        return x
    return _MaterializeFn.apply(x, in_link.scale, in_link.shift)


class _GradMaskFn(torch.autograd.Function):
    """Identity whose gradient is zeroed where ``x <= 0`` (the ``mask_dx`` contract for code paths that
    do not run the fused op)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * (x > 0).to(g.dtype)


def grad_mask(x: torch.Tensor) -> torch.Tensor:
    return _GradMaskFn.apply(x)


# ------------------------------------------------------------------------------------ stem
def _stem_geometry(N, H, W, R, S, stride, pad):
    """Geometry of the stem conv as an implicit GEMM over a zero-bordered 4-channel image.

    The bordered image is [N][Hp][Wp][4]; one GEMM K-step = one filter row r = 8
    consecutive pixels x 4 channels = 32 contiguous values, so the gathered "tensor"
    has Ci=32, R taps, S=1 and a pixel stride of stride*4 elements.
    """
    assert S <= 8 and stride[1] * 4 % 8 == 0, "stem kernel wants filter width <= 8 and an even stride"
    Ho, Wo = C.out_size(H, R, stride[0], pad[0]), C.out_size(W, S, stride[1], pad[1])
    Hp = H + 2 * pad[0]
    Wp = max(W + 2 * pad[1], (Wo - 1) * stride[1] + 8)
    Wp += Wp % 2
    geom = (N, Hp, 1 << 30, 32, Hp * Wp * 4, Wp * 4, stride[1] * 4, Ho, Wo, R, 1)
    return Hp, Wp, Ho, Wo, geom


def stem_weight(weight: torch.Tensor, dtype) -> torch.Tensor:
    """[K,3,R,S] master weight -> [K][R][8][4] (zero padded) GEMM B operand."""
    K, Cin, R, S = weight.shape
    w = torch.zeros((K, R, 8, 4), device=weight.device, dtype=dtype)
    w[:, :, :S, :Cin] = weight.detach().permute(0, 2, 3, 1).to(dtype)
    return w


def input_to_nhwc(images: torch.Tensor, dtype, cp: int, pad_hw=(0, 0), wp: Optional[int] = None, mean=None,
                  inv_std=None, pre_scale: float = 1.0) -> torch.Tensor:
    """NCHW (uint8/fp32/fp16) -> zero-bordered NHWC[cp] in the compute dtype, normalised."""
    l = _lib.lib()
    images = images.contiguous()
    N, Cc, H, W = images.shape
    Hp = H + 2 * pad_hw[0]
    Wp = wp if wp is not None else W + 2 * pad_hw[1]
    out = torch.empty((N, Hp, Wp, cp), device=images.device, dtype=dtype)
    _lib.check(l.pda_nchw_to_nhwc(_lib.dt(images), _lib.dt(dtype), images.data_ptr(), out.data_ptr(), N, Cc, H, W,
                                  cp, Hp, Wp, pad_hw[0], pad_hw[1], _lib.ptr(mean), _lib.ptr(inv_std),
                                  float(pre_scale), _lib.stream()), "pda_nchw_to_nhwc")
    return out


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, weight, gamma, beta, stride, pad, st: Nm.BNState, dtype, pool, mean, inv_std,
                pre_scale):
        N, Cin, H, W = images.shape
        K, _, R, S = weight.shape
        Hp, Wp, Ho, Wo, geom = _stem_geometry(N, H, W, R, S, stride, pad)
        xp = input_to_nhwc(images, dtype, 4, (pad[0], pad[1]), Wp, mean, inv_std, pre_scale)
        w_st = stem_weight(weight, dtype)
        train_stats = st.training or st.running_mean is None
        r = C.fprop_gpu(xp, w_st.view(K, R, 1, 32), (stride[0], 1), (0, 0), stats=train_stats, geom=geom)
        if r is None:
            raise RuntimeError("stem conv: unsupported configuration")
        if train_stats:
            y, psum, psq = r
            mean_, invstd, scale, shift, count = Nm.finalize(psum, psq, float(y.numel() // K), gamma, beta, st)
        else:
            y = r
            scale, shift = Nm.eval_coeffs(gamma, beta, st)
            mean_ = invstd = None
            if gamma is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
                mean_, invstd = Nm.eval_moments(st)
            count = float(y.numel() // K)
        if pool is not None and train_stats:
            # normalise + ReLU + pool in one pass: the full-resolution activation is never written (the
            # backward needs y and the arg-max bytes only).  The backward keeps the materialised max-pool
            # gradient: gathering it through the arg-max bytes inside both BatchNorm-backward passes was
            # measured slower (1.23 vs 1.01 ms at batch 400) than writing it once.
            act = None
            out, idx = P.maxpool_fwd_gpu(y, *pool, scale=scale, shift=shift)
        else:
            act = Nm.apply(y, scale, shift, None, True)
            out, idx = P.maxpool_fwd_gpu(act, *pool) if pool is not None else (act, None)
        ctx.cfg = (stride, pad, geom, count, st.group, pool, tuple(y.shape), train_stats, weight.shape)
        ctx.beta = beta
        ctx.save_for_backward(xp, y if (train_stats or mean_ is not None) else None, idx, mean_, invstd, scale, shift, gamma,
                              act if not train_stats else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        xp, y, idx, mean_, invstd, scale, shift, gamma, act = ctx.saved_tensors
        stride, pad, geom, count, group, pool, act_shape, train_stats, wshape = ctx.cfg
        dout = dout.contiguous()
        K, Cin, R, S = wshape
        dact = P.maxpool_bwd_gpu(dout, idx, act_shape, *pool) if pool is not None else dout
        if (train_stats and _FUSE_STEM_BWD and ctx.needs_input_grad[1] and dact.element_size() == 2
                and (K, R) == (64, 7) and not backends.deterministic):
            # BatchNorm backward apply + weight gradient in one pass: dy is formed per tile on chip, never stored
            coef, _, dg, db = Nm.backward(dact, y, None, mean_, invstd, scale, shift, gamma, count, True, False, group,
                                          ctx.beta, coef_only=True)
            dwk = C.stem_bwd_fused(dact, y, coef, xp, geom, stride[0])
            if dwk is not None:
                dw = dwk.view(K, R, 8, 4)[:, :, :S, :Cin].permute(0, 3, 1, 2)
                return None, dw, dg, db, None, None, None, None, None, None, None, None
            dy = Nm.apply_backward_coefs(dact, y, coef, True)
        elif train_stats:
            dy, _, dg, db = Nm.backward(dact, y, None, mean_, invstd, scale, shift, gamma, count, True, False, group,
                                        ctx.beta)
        else:
            g = dact * (act > 0).to(dact.dtype)
            dy = (g.float() * scale).to(dact.dtype)
            dg = db = None
            if mean_ is not None:
                dg, db = Nm.eval_affine_grads(g, y, mean_, invstd)
        dw = None
        if ctx.needs_input_grad[1]:
            dwk = C.wgrad_gpu(dy, xp, R, 1, (stride[0], 1), (0, 0), geom=geom)  # [K][R][1][32]
            if dwk is None:
                raise RuntimeError("stem wgrad: unsupported configuration")
            dw = dwk.view(K, R, 8, 4)[:, :, :S, :Cin].permute(0, 3, 1, 2)
        return None, dw, dg, db, None, None, None, None, None, None, None, None


def stem_conv_bn_relu_pool(images, weight, gamma, beta, stride, pad, st: Nm.BNState, dtype, pool=(3, 2, 1),
                           mean=None, inv_std=None, pre_scale: float = 1.0):
    """images NCHW -> NHWC feature map after conv+BN+ReLU(+maxpool)."""
    if not images.is_cuda:
        x = images.float() * pre_scale
        if mean is not None:
            x = (x - mean.view(1, -1, 1, 1)) * inv_std.view(1, -1, 1, 1)
        x = x.to(dtype).permute(0, 2, 3, 1)
        y = C.conv2d_ref(x, weight, None, stride, pad)
        a = Nm.batch_norm_act_ref(y, gamma, beta, None, True, st)
        return P.max_pool2d(a, *pool) if pool is not None else a
    return _StemFn.apply(images, weight, gamma, beta, tuple(stride), tuple(pad), st, dtype, pool, mean, inv_std,
                         pre_scale)
