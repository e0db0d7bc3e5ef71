"""apex-amp-shaped mixed-precision front end.

    model, optimizer = amp.initialize(model, optimizer, opt_level="O1")
    with amp.scale_loss(loss, optimizer) as scaled_loss:
        scaled_loss.backward()
    optimizer.step()

is the reference's usage (/root/reference/4.apex_distributed2.py:177,289-291).

opt levels
  O0  fp32 everything (loss scale 1).
  O1  convolutions / linears take fp16 (or bf16) inputs on the MFMA units with fp32
      accumulation; BatchNorm statistics, softmax/cross-entropy and reductions are
      fp32 inside the kernels; parameters stay fp32.  For this package's models that
      is ``set_compute_dtype``; any other ``nn.Module`` runs under ``torch.autocast``.
      Dynamic loss scaling.
  O2  O1 + the optimizer owns fp32 master weights and writes the half-precision
      model copy of every conv weight inside the fused update kernel (no per-step
      cast kernels).  BatchNorm stays fp32.  Dynamic loss scaling.
  O3  pure half precision compute with a static loss scale of 1 (speed baseline).
"""
from __future__ import annotations

import contextlib
import functools
from typing import Optional, Union

import torch

from ..ops import multi_tensor as mt
from ..ops.conv import WeightCache, krsc
from .scaler import LossScaler

_STATE = {"scalers": [], "opt_level": None, "dtype": None, "verbosity": 1}


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model


def _apply_policy(model: torch.nn.Module, opt_level: str, dtype: torch.dtype) -> None:
    from .. import nn as pnn
    inner = _unwrap(model)
    low = dtype if opt_level in ("O1", "O2", "O3") else torch.float32
    has_native = False
    for m in inner.modules():
        if hasattr(m, "set_compute_dtype"):
            m.set_compute_dtype(low)
            has_native = True
    if not has_native and low != torch.float32:
        # generic torch module: run its forward under autocast (whitelist/blacklist policy of torch.amp)
        fwd = inner.forward
        dev_type = next(inner.parameters()).device.type if any(True for _ in inner.parameters()) else "cpu"

        @functools.wraps(fwd)
        def autocast_forward(*a, **k):
            with torch.autocast(device_type=dev_type, dtype=low if dev_type == "cuda" else torch.bfloat16):
                return fwd(*a, **k)
        autocast_forward._pda_autocast_dtype = low
        inner.forward = autocast_forward


def _install_master_copies(model: torch.nn.Module, optimizer, dtype: torch.dtype) -> None:
    """O2: persistent half copies of the conv and linear weights, refreshed by the optimizer kernel."""
    from .. import nn as pnn
    if not hasattr(optimizer, "register_model_copy"):
        return
    for m in _unwrap(model).modules():
        if isinstance(m, pnn.Conv2d) and m.weight.is_cuda:
            w = krsc(m.weight.detach())
            half = w.to(dtype)
            cache: WeightCache = m._cache
            cache.pinned_lp = (dtype, half, m.weight._version)
            from ..ops import conv as _conv
            _conv._PINNED_SYNCED[half.data_ptr()] = _conv.param_generation()
            optimizer.register_model_copy(m.weight, half)
        elif isinstance(m, pnn.Linear) and m.weight.is_cuda and m.weight.dtype == torch.float32:
            from ..ops import conv as _conv
            from ..ops.linear import LinearCache
            lcache = m.__dict__.get("_pda_cache")
            if lcache is None:
                lcache = m.__dict__["_pda_cache"] = LinearCache()
            half = m.weight.detach().to(dtype)
            lcache.pinned_lp = (dtype, half, m.weight._version)
            _conv._PINNED_SYNCED[half.data_ptr()] = _conv.param_generation()
            optimizer.register_model_copy(m.weight, half)


def initialize(model, optimizer=None, opt_level: str = "O1", loss_scale: Union[str, float, None] = None,
               dtype: torch.dtype = torch.float16, keep_batchnorm_fp32: Optional[bool] = None,
               master_weights: Optional[bool] = None, min_loss_scale: Optional[float] = None,
               max_loss_scale: float = 2.0 ** 24, enabled: bool = True, verbosity: int = 1):
    """Configure ``model`` / ``optimizer`` for mixed precision and return them."""
    if opt_level not in ("O0", "O1", "O2", "O3"):
        raise ValueError(f"unknown opt_level {opt_level!r}")
    if not enabled:
        opt_level = "O0"
    if loss_scale is None:
        loss_scale = "dynamic" if opt_level in ("O1", "O2") and dtype == torch.float16 else 1.0
    models = model if isinstance(model, (list, tuple)) else [model]
    for m in models:
        _apply_policy(m, opt_level, dtype)
    try:
        device = next(_unwrap(models[0]).parameters()).device
    except StopIteration:
        device = torch.device("cpu")
    opts = [] if optimizer is None else (list(optimizer) if isinstance(optimizer, (list, tuple)) else [optimizer])
    scaler = LossScaler(loss_scale, min_loss_scale=min_loss_scale, max_loss_scale=max_loss_scale, device=device,
                        enabled=(opt_level != "O0" or loss_scale not in (1.0, 1)))
    _STATE["scalers"] = [scaler]
    _STATE["opt_level"], _STATE["dtype"], _STATE["verbosity"] = opt_level, dtype, verbosity
    for opt in opts:
        _attach(opt, scaler)
        if opt_level == "O2" or master_weights:
            for m in models:
                _install_master_copies(m, opt, dtype)
    if verbosity and _rank0():
        print(f"[amp] opt_level={opt_level} compute_dtype={dtype if opt_level != 'O0' else torch.float32} "
              f"loss_scale={loss_scale}")
    return model, optimizer


def _rank0() -> bool:
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _attach(optimizer, scaler: LossScaler) -> None:
    optimizer._amp_scaler_ref = scaler
    if hasattr(optimizer, "attach_scaler"):
        optimizer.attach_scaler(scaler)   # fused path: skip decision + unscale happen in the kernel
        return
    if getattr(optimizer, "_amp_patched", False):
        return
    real_step = optimizer.step

    @functools.wraps(real_step)
    def step(*a, **k):
        # stock optimizer: the skip decision needs the flag on the host (one sync per step, as apex does)
        overflow = scaler.enabled and bool(scaler.overflow_flag.item() != 0)
        out = None
        if overflow:
            if _STATE["verbosity"] and _rank0():
                print("Gradient overflow.  Skipping step, loss scaler reducing loss scale to "
                      f"{scaler.loss_scale() * scaler.backoff_factor}")
        else:
            out = real_step(*a, **k)
        if scaler.enabled:
            scaler.after_step()
        return out
    optimizer.step = step
    optimizer._amp_patched = True


def _grads_of(optimizer):
    """The gradients the optimizer is going to consume (for a fused optimizer behind a fused-wire
    DistributedDataParallel these are views of the all-reduced wire buckets, not ``p.grad``)."""
    if hasattr(optimizer, "attach_scaler"):
        from ..optim.fused_sgd import effective_grad
        return [effective_grad(p)[0] for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    return [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]


def _overflow_check(opt, scaler: LossScaler) -> None:
    """Run the unscale / overflow pass over the gradients ``opt`` is going to consume.

    A fused optimizer divides by the loss scale inside its update kernel, so for it the pass only reads
    (``check_only``: it must not write a gradient back, since it has no business changing one)."""
    grads = _grads_of(opt)
    if not grads:
        return
    by_dtype = {}
    for g in grads:
        by_dtype.setdefault(g.dtype, []).append(g)
    fused = hasattr(opt, "attach_scaler")
    for gl in by_dtype.values():
        key = ("amp_unscale", tuple(g.data_ptr() for g in gl))
        cache = opt.__dict__.setdefault("_amp_tl_cache", {})
        tl = cache.get(key)
        if tl is None:
            if len(cache) > 16:
                cache.clear()
            tl = cache[key] = mt.TensorLists([gl, gl])
        if fused:
            mt.scale(tl, 1.0, noop_flag=scaler.overflow_flag, check_only=True)
        else:
            mt.scale(tl, 1.0, dev_scale=scaler.scale_tensor, invert=True, noop_flag=scaler.overflow_flag)
    scaler.grads_scaled = fused


def _finish_reductions(opt) -> None:
    """Gradient reductions an optimizer wrapper launched from backward hooks and only waits for inside
    ``step()`` (``hvd.DistributedOptimizer``) have to be finished BEFORE the overflow pass: the pass must see
    the reduced gradients - every rank then takes the same skip decision - and must not run next to the
    engine's unpack kernels that are still replacing ``p.grad``.  This is Horovod's recipe for apex amp
    (``optimizer.synchronize()`` inside ``scale_loss``, ``skip_synchronize()`` around ``step()``), done here
    so that the reference's plain ``scale_loss`` + ``step()`` loop
    (/root/reference/4.apex_distributed2.py:289-291 on /root/reference/5.horovod_distributed.py:125) is right."""
    sync = getattr(opt, "synchronize", None)
    if callable(sync) and hasattr(opt, "_handles"):
        sync()
        opt._synchronized_by_amp = True


@contextlib.contextmanager
def scale_loss(loss: torch.Tensor, optimizers, loss_id: int = 0, delay_unscale: bool = False):
    """Yield ``loss * scale``; on exit unscale the gradients in place and record overflow.

    With a fused optimizer the unscale is deferred into the update kernel: the exit
    only runs the overflow check (one read-only pass over the gradients), so the grads
    seen between the context manager and ``optimizer.step()`` are still scaled
    unless ``PDA_AMP_EAGER_UNSCALE=1`` / ``delay_unscale=False`` on a stock optimizer.
    """
    opts = list(optimizers) if isinstance(optimizers, (list, tuple)) else [optimizers]
    scaler: Optional[LossScaler] = getattr(opts[0], "_amp_scaler_ref", None) or (
        _STATE["scalers"][0] if _STATE["scalers"] else None)
    if scaler is None or not scaler.enabled:
        yield loss
        return
    yield loss.float() * scaler.scale_tensor.to(loss.device)[0]
    if delay_unscale:
        scaler.grads_scaled = True
        return
    for opt in opts:
        _finish_reductions(opt)
        _overflow_check(opt, scaler)


def state_dict() -> dict:
    return {f"loss_scaler{i}": s.state_dict() for i, s in enumerate(_STATE["scalers"])}


def load_state_dict(sd: dict) -> None:
    for i, s in enumerate(_STATE["scalers"]):
        if f"loss_scaler{i}" in sd:
            s.load_state_dict(sd[f"loss_scaler{i}"])


def master_params(optimizer):
    """Iterator over the fp32 parameters the optimizer updates (apex API)."""
    for g in optimizer.param_groups:
        yield from g["params"]
