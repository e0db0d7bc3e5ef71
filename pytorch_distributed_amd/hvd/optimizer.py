"""``hvd.DistributedOptimizer``: wraps an optimizer so that every gradient is all-reduced
(averaged exactly once) while backward is still running.

API and behaviour follow what the reference scripts use (/root/reference/5.horovod_distributed.py:125,
/root/reference/5.2.horovod_pytorch_mnist.py:181-185): the returned object is an instance of a dynamic
subclass of the wrapped optimizer's class (``isinstance(opt, torch.optim.SGD)`` keeps holding) that
shares the wrapped optimizer's state; a post-accumulate hook per parameter issues
``allreduce_async_(grad, name='allreduce.<param name>')`` with the chosen compression, and ``step()``
is ``synchronize()`` followed by the wrapped step.  ``gradient_predivide_factor f`` scales by 1/f before
and f/size after the sum; ``backward_passes_per_step`` delays the reduction for local accumulation.
"""
from __future__ import annotations

import contextlib
import warnings

import torch

from . import mpi_ops as M


class _Reducing:
    """Mixed into the wrapped optimizer's class; carries only the gradient-reduction behaviour."""

    def _hvd_setup(self, named_parameters, compression, backward_passes_per_step, op, gradient_predivide_factor):
        params = [p for g in self.param_groups for p in g["params"]]
        if named_parameters is None:
            named = [(f"allreduce.noname.{i}", p) for i, p in enumerate(params)]
        else:
            named = list(named_parameters)
            if any(not (isinstance(e, tuple) and len(e) == 2) for e in named):
                raise ValueError("named_parameters should be a sequence of (name, parameter) tuples, "
                                 "usually model.named_parameters()")
        names = [n for n, _ in named]
        if len(set(names)) != len(names):
            raise ValueError("parameter names in named_parameters must be unique")
        unnamed = {id(p) for p in params} - {id(p) for _, p in named}
        if unnamed:
            raise ValueError(f"named_parameters misses {len(unnamed)} of the optimizer's parameters")
        self._compression = compression
        self._name_of = {p: n for n, p in named}
        self.backward_passes_per_step = int(backward_passes_per_step)
        self._passes_left = {p: self.backward_passes_per_step for _, p in named}
        self.op = op
        self.gradient_predivide_factor = float(gradient_predivide_factor)
        self._handles = {}
        self._synchronized = False
        self._synchronized_by_amp = False
        self._should_synchronize = True
        self._reduced_params = [p for p in params if p.requires_grad]
        self._direct_fired = set()
        self._hooks = [p.register_post_accumulate_grad_hook(self._hvd_autograd_hook) for p in self._reduced_params]
        # This package's fused conv / BatchNorm ops accumulate weight gradients straight into an existing
        # ``.grad`` (no AccumulateGrad node, so no autograd hook): they announce such a gradient through
        # ``_pda_grad_ready`` instead, which keeps the reduction overlapped with the rest of backward.
        for p in self._reduced_params:
            ddp = getattr(p, "_pda_ddp", None)
            if ddp is not None and ddp() is not None:
                continue            # a live DistributedDataParallel wrapper owns this parameter's announcements
            old = getattr(getattr(p, "_pda_grad_ready", None), "_hvd_owner", None)
            old = old() if old is not None else None
            if old is not None and old is not self:
                old._hvd_release()  # a re-wrap (resume, second wrapper in one process): the old wrapper lets go
            p._pda_grad_ready = self._make_direct_ready(p)

    def _hvd_release(self):
        """Detach from the parameters: remove the autograd hooks and the direct-announcement callbacks."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self._reduced_params:
            if getattr(getattr(p, "_pda_grad_ready", None), "_hvd_owner", None) is not None \
                    and p._pda_grad_ready._hvd_owner() is self:
                del p._pda_grad_ready

    def _make_direct_ready(self, p):
        def ready():
            if p.grad is None:
                return
            # the gradient may have been launched on the backward side stream: enqueue the reduction from that
            # stream (the engine orders itself after an event of the enqueuing stream), so the input-gradient
            # chain on the main stream does not wait for weight gradients it never reads
            from ..ops import streams
            side = streams.pending(p.grad.device) if p.grad.is_cuda else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(p.grad.device))
                with torch.cuda.stream(side):
                    self._hvd_hook(p)
            else:
                self._hvd_hook(p)
            self._direct_fired.add(p)
        import weakref
        ready._hvd_owner = weakref.ref(self)
        return ready

    def _hvd_autograd_hook(self, p):
        # autograd runs a parameter's AccumulateGrad node - and with it this hook - even when the op returned no
        # gradient for it because it had already accumulated into ``.grad`` and announced that itself
        if p in self._direct_fired:
            self._direct_fired.discard(p)
            return
        self._hvd_hook(p)

    # -- gradient hooks -----------------------------------------------------------------------
    def _hvd_launch(self, p):
        g = p.grad
        if not g.is_contiguous():
            g = torch.as_strided(g, (g.numel(),), (1,))  # dense but permuted storage: reduce the buffer itself
        name = self._name_of.get(p)
        if self.op == M.Average:
            f = self.gradient_predivide_factor
            # sum(g / f) * (f / size): the division by size() happens exactly once
            return M.allreduce_async_(g, name=name, op=M.Average, prescale_factor=1.0 / f, postscale_factor=f,
                                      compression=self._compression)
        return M.allreduce_async_(g, name=name, op=self.op, compression=self._compression)

    def _hvd_hook(self, p):
        if self._handles.get(p) is not None:
            raise AssertionError(f"the gradient of {self._name_of.get(p, '<unnamed>')!r} was produced again before "
                                 "step()/synchronize() consumed the previous reduction: raise "
                                 "backward_passes_per_step to accumulate locally")
        self._passes_left[p] -= 1
        self._handles[p] = self._hvd_launch(p) if self._passes_left[p] <= 0 else None

    # -- public API ---------------------------------------------------------------------------
    def synchronize(self):
        """Wait for every outstanding gradient reduction (launching the ones whose hook never fired)."""
        for p in self._reduced_params:
            if p.grad is not None and self._handles.get(p) is None:
                self._handles[p] = self._hvd_launch(p)
        for p, h in self._handles.items():
            if h is not None:
                M.synchronize(h)
                self._passes_left[p] = self.backward_passes_per_step
        self._handles.clear()
        self._direct_fired.clear()
        self._synchronized = True

    @contextlib.contextmanager
    def skip_synchronize(self):
        """``with opt.skip_synchronize(): opt.step()`` after an explicit ``opt.synchronize()``
        (gradient clipping between the two)."""
        self._should_synchronize = False
        try:
            yield
        finally:
            self._should_synchronize = True

    def step(self, closure=None):
        if self._should_synchronize:
            if self._synchronized_by_amp and not self._pending_work():
                pass    # amp.scale_loss has already waited for this step's reductions (before its overflow check)
            else:
                if self._synchronized:
                    warnings.warn("optimizer.step() after optimizer.synchronize() without skip_synchronize(): "
                                  "gradients are reduced twice as slowly as needed")
                self.synchronize()
        self._synchronized = False
        self._synchronized_by_amp = False
        if M._Diag.on():
            ps = [p for p in self._reduced_params if p.grad is not None]
            if ps and ps[0].grad.is_cuda:
                torch.cuda.synchronize()
            M._Diag.step += 1
            M._Diag.write({"step": M._Diag.step - 1, "step_grads": {self._name_of.get(p, "?"): v
                                          for p, v in zip(ps, M._Diag.sums([p.grad for p in ps]))}})
        return super().step(closure)

    def _pending_work(self) -> bool:
        return any(h is not None for h in self._handles.values())

    def _hvd_super_step(self, closure=None):
        """The wrapped optimizer's own step, without gradient reduction (state materialisation in
        ``broadcast_optimizer_state``)."""
        return super().step(closure)

    def zero_grad(self, *a, **k):
        if any(h is not None for h in self._handles.values()):
            raise AssertionError("optimizer.zero_grad() between loss.backward() and optimizer.step(): the gradients "
                                 "are still being reduced")
        return super().zero_grad(*a, **k)


def DistributedOptimizer(optimizer, named_parameters=None, compression=M.Compression.none,
                         backward_passes_per_step: int = 1, op=M.Average, gradient_predivide_factor: float = 1.0):
    if gradient_predivide_factor != 1.0 and op != M.Average:
        raise ValueError("gradient_predivide_factor not supported with op != Average")
    base = optimizer.__class__
    cls = type(base.__name__, (_Reducing, base), {})
    obj = cls.__new__(cls)
    obj.__dict__.update(optimizer.__dict__)     # share param_groups / state / hyper-parameters with the wrapped one
    if obj.__dict__.get("_amp_patched") and "step" in obj.__dict__:
        # amp.initialize() patched the wrapped optimizer's instance with a step() that checks the overflow flag and
        # then calls the wrapped step directly; copied as it is, it would shadow this class's step() and with it
        # synchronize().  Patch the wrapper instead: flag check -> synchronize -> wrapped step
        del obj.__dict__["step"]
        obj._amp_patched = False
        from ..amp.frontend import _attach
        _attach(obj, obj._amp_scaler_ref)
    obj._hvd_setup(named_parameters, compression, backward_passes_per_step, op, gradient_predivide_factor)
    return obj
