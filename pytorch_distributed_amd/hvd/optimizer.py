"""``hvd.DistributedOptimizer``: wraps an optimizer so that every gradient is all-reduced
(averaged exactly once) while backward is still running.

Same construction as Horovod (/root/reference/5.horovod_distributed.py:125,
/root/reference/5.2.horovod_pytorch_mnist.py:181-185): a dynamic subclass of the wrapped
optimizer's class (``isinstance(opt, torch.optim.SGD)`` keeps holding), a post-accumulate hook
per parameter that issues ``allreduce_async_(grad, name='allreduce.<param name>')`` with the
chosen compression, and ``step()`` = ``synchronize()`` + the wrapped step.
``gradient_predivide_factor f`` scales by 1/f before and f/size after the sum;
``backward_passes_per_step`` delays the reduction for local gradient accumulation.
"""
from __future__ import annotations

import contextlib
import warnings

import torch

from . import mpi_ops as M


class _DistributedOptimizer(torch.optim.Optimizer):
    def __init__(self, params, named_parameters, compression, backward_passes_per_step=1, op=M.Average,
                 gradient_predivide_factor=1.0):
        super(self.__class__, self).__init__(params)
        self._compression = compression
        if named_parameters is not None:
            named_parameters = list(named_parameters)
        else:
            named_parameters = [(f"allreduce.noname.{i}.{j}", v) for i, g in enumerate(self.param_groups)
                                for j, v in enumerate(g["params"])]
        if any(not isinstance(p, tuple) for p in named_parameters):
            raise ValueError("named_parameters should be a sequence of tuples (name, parameter), "
                             "usually produced by model.named_parameters().")
        names = [n for n, _ in named_parameters]
        if len(set(names)) != len(names):
            raise ValueError("Parameter names in named_parameters must be unique.")
        all_ids = {id(v) for g in self.param_groups for v in g["params"]}
        named_ids = {id(v) for _, v in named_parameters}
        if all_ids - named_ids:
            raise ValueError("named_parameters was specified, but one or more model parameters were not named. "
                             "Python object ids: " + ", ".join(str(i) for i in (all_ids - named_ids)))
        self._parameter_names = {v: k for k, v in sorted(named_parameters)}
        self.backward_passes_per_step = backward_passes_per_step
        self._allreduce_delay = {v: backward_passes_per_step for _, v in sorted(named_parameters)}
        self.op = op
        self.gradient_predivide_factor = gradient_predivide_factor
        self._handles = {}
        self._hooks = []
        self._requires_update = set()
        self._synchronized = False
        self._should_synchronize = True
        if M.size() > 1 or True:
            self._register_hooks()

    def _register_hooks(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    self._requires_update.add(p)
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(p)))

    def _allreduce_grad_async(self, p):
        name = self._parameter_names.get(p)
        g = p.grad
        if not g.is_contiguous():
            g = torch.as_strided(g, (g.numel(),), (1,))  # dense, permuted storage: reduce the buffer itself
        if self.op == M.Average:
            f = self.gradient_predivide_factor
            # sum(g / f) * (f / size): the division by size() happens exactly once
            pre, post = 1.0 / f, f
            return M.allreduce_async_(g, name=name, op=M.Average, prescale_factor=pre, postscale_factor=post,
                                      compression=self._compression)
        return M.allreduce_async_(g, name=name, op=self.op, compression=self._compression)

    def _make_hook(self, p):
        def hook(*ignore):
            if p in self._handles and self._handles[p] is not None:
                if self._allreduce_delay[p] <= 0:
                    raise AssertionError("Gradients were computed more than backward_passes_per_step times before "
                                         "call to step(). Increase backward_passes_per_step to accumulate gradients "
                                         "locally.")
            assert not p.grad.requires_grad
            assert self._allreduce_delay[p] > 0
            handle = None
            self._allreduce_delay[p] -= 1
            if self._allreduce_delay[p] == 0:
                handle = self._allreduce_grad_async(p)
            self._handles[p] = handle
        return hook

    def synchronize(self):
        missing = self._requires_update - set(self._handles.keys())
        for p in missing:
            if p.grad is None:
                continue
            self._handles[p] = self._allreduce_grad_async(p)
        for p, handle in list(self._handles.items()):
            if handle is None and p.grad is not None:
                self._handles[p] = self._allreduce_grad_async(p)
        for p, handle in self._handles.items():
            if handle is None:
                continue
            M.synchronize(handle)
            self._allreduce_delay[p] = self.backward_passes_per_step
        self._handles.clear()
        self._synchronized = True

    @contextlib.contextmanager
    def skip_synchronize(self):
        self._should_synchronize = False
        try:
            yield
        finally:
            self._should_synchronize = True

    def step(self, closure=None):
        if self._should_synchronize:
            if self._synchronized:
                warnings.warn("optimizer.step() called without optimizer.skip_synchronize() context after "
                              "optimizer.synchronize(). This can cause training slowdown.")
            self.synchronize()
        self._synchronized = False
        return super(self.__class__, self).step(closure)

    def _hvd_super_step(self, closure=None):
        return super(self.__class__, self).step(closure)

    def zero_grad(self, *a, **k):
        if self._handles:
            raise AssertionError("optimizer.zero_grad() was called after loss.backward() but before optimizer.step() "
                                 "or optimizer.synchronize(). This is prohibited as it can cause a race condition.")
        return super(self.__class__, self).zero_grad(*a, **k)


def DistributedOptimizer(optimizer, named_parameters=None, compression=M.Compression.none,
                         backward_passes_per_step: int = 1, op=M.Average, gradient_predivide_factor: float = 1.0):
    if gradient_predivide_factor != 1.0 and op != M.Average:
        raise ValueError("gradient_predivide_factor not supported with op != Average")
    cls = type(optimizer.__class__.__name__, (optimizer.__class__,), dict(_DistributedOptimizer.__dict__))
    # carry the wrapped optimizer's state over by sharing its param_groups
    obj = cls.__new__(cls)
    obj.__dict__.update(optimizer.__dict__)
    _DistributedOptimizer_init(obj, named_parameters, compression, backward_passes_per_step, op,
                               gradient_predivide_factor)
    return obj


def _DistributedOptimizer_init(self, named_parameters, compression, backward_passes_per_step, op,
                               gradient_predivide_factor):
    # same body as _DistributedOptimizer.__init__ minus the base-class constructor (state already copied)
    self._compression = compression
    if named_parameters is not None:
        named_parameters = list(named_parameters)
    else:
        named_parameters = [(f"allreduce.noname.{i}.{j}", v) for i, g in enumerate(self.param_groups)
                            for j, v in enumerate(g["params"])]
    if any(not isinstance(p, tuple) for p in named_parameters):
        raise ValueError("named_parameters should be a sequence of tuples (name, parameter), "
                         "usually produced by model.named_parameters().")
    names = [n for n, _ in named_parameters]
    if len(set(names)) != len(names):
        raise ValueError("Parameter names in named_parameters must be unique.")
    all_ids = {id(v) for g in self.param_groups for v in g["params"]}
    named_ids = {id(v) for _, v in named_parameters}
    if all_ids - named_ids:
        raise ValueError("named_parameters was specified, but one or more model parameters were not named.")
    self._parameter_names = {v: k for k, v in sorted(named_parameters, key=lambda kv: kv[0])}
    self.backward_passes_per_step = backward_passes_per_step
    self._allreduce_delay = {v: backward_passes_per_step for _, v in named_parameters}
    self.op = op
    self.gradient_predivide_factor = gradient_predivide_factor
    self._handles = {}
    self._hooks = []
    self._requires_update = set()
    self._synchronized = False
    self._should_synchronize = True
    self._register_hooks()
