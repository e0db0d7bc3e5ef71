"""DistributedDataParallel: bucketed gradient all-reduce over RCCL/xGMI, overlapped
with backward.

What the reference gets from ``torch.nn.parallel.DistributedDataParallel(model,
device_ids=[local_rank])`` (/root/reference/2.distributed.py:114,
/root/reference/3.multiprocessing_distributed.py:120, /root/reference/6.distributed_slurm_main.py:116)
and from ``apex.parallel.DistributedDataParallel(model)``
(/root/reference/4.apex_distributed2.py:178), redesigned for an 8-GPU xGMI mesh:

* construction broadcasts rank 0's parameters and buffers (coalesced: one flatten
  kernel + one broadcast per dtype) after checking that every rank holds the same
  parameter shapes;
* gradients live *inside* persistent flat bucket buffers (``param.grad`` is a view),
  so there is no copy-in/copy-out and ``zero_grad`` is one memset per bucket;
* the bucket plan and the ready-tracking run in the C++ reducer core
  (``csrc/runtime/reducer.cc``): reverse registration order, 25 MiB cap with a 1 MiB
  first bucket (torch's defaults) - buckets are padded so that their 8 shards stay
  vector aligned, and ``bucket_cap_mb`` in the 16-32 MiB range keeps every xGMI
  link's share of a direct reduce-scatter/all-gather above a few MiB;
* a bucket's all-reduce is issued (async, on RCCL's own stream, which first waits for
  the producing kernels) the moment its last gradient has been produced, strictly in
  bucket order on every rank; the end-of-backward callback makes the compute stream
  wait for the outstanding collectives and applies the 1/world average;
* ``wire_dtype`` (default ``"auto"``: the 16-bit compute dtype ``amp.initialize`` gave the model under
  O1-O3, else the gradient dtype) halves the bytes on the links - the Horovod ``Compression.fp16`` idea
  (/root/reference/5.horovod_distributed.py:123).  A ready bucket is cast into its persistent wire
  buffer by ONE multi-tensor launch that also applies the pre-division (1/world for fp16 so the sum
  cannot leave the fp16 range) and flags overflow; after the all-reduce one launch per bucket unpacks the
  reduced gradient into ``param.grad`` - which therefore holds the averaged gradient after backward, as it
  does under torch's and apex's wrappers (clipping, logging and any optimizer see the same thing).  With
  ``fused_wire=True`` (opt-in; env ``PDA_DDP_FUSED_WIRE=1``) a fused optimizer (``optim.FusedSGD``) reads the
  all-reduced wire buffer directly instead, with the remaining 1/world folded into its update kernel, and the
  unpack pass is skipped: ``param.grad`` then keeps the LOCAL gradient, and code between backward and step
  has to go through ``reduced_grad()`` / ``clip_grad_norm_()`` of this wrapper;
* after the first backward the plan is rebuilt in gradient-arrival order (rank 0's
  order is broadcast so every rank rebuilds identically);
* module buffers (BatchNorm running stats) are re-broadcast from rank 0 before each
  training forward when ``broadcast_buffers=True`` (torch's default behaviour).

``.module`` exposes the wrapped model and ``state_dict()`` keys carry the
``module.`` prefix exactly like torch's wrapper.
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Variable

from ..ops import multi_tensor as mt, streams as _streams
from . import _pyext
from .reducer import BucketPlan

_MiB = 1 << 20
_P2P_SEQ = 0


def _physical_flat(t: torch.Tensor) -> torch.Tensor:
    return torch.as_strided(t, (t.numel(),), (1,))


_HOST_T: Dict[str, List[float]] = {}


def _install_host_timers(ddp) -> None:
    """Diagnostics (``PDA_DDP_HOST_TIMERS=1``): host seconds and call counts of the wrapper's own work per phase,
    printed at exit - the hooks run on autograd's thread, where cProfile does not see them."""
    import atexit
    import time

    def timed(name, fn):
        rec = _HOST_T.setdefault(name, [0.0, 0])

        def run(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                rec[0] += time.perf_counter() - t0
                rec[1] += 1
        return run

    for name in ("_launch", "_pack_and_reduce", "_finalize_backward", "_first_ready", "_broadcast_buffers_cached"):
        setattr(ddp, name, timed(name, getattr(ddp, name)))
    coll = ddp._coll
    for name in ("all_reduce", "broadcast"):
        try:
            setattr(coll, name, timed("coll." + name, getattr(coll, name)))
        except Exception:
            pass
    atexit.register(lambda: print("DDP host timers: " + ", ".join(
        f"{k} {1e3 * v[0]:.1f} ms / {v[1]} calls" for k, v in _HOST_T.items()), flush=True))


class _Bucket:
    __slots__ = ("index", "flat", "params", "views", "work", "wire", "launched", "pack", "unpack", "wire_views")

    def __init__(self, index, flat, params):
        self.index, self.flat, self.params = index, flat, params
        self.views: List[torch.Tensor] = []
        self.work = None
        self.wire = None          # persistent low-precision copy that travels over the links
        self.pack = None          # cached multi-tensor tables: flat -> wire
        self.unpack = None        # wire -> flat
        self.wire_views = None    # per-parameter views of the wire buffer (fused optimizer hand-over)
        self.launched = False


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, output_device=None, dim: int = 0,
                 broadcast_buffers: bool = True, process_group=None, bucket_cap_mb: float = 25.0,
                 first_bucket_mb: float = 1.0, find_unused_parameters: bool = False,
                 gradient_as_bucket_view: bool = True, wire_dtype="auto", fused_wire: Optional[bool] = None,
                 gradient_average: bool = True, gradient_predivide_factor: float = 1.0,
                 delay_allreduce: bool = False, message_size: Optional[int] = None,
                 rebuild_buckets: bool = True, check_shapes: bool = True, use_avg_op: bool = False,
                 comm=None, force_comm: Optional[bool] = None):
        """``comm``: collective back-end.  ``None`` (default; env ``PDA_DDP_COMM`` overrides) = this
        package's own RCCL communicator (``comm.rccl.RcclCommunicator``: own comm stream, overlap with
        backward) when the parameters live on a GPU, the ``torch.distributed`` group ``process_group``
        (gloo) when they live on the CPU; ``"native"`` / ``"torch"`` / ``"p2p2"`` force one of the own RCCL
        communicator, ``torch.distributed`` (ProcessGroupNCCL on GPUs) or the peer-memory two-shot
        all-reduce (``comm.p2p.P2PCollectives``); any object with that interface is used as is.
        ``force_comm`` (default: env ``PDA_DDP_FORCE_COMM=1``) runs the collectives even for a single
        rank, which is how the communication path is exercised on a one-GPU machine."""
        super().__init__()
        from ..comm.rccl import TorchCollectives
        if comm is None:
            comm = os.environ.get("PDA_DDP_COMM") or None
        if comm is None or isinstance(comm, str):
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("DistributedDataParallel needs an initialised process group "
                                   "(pytorch_distributed_amd.comm.init_process_group)")
            base = TorchCollectives(process_group)
            dev = next(p for p in module.parameters() if p.requires_grad).device
            if comm is None:
                comm = "native" if dev.type == "cuda" else "torch"
            if comm == "native":
                from ..comm import device_communicator
                comm = device_communicator(dev, group=process_group)
                if comm is None:
                    raise RuntimeError("no RCCL communicator could be created for " + str(dev))
            elif comm == "p2p2":
                from ..comm.p2p import P2PCollectives
                from ..comm import native_store, PrefixStore
                st = native_store()
                # (a key prefix of its own per wrapper: every rank builds its wrappers in the same order, and a
                # second set of peer regions must not find the IPC handles the first one published)
                global _P2P_SEQ
                _P2P_SEQ += 1
                comm = P2PCollectives(base.rank, base.world_size, device=dev, overlap=True,
                                      store=PrefixStore(f"ddp_p2p{_P2P_SEQ}", st) if st is not None else None,
                                      key=f"pda/p2pc{_P2P_SEQ}")
            elif comm == "torch":
                comm = base
            else:
                raise ValueError(f"unknown comm {comm!r} (native, torch or p2p2)")
            self.process_group = base.group
        else:
            self.process_group = process_group
        self._coll = comm
        self.module = module
        self.world_size = comm.world_size
        self.rank = comm.rank
        if force_comm is None:
            force_comm = os.environ.get("PDA_DDP_FORCE_COMM", "0") == "1"
        self._active = self.world_size > 1 or bool(force_comm)
        self.device_ids = device_ids
        self.output_device = output_device
        self.broadcast_buffers = broadcast_buffers
        self.find_unused_parameters = find_unused_parameters
        self.gradient_average = gradient_average
        self.predivide = float(gradient_predivide_factor)
        self.delay_allreduce = delay_allreduce
        self._hook_state = None     # shared flags of the C ready hooks (set below when the extension is built)
        self._require_sync = True
        self._rebuild_pending = rebuild_buckets and not delay_allreduce and self._active
        self._params: List[nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        self._index: Dict[int, int] = {id(p): i for i, p in enumerate(self._params)}
        if not self._params:
            raise RuntimeError("DistributedDataParallel: module has no parameter that requires grad")
        self.device = self._params[0].device
        self._on_device = bool(comm.device_collectives)
        self._exposed_probe = None
        self._buf_list = None
        self._frozen_params = [p for p in module.parameters() if not p.requires_grad]
        if isinstance(wire_dtype, str):
            if wire_dtype != "auto":
                wire_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": None, "none": None}[wire_dtype]
            else:
                # mixed precision on (amp.initialize switched a model of this package to a 16-bit compute
                # dtype, or wrapped a foreign module's forward in autocast): gradients travel in that dtype
                low = [getattr(m, "compute_dtype", None) for m in module.modules()]
                low = [d for d in low if d in (torch.float16, torch.bfloat16)]
                if not low and getattr(module.forward, "_pda_autocast_dtype", None) is not None:
                    low = [module.forward._pda_autocast_dtype]
                wire_dtype = low[0] if low and self._active else None
        if wire_dtype == self._params[0].dtype:
            wire_dtype = None
        self.wire_dtype = wire_dtype
        # fp16 on the wire: pre-divide by the world size so that the SUM stays in range (apex's
        # gradient_predivide_factor, chosen automatically unless the caller picked one)
        self._pre = self.predivide
        if wire_dtype == torch.float16 and self.predivide == 1.0 and gradient_average:
            self._pre = float(self.world_size)
        # fused_wire (opt-in): the hand-over starts as soon as a fused optimizer announces itself (FusedSGD does,
        # whether it is built before or after this wrapper); default: always unpack into param.grad
        if fused_wire is None:
            fused_wire = os.environ.get("PDA_DDP_FUSED_WIRE", "0") == "1"
        self._fused_wire_allowed = bool(fused_wire)
        self.fused_wire = False
        # SUM + one scale pass by default; RCCL's AVG op saves that pass but is opt-in (use_avg_op)
        self._use_avg_op = (use_avg_op and self._on_device and gradient_average and self.predivide == 1.0
                            and wire_dtype is None)
        cap = bucket_cap_mb * _MiB
        first = first_bucket_mb * _MiB
        if message_size is not None:  # apex flavour: bucket size given in elements (default 1e7 there)
            cap = first = message_size * self._params[0].element_size()
        if delay_allreduce:
            cap = first = 1 << 62
        if check_shapes and self._active:
            self._verify_shapes()
        self._sync_module_states()
        self._callback_queued = False
        self._in_backward = False
        self._grad_dtype = self._params[0].dtype
        sizes = [p.numel() * p.element_size() for p in self._params]
        self._plan = BucketPlan(sizes, int(cap), int(first))
        self._buckets: List[_Bucket] = []
        self._build_buckets()
        if os.environ.get("PDA_DDP_HOST_TIMERS") == "1":
            _install_host_timers(self)
        self._hooks = []
        # The per-parameter ready hook as a C callable (csrc/pyext/ready_hook.cc) when the extension is built:
        # the .grad identity check and the plan's mark_ready() without a Python frame or a ctypes transition.
        ext = _pyext.module() if os.environ.get("PDA_DDP_NATIVE_HOOKS", "1") != "0" else None
        self._native_hooks = []
        if ext is not None:
            self._hook_state = ext.HookState()
            self._hook_state.enabled = int(self._require_sync and self._active)
            mark = self._plan.mark_ready_address()
            n = len(self._params)
            self._native_hooks = [ext.ReadyHook(self._hook_state, self._plan.handle(), mark, i, n, self._view_of[i],
                                                self._first_ready, self._launch_index, self._fold_grad)
                                  for i in range(n)]
        for i, p in enumerate(self._params):
            hook = self._native_hooks[i] if self._native_hooks else None
            self._hooks.append(p.register_post_accumulate_grad_hook(hook if hook is not None else self._make_hook(i)))
            # convolution ops of this package may write straight into .grad and call this instead
            p._pda_direct_grad = True
            p._pda_grad_ready = hook if hook is not None else self._make_direct_ready(i)
            p._pda_ddp = weakref.ref(self)
            if getattr(p, "_pda_fused_opt", False):
                self.enable_fused_wire()

    # ------------------------------------------------------------------ construction helpers
    def _verify_shapes(self) -> None:
        sig = torch.tensor([len(self._params)] + [p.numel() for p in self._params], dtype=torch.int64)
        dev = self.device if self._on_device else torch.device("cpu")
        sig = sig.to(dev)
        ref = sig.clone()
        self._coll.broadcast(ref, 0)
        if ref.shape != sig.shape or not torch.equal(ref, sig):
            raise RuntimeError(f"rank {self.rank}: parameter shapes differ from rank 0's "
                               "(DistributedDataParallel needs identical replicas)")

    def _broadcast_buffers_cached(self, bufs: List[torch.Tensor]) -> None:
        """The per-forward buffer broadcast with everything that does not change from step to step kept:
        one persistent flat buffer per dtype, the device address tables of the pack / unpack launches, and
        a multi-tensor copy for the integer buffers (53 ``num_batches_tracked`` scalars in a ResNet-50 are
        one launch each way instead of 53 eight-byte copies)."""
        key = tuple(b.data_ptr() for b in bufs)
        plan = getattr(self, "_buf_plan", None)
        if plan is None or plan[0] != key:
            groups = []
            by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
            for t in bufs:
                by_dtype.setdefault(t.dtype, []).append(t.detach())
            for dt, ts in by_dtype.items():
                flat = torch.empty(sum(t.numel() for t in ts), device=ts[0].device, dtype=dt)
                views, off = [], 0
                for t in ts:
                    views.append(flat[off:off + t.numel()].view(t.shape) if t.is_contiguous()
                                 else flat[off:off + t.numel()])
                    off += t.numel()
                if dt in (torch.float32, torch.float16, torch.bfloat16) and ts[0].is_cuda:
                    phys = [_physical_flat(t) for t in ts]
                    fviews = [v.reshape(-1) for v in views]
                    groups.append(("mt", flat, mt.TensorLists([phys, fviews]), mt.TensorLists([fviews, phys])))
                else:
                    groups.append(("foreach", flat, ts, views))
            plan = (key, groups)
            self._buf_plan = plan
        for kind, flat, a, b in plan[1]:
            if kind == "mt":
                mt.scale(a, 1.0)
                self._coll.broadcast(flat, 0)
                mt.scale(b, 1.0)
            else:
                torch._foreach_copy_(b, a)
                self._coll.broadcast(flat, 0)
                torch._foreach_copy_(a, b)

    def _broadcast_coalesced(self, tensors: List[torch.Tensor]) -> None:
        """rank 0 -> all, one flat buffer per dtype (flatten / unflatten are single launches)."""
        if not self._active or not tensors:
            return
        by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dt, ts in by_dtype.items():
            if dt.is_floating_point and dt in (torch.float32, torch.float16, torch.bfloat16):
                flats = [_physical_flat(t.detach()) for t in ts]
                buf = mt.flatten(flats)
                self._coll.broadcast(buf, 0)
                mt.unflatten(buf, flats)
            else:
                buf = torch.cat([t.detach().reshape(-1) for t in ts])
                self._coll.broadcast(buf, 0)
                off = 0
                for t in ts:
                    t.detach().copy_(buf[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def _sync_module_states(self) -> None:
        with torch.no_grad():
            self._broadcast_coalesced([p for p in self.module.parameters()] + [b for b in self.module.buffers()])
        from ..ops.conv import bump_param_generation
        bump_param_generation()

    def _build_buckets(self) -> None:
        old = {id(p): p.grad for p in self._params}
        self._buckets = []
        self._view_of: Dict[int, torch.Tensor] = {}
        for b in range(self._plan.num_buckets):
            idxs = self._plan.bucket_params(b)
            nbytes = self._plan.bucket_bytes(b)
            esz = self._params[0].element_size()
            flat = torch.zeros(nbytes // esz, device=self.device, dtype=self._grad_dtype)
            bk = _Bucket(b, flat, idxs)
            for i in idxs:
                p = self._params[i]
                off = self._plan.offset_of(i) // esz
                view = torch.as_strided(flat, p.shape, p.stride(), off) if p.is_contiguous() or _dense(p) \
                    else flat[off:off + p.numel()].view_as(p)
                prev = old.get(id(p))
                if prev is not None:
                    view.copy_(prev)
                p.grad = view
                bk.views.append(view)
                self._view_of[i] = view
            self._buckets.append(bk)
        for i, h in enumerate(getattr(self, "_native_hooks", ())):
            h.view = self._view_of[i]

    # ------------------------------------------------------------------ hooks
    @property
    def require_backward_grad_sync(self) -> bool:
        return self._require_sync

    @require_backward_grad_sync.setter
    def require_backward_grad_sync(self, on: bool) -> None:
        self._require_sync = bool(on)
        if self._hook_state is not None:
            self._hook_state.enabled = int(self._require_sync and self._active)

    # callbacks of the C hooks: first ready gradient of a backward, a completed bucket, a replaced .grad
    def _first_ready(self) -> None:
        self._callback_queued = True
        self._in_backward = True
        Variable._execution_engine.queue_callback(self._finalize_backward)

    def _launch_index(self, b: int) -> None:
        self._launch(self._buckets[b])

    def _fold_grad(self, i: int, param) -> None:
        view = self._view_of[i]
        g = param.grad
        if g is not None and g is not view and g.data_ptr() != view.data_ptr():
            view.copy_(g)
            param.grad = view
        elif g is not None and g is not view:
            param.grad = view   # another Python object over the same memory: restore the identity the hook checks

    def _make_hook(self, i: int):
        def hook(param):
            self._on_grad_ready(i, param)
        return hook

    def _make_direct_ready(self, i: int):
        def ready():
            self._on_grad_ready(i, self._params[i])
        return ready

    def _on_grad_ready(self, i: int, param) -> None:
        if not self.require_backward_grad_sync or not self._active:
            return
        view = self._view_of[i]
        g = param.grad
        if g is not view and g is not None and g.data_ptr() != view.data_ptr():
            # someone replaced .grad (zero_grad(set_to_none=True) or a fresh accumulation): fold it back
            view.copy_(g)
            param.grad = view
        if not self._callback_queued:
            self._callback_queued = True
            self._in_backward = True
            Variable._execution_engine.queue_callback(self._finalize_backward)
        for b in self._plan.mark_ready(i):
            self._launch(self._buckets[b])

    def enable_fused_wire(self) -> None:
        """Called by a fused optimizer that consumes ``optim.fused_sgd.effective_grad``: from now on the
        reduced gradients stay in the wire buckets (``param.grad`` keeps the local gradient)."""
        if self._fused_wire_allowed and self.wire_dtype is not None:
            self.fused_wire = True

    def _wire_of(self, bk: _Bucket) -> torch.Tensor:
        if bk.wire is None:
            bk.wire = torch.zeros_like(bk.flat, dtype=self.wire_dtype)
            bk.pack = mt.TensorLists([[bk.flat], [bk.wire]])
            bk.unpack = mt.TensorLists([[bk.wire], [bk.flat]])
            esz = self._params[0].element_size()
            bk.wire_views = []
            for i in bk.params:
                p = self._params[i]
                off = self._plan.offset_of(i) // esz
                bk.wire_views.append(torch.as_strided(bk.wire, p.shape, p.stride(), off)
                                     if p.is_contiguous() or _dense(p) else bk.wire[off:off + p.numel()].view_as(p))
        return bk.wire

    def _launch(self, bk: _Bucket) -> None:
        # The bucket's weight gradients may still be running on the backward pass' side stream.  The pack and the
        # collective launch then go to that stream too (after whatever the main stream has written into the bucket
        # so far): the all-reduce is ordered behind the gradients it sends while the input-gradient chain on the
        # main stream keeps running.
        side = _streams.pending(bk.flat.device) if self._on_device else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(bk.flat.device))
            with torch.cuda.stream(side):
                self._pack_and_reduce(bk)
        else:
            self._pack_and_reduce(bk)
        bk.launched = True

    def _pack_and_reduce(self, bk: _Bucket) -> None:
        buf = bk.flat
        if self.wire_dtype is not None:
            wire = self._wire_of(bk)
            mt.scale(bk.pack, 1.0 / self._pre)      # cast + pre-division, one launch for the whole bucket
            buf = wire
        elif self._pre != 1.0:
            buf.mul_(1.0 / self._pre)
        bk.work = self._coll.all_reduce(buf, "avg" if self._use_avg_op else "sum", async_op=True)

    def _finalize_backward(self) -> None:
        self._callback_queued = False
        self._in_backward = False
        if self._hook_state is not None:
            self._hook_state.queued = 0
        _streams.join()
        if not self.require_backward_grad_sync or not self._active:
            return
        if self._plan.num_unready():
            if not self.find_unused_parameters and not self.delay_allreduce:
                # parameters that received no gradient this pass: their (zero) slots still take part
                pass
            for b in self._plan.flush_unready():
                self._launch(self._buckets[b])
        probe = self._exposed_probe if self._on_device else None
        if probe is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
        post = 1.0
        if self.gradient_average and not self._use_avg_op:
            post = self._pre / self.world_size
        elif not self.gradient_average and self._pre != 1.0:
            post = self._pre
        for bk in self._buckets:
            if bk.work is None:
                continue
            bk.work.wait()
            bk.work = None
            if bk.wire is not None and self.wire_dtype is not None:
                if self.fused_wire:
                    for i, v in zip(bk.params, bk.wire_views):
                        self._params[i]._pda_wire_grad = (v, post)
                else:
                    mt.scale(bk.unpack, post)
            elif post != 1.0:
                bk.flat.mul_(post)
            bk.launched = False
        if probe is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            probe.append((e1, e2))
            if len(probe) > 512:
                del probe[:256]
        order = self._plan.arrival_order() if self._rebuild_pending else None
        self._plan.reset()
        if self._rebuild_pending:
            self._rebuild_pending = False
            self._plan.record_arrival(False)
            self._rebuild(order)

    def _rebuild(self, order: List[int]) -> None:
        n = len(self._params)
        t = torch.full((n,), -1, dtype=torch.int64)
        t[:len(order)] = torch.tensor(order, dtype=torch.int64)
        dev = self.device if self._on_device else torch.device("cpu")
        t = t.to(dev)
        self._coll.broadcast(t, 0)
        order = [int(v) for v in t.tolist() if v >= 0]
        self._plan.rebuild(order)
        self._build_buckets()

    # ------------------------------------------------------------------ public API
    def forward(self, *inputs, **kwargs):
        if self.broadcast_buffers and self.training and self._active and torch.is_grad_enabled():
            # (the walk over the module tree is kept between steps: ~100 buffers of a ResNet cost 0.1 ms of host
            # time per forward to enumerate; anything that replaces buffers goes through _apply / load_state_dict /
            # register_buffer on a sub-module, after which refresh_buffers() - or the next train() call - re-walks)
            bufs = self._buf_list
            if bufs is None:
                bufs = self._buf_list = [b for b in self.module.buffers()]
            if bufs:
                with torch.no_grad():
                    self._broadcast_buffers_cached(bufs)
        return self.module(*inputs, **kwargs)

    def refresh_buffers(self) -> None:
        """Re-enumerate the module's buffers at the next forward (after adding / replacing buffers by hand)."""
        self._buf_list = None

    def _apply(self, fn, *a, **k):
        self._buf_list = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._buf_list = None
        return super().load_state_dict(*a, **k)

    @contextlib.contextmanager
    def no_sync(self):
        """Accumulate gradients locally; the next backward outside the context reduces them."""
        old = self.require_backward_grad_sync
        self.require_backward_grad_sync = False
        self._drop_wire_handover()
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    def _drop_wire_handover(self) -> None:
        for p in self._params:
            if getattr(p, "_pda_wire_grad", None) is not None:
                p._pda_wire_grad = None

    def reduced_grad(self, param: torch.Tensor) -> torch.Tensor:
        """The all-reduced (averaged) gradient of ``param`` as an fp32 tensor, wherever it currently lives
        (``param.grad``, or the wire bucket in fused-wire mode)."""
        w = getattr(param, "_pda_wire_grad", None)
        if w is not None:
            return w[0].float() * w[1]
        return param.grad

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float, eps: float = 1e-6) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_`` over the REDUCED gradients, wherever they live (``param.grad``, or
        the wire buckets in fused-wire mode, whose pending 1/world factor is accounted for).  No host
        synchronisation: the norm and the clip coefficient stay on the device.  Returns the total norm."""
        plain, wired = [], {}
        for p in self._params:
            w = getattr(p, "_pda_wire_grad", None)
            if w is not None:
                wired.setdefault(float(w[1]), []).append(w[0])
            elif p.grad is not None:
                plain.append(p.grad)
        sq = []
        if plain:
            sq.append(mt.l2norm(mt.TensorLists([[_physical_flat(g) for g in plain]]))[0].float() ** 2)
        for post, gs in wired.items():
            sq.append((mt.l2norm(mt.TensorLists([[_physical_flat(g) for g in gs]]))[0].float() * post) ** 2)
        total = torch.sqrt(torch.stack(sq).sum()) if sq else torch.zeros((), device=self.device)
        coef = torch.clamp(max_norm / (total + eps), max=1.0)
        for gs in [plain] + list(wired.values()):
            for dt in {g.dtype for g in gs}:
                torch._foreach_mul_([g for g in gs if g.dtype == dt], coef.to(dt))
        return total

    def zero_grad(self, set_to_none: bool = False) -> None:
        """One memset per bucket; keeps ``param.grad`` pointing into the buckets."""
        if self.fused_wire:
            self._drop_wire_handover()   # a step() without a backward in between must not reuse old wire gradients
        for bk in self._buckets:
            bk.flat.zero_()
        for bk in self._buckets:
            for i, v in zip(bk.params, bk.views):
                p = self._params[i]
                g = p.grad
                if g is not v and (g is None or g.data_ptr() != v.data_ptr()):
                    p.grad = v
        for p in self._frozen_params:
            if p.grad is not None:
                p.grad = None

    def measure_exposed_comm(self, on: bool = True) -> None:
        """Bracket the wait for the bucket all-reduces at the end of every backward pass with two device events:
        the time between them is communication the backward kernels did NOT hide (0 when every bucket had
        finished before the last gradient kernel did)."""
        self._exposed_probe = [] if on else None

    def exposed_comm_ms(self, last: int = 0) -> Optional[float]:
        """Mean exposed communication time per backward pass over the recorded (or the ``last`` n) passes;
        synchronises the device."""
        probe = self._exposed_probe
        if not probe:
            return None
        torch.cuda.synchronize(self.device)
        sel = probe[-last:] if last else probe
        return sum(a.elapsed_time(b) for a, b in sel) / len(sel)

    def bucket_summary(self) -> List[dict]:
        wsz = torch.empty((), dtype=self.wire_dtype).element_size() if self.wire_dtype is not None else None
        return [{"bucket": bk.index, "bytes": bk.flat.numel() * bk.flat.element_size(), "params": len(bk.params),
                 "wire_bytes": bk.flat.numel() * (wsz or bk.flat.element_size())} for bk in self._buckets]

    def train(self, mode: bool = True):
        self._buf_list = None
        super().train(mode)
        return self


def _dense(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)
