"""hipGraph capture of a whole training step.

Small-image configurations (ResNet-18 on 32x32 inputs: ~200 kernels of a few microseconds each,
the workload of /root/reference/2.distributed.py with its CIFAR10 data) are bound by the host's launch
path, not by the GPU.  Everything this package does inside a step is capturable - the loss scale, the
overflow flag, the optimizer hyper-parameters and the metric accumulators live in device memory, the
gradients stay in one persistent arena, no kernel result is read back by the host - so the step is
recorded once into a hipGraph (through ``torch.cuda.CUDAGraph``, which on ROCm is hipGraph) and then
replayed with a single launch per iteration.

    step = GraphedStep(lambda: train_step(static_images, static_target), warmup=3)
    for images, target in loader:
        static_images.copy_(images, non_blocking=True); static_target.copy_(target, non_blocking=True)
        loss = step()          # replays forward + backward + optimizer update

Constraints (those of stream capture): fixed shapes, inputs in fixed buffers, single device, no
``.item()`` / printing of device values inside ``fn``.  Values returned by ``fn`` are static tensors that
are overwritten by every replay.

A step under ``DistributedDataParallel`` records as well (pass the wrapper as ``ddp=``): the bucket pack, the
all-reduce and the hand-over to the fused optimizer are stream work - the RCCL call on the communicator's own
stream joins the capture through the event the wrapper records behind the bucket's last gradient, and the
peer-memory kernels keep their sequence numbers in device memory - so every replay runs the same collectives
in the same order on every rank.  The eager warm-up has to cover the one-time bucket rebuild (two steps);
the exposed-communication probe is switched off because its timing events cannot live inside a graph.  The
Horovod-style engine cannot be recorded (its collectives are launched from a thread of its own, when the
negotiation completes).
"""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch


class GraphedStep:
    """Capture ``fn()`` after ``warmup`` eager runs; calling the object replays the graph."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 3, device: Optional[torch.device] = None,
                 enabled: bool = True, ddp=None):
        self._fn = fn
        if ddp is not None:
            if getattr(ddp, "_rebuild_pending", False):
                warmup = max(int(warmup), 2)   # the one-time bucket rebuild happens in the first backward passes
            if getattr(ddp, "_exposed_probe", None) is not None:
                ddp.measure_exposed_comm(False)
        self._graph = None
        self._out = None
        self.enabled = bool(enabled) and torch.cuda.is_available()
        if not self.enabled:
            return
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        # Eager runs on the capture stream first: lazily created state (momentum buffers, gradient
        # arena, multi-tensor tables, weight-layout tables, workspaces) must exist before recording.
        with torch.cuda.stream(self._stream):
            for _ in range(max(int(warmup), 0)):  # 0: the caller has already run the step eagerly
                fn()
        self._stream.synchronize()
        if ddp is not None and getattr(ddp, "_rebuild_pending", False):
            raise RuntimeError("GraphedStep: the DistributedDataParallel wrapper has not rebuilt its buckets yet "
                               "(the warm-up steps must run a backward pass)")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._stream):
            self._out = fn()
            # dropout masks: the host-side Philox offsets are constants of the recording; the device-side counter the
            # kernels add to them moves on here, once per replay
            from ..ops.elementwise import advance_replay_counter
            advance_replay_counter(dev)
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        self._graph = graph

    def __call__(self):
        if self._graph is None:
            return self._fn()
        self._graph.replay()
        return self._out

    @property
    def captured(self) -> bool:
        return self._graph is not None
