"""Second compute stream for the weight-gradient kernels of the backward pass.

In ``loss.backward()`` (/root/reference/2.distributed.py:231) the input-gradient chain
(BatchNorm backward -> dgrad -> the previous layer's BatchNorm backward -> ...) is the critical path; a
layer's weight gradient depends only on that layer's ``dy`` and saved input and nothing consumes it before
the gradient all-reduce / the optimizer.  The wgrad launches therefore go to a side stream, where the
hardware runs them next to the memory-bound BatchNorm passes and the tails of the dgrad launches of
the layers below instead of in line with them.

Protocol (all on one device, same stream priority as the compute stream: a wait between queues of
different priorities can starve the queue that is waited for):

* ``fork()``: the side stream waits for everything the calling stream has been given so far
  (the ``dy`` just produced); returns the side stream, or ``None`` when the feature is off or the tensors
  are not on a GPU.
* the caller launches on the side stream, calls ``keep_alive(...)`` for every tensor read there and
  ``mark()`` afterwards.
* ``join()`` makes the calling stream wait for the side work launched so far.  It runs from an autograd
  final callback queued by ``mark()`` (so ``backward()`` returns with the stream joined), and
  from gradient consumers that run inside backward (the DDP bucket launch).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
from torch.autograd import Variable

_ENABLED = os.environ.get("PDA_WGRAD_STREAM", "1") == "1"
# Under hipGraph capture the fork / join events become graph edges, so the recorded step keeps the two branches.
_IN_GRAPH = os.environ.get("PDA_WGRAD_STREAM_IN_GRAPH", "1") == "1"
_SIDE: Dict[int, "torch.cuda.Stream"] = {}
_DIRTY: Dict[int, bool] = {}
_ORIGIN: Dict[int, "torch.cuda.Stream"] = {}   # the stream the latest fork came from (the backward pass' stream)
launches = 0   # wgrad launches that went to the side stream (tests / diagnostics)


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> bool:
    """Switch the side stream on / off; returns the previous setting."""
    global _ENABLED
    old, _ENABLED = _ENABLED, bool(flag)
    return old


def fork(device: torch.device) -> Optional["torch.cuda.Stream"]:
    if not _ENABLED or device.type != "cuda":
        return None
    if not _IN_GRAPH and torch.cuda.is_current_stream_capturing():
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _SIDE.get(idx)
    if side is None:
        side = torch.cuda.Stream(device=idx)    # default priority, as the compute stream
        _SIDE[idx] = side
    cur = torch.cuda.current_stream(idx)
    if cur == side:
        return None
    side.wait_stream(cur)
    _ORIGIN[idx] = cur
    return side


def keep_alive(side: "torch.cuda.Stream", *tensors) -> None:
    """The caching allocator must not hand a tensor's memory to the compute stream while the side stream
    still reads it."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(side)


def mark(device: torch.device) -> None:
    global launches
    idx = device.index if device.index is not None else torch.cuda.current_device()
    launches += 1
    if not _DIRTY.get(idx, False):
        _DIRTY[idx] = True
        # one callback per fork-join episode (a flag that outlives a backward pass which raised would
        # silence every later one)
        try:
            Variable._execution_engine.queue_callback(join)
        except RuntimeError:
            # not inside a backward pass (an op's backward called by hand): join right away
            join()


def pending(device: torch.device) -> Optional["torch.cuda.Stream"]:
    """The side stream of ``device`` if work launched there has not been joined yet, else ``None``.  A consumer
    that can run on the side stream itself (the gradient-bucket pack + all-reduce launch) orders itself after the
    weight gradients that way without making the backward pass' stream wait for them."""
    if device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _SIDE[idx] if _DIRTY.get(idx, False) else None


def join() -> None:
    """The stream that forked (on every device with outstanding side work) waits for that work.  (Not "the
    current stream": the final callback of a backward pass runs after autograd has restored the thread's
    stream, which need not be the one the pass ran on.)"""
    for idx, dirty in list(_DIRTY.items()):
        if dirty:
            _ORIGIN[idx].wait_stream(_SIDE[idx])
            _DIRTY[idx] = False
