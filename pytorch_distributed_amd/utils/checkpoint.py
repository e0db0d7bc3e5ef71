"""Checkpoints in the reference's on-disk format, plus the resume path it lacks.

``save_checkpoint(state, is_best, filename)`` writes ``<arch>.checkpoint.pth.tar`` with
``torch.save`` and copies it to ``<arch>.model_best.pth.tar`` when ``is_best``
(/root/reference/2.distributed.py:288-293; un-prefixed names as in
/root/reference/5.horovod_distributed.py:275-278 with ``prefix_arch=False``).  The
payload keeps the reference's keys ``epoch, arch, state_dict, best_acc1`` and may carry
``optimizer``, ``amp`` and ``sampler_epoch`` as a superset.  ``state_dict`` values are
plain torchvision-named tensors, so the files load into stock torchvision ResNets
(and DDP-wrapped models round-trip the ``module.`` prefix).
"""
from __future__ import annotations

import os
import shutil
from typing import Optional

import torch


def checkpoint_paths(arch: Optional[str], filename: str = "checkpoint.pth.tar", directory: str = ".",
                     prefix_arch: bool = True):
    base = f"{arch}.{filename}" if (prefix_arch and arch) else filename
    best = f"{arch}.model_best.pth.tar" if (prefix_arch and arch) else "model_best.pth.tar"
    return os.path.join(directory, base), os.path.join(directory, best)


def save_checkpoint(state: dict, is_best: bool, filename: str = "checkpoint.pth.tar", directory: str = ".",
                    prefix_arch: bool = True) -> str:
    path, best = checkpoint_paths(state.get("arch"), filename, directory, prefix_arch)
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)  # never leave a torn file behind if the job dies mid-write
    if is_best:
        shutil.copyfile(path, best)
    return path


def strip_module_prefix(sd: dict) -> dict:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def add_module_prefix(sd: dict) -> dict:
    return {(k if k.startswith("module.") else "module." + k): v for k, v in sd.items()}


def build_state(model, epoch: int, arch: str, best_acc1: float, optimizer=None, amp_state: Optional[dict] = None,
                extra: Optional[dict] = None) -> dict:
    inner = model.module if hasattr(model, "module") else model
    state = {"epoch": epoch + 1, "arch": arch, "state_dict": inner.state_dict(), "best_acc1": best_acc1}
    if optimizer is not None:
        state["optimizer"] = optimizer.state_dict()
    if amp_state is not None:
        state["amp"] = amp_state
    if extra:
        state.update(extra)
    return state


def load_checkpoint(path: str, model=None, optimizer=None, map_location="cpu", strict: bool = True,
                    trusted: bool = False) -> dict:
    """Load a checkpoint written by ``save_checkpoint`` (or by the reference scripts) and
    restore model / optimizer / amp state.  Returns the payload (``epoch`` = next epoch).

    The file is read with the tensors-and-plain-containers loader (``weights_only=True``): the payload
    is tensors, numbers, strings, dicts and lists, so nothing in a checkpoint needs code execution and a
    file that carries any other pickled object is refused.  ``trusted=True`` (a file this job wrote
    itself, carrying custom objects in ``extra``) opts into the full unpickler."""
    ckpt = torch.load(path, map_location=map_location, weights_only=not trusted)
    if model is not None:
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        wrapped = hasattr(model, "module")
        has_prefix = any(k.startswith("module.") for k in sd)
        if wrapped and not has_prefix:
            sd = add_module_prefix(sd)
        elif not wrapped and has_prefix:
            sd = strip_module_prefix(sd)
        model.load_state_dict(sd, strict=strict)
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    if "amp" in ckpt:
        from .. import amp
        amp.load_state_dict(ckpt["amp"])
    return ckpt
