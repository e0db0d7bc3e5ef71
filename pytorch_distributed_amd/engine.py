"""Shared training / evaluation loop and CLI for the example scripts.

Every reference script is the same ImageNet-example loop wired to a different
parallel engine (/root/reference/2.distributed.py:166-285 is the canonical copy).  Here the loop
lives once: ``build_parser`` reproduces the reference's flag surface (names and defaults,
/root/reference/2.distributed.py:25-68) plus this framework's extras; ``train`` / ``validate`` keep
the reference's meter text format but accumulate loss and accuracy on the device and all-reduce
ONE 4-float vector every ``print_freq`` steps instead of barrier + three scalar all-reduces + four
host syncs per step; ``fit`` is the epoch driver (set_epoch, LR schedule, train, validate, rank-0
checkpoint, CSV / JSONL logs, optional resume).
"""
from __future__ import annotations

import argparse
import time
from typing import Callable, Optional

import torch

from . import amp as _amp
from . import models, ops, utils
from .data import (DistributedSampler, make_loader, cifar10, synthetic_imagenet, DevicePrefetcher,
                   IMAGENET_MEAN, IMAGENET_STD, imagenet, imagenet_available)


def build_parser(description: str = "PyTorch ImageNet Training", arch: str = "resnet18", epochs: int = 2,
                 batch_size: int = 3200, local_rank_flag: bool = False, dist_file_flag: bool = False):
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--data", metavar="DIR", default="./data/cifar10", help="path to dataset")
    p.add_argument("-a", "--arch", metavar="ARCH", default=arch, choices=models.model_names,
                   help="model architecture: " + " | ".join(models.model_names) + f" (default: {arch})")
    p.add_argument("-j", "--workers", default=4, type=int, metavar="N", help="data loading workers (honoured)")
    p.add_argument("--epochs", default=epochs, type=int, metavar="N")
    p.add_argument("--start-epoch", default=0, type=int, metavar="N")
    p.add_argument("-b", "--batch-size", default=batch_size, type=int, metavar="N",
                   help="total batch size of all GPUs on the current node")
    p.add_argument("--lr", "--learning-rate", default=0.1, type=float, metavar="LR", dest="lr")
    p.add_argument("--momentum", default=0.9, type=float, metavar="M")
    p.add_argument("--wd", "--weight-decay", default=1e-4, type=float, metavar="W", dest="weight_decay")
    p.add_argument("-p", "--print-freq", default=10, type=int, metavar="N")
    p.add_argument("-e", "--evaluate", dest="evaluate", action="store_true")
    p.add_argument("--pretrained", dest="pretrained", action="store_true")
    p.add_argument("--seed", default=None, type=int)
    if local_rank_flag:
        p.add_argument("--local_rank", "--local-rank", default=-1, type=int, help="node rank for distributed training")
    if dist_file_flag:
        p.add_argument("--dist-file", default=None, type=str, help="rendezvous file on a shared filesystem")
    # framework extras
    p.add_argument("--synthetic", action=argparse.BooleanOptionalAction, default=None,
                   help="synthetic data of the dataset's shape (default: automatic when the files are absent)")
    p.add_argument("--dataset", default="cifar10", choices=["cifar10", "imagenet"])
    p.add_argument("--num-classes", default=1000, type=int, help="head size (reference keeps 1000 even for CIFAR)")
    p.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16"])
    p.add_argument("--opt-level", default=None, choices=["O0", "O1", "O2", "O3"])
    p.add_argument("--bucket-mb", default=25.0, type=float)
    p.add_argument("--sync-bn", action="store_true")
    p.add_argument("--resume", default="", type=str, metavar="PATH")
    p.add_argument("--steps-per-epoch", default=0, type=int, help="cap iterations per epoch (0 = full)")
    p.add_argument("--jsonl", default="", type=str, help="per-step JSONL log (rank 0)")
    p.add_argument("--graph", action="store_true",
                   help="GPU runs (one process, or one rank per GPU under DistributedDataParallel): record the training step "
                        "into a hipGraph and replay it")
    p.add_argument("--cpu", action="store_true", help="force CPU + gloo (functional testing)")
    p.add_argument("--exact-eval", action="store_true", help="mask DistributedSampler padding in validation")
    return p


def make_datasets(args):
    """``--dataset imagenet --data DIR``: ``DIR/train`` and ``DIR/val`` image folders with the reference's
    augmentation (/root/reference/6.distributed_slurm_main.py:126-159) when they exist (or ``--no-synthetic``
    insists on them); the batches are raw uint8 and ``fit`` arranges for the normalisation to happen on the
    device.  Otherwise synthetic data of the dataset's shape."""
    if args.dataset == "imagenet":
        if args.synthetic is False or (args.synthetic is None and imagenet_available(args.data)):
            args.input_uint8 = (IMAGENET_MEAN, IMAGENET_STD)
            return imagenet(args.data, True), imagenet(args.data, False)
        n = args.steps_per_epoch * args.batch_size if args.steps_per_epoch else None
        return synthetic_imagenet(True, n), synthetic_imagenet(False, min(n or 50000, 50000))
    return cifar10(args.data, True, args.synthetic), cifar10(args.data, False, args.synthetic)


def make_model(args, device):
    model = models.get_model(args.arch, pretrained=args.pretrained, num_classes=args.num_classes) \
        if args.arch != "mnist_net" else models.Net()
    if args.sync_bn:
        from .nn import SyncBatchNorm
        SyncBatchNorm.convert_sync_batchnorm(model)
    return model.to(device)


def make_optimizer(args, model, fused: bool = True):
    from .optim import FusedSGD
    cls = FusedSGD if fused else torch.optim.SGD
    return cls(model.parameters(), args.lr, momentum=args.momentum, weight_decay=args.weight_decay)


def setup_amp(args, model, optimizer):
    level = args.opt_level or ("O0" if args.precision == "fp32" else "O1")
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float16
    return _amp.initialize(model, optimizer, opt_level=level, dtype=dtype, verbosity=0)


def _world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def train(loader, model, criterion, optimizer, epoch: int, device, args, zero_grad: Optional[Callable] = None,
          scale_loss: bool = True, jsonl: Optional[utils.JsonlLogger] = None):
    batch_time = utils.AverageMeter("Time", ":6.3f")
    data_time = utils.AverageMeter("Data", ":6.3f")
    losses = utils.AverageMeter("Loss", ":.4e")
    top1 = utils.AverageMeter("Acc@1", ":6.2f")
    top5 = utils.AverageMeter("Acc@5", ":6.2f")
    thr = utils.ThroughputMeter()
    nb = len(loader) if not args.steps_per_epoch else min(len(loader), args.steps_per_epoch)
    progress = utils.ProgressMeter(nb, [batch_time, data_time, losses, top1, top5, thr], prefix=f"Epoch: [{epoch}]",
                                   rank0_only=True)
    # --graph: GPU runs replay the whole step as one hipGraph - a single process, or one rank per GPU under this
    # package's DistributedDataParallel (its bucket pack, all-reduce and hand-over to the optimizer are stream work
    # that records with the step; every rank records after the same three eager steps).  The recorded graph, its
    # static input buffers and the metric accumulator it writes to live on the optimizer across epochs.
    graph_ddp = _graphable_ddp(model) if _world() > 1 else None
    use_graph = (getattr(args, "graph", False) and torch.device(device).type == "cuda"
                 and (_world() == 1 or graph_ddp is not None) and _takes_metrics(criterion))
    gstate = getattr(optimizer, "_pda_graph_state", None) if use_graph else None
    metrics = gstate["metrics"] if gstate is not None else utils.DeviceMetrics(device)
    metrics.reset()
    if use_graph and hasattr(optimizer, "sync_hyperparameters"):
        optimizer.sync_hyperparameters()  # the epoch's learning rate reaches the replayed kernels
    model.train()
    end = time.time()
    seen = 0

    def run_step(images, target):
        with utils.annotate("forward"):
            output = model(images)
            loss = criterion(output, target, metrics.buf) if _takes_metrics(criterion) else criterion(output, target)
            if not _takes_metrics(criterion):
                ops.loss.update_metrics_ref(output.detach().float(), target, metrics.buf)
        (zero_grad or optimizer.zero_grad)()
        with utils.annotate("backward"):
            if scale_loss and getattr(optimizer, "_amp_scaler_ref", None) is not None:
                with _amp.scale_loss(loss, optimizer) as scaled:
                    scaled.backward()
            else:
                loss.backward()
        with utils.annotate("step"):
            optimizer.step()

    for i, (images, target) in enumerate(loader):
        if i >= nb:
            break
        data_time.update(time.time() - end)
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        if use_graph and gstate is None:
            gstate = {"metrics": metrics, "x": images.clone(), "t": target.clone(), "eager": 0, "graph": None}
            optimizer._pda_graph_state = gstate
        if gstate is not None and images.shape == gstate["x"].shape and target.shape == gstate["t"].shape:
            gstate["x"].copy_(images, non_blocking=True)
            gstate["t"].copy_(target, non_blocking=True)
            if gstate["graph"] is None and gstate["eager"] >= 3:
                # three ordinary steps have created every lazily built piece of state: record now
                gx, gt = gstate["x"], gstate["t"]
                gstate["graph"] = utils.GraphedStep(lambda: run_step(gx, gt), warmup=0, device=device, ddp=graph_ddp)
            if gstate["graph"] is not None:
                gstate["graph"]()
            else:
                run_step(gstate["x"], gstate["t"])
                gstate["eager"] += 1
        else:
            run_step(images, target)
        seen += images.size(0)
        if i % args.print_freq == 0 or i == nb - 1:
            # ONE all-reduce of {loss_sum, correct1, correct5, n}; the only host sync of the window
            l, a1, a5, n = metrics.snapshot()
            dt = time.time() - end
            losses.update(l, n)
            top1.update(a1, n)          # fractions in [0,1], as the reference's accuracy() reports them
            top5.update(a5, n)
            steps = 1 if i == 0 else (args.print_freq if i % args.print_freq == 0 else (nb - 1) % args.print_freq)
            batch_time.update(dt / max(steps, 1), max(steps, 1))
            thr.update(seen * _world(), dt)
            seen = 0
            end = time.time()
            progress.display(i)
            if jsonl is not None and _rank() == 0:
                jsonl.log(epoch=epoch, step=i, loss=l, acc1=a1, acc5=a5, images_per_sec=thr.val)
    return losses.avg, top1.avg


def _graphable_ddp(model):
    """The wrapper whose step can be recorded on several ranks: this package's DistributedDataParallel over a model
    without SyncBatchNorm (its statistics all-reduce goes through the process group, outside the recorded streams)."""
    from .nn import SyncBatchNorm
    from .parallel import DistributedDataParallel
    if not isinstance(model, DistributedDataParallel):
        return None
    if any(isinstance(m, SyncBatchNorm) for m in model.modules()):
        return None
    return model


def _takes_metrics(criterion) -> bool:
    from .nn import CrossEntropyLoss
    return isinstance(criterion, CrossEntropyLoss)


def validate(loader, model, criterion, device, args, sampler: Optional[DistributedSampler] = None):
    """Distributed evaluation: every rank scores its shard, loss/accuracy are summed on the device and
    all-reduced once at the end (the per-epoch pattern of /root/reference/5.2.horovod_pytorch_mnist.py:81-107;
    the per-batch barrier + 3 all-reduces of /root/reference/2.distributed.py:265-273 is not needed)."""
    batch_time = utils.AverageMeter("Time", ":6.3f")
    losses = utils.AverageMeter("Loss", ":.4e")
    top1 = utils.AverageMeter("Acc@1", ":6.2f")
    top5 = utils.AverageMeter("Acc@5", ":6.2f")
    nb = len(loader) if not args.steps_per_epoch else min(len(loader), args.steps_per_epoch)
    progress = utils.ProgressMeter(nb, [batch_time, losses, top1, top5], prefix="Test: ", rank0_only=True)
    metrics = utils.DeviceMetrics(device)
    model.eval()
    pad = sampler.padding_mask() if (sampler is not None and args.exact_eval) else None
    pos = 0
    with torch.no_grad():
        end = time.time()
        for i, (images, target) in enumerate(loader):
            if i >= nb:
                break
            images = images.to(device, non_blocking=True)
            target = target.to(device, non_blocking=True)
            if pad is not None:  # drop wrap-around duplicates so every sample counts once
                keep = torch.tensor([not p for p in pad[pos:pos + target.numel()]], device=device)
                pos += target.numel()
                if not bool(keep.all()):
                    images, target = images[keep], target[keep]
                    if target.numel() == 0:
                        continue
            output = model(images)
            if _takes_metrics(criterion):
                criterion(output, target, metrics.buf)
            else:
                ops.loss.update_metrics_ref(output.float(), target, metrics.buf)
            batch_time.update(time.time() - end)
            end = time.time()
            if i % args.print_freq == 0:
                l, a1, a5, n = metrics.snapshot(reset=False)   # running totals so far, all ranks
                losses.val = losses.avg = l
                top1.val = top1.avg = a1
                top5.val = top5.avg = a5
                progress.display(i)
    losses.reset(), top1.reset(), top5.reset()
    l, a1, a5, n = metrics.snapshot()
    losses.update(l, n)
    top1.update(a1, n)
    top5.update(a5, n)
    if _rank() == 0:
        print(" * Acc@1 {top1.avg:.3f} Acc@5 {top5.avg:.3f}".format(top1=top1, top5=top5), flush=True)
    return top1.avg


def fit(args, model, optimizer, device, train_set, val_set, per_rank_batch: int, distributed: bool,
        zero_grad: Optional[Callable] = None, csv_path: Optional[str] = None, save_prefix_arch: bool = True,
        is_chief: Optional[bool] = None, samplers: Optional[tuple] = None):
    from .nn import CrossEntropyLoss
    from . import backends
    criterion = CrossEntropyLoss()
    # the reference's `cudnn.benchmark = True` (/root/reference/2.distributed.py:122): per-shape selection of the
    # conv tile configuration by timing, during the first iteration
    backends.benchmark = device.type == "cuda"
    if samplers is not None:
        train_sampler, val_sampler = samplers
    elif distributed:
        train_sampler, val_sampler = DistributedSampler(train_set), DistributedSampler(val_set)
    else:
        train_sampler = val_sampler = None
    pin = device.type == "cuda"
    train_loader = make_loader(train_set, per_rank_batch, train_sampler, args.workers, pin, shuffle=True)
    val_loader = make_loader(val_set, per_rank_batch, val_sampler, args.workers, pin)
    # raw uint8 batches (image folders): normalise on the device - inside the stem kernel when the model
    # has one that does it, else on the prefetcher's side stream
    norm = getattr(args, "input_uint8", None)
    prefetch_norm = None
    if norm is not None:
        inner = model.module if hasattr(model, "module") else model
        if hasattr(inner, "set_input_normalization"):
            inner.set_input_normalization(norm[0], norm[1], 1.0 / 255.0)
        else:
            prefetch_norm = norm
            val_loader = DevicePrefetcher(val_loader, device, normalize=norm, channels_scale=255.0)
    chief = (_rank() == 0) if is_chief is None else is_chief
    best_acc1 = 0.0
    if args.resume:
        ck = utils.load_checkpoint(args.resume, model, optimizer, map_location="cpu")
        args.start_epoch, best_acc1 = ck.get("epoch", 0), ck.get("best_acc1", 0.0)
        if chief:
            print(f"=> resumed from '{args.resume}' (epoch {args.start_epoch})")
    if args.evaluate:
        return validate(val_loader, model, criterion, device, args, val_sampler)
    csv = utils.EpochCSV(csv_path) if (csv_path and chief) else None
    jsonl = utils.JsonlLogger(args.jsonl) if args.jsonl else None
    # failure detection for the whole run: a rank that dies (or a link that goes away under a collective) ends
    # the job with a non-zero exit code instead of leaving the others inside their next all-reduce, which is what
    # happens to the reference's loop (/root/reference/2.distributed.py:219-223)
    from .comm.watchdog import start_watchdog, stop_watchdog
    if distributed:
        start_watchdog()
    try:
        best_acc1 = _fit_epochs(args, model, optimizer, device, criterion, train_loader, val_loader, train_sampler,
                                val_sampler, prefetch_norm, pin, zero_grad, csv, jsonl, chief, best_acc1,
                                save_prefix_arch)
    finally:
        stop_watchdog()
    return best_acc1


def _fit_epochs(args, model, optimizer, device, criterion, train_loader, val_loader, train_sampler, val_sampler,
                prefetch_norm, pin, zero_grad, csv, jsonl, chief, best_acc1, save_prefix_arch):
    for epoch in range(args.start_epoch, args.epochs):
        t0 = time.time()
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
            val_sampler.set_epoch(epoch)
        utils.adjust_learning_rate(optimizer, epoch, args.lr)
        tl = DevicePrefetcher(train_loader, device, normalize=prefetch_norm, channels_scale=255.0) \
            if (pin or prefetch_norm is not None) else train_loader
        train(tl, model, criterion, optimizer, epoch, device, args, zero_grad, jsonl=jsonl)
        acc1 = validate(val_loader, model, criterion, device, args, val_sampler)
        is_best = acc1 > best_acc1
        best_acc1 = max(acc1, best_acc1)
        if csv is not None:
            csv.write(t0, time.time() - t0)
        if chief:
            state = utils.build_state(model, epoch, args.arch, best_acc1, optimizer, _amp.state_dict())
            utils.save_checkpoint(state, is_best, prefix_arch=save_prefix_arch)
    return best_acc1
