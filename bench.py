#!/usr/bin/env python
"""Headline benchmark: ResNet-50 training throughput (images/sec, whole job) on
synthetic ImageNet-shaped data, AMP fp16 with dynamic loss scaling, bucketed-all-reduce
DDP over RCCL when launched on more than one GPU.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 30 --warmup 10

One timed step = zero_grad + forward + loss + scaled backward (+ gradient all-reduce
overlapped with backward) + overflow check + fused momentum-SGD update of all
25.6 M parameters.  Weak scaling: the per-GPU batch is fixed.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--batch-size", type=int, default=400,
                    help="per-GPU batch (default 400 = the reference's global -b 3200 over 8 GPUs, "
                         "/root/reference/2.distributed.py:41-48,113)")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--num-classes", type=int, default=1000)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--opt-level", default="O2", choices=["O0", "O1", "O2", "O3"])
    ap.add_argument("--engine", default="native", choices=["native", "stock"],
                    help="native = this framework's kernels/DDP/AMP; stock = plain PyTorch-ROCm comparator")
    ap.add_argument("--bucket-mb", type=float, default=25.0)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--channels-last", action="store_true", help="stock engine only")
    ap.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    ap.add_argument("--graph", action="store_true",
                    help="record the step into a hipGraph after warm-up and replay it (single GPU, native engine)")
    ap.add_argument("--native-comm", action="store_true",
                    help="gradient all-reduce through this package's own RCCL communicator instead of torch.distributed")
    ap.add_argument("--autotune", action=argparse.BooleanOptionalAction, default=True,
                    help="backends.benchmark: time the candidate conv tile configurations per layer shape during "
                         "warm-up (the reference sets cudnn.benchmark = True, /root/reference/2.distributed.py:122)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="wrap the model in DistributedDataParallel even for one rank (exercises the bucketed "
                         "all-reduce path on a single GPU)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(max(args.local_rank, 0))))
    # PDA_BENCH_DEVICE=cpu: control-flow rehearsal of the multi-rank path (gloo, CPU reference ops) for the
    # test-suite; numbers from it are not benchmark results and the JSON says so in "device".
    on_cpu = os.environ.get("PDA_BENCH_DEVICE", "") == "cpu"
    if not on_cpu and not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; bench.py needs an MI355X"}))
        return 2
    if on_cpu:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if on_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pytorch_distributed_amd import models, ops, amp, backends
    backends.benchmark = bool(args.autotune) and args.warmup > 0   # the tuning runs belong to the warm-up steps
    from pytorch_distributed_amd.optim import FusedSGD

    torch.manual_seed(1234 + rank)
    B, S = args.batch_size, args.image_size
    images = torch.randn(B, 3, S, S, device=dev)
    target = torch.randint(0, args.num_classes, (B,), device=dev)
    cdtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.precision]

    if args.engine == "native":
        model = models.get_model(args.arch, num_classes=args.num_classes).to(dev)
        if args.sync_bn:
            from pytorch_distributed_amd.nn import SyncBatchNorm
            SyncBatchNorm.convert_sync_batchnorm(model)
        opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        level = "O0" if args.precision == "fp32" else args.opt_level
        model, opt = amp.initialize(model, opt, opt_level=level, dtype=cdtype if level != "O0" else torch.float16,
                                    verbosity=0)
        if ddp:
            from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
            model = DDP(model, device_ids=None if on_cpu else [local_rank], bucket_cap_mb=args.bucket_mb,
                        force_comm=args.force_ddp,
                        comm="native" if args.native_comm else None)
        metrics = ops.new_metrics(dev)

        def step():
            model.zero_grad() if ddp else opt.zero_grad(set_to_none=False)
            loss = ops.cross_entropy(model(images), target, metrics)
            with amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
            opt.step()
            return loss
    else:
        model = models.torch_resnet(args.arch, args.num_classes).to(dev)
        x = images
        if args.channels_last:
            model = model.to(memory_format=torch.channels_last)
            x = images.contiguous(memory_format=torch.channels_last)
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        scaler = torch.amp.GradScaler("cuda", enabled=(cdtype == torch.float16))
        if ddp:
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=None if on_cpu else [local_rank])
        crit = torch.nn.CrossEntropyLoss()

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=cdtype, enabled=(cdtype != torch.float32)):
                loss = crit(model(x), target)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            return loss

    def fence():
        if world > 1:
            dist.barrier()
        if not on_cpu:
            torch.cuda.synchronize()

    graphed = False
    if args.graph and args.engine == "native" and not ddp:
        from pytorch_distributed_amd.utils.graph import GraphedStep
        eager_step = step
        step = GraphedStep(eager_step, warmup=min(3, max(args.warmup, 1)))
        graphed = step.captured

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss.item())
    if rank == 0:
        ips = world * B * args.steps / dt
        out = {
            "metric": ("images/sec (whole node) ResNet-50 fp16 training, synthetic ImageNet 224x224"
                       if (args.arch, args.precision, S) == ("resnet50", "fp16", 224) else
                       f"images/sec (whole node) {args.arch} {args.precision} training, synthetic {S}x{S}"),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"model": args.arch, "global_batch": world * B, "per_gpu_batch": B, "image_size": S,
                       "seq_len": None, "parallelism": f"dp{world}", "engine": args.engine,
                       "amp_opt_level": args.opt_level if args.precision != "fp32" else "O0",
                       "optimizer": "momentum-SGD (fused)" if args.engine == "native" else "torch.optim.SGD",
                       "loss_scaling": "dynamic" if args.precision == "fp16" else "none",
                       "device": "cpu (rehearsal, not a result)" if on_cpu else torch.cuda.get_device_name(dev),
                       "weights": "random-init", "conv_autotune": backends.benchmark if args.engine == "native" else None, "final_loss": round(final_loss, 4), "hip_graph": graphed,
                       "baseline_note": "reference publishes no number; vs_baseline is null"},
        }
        print(json.dumps(out), flush=True)
    if ddp:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    try:
        rc = main()
    except Exception as e:  # noqa: BLE001 - a failed run must still say why, in the one place the driver reads
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"error": f"{type(e).__name__}: {e}"[:500]}), flush=True)
        rc = 1
    sys.exit(rc)
