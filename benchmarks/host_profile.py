#!/usr/bin/env python
"""Where the HOST spends its time issuing one training step (cProfile over a few steps of the benchmark's
ResNet-50 configuration; the device runs asynchronously, so this is launch-path time only).

    python benchmarks/host_profile.py --batch-size 400 --steps 10
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd import amp, backends, models, ops
from pytorch_distributed_amd.optim import FusedSGD


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=400)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--ddp", action="store_true", help="single-rank DistributedDataParallel on the own RCCL communicator")
    ap.add_argument("--sort", default="tottime")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    backends.benchmark = True
    model = models.get_model(a.arch).to(dev)
    opt = FusedSGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    model, opt = amp.initialize(model, opt, opt_level="O2", verbosity=0)
    if a.ddp:
        from pytorch_distributed_amd import comm as pcomm
        from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        pcomm.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
        model = DDP(model, device_ids=[0], force_comm=True, fused_wire=True)
    x = torch.randn(a.batch_size, 3, a.image_size, a.image_size, device=dev)
    t = torch.randint(0, 1000, (a.batch_size,), device=dev)
    metrics = ops.new_metrics(dev)

    def step():
        opt.zero_grad(set_to_none=False)
        loss = ops.cross_entropy(model(x), t, metrics)
        with amp.scale_loss(loss, opt) as scaled:
            scaled.backward()
        opt.step()

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"host issue time {1e3 * host / a.steps:.2f} ms/step (under cProfile), device-complete {1e3 * total / a.steps:.2f} ms/step")
    st = pstats.Stats(pr)
    st.sort_stats(a.sort).print_stats(a.top)


if __name__ == "__main__":
    main()
