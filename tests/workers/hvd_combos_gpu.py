"""hvd.DistributedOptimizer combinations on one GPU (one rank): fused / stock optimizer, mixed precision, local
gradient accumulation (backward_passes_per_step), fp16 compression; each must train (loss falls) without tripping
the wrapper's once-per-step bookkeeping."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
import torch
import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import models, amp, ops
from pytorch_distributed_amd.optim import FusedSGD

hvd.init()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
x = torch.randn(32, 3, 64, 64, device=dev); t = torch.randint(0, 16, (32,), device=dev)


def run(name, fused, level, passes, compression):
    torch.manual_seed(0)
    m = models.get_model("resnet18", num_classes=16).to(dev)
    cls = FusedSGD if fused else torch.optim.SGD
    opt = cls(m.parameters(), lr=0.02, momentum=0.9)
    if level:
        m, opt = amp.initialize(m, opt, opt_level=level, verbosity=0, loss_scale=128.0)
    opt = hvd.DistributedOptimizer(opt, named_parameters=m.named_parameters(), compression=compression,
                                   backward_passes_per_step=passes)
    losses = []
    for it in range(6):
        opt.zero_grad()
        for _ in range(passes):
            loss = ops.cross_entropy(m(x), t)
            if level and fused:
                with amp.scale_loss(loss, opt) as s:
                    s.backward()
            else:
                loss.backward()
        opt.step()
        losses.append(loss.item())
    ok = all(l == l for l in losses) and losses[-1] < losses[0]
    print(f"{name:34s} {'ok  ' if ok else 'FAIL'} {losses[0]:.3f} -> {losses[-1]:.3f}", flush=True)
    return ok


good = True
good &= run("fused fp32", True, None, 1, hvd.Compression.none)
good &= run("fused fp32 fp16-compression", True, None, 1, hvd.Compression.fp16)
good &= run("fused O2", True, "O2", 1, hvd.Compression.fp16)
good &= run("fused O2 2 passes per step", True, "O2", 2, hvd.Compression.none)
good &= run("fused fp32 3 passes per step", True, None, 3, hvd.Compression.none)
good &= run("stock fp32", False, None, 1, hvd.Compression.fp16)
good &= run("stock fp32 2 passes per step", False, None, 2, hvd.Compression.none)
hvd.shutdown()
sys.exit(0 if good else 1)
