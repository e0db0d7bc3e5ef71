"""DistributedDataParallel combinations on one GPU (one rank, the package's own RCCL communicator): fused / stock
optimizer, O0 / O1 / O2, delayed all-reduce (apex flavour), local accumulation under no_sync(), SyncBatchNorm;
each must train (loss falls) and leave every parameter with a finite value."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29563")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
from pytorch_distributed_amd import models, amp, ops, comm
from pytorch_distributed_amd.nn import SyncBatchNorm
from pytorch_distributed_amd.optim import FusedSGD
from pytorch_distributed_amd.parallel import DistributedDataParallel, ApexDistributedDataParallel

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
comm.init_process_group("cpu:gloo,cuda:nccl", rank=0, world_size=1)
x = torch.randn(32, 3, 64, 64, device=dev); t = torch.randint(0, 16, (32,), device=dev)


def run(name, fused, level, wrap, accumulate=1, sync_bn=False):
    torch.manual_seed(0)
    m = models.get_model("resnet18", num_classes=16)
    if sync_bn:
        SyncBatchNorm.convert_sync_batchnorm(m)
    m = m.to(dev)
    opt = (FusedSGD if fused else torch.optim.SGD)(m.parameters(), lr=0.02, momentum=0.9)
    if level:
        m, opt = amp.initialize(m, opt, opt_level=level, verbosity=0, loss_scale=128.0)
    m = wrap(m)
    losses = []
    for it in range(6):
        opt.zero_grad()
        for k in range(accumulate):
            ctx = m.no_sync() if k + 1 < accumulate else torch.enable_grad()
            with ctx:
                loss = ops.cross_entropy(m(x), t)
                if level:
                    with amp.scale_loss(loss, opt) as s:
                        s.backward()
                else:
                    loss.backward()
        opt.step()
        losses.append(loss.item())
    finite = all(bool(torch.isfinite(p).all()) for p in m.parameters())
    ok = finite and all(l == l for l in losses) and losses[-1] < losses[0]
    print(f"{name:40s} {'ok  ' if ok else 'FAIL'} {losses[0]:.3f} -> {losses[-1]:.3f}", flush=True)
    return ok


ddp = lambda m: DistributedDataParallel(m, device_ids=[0])
good = True
good &= run("ddp fused fp32", True, None, ddp)
good &= run("ddp fused O2", True, "O2", ddp)
good &= run("ddp fused O1", True, "O1", ddp)
good &= run("ddp stock fp32", False, None, ddp)
good &= run("ddp stock O1", False, "O1", ddp)
good &= run("ddp fused O2, 2 micro-batches no_sync", True, "O2", ddp, accumulate=2)
good &= run("ddp stock fp32, 3 micro-batches no_sync", False, None, ddp, accumulate=3)
good &= run("apex-style ddp delay_allreduce O2", True, "O2", lambda m: ApexDistributedDataParallel(m, delay_allreduce=True))
good &= run("apex-style ddp message_size O1", False, "O1", lambda m: ApexDistributedDataParallel(m, message_size=1_000_000))
good &= run("ddp fused O2 SyncBatchNorm", True, "O2", ddp, sync_bn=True)
comm.destroy_process_group()
sys.exit(0 if good else 1)
