"""2 ranks on GPU 0: where does the fp16 SyncBatchNorm model leave the full-batch fp16 reference?
Per top-level stage: forward output (this rank's slice) and the gradient w.r.t. it.  argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd import amp, comm, models, ops
from pytorch_distributed_amd.nn import SyncBatchNorm
from pytorch_distributed_amd.optim import FusedSGD

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
level = os.environ.get("DBG_LEVEL", "O2")
per, hw = int(os.environ.get("DBG_PER", "16")), int(os.environ.get("DBG_HW", "64"))
torch.cuda.set_device(0)
comm.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
dev = torch.device("cuda", 0)


import faulthandler
faulthandler.dump_traceback_later(60, exit=True)


def build(sync, dtype):
    torch.manual_seed(3)
    m = models.resnet18(num_classes=10).to(dev)
    if sync:
        SyncBatchNorm.convert_sync_batchnorm(m)
        m = m.to(dev)
    m.set_compute_dtype(dtype)
    return m, None


def run(m, x, t, tag):
    acts, grads = {}, {}
    hs = []
    for name, mod in m.named_children():
        def hook(_m, _i, out, name=name):
            o = out[0] if isinstance(out, tuple) else out
            acts[name] = o.detach().float().clone()
            if o.requires_grad:
                o.register_hook(lambda g, name=name: grads.__setitem__(name, g.detach().float().clone()))
        hs.append(mod.register_forward_hook(hook))
    for p in m.parameters():
        p.grad = None
    loss = ops.cross_entropy(m(x), t)
    loss.backward()
    for h in hs:
        h.remove()
    pg = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
    return acts, grads, pg


torch.manual_seed(11)
x = torch.randn(world * per, 3, hw, hw, device=dev)
t = torch.randint(0, 10, (world * per,), device=dev)
sl = slice(rank * per, (rank + 1) * per)
import torch.distributed as dist


def rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def grads_of(sync, dtype):
    m, _ = build(sync, dtype)
    _, _, pg = run(m, x[sl] if sync else x, t[sl] if sync else t, "")
    if sync:
        for k in pg:
            g = pg[k].cpu()
            dist.all_reduce(g)
            pg[k] = (g / world).to(dev)
    torch.cuda.synchronize()
    print(f"rank {rank}: sync={sync} {dtype} done", flush=True)
    return pg


ref32 = grads_of(False, torch.float32)
ref16 = grads_of(False, torch.float16)
net16 = grads_of(True, torch.float16)
net32 = grads_of(True, torch.float32)
if rank == 0:
    keys = list(ref32)[::-1][:16]
    for label, a, b in (("ref16 vs ref32", ref16, ref32), ("net16 vs ref32", net16, ref32),
                        ("net16 vs ref16", net16, ref16), ("net32 vs ref32", net32, ref32)):
        print(label, [(round(rel(a[k], b[k]), 4), k) for k in keys], flush=True)
comm.destroy_process_group()
