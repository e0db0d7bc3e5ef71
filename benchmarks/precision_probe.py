"""Gradient-direction agreement of low-precision runs with the fp32 oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pytorch_distributed_amd import models, ops

def cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-20)).item()

def grads(model, x, t, ls, native, autocast=None):
    model.zero_grad()
    if native:
        loss = ops.cross_entropy(model(x), t)
    else:
        with torch.autocast("cuda", dtype=autocast, enabled=autocast is not None):
            loss = F.cross_entropy(model(x), t)
    (loss * ls).backward()
    return {k: p.grad.detach().clone() / ls for k, p in model.named_parameters()}

dev = "cuda"
arch, hw, batch = sys.argv[1] if len(sys.argv) > 1 else "resnet50", 64, 32
torch.manual_seed(12)
m = models.get_model(arch).to(dev)
ref = models.torch_resnet(arch).to(dev)
ref.load_state_dict(m.state_dict())
x = torch.randn(batch, 3, hw, hw, device=dev); t = torch.randint(0, 1000, (batch,), device=dev)
oracle = grads(ref, x, t, 1.0, False)
def report(name, g):
    allg = torch.cat([g[k].float().flatten() for k in oracle]); allo = torch.cat([oracle[k].float().flatten() for k in oracle])
    cs = sorted(cos(g[k], oracle[k]) for k in oracle)
    print(f"{name:34s} total={cos(allg, allo):.4f} median={cs[len(cs)//2]:.4f} min={cs[0]:.4f}", flush=True)
report("native fp32", grads(m.set_compute_dtype(torch.float32), x, t, 1.0, True))
for ls in (1.0, 128.0, 1024.0, 65536.0):
    report(f"native fp16 ls={ls:g}", grads(m.set_compute_dtype(torch.float16), x, t, ls, True))
    report(f"stock autocast fp16 ls={ls:g}", grads(ref, x, t, ls, False, torch.float16))
report("native bf16", grads(m.set_compute_dtype(torch.bfloat16), x, t, 1.0, True))
report("stock autocast bf16", grads(ref, x, t, 1.0, False, torch.bfloat16))
