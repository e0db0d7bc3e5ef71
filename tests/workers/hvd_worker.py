import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import models
hvd.init()
r, w = hvd.rank(), hvd.size()
t = torch.full((5,), float(r + 1))
out = hvd.allreduce(t, name="x")
assert torch.allclose(out, torch.full((5,), sum(range(1, w + 1)) / w)), out
assert torch.allclose(t, torch.full((5,), float(r + 1)))          # out-of-place
hvd.allreduce_(t, op=hvd.Sum, name="y")
assert torch.allclose(t, torch.full((5,), float(sum(range(1, w + 1)))))
# different enqueue order on different ranks + fusion
names = ["a", "b", "c", "d"]
order = names if r == 0 else names[::-1]
ts = {n: torch.full((1000,), float(ord(n)) + r) for n in names}
hs = {n: hvd.allreduce_async_(ts[n], name=n, compression=hvd.Compression.fp16) for n in order}
for n in names:
    hvd.synchronize(hs[n])
    exp = float(ord(n)) + (w - 1) / 2
    assert torch.allclose(ts[n], torch.full((1000,), exp), atol=0.1), (n, ts[n][0], exp)
b = torch.arange(4.0) if r == 0 else torch.zeros(4)
hvd.broadcast_(b, 0, name="b0"); assert torch.equal(b, torch.arange(4.0))
g = hvd.allgather(torch.tensor([[float(r)]])); assert g.flatten().tolist() == [float(i) for i in range(w)]
torch.manual_seed(r)
net = models.Net()
hvd.broadcast_parameters(net.state_dict(), root_rank=0)
chk = hvd.allreduce(net.fc1.weight.detach().clone(), name="chk")
assert torch.allclose(chk, net.fc1.weight), "params differ after broadcast"
opt = torch.optim.SGD(net.parameters(), lr=0.01 * w, momentum=0.5)
hvd.broadcast_optimizer_state(opt, root_rank=0)
opt = hvd.DistributedOptimizer(opt, named_parameters=net.named_parameters(), compression=hvd.Compression.fp16, op=hvd.Average, gradient_predivide_factor=1.0)
assert isinstance(opt, torch.optim.SGD)
torch.manual_seed(100 + r)
x = torch.randn(16, 1, 28, 28); y = torch.randint(0, 10, (16,))
net.train()
for i in range(3):
    opt.zero_grad()
    loss = torch.nn.functional.nll_loss(net(x), y); loss.backward(); opt.step()
chk = hvd.allreduce(net.fc2.weight.detach().clone(), name="chk2")
assert torch.allclose(chk, net.fc2.weight, atol=1e-6), "replicas diverged"
# a host tensor larger than one piece / slice of either control plane, and 16-bit host tensors
gens = [torch.Generator().manual_seed(50 + i) for i in range(w)]
big = [torch.randn(3_000_001, generator=g) for g in gens]
got = hvd.allreduce(big[r].clone(), op=hvd.Sum, name="big")
assert torch.allclose(got, sum(big), atol=1e-5), (got - sum(big)).abs().max()
for dt in (torch.float16, torch.bfloat16):
    vals = [(b[:4099] * 3).to(dt) for b in big]
    got = hvd.allreduce(vals[r].clone(), op=hvd.Sum, name=f"half.{dt}")
    exp = sum(v.float() for v in vals)
    assert got.dtype == dt and torch.allclose(got.float(), exp, rtol=1e-2, atol=1e-2), (dt, (got.float() - exp).abs().max())
    mx = hvd.allreduce(vals[r].clone(), op=hvd.Max, name=f"halfmax.{dt}")
    assert torch.equal(mx, torch.stack(vals).max(0).values)
ints = hvd.allreduce(torch.arange(10, dtype=torch.int64) * (r + 1), op=hvd.Sum, name="ints")
assert torch.equal(ints, torch.arange(10, dtype=torch.int64) * sum(range(1, w + 1)))
a = torch.tensor([1.0, 0.0]) if r == 0 else torch.tensor([0.0, 1.0])
if w == 2:
    ad = hvd.allreduce(a, op=hvd.Adasum, name="ad"); assert torch.allclose(ad, torch.tensor([1.0, 1.0])), ad
print("rank", r, "hvd OK", hvd.engine_stats(), flush=True)
hvd.shutdown()
