"""amp.scale_loss + hvd.DistributedOptimizer on the CPU engine (shared-memory backend), launched with
`python -m pytorch_distributed_amd.launch -np 2`.  Rank 1 alone produces an infinite gradient in step 1: the
overflow pass at the exit of scale_loss must look at REDUCED gradients, so every rank skips that step and the
replicas stay identical.  argv: fused|stock"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import amp, models
from pytorch_distributed_amd.optim import FusedSGD
kind = sys.argv[1]
hvd.init()
r, w = hvd.rank(), hvd.size()
torch.manual_seed(r)
net = models.Net().eval()
hvd.broadcast_parameters(net.state_dict(), root_rank=0)
opt = (FusedSGD if kind == "fused" else torch.optim.SGD)(net.parameters(), lr=0.05, momentum=0.5)
net, opt = amp.initialize(net, opt, opt_level="O0", loss_scale="dynamic", verbosity=0)
hvd.broadcast_optimizer_state(opt, root_rank=0)
opt = hvd.DistributedOptimizer(opt, named_parameters=net.named_parameters())
scaler = opt._amp_scaler_ref
torch.manual_seed(100 + r)
x = torch.randn(16, 1, 28, 28); y = torch.randint(0, 10, (16,))
w0 = net.fc2.weight.detach().clone()
for step in range(3):
    opt.zero_grad()
    loss = torch.nn.functional.nll_loss(net(x), y)
    with amp.scale_loss(loss, opt) as scaled:
        (scaled * float("inf") if (step == 1 and r == 1) else scaled).backward()
    if step == 1:
        before = net.fc2.weight.detach().clone()
    opt.step()
    if step == 1:
        assert torch.equal(before, net.fc2.weight), f"rank {r} stepped on an overflowed step"
assert scaler.skipped_steps() == 1, (r, scaler.skipped_steps())
assert scaler.loss_scale() == 2.0 ** 15, scaler.loss_scale()
assert torch.isfinite(net.fc2.weight).all() and not torch.equal(w0, net.fc2.weight)
chk = hvd.allreduce(net.fc2.weight.detach().clone(), name="chk")
assert torch.allclose(chk, net.fc2.weight, atol=1e-7), "replicas diverged"
# a second wrapper over the same parameters takes the announcements over from the first
opt2 = hvd.DistributedOptimizer(torch.optim.SGD(net.parameters(), lr=0.01), named_parameters=net.named_parameters())
opt2.zero_grad()
torch.nn.functional.nll_loss(net(x), y).backward()
opt2.step()
assert not opt._handles, "the discarded wrapper still received gradient announcements"
print("rank", r, "hvd amp OK", kind, flush=True)
hvd.shutdown()
