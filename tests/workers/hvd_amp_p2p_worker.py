"""amp.scale_loss + hvd.DistributedOptimizer(FusedSGD) on 2+ ranks sharing GPU 0 (engine collectives through the
peer-memory backend), the loop of /root/reference/5.horovod_distributed.py:219-221 with the mixed-precision
context of /root/reference/4.apex_distributed2.py:289-291 around backward.

The engine thread is slowed down (PDA_HVD_TEST_EXEC_DELAY_MS) so that every gradient reduction is still in flight
when backward returns: the overflow pass at the exit of ``scale_loss`` has to wait for the reductions first, or it
looks at local, un-reduced gradients (ranks then take different skip decisions) next to kernels that are still
replacing them.

modes
  fused            ResNet-18, amp O2, static scale, four steps: identical parameters on all ranks, loss falls
  overflow         dynamic scale; rank 1 alone produces an infinite gradient in step 1: EVERY rank skips that step
  legacy-overflow  the old order (overflow pass before the reductions are waited for) on the same input: the
                   ranks must be seen to disagree - proof that ``overflow`` tests what it claims to
argv: rank world port mode"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "fused"
os.environ.update(HOROVOD_RANK=str(rank), HOROVOD_SIZE=str(world), HOROVOD_LOCAL_RANK=str(rank),
                  HOROVOD_LOCAL_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                  PDA_HVD_NO_CUDA="1", PDA_HVD_DEVICE_COMM="p2p", PDA_HVD_P2P_DEVICE="0")
os.environ.setdefault("PDA_HVD_TEST_EXEC_DELAY_MS", "40")
import torch

import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import amp, models, ops
from pytorch_distributed_amd.amp import frontend
from pytorch_distributed_amd.optim import FusedSGD


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    hvd.init()
    torch.manual_seed(200 + rank)
    net = models.get_model("resnet18", num_classes=16).to(dev)
    hvd.broadcast_parameters(net.state_dict(), root_rank=0)
    opt = FusedSGD(net.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    dynamic = mode != "fused"
    net, opt = amp.initialize(net, opt, opt_level="O2", verbosity=0, loss_scale="dynamic" if dynamic else 128.0)
    hvd.broadcast_optimizer_state(opt, root_rank=0)
    opt = hvd.DistributedOptimizer(opt, named_parameters=net.named_parameters(), compression=hvd.Compression.fp16)
    scaler = opt._amp_scaler_ref
    if dynamic:
        scaler.state[0] = 128.0
    torch.manual_seed(300 + rank)          # every rank its own shard
    x = torch.randn(16, 3, 64, 64, device=dev)
    t = torch.randint(0, 16, (16,), device=dev)
    losses = []
    for step in range(4):
        opt.zero_grad()
        loss = ops.cross_entropy(net(x), t)
        poison = dynamic and step == 1 and rank == 1
        if mode == "legacy-overflow":
            (loss.float() * scaler.scale_tensor[0] * (float("inf") if poison else 1.0)).backward()
            frontend._overflow_check(opt, scaler)      # looks at gradients whose reductions have not run yet
        else:
            with amp.scale_loss(loss, opt) as scaled:
                (scaled * float("inf") if poison else scaled).backward()
        opt.step()
        losses.append(loss.item())
    names = [n for n, _ in net.named_parameters()]
    sums = torch.stack([p.detach().double().abs().sum() for p in net.parameters()]).float()
    sums = torch.nan_to_num(sums, nan=-1.0, posinf=-2.0, neginf=-3.0)
    allp = hvd.allgather(sums.view(1, -1), name="probe")
    diff = (allp - allp[0:1]).abs().max(0).values
    same = bool(diff.max() == 0)
    differing = [names[i] for i in torch.nonzero(diff).flatten().tolist()][:6]
    finite = all(bool(torch.isfinite(p).all()) for p in net.parameters())
    skipped = hvd.allgather(torch.tensor([[float(scaler.skipped_steps())]], device=dev), name="skipped").flatten().tolist()
    if mode == "fused":
        ok = same and finite and losses[-1] < losses[0] and skipped == [0.0] * world
    elif mode == "overflow":
        ok = same and finite and skipped == [1.0] * world
    else:
        # the defect this mode keeps alive: only the rank that produced the infinity skips
        ok = not (same and finite and skipped == [1.0] * world)
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} mode {mode} ranks agree {same} {differing}, finite {finite}, "
          f"skipped per rank {skipped}, loss {losses[0]:.3f} -> {losses[-1]:.3f}", flush=True)
    hvd.shutdown()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
