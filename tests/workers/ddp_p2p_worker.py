"""DistributedDataParallel on 2 ranks that share GPU 0 (collectives through the peer-memory backend):
amp O2 + FusedSGD + SyncBatchNorm ResNet-18, two optimizer steps, against one process running plain
BatchNorm on the concatenated batch.  Exercises the device-side multi-rank path of the reducer: parameter
broadcast at wrap time, bucket views as .grad, wgrad accumulating straight into the buckets, all-reduce
launched from the ready hooks, bucket rebuild after the first iteration, per-forward buffer broadcast.
argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PDA_P2P_SMALL"] = "1"
import torch

from pytorch_distributed_amd import amp, comm, models, ops
from pytorch_distributed_amd.comm.p2p import P2PCollectives
from pytorch_distributed_amd.nn import SyncBatchNorm
from pytorch_distributed_amd.optim import FusedSGD
from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP


def build(dev, sync):
    torch.manual_seed(3)
    m = models.resnet18(num_classes=10).to(dev)
    if sync:
        SyncBatchNorm.convert_sync_batchnorm(m)
        m = m.to(dev)
    opt = FusedSGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    return amp.initialize(m, opt, opt_level="O0", verbosity=0)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    comm.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    dev = torch.device("cuda", 0)
    per = 4
    ref, ref_opt = build(dev, False)
    net, opt = build(dev, True)
    if rank != 0:   # DDP has to repair this from rank 0
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.5)
    coll = P2PCollectives(rank, world, store=comm.PrefixStore("ddpc", comm.native_store()), device=dev)
    ddp = DDP(net, device_ids=[0], bucket_cap_mb=8, comm=coll)
    torch.manual_seed(11)
    ok = True
    msgs = []
    w0 = [q.detach().clone() for q in ref.parameters()]   # both models start here (DDP broadcast rank 0's copy)
    for step in range(2):
        x = torch.randn(world * per, 3, 32, 32, device=dev)
        t = torch.randint(0, 10, (world * per,), device=dev)
        sl = slice(rank * per, (rank + 1) * per)
        ddp.zero_grad()
        loss = ops.cross_entropy(ddp(x[sl]), t[sl])
        loss.backward()
        opt.step()
        ref_opt.zero_grad(set_to_none=False)
        loss_ref = ops.cross_entropy(ref(x), t)
        loss_ref.backward()
        ref_opt.step()
        worst, wk = 0.0, ""
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            e = ((p - q).norm() / (q.norm() + 1e-12)).item()
            if e > worst:
                worst, wk = e, k
        # the UPDATE is what the gradients produced: compare it as one vector (weights barely move in two
        # steps, so a weight comparison alone would not notice a wrong reduction)
        du = torch.cat([(p.detach() - a).flatten() for p, a in zip(net.parameters(), w0)])
        dr = torch.cat([(q.detach() - a).flatten() for q, a in zip(ref.parameters(), w0)])
        upd = ((du - dr).norm() / dr.norm()).item()
        msgs.append(f"step {step}: worst weight diff {worst:.2e} ({wk}), update rel err {upd:.2e}")
        ok = ok and worst < 2e-2 and upd < 2e-2   # (biases start at 0: their relative difference is the noisy one)
    coll.check()
    nb = len(ddp.bucket_summary())
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} buckets {nb}; " + "; ".join(msgs), flush=True)
    torch.cuda.synchronize()
    coll.close()
    comm.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
