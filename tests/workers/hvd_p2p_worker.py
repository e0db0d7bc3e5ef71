"""hvd.DistributedOptimizer with device tensors on 2 ranks sharing GPU 0: the engine's device executor runs
its fused all-reduces and broadcasts through the peer-memory backend (PDA_HVD_DEVICE_COMM=p2p).  MNIST net
(dropout off), fp16 wire compression, two SGD steps, against one process on the concatenated batch.
argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
os.environ.update(HOROVOD_RANK=str(rank), HOROVOD_SIZE=str(world), HOROVOD_LOCAL_RANK=str(rank),
                  HOROVOD_LOCAL_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                  PDA_HVD_NO_CUDA="1", PDA_HVD_DEVICE_COMM="p2p", PDA_HVD_P2P_DEVICE="0")
import torch

import pytorch_distributed_amd.hvd as hvd
from pytorch_distributed_amd import models, ops


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    hvd.init()
    assert hvd.size() == world and hvd.rank() == rank
    torch.manual_seed(100 + rank)          # different initial weights per rank: broadcast_parameters repairs them
    net = models.Net().to(dev).eval()      # eval(): dropout off, the net has no BatchNorm
    hvd.broadcast_parameters(net.state_dict(), root_rank=0)
    torch.manual_seed(100)
    ref = models.Net().to(dev).eval()
    start = max((p - q).abs().max().item() for p, q in zip(net.parameters(), ref.parameters()))
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.5)
    hvd.broadcast_optimizer_state(opt, root_rank=0)
    opt = hvd.DistributedOptimizer(opt, named_parameters=net.named_parameters(), compression=hvd.Compression.fp16)
    ref_opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.5)
    w0 = [q.detach().clone() for q in ref.parameters()]
    per = 8
    torch.manual_seed(5)
    upd = 0.0
    for _ in range(2):
        x = torch.randn(world * per, 1, 28, 28, device=dev)
        t = torch.randint(0, 10, (world * per,), device=dev)
        sl = slice(rank * per, (rank + 1) * per)
        opt.zero_grad()
        ops.nll_loss(net(x[sl]), t[sl]).backward()
        opt.step()
        ref_opt.zero_grad()
        ops.nll_loss(ref(x), t).backward()
        ref_opt.step()
        du = torch.cat([(p.detach() - a).flatten() for p, a in zip(net.parameters(), w0)])
        dr = torch.cat([(q.detach() - a).flatten() for q, a in zip(ref.parameters(), w0)])
        upd = max(upd, ((du - dr).norm() / dr.norm()).item())
    avg = hvd.allreduce(torch.tensor([float(rank + 1)], device=dev), name="probe").item()
    gath = hvd.allgather(torch.full((2, 3), float(rank), device=dev), name="g")
    gath_ok = gath.shape == (2 * world, 3) and all(float(gath[2 * r, 0]) == r for r in range(world))
    # device Adasum (halving / doubling exchange + HIP dot / combine kernels) against the fp64 tree on the host
    from pytorch_distributed_amd.hvd.adasum import adasum_tree
    g = torch.Generator().manual_seed(77)
    vecs = [torch.randn(10007, generator=g) for _ in range(world)]
    vecs[1 % world] = vecs[1 % world] + 0.5 * vecs[0]            # correlated gradients: the coefficients matter
    ada = hvd.allreduce(vecs[rank].to(dev), op=hvd.Adasum, name="ada")
    ada_ref = adasum_tree(vecs)
    ada_err = ((ada.cpu().double() - ada_ref.double()).norm() / ada_ref.double().norm()).item()
    st = hvd.engine_stats()
    # a fused device group costs one pack(+cast to fp16), one collective and one unpack(+cast back, average)
    launches_ok = st["device_groups"] > 0 and st["max_launches_per_group"] <= 3
    ok = (start == 0.0 and upd < 5e-3 and abs(avg - (world + 1) / 2) < 1e-6 and gath_ok and launches_ok
          and ada_err < 1e-5)
    print(f"rank {rank}: engine {st} allgather_ok {gath_ok} adasum rel err {ada_err:.2e}", flush=True)
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} initial diff {start:.1e}, update rel err {upd:.2e}, avg {avg}", flush=True)
    hvd.shutdown()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
