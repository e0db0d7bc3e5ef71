"""A DistributedDataParallel step recorded into a hipGraph against the same step run eagerly: amp O2 + FusedSGD
ResNet-18 on 32x32 inputs (the launch-bound shape of /root/reference/2.distributed.py:192-239), fp16 gradients
on the wire handed to the fused optimizer, weight gradients on the backward side stream.  With the run-to-run
reproducible weight-gradient reduction switched on, the parameters after warm-up + 3 steps must be bit-identical.
world > 1: the ranks share GPU 0 and reduce through the peer-memory two-shot kernel, whose sequence numbers
live in device memory so that a replayed launch stays in step with its peers.
argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from pytorch_distributed_amd import amp, backends, comm, models, ops
from pytorch_distributed_amd.comm.p2p import P2PCollectives
from pytorch_distributed_amd.optim import FusedSGD
from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP
from pytorch_distributed_amd.utils.graph import GraphedStep

WARM, STEPS = 3, 3


def build(dev, rank, world, tag):
    torch.manual_seed(3)
    m = models.resnet18(num_classes=10).to(dev)
    opt = FusedSGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    m, opt = amp.initialize(m, opt, opt_level="O2", loss_scale=128.0, verbosity=0)
    if world > 1:
        coll = P2PCollectives(rank, world, store=comm.PrefixStore(tag, comm.native_store()), device=dev, overlap=True)
        ddp = DDP(m, device_ids=[0], bucket_cap_mb=8, comm=coll, fused_wire=True)
    else:
        coll = None
        ddp = DDP(m, device_ids=[0], bucket_cap_mb=8, force_comm=True, fused_wire=True)      # own RCCL communicator, one rank
    assert ddp.wire_dtype == torch.float16 and ddp.fused_wire
    return m, opt, ddp, coll


def main():
    import faulthandler
    faulthandler.dump_traceback_later(80, exit=True)     # a hang names its line
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm.init_process_group("gloo" if world > 1 else "cpu:gloo,cuda:nccl", f"tcp://127.0.0.1:{port}",
                            world_size=world, rank=rank)
    backends.deterministic = True
    torch.manual_seed(50 + rank)
    x = torch.randn(16, 3, 32, 32, device=dev)
    t = torch.randint(0, 10, (16,), device=dev)

    def make_step(ddp, opt):
        def step():
            ddp.zero_grad()
            loss = ops.cross_entropy(ddp(x), t)
            with amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
            opt.step()
            return loss
        return step

    m1, o1, d1, c1 = build(dev, rank, world, "eager")
    s1 = make_step(d1, o1)
    for _ in range(WARM + STEPS):
        l1 = s1()
    torch.cuda.synchronize()
    m2, o2, d2, c2 = build(dev, rank, world, "graph")
    if os.environ.get("PDA_TEST_WARM_ON_DEFAULT_STREAM") == "1":
        # the way engine.train gets there: eager steps on the thread's own stream, then a recording without warm-up
        st = make_step(d2, o2)
        for _ in range(WARM):
            st()
        s2 = GraphedStep(st, warmup=0, ddp=d2)
    else:
        s2 = GraphedStep(make_step(d2, o2), warmup=WARM, ddp=d2)
    assert s2.captured
    for _ in range(STEPS):
        l2 = s2()
    torch.cuda.synchronize()
    diff = [k for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()) if not torch.equal(p, q)]
    moved = not torch.equal(m2.fc.weight, models.resnet18(num_classes=10).fc.weight.to(dev))
    for c in (c1, c2):
        if c is not None:
            c.check()
    ok = not diff and moved and float(l1) == float(l2) and float(l2) == float(l2)
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} graphed vs eager after {WARM}+{STEPS} steps: {len(diff)} parameters "
          f"differ {diff[:4]}, loss {float(l1):.6f} / {float(l2):.6f}, buckets {len(d2.bucket_summary())}", flush=True)
    for c in (c1, c2):
        if c is not None:
            c.close()
    comm.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
