"""engine.train with --graph on several ranks: the step of a DistributedDataParallel model (amp O2 + FusedSGD,
ResNet-18 on 32x32 inputs - the launch-bound default workload of /root/reference/2.distributed.py:192-239) is
recorded after three eager steps and replayed; with the reproducible weight-gradient reduction on, the parameters
after the epoch equal those of the same epoch run eagerly, bit for bit.  The ranks share GPU 0 and reduce through the
peer-memory two-shot kernel.  argv: rank world port"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from pytorch_distributed_amd import amp, backends, comm, engine, models, nn as pnn
from pytorch_distributed_amd.comm.p2p import P2PCollectives
from pytorch_distributed_amd.optim import FusedSGD
from pytorch_distributed_amd.parallel import DistributedDataParallel as DDP

STEPS = 7


def run(dev, rank, world, tag, graph):
    torch.manual_seed(3)
    m = models.resnet18(num_classes=10).to(dev)
    opt = FusedSGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    m, opt = amp.initialize(m, opt, opt_level="O2", loss_scale=128.0, verbosity=0)
    coll = P2PCollectives(rank, world, store=comm.PrefixStore(tag, comm.native_store()), device=dev, overlap=True)
    ddp = DDP(m, device_ids=[0], bucket_cap_mb=8, comm=coll, fused_wire=True)
    g = torch.Generator().manual_seed(100 + rank)
    batches = [(torch.randn(16, 3, 32, 32, generator=g), torch.randint(0, 10, (16,), generator=g)) for _ in range(STEPS)]
    args = argparse.Namespace(graph=graph, steps_per_epoch=0, print_freq=int(os.environ.get("PDA_TEST_PRINT_FREQ", "3")))
    loss, _ = engine.train(batches, ddp, pnn.CrossEntropyLoss(), opt, 0, dev, args, zero_grad=ddp.zero_grad)
    torch.cuda.synchronize()
    state = getattr(opt, "_pda_graph_state", None)
    captured = state is not None and state["graph"] is not None and state["graph"].captured
    coll.check()
    return m, loss, captured, coll


def main():
    import faulthandler
    faulthandler.dump_traceback_later(70, exit=True)     # a hang names its line
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm.init_process_group("gloo", f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    backends.deterministic = True
    m1, l1, c1, k1 = run(dev, rank, world, "eager", False)
    m2, l2, c2, k2 = run(dev, rank, world, "graph", True)
    diff = [k for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()) if not torch.equal(p, q)]
    # (the loss meter sums per-row losses with float atomics: equal up to the summation order)
    ok = (not c1) and bool(c2) and not diff and abs(l1 - l2) < 1e-4 * max(abs(l1), 1.0)
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} engine.train graphed vs eager over {STEPS} steps: captured {c2} "
          f"(eager run: {c1}), {len(diff)} parameters differ {diff[:4]}, loss {l1:.6f} / {l2:.6f}", flush=True)
    k1.close()
    k2.close()
    comm.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
