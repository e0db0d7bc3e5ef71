"""Stress of the peer-memory collectives on their own (no Horovod engine): two ranks on GPU 0 issue thousands of
back-to-back in-place all-reduces of gradient-group sizes (fp16 and fp32, the one-shot and the two-shot kernel mixed,
from a side stream while the main stream runs matmuls) and compare digests of every result across the ranks.
argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pytorch_distributed_amd.comm.p2p import P2PCollectives
from pytorch_distributed_amd.comm.store import TCPStore, PrefixStore


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    store = TCPStore("127.0.0.1", port, world, is_master=(rank == 0), timeout=60.0)
    coll = P2PCollectives(rank, world, store=PrefixStore("stress", store), device=dev)
    sizes = [64, 128, 36864 + 128, 3 * 36864, 73728 + 256, 147456, 8192, 16384, 16385, 589824 + 512, 2359296, 512 + 16]
    side = torch.cuda.Stream(dev)
    a = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
    digests = []
    rounds = int(os.environ.get("PDA_STRESS_ROUNDS", "300"))
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    bufs = {dt: torch.empty(max(sizes), device=dev, dtype=dt) for dt in (torch.float16, torch.float32)}
    for it in range(rounds):
        b = a @ a                                   # compute on the main stream, as backward would
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for k, n in enumerate(sizes):
                dt = torch.float16 if (it + k) % 2 == 0 else torch.float32
                flat = bufs[dt][:n]
                flat.normal_()                      # stands in for the pack kernel writing the shared fusion buffer
                coll.all_reduce(flat, "sum")
                digests.append(flat.double().sum().reshape(1))
        torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    mine = torch.cat(digests).cpu()
    store.set(f"dg{rank}", mine.numpy().tobytes())
    import numpy as np
    bad = 0
    for r in range(world):
        other = torch.from_numpy(np.frombuffer(bytes(store.get(f"dg{r}")), dtype=np.float64).copy())
        diff = torch.nonzero(other != mine).flatten()
        if diff.numel():
            bad += diff.numel()
            if rank == 0:
                first = int(diff[0])
                print(f"rank 0 vs {r}: {diff.numel()} of {mine.numel()} results differ, first: round {first // len(sizes)}, "
                      f"size {sizes[first % len(sizes)]}", flush=True)
    print(f"rank {rank}: {'OK' if bad == 0 else 'FAILED'} ({mine.numel()} all-reduces)", flush=True)
    coll.close() if hasattr(coll, "close") else None
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
