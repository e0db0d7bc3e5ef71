"""One rank of the two-shot all-reduce test: all ranks share GPU 0 (IPC hand-shake, per-workgroup flag
protocol, parity buffers, shard / slice arithmetic; the xGMI hops themselves need a multi-GPU node).
Sizes from one element to several region capacities, fp32 / fp16 / bf16, sum and avg, in place and out of
place, against an fp64 sum of the per-rank inputs.  argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from pytorch_distributed_amd.comm.p2p import P2PTwoShot
from pytorch_distributed_amd.comm.store import TCPStore


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    store = TCPStore("127.0.0.1", port, world, is_master=(rank == 0), timeout=60.0)
    cap = 1 << 20                       # 1 MiB region capacity: the larger cases run in several pieces
    ar = P2PTwoShot(rank, world, store, cap_bytes=cap, timeout_ms=8000, nblk=8)
    ok = True
    sizes = [1, 7, 8, 255, 1024, 4097, 65536, 262144 - 3, 262144, 1 << 20, (1 << 20) + 12345, 3 * (1 << 20) + 1]
    it = 0
    for dtype, tol in ((torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        for n in sizes:
            it += 1
            g = torch.Generator(device="cpu").manual_seed(1000 + it)
            per_rank = [torch.randn(n, generator=g) for _ in range(world)]      # every rank draws all inputs
            mine = per_rank[rank].to(dev).to(dtype)
            want = sum(p.to(dtype).double() for p in per_rank)
            op = "avg" if it % 2 else "sum"
            if op == "avg":
                want = want / world
            if it % 3 == 0:
                out = torch.full((n,), float("nan"), device=dev, dtype=dtype)
                keep = mine.clone()
                ar.all_reduce(mine, op, out=out)
                if not torch.equal(mine, keep):
                    ok = False
                    print(f"rank {rank}: input modified by an out-of-place call (n={n}, {dtype})", flush=True)
            else:
                out = ar.all_reduce(mine, op)
            err = ((out.double().cpu() - want).abs().max() / (want.abs().max() + 1e-30)).item()
            if not err < tol:
                ok = False
                print(f"rank {rank}: n={n} {dtype} {op}: rel err {err:.3e}", flush=True)
                break
    # every rank must hold bit-identical results (each shard is reduced by exactly one rank)
    t = torch.randn(100003, generator=torch.Generator().manual_seed(rank)).to(dev).half()
    ar.all_reduce(t, "sum")
    torch.cuda.synchronize()
    digest = t.view(torch.int16).to(torch.int64).sum().item()
    store.set(f"digest{rank}", str(digest).encode())
    same = all(int(bytes(store.get(f"digest{r}")).decode()) == digest for r in range(world))
    ok = ok and same
    ar.check()
    ar.close()
    print(f"rank {rank}: {'OK' if ok else 'FAILED'} identical={same}", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
