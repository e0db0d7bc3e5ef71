"""One rank of the P2PAllReduce test: all ranks share GPU 0 (the peer mapping then stays on one device,
which exercises the IPC hand-shake, the flag protocol and the parity buffers; the xGMI hop itself needs
a multi-GPU node).  argv: rank world port"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from pytorch_distributed_amd.comm.p2p import P2PAllReduce
from pytorch_distributed_amd.comm.store import TCPStore


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    store = TCPStore("127.0.0.1", port, world, is_master=(rank == 0), timeout=60.0)
    ar = P2PAllReduce(rank, world, store, max_elems=4096, timeout_ms=5000)
    dev = torch.device("cuda", 0)
    ok = True
    for it in range(200):
        n = [1, 3, 64, 777, 4096][it % 5]
        base = torch.arange(n, device=dev, dtype=torch.float32) * 0.25 + it
        t = base * (rank + 1)
        want = base * sum(r + 1 for r in range(world))
        if it % 2:
            out = torch.empty_like(t)
            ar.all_reduce(t, "avg", out=out)
            want = want / world
        else:
            out = ar.all_reduce(t, "sum")
        if not torch.allclose(out, want, rtol=1e-6, atol=1e-6):
            ok = False
            print(f"rank {rank}: mismatch at iteration {it} (n={n}): {out[:4].tolist()} vs {want[:4].tolist()}", flush=True)
            break
    ar.check()
    ar.close()
    print(f"rank {rank}: {'OK' if ok else 'FAILED'}", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
