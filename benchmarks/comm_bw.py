#!/usr/bin/env python
"""All-reduce bandwidth: this package's RCCL communicator vs the peer-memory two-shot kernel.

    python -m pytorch_distributed_amd.launch --nproc_per_node 8 benchmarks/comm_bw.py --comm rccl,p2p2
    python -m pytorch_distributed_amd.launch --nproc_per_node 4 benchmarks/comm_bw.py --comm p2p2 --same-gpu

Rank 0 prints one JSON line per (backend, dtype, size): algorithm bandwidth (message bytes / time) and bus
bandwidth (x 2(N-1)/N, the bytes each GPU has to move per byte of message - comparable across N and with
link speed).  ``--same-gpu`` puts every rank on GPU 0 (only the peer-memory backend allows that): the
"links" are then HBM, which still measures the kernel's own overhead and shows it working at N > 1 on a
one-GPU machine.  Sizes follow the gradient-bucket range of DistributedDataParallel (the collective behind
`loss.backward()` at /root/reference/2.distributed.py:231).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--comm", default="rccl,p2p2")
    ap.add_argument("--sizes-mb", default="0.001,0.064,1,4,16,25,64")
    ap.add_argument("--dtypes", default="fp16,fp32")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--same-gpu", action="store_true")
    ap.add_argument("--nblk", type=int, default=16)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from pytorch_distributed_amd import comm as pcomm
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pcomm.init_process_group("gloo", rank=rank, world_size=world)
    import torch.distributed as dist
    backends = {}
    for name in a.comm.split(","):
        if name == "rccl":
            if a.same_gpu and world > 1:
                continue                       # RCCL refuses two ranks on one device
            backends[name] = pcomm.device_communicator(dev)
        elif name == "p2p2":
            from pytorch_distributed_amd.comm.p2p import P2PTwoShot
            backends[name] = P2PTwoShot(rank, world, store=pcomm.PrefixStore("bw", pcomm.native_store()), device=dev,
                                        cap_bytes=64 << 20, nblk=a.nblk)
    lines = []
    for name, c in backends.items():
        for dn in a.dtypes.split(","):
            dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[dn]
            for mb in [float(x) for x in a.sizes_mb.split(",")]:
                n = max(1, int(mb * (1 << 20)) // torch.empty((), dtype=dt).element_size())
                t = torch.randn(n, device=dev).to(dt)
                for _ in range(a.warmup):
                    c.all_reduce(t, "avg")
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    c.all_reduce(t, "avg")
                torch.cuda.synchronize()
                dt_s = (time.perf_counter() - t0) / a.iters
                tm = torch.tensor([dt_s], dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                nbytes = n * t.element_size()
                if rank == 0:
                    algbw = nbytes / float(tm) / 1e9
                    rec = {"backend": name, "dtype": dn, "bytes": nbytes, "n_ranks": world, "same_gpu": a.same_gpu,
                           "us": round(float(tm) * 1e6, 1), "algbw_GBps": round(algbw, 2),
                           "busbw_GBps": round(algbw * 2 * (world - 1) / max(world, 1), 2)}
                    lines.append(json.dumps(rec))
                    print(lines[-1], flush=True)
    if rank == 0 and a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    for c in backends.values():
        if hasattr(c, "check"):
            c.check()
    torch.cuda.synchronize()
    dist.barrier()
    for name, c in backends.items():
        if name == "p2p2":
            c.close()
    pcomm.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
