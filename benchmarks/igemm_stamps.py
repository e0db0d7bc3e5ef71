#!/usr/bin/env python
"""Where one implicit-GEMM workgroup spends its shader cycles (diagnostic build path of conv_igemm.hip).

    PDA_IGEMM_STAMPS=1 python benchmarks/igemm_stamps.py --h 14 --cin 256 --cout 256 --k 3 --batch 400 --cfg 5

Every workgroup's thread 0 stamps s_memtime at: entry, set-up done, first tile staged, main loop done, write-out
done.  Printed: median cycles of each phase over the workgroups, the span of the launch (first entry to last
exit) and how many workgroups were resident at the median entry time."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDA_IGEMM_STAMPS", "1")
import numpy as np
import torch

from pytorch_distributed_amd.ops import _lib
from pytorch_distributed_amd.ops import conv as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="fprop", choices=["fprop", "dgrad"])
    ap.add_argument("--h", type=int, default=14)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--batch", type=int, default=400)
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=50)
    a = ap.parse_args()
    dev = "cuda"
    s, k, pad = a.stride, a.k, a.k // 2
    ho = (a.h + 2 * pad - k) // s + 1
    x = torch.randn(a.batch, a.h, a.h, a.cin, device=dev).half()
    w = (torch.randn(a.cout, k, k, a.cin, device=dev) / (k * k * a.cin) ** 0.5).half()
    wt = C.krsc_to_crsk(w, torch.float16)
    dy = torch.randn(a.batch, ho, ho, a.cout, device=dev).half()

    def run():
        if a.op == "fprop":
            C.fprop_gpu(x, w, (s, s), (pad, pad), stats=True, cfg=a.cfg)
        else:
            C.dgrad_gpu(dy, wt, (a.h, a.h), (s, s), (pad, pad), cfg=a.cfg)

    for _ in range(a.warmup):
        run()
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    l = _lib.lib()
    l.pda_conv_stamps_read.argtypes = [ctypes.c_void_p]
    buf = np.zeros((8192, 8), dtype=np.uint64)
    rc = l.pda_conv_stamps_read(buf.ctypes.data)
    assert rc == 0, rc
    st = buf[buf[:, 0] > 0].astype(np.int64)
    t0 = st[:, 0].min()
    ph = ["set-up", "first tile", "main loop", "write-out"]
    print(f"cfg {a.cfg}: {len(st)} workgroups; launch span {st[:, 4].max() - t0} cycles")
    for i, name in enumerate(ph):
        d = st[:, i + 1] - st[:, i]
        print(f"  {name:10s} median {int(np.median(d)):7d}  p10 {int(np.percentile(d, 10)):7d}  p90 {int(np.percentile(d, 90)):7d}")
    life = st[:, 4] - st[:, 0]
    print(f"  lifetime   median {int(np.median(life)):7d}; entry times (rel.) p50 {int(np.median(st[:, 0] - t0))} max {int((st[:, 0] - t0).max())}")


if __name__ == "__main__":
    main()
