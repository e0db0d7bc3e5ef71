"""Block-end pass + next conv1 as two kernels (norm.apply with sign bits, then the A-stationary pointwise kernel)
against the pointwise kernel's MAT mode that does both, at the ResNet-50 shapes (batch 400)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytorch_distributed_amd.ops import conv as C, norm as Nm  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    for hw, ci, co in [(56, 256, 64), (56, 256, 128), (28, 512, 128), (28, 512, 256)]:
        y3 = torch.randn(400, hw, hw, ci, device=dev, dtype=torch.float16)
        res = torch.randn(400, hw, hw, ci, device=dev, dtype=torch.float16)
        sc, sf = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.3
        w = (torch.randn(co, 1, 1, ci, device=dev) / ci ** 0.5).half()

        def two():
            out, bits = Nm.apply(y3, sc, sf, res, True, want_mask=True)
            return C.fprop_gpu(out, w, (1, 1), (0, 0), stats=True, cfg=C._FPROP_PW_CFG)

        def one():
            return C.pw_fprop_materialize(y3, res, sc, sf, None, w, stats=True)

        t_apply = timeit(lambda: Nm.apply(y3, sc, sf, res, True, want_mask=True))
        t2, t1 = timeit(two), timeit(one)
        print(f"{hw:3d} {ci:4d} -> {co:4d} | apply {t_apply:6.1f} us, apply + conv {t2:6.1f} us | one kernel {t1:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
