import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd.ops import conv as C

DEV = "cuda"
for ci, co, shape, stats in ((512, 256, (64, 28, 28), True), (512, 256, (64, 28, 28), False), (512, 256, (16, 28, 28), True),
                             (512, 192, (64, 28, 28), True), (256, 256, (64, 28, 28), True), (512, 512, (64, 28, 28), True)):
    torch.manual_seed(5)
    n, h, w_ = shape
    x = torch.randn(n, h, w_, ci, device=DEV).half()
    wk = (torch.randn(co, 1, 1, ci, device=DEV) / ci ** 0.5).half()
    y32 = x.float().reshape(-1, ci) @ wk.float().reshape(co, ci).t()
    for rep in range(int(os.environ.get("REPS", "3"))):
        r = C.fprop_gpu(x, wk, (1, 1), (0, 0), stats=stats, cfg=C._FPROP_PW_CFG)
        y = r[0] if stats else r
        d = (y.float().reshape(-1, co) - y32).abs()
        bad = d > 0.05
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        blocks = sorted(set((int(r_) // 32) for r_ in rows.tolist()))
        print(f"ci {ci} co {co} {shape} stats {stats} rep {rep}: bad elems {int(bad.sum())}, row blocks(32) {blocks[:10]}{'...' if len(blocks) > 10 else ''} n={len(blocks)}, "
              f"col tiles {sorted(set(int(c) // 64 for c in cols.tolist()))}", flush=True)
