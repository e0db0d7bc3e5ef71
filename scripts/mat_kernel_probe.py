import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd.ops import conv as C, norm as Nm

DEV = "cuda"
for ci, co, shape in ((256, 64, (64, 56, 56)), (512, 128, (64, 28, 28)), (512, 128, (3, 14, 14)), (512, 256, (64, 28, 28))):
    torch.manual_seed(5)
    n, h, w_ = shape
    y3 = torch.randn(n, h, w_, ci, device=DEV).half()
    res = torch.randn(n, h, w_, ci, device=DEV).half()
    sc, sf = torch.rand(ci, device=DEV) + 0.5, torch.randn(ci, device=DEV) * 0.3
    wk = (torch.randn(co, 1, 1, ci, device=DEV) / ci ** 0.5).half()
    r = C.pw_fprop_materialize(y3, res, sc, sf, None, wk, stats=True)
    out, bits, y, psum, psq = r
    ref_out, ref_bits = Nm.apply(y3, sc, sf, res, True, want_mask=True)
    d = (out.float() - ref_out.float()).abs()
    print(f"ci {ci} co {co} {shape}: out differing {int((out != ref_out).sum())} of {out.numel()}, max {float(d.max()):.4f}")
    ref_y, ref_ps, ref_pq = C.fprop_gpu(out, wk, (1, 1), (0, 0), stats=True, cfg=C._FPROP_PW_CFG)
    dy = (y.float() - ref_y.float()).abs()
    print(f"   y vs plain kernel on its own out: differing {int((y != ref_y).sum())}, max {float(dy.max()):.4f}")
    y32 = out.float().reshape(-1, ci) @ wk.float().reshape(co, ci).t()
    print(f"   y vs fp32 product: rel {float((y.float().reshape(-1, co) - y32).norm() / y32.norm()):.5f}; plain: "
          f"{float((ref_y.float().reshape(-1, co) - y32).norm() / y32.norm()):.5f}")
    rows = (dy.reshape(-1, co).max(1).values > 0).nonzero().flatten()
    if rows.numel():
        print("   rows differing:", rows.numel(), rows[:12].tolist(), "...", rows[-4:].tolist())
    print("   mask ok:", bool(torch.equal(C.unpack_mask_bits(bits, out) > 0, out > 0)),
          " psum rel", float((psum.sum(0) - ref_ps.sum(0)).norm() / ref_ps.sum(0).norm()))
