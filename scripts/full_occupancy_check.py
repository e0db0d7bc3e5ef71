"""Every element of every conv kernel candidate at chip-filling sizes, several launches each, against fp32 autograd on
the same 16-bit values - and, for the outputs that no atomics touch (fprop, dgrad, the fused kernels' dx), bit for bit
against the first launch.  Unit tests run small shapes; a race that needs two busy workgroups per CU only shows here.

    python scripts/full_occupancy_check.py [--batch 128] [--reps 4]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pytorch_distributed_amd.ops import conv as C

DEV = "cuda"
LAYERS = [(56, 64, 64, 3, 1), (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (56, 128, 128, 3, 2), (56, 256, 512, 1, 2),
          (28, 128, 128, 3, 1), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (28, 256, 256, 3, 2), (14, 256, 256, 3, 1),
          (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (14, 512, 512, 3, 2), (7, 512, 512, 3, 1), (7, 512, 2048, 1, 1),
          (7, 2048, 512, 1, 1)]
FCFGS = (1, 2, 9, 12, 5, 6, 7, 13, 33, 34)
DCFGS = (1, 2, 9, 12, 5, 6, 7, 13)
WCFGS = (1, 12, 65, 66, 68, 69, 70)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--batch", type=int, default=128)
    p.add_argument("--reps", type=int, default=4)
    a = p.parse_args()
    total_bad = 0
    for (hw, ci, co, k, s) in LAYERS:
        torch.manual_seed(hw + ci + co)
        pad = k // 2
        x = torch.randn(a.batch, hw, hw, ci, device=DEV).half()
        w = (torch.randn(co, k, k, ci, device=DEV) / (k * k * ci) ** 0.5).half()
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        wr = w.float().permute(0, 3, 1, 2).requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride=s, padding=pad)
        dy = torch.randn(yr.shape[0], yr.shape[2], yr.shape[3], co, device=DEV).half()
        yr.backward(dy.float().permute(0, 3, 1, 2))
        ref_y = yr.detach().permute(0, 2, 3, 1)
        ref_dx = xr.grad.permute(0, 2, 3, 1)
        ref_dw = wr.grad.permute(0, 2, 3, 1)
        w_t = C.krsc_to_crsk(w, torch.float16)
        tol_y = 0.02 * float(ref_y.abs().max())
        tol_dx = 0.02 * float(ref_dx.abs().max())
        tol_dw = 2e-3 * float(ref_dw.abs().max())
        line = []
        for kind, cfgs in (("fprop", FCFGS), ("dgrad", DCFGS), ("wgrad", WCFGS)):
            for cfg in cfgs:
                worst, nbad, ran = 0.0, 0, True
                first = None      # fprop / dgrad have no atomics on their output: every launch must give the same bits
                for rep in range(a.reps):
                    if kind == "fprop":
                        r = C.fprop_gpu(x, w, (s, s), (pad, pad), stats=True, cfg=cfg)
                        if r is None:
                            ran = False
                            break
                        d = (r[0].float() - ref_y).abs()
                        tol = tol_y
                        if first is None:
                            first = r[0]
                        elif not torch.equal(first, r[0]):
                            nbad += int((first != r[0]).sum())
                    elif kind == "dgrad":
                        r = C.dgrad_gpu(dy, w_t, (hw, hw), (s, s), (pad, pad), cfg=cfg)
                        if r is None:
                            ran = False
                            break
                        d = (r.float() - ref_dx).abs()
                        tol = tol_dx
                        if first is None:
                            first = r
                        elif not torch.equal(first, r):
                            nbad += int((first != r).sum())
                    else:
                        r = C.wgrad_gpu(dy, x, k, k, (s, s), (pad, pad), cfg=cfg)
                        if r is None:
                            ran = False
                            break
                        d = (r - ref_dw).abs()
                        tol = tol_dw
                    worst = max(worst, float(d.max()) / tol)
                    nbad += int((d > tol).sum())
                if ran:
                    line.append(f"{kind[0]}{cfg}:{'ok' if nbad == 0 else 'BAD ' + str(nbad)}({worst:.2f})")
                    total_bad += nbad
        print(f"{hw:3d} {ci:5d}->{co:5d} k{k} s{s}: " + " ".join(line), flush=True)
        del x, w, xr, wr, yr, dy, ref_y, ref_dx, ref_dw
        torch.cuda.empty_cache()
    total_bad += fused_kernels(a)
    total_bad += dgrad_epilogues(a)
    total_bad += grouped_convs(a)
    total_bad += stem_backward(a)
    print("TOTAL BAD ELEMENTS:", total_bad)
    return 1 if total_bad else 0


def fused_kernels(a) -> int:
    """The one-pass bottleneck-tail backward and the masked input gradient of the weight-stationary 3x3 kernel, every
    element of dx, at the sizes they run at in ResNet-50."""
    bad_total = 0
    for (Ci, Co, hw) in ((64, 256, 56), (128, 512, 28)):
        torch.manual_seed(Ci)
        M = a.batch * hw * hw
        dz = torch.randn(M, Co, device=DEV).half()
        y3 = torch.randn(M, Co, device=DEV).half()
        coef = torch.randn(5, Co, device=DEV) * torch.tensor([1.0, 0.1, 0.05, 1, 1], device=DEV).view(5, 1)
        x2 = torch.randn(M, Ci, device=DEV).half()
        scale = (torch.rand(Ci, device=DEV) + 0.5).half()
        shift = (torch.rand(Ci, device=DEV) - 0.5).half()
        w = (torch.randn(Co, 1, 1, Ci, device=DEV) / Ci ** 0.5).half()
        wt = C.krsc_to_crsk(w, torch.float16)
        dy3 = (coef[0] * dz.float() + coef[1] * y3.float() + coef[2]).half()
        a2 = torch.relu(torch.addcmul(shift, x2, scale))
        ref = (dy3.float() @ w.view(Co, Ci).float()) * (a2 > 0)
        ref_dw = dy3.float().t() @ a2.float()
        tol, tol_w = 0.02 * float(ref.abs().max()), 2e-3 * float(ref_dw.abs().max())
        nbad, worst = 0, 0.0
        for rep in range(a.reps):
            dw = torch.zeros(Co, 1, 1, Ci, device=DEV)
            dx, sums, dy = C.bottleneck_tail_bwd(dz, y3, coef, x2, (scale, shift), wt, dw if Ci == 64 else None)
            d = (dx.float() - ref).abs()
            nbad += int((d > tol).sum())
            worst = max(worst, float(d.max()) / tol)
            if rep == 0:
                first = dx
            else:
                nbad += int((first != dx).sum())          # dx carries no atomics: the same bits on every launch
            if Ci == 64:
                dd = (dw.view(Co, Ci) - ref_dw).abs()
                nbad += int((dd > tol_w).sum())
                worst = max(worst, float(dd.max()) / tol_w)
        print(f"tail backward {Ci:4d} -> {Co:4d} at {hw}x{hw}: {'ok' if nbad == 0 else 'BAD ' + str(nbad)}({worst:.2f})", flush=True)
        bad_total += nbad
        del dz, y3, x2, ref, dy3, a2
        torch.cuda.empty_cache()
    # weight-stationary 3x3 (64 -> 64): masked dgrad with the BatchNorm sums, against fp32
    torch.manual_seed(3)
    n, hw = a.batch, 56
    dy = torch.randn(n, hw, hw, 64, device=DEV).half()
    w = (torch.randn(64, 3, 3, 64, device=DEV) / 24.0).half()
    wt = C.krsc_to_crsk(w, torch.float16)
    bn_in = torch.randn(n, hw, hw, 64, device=DEV).half()
    sc, sf = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3
    ref = torch.nn.grad.conv2d_input((n, 64, hw, hw), w.float().permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2),
                                     stride=1, padding=1).permute(0, 2, 3, 1)
    ref = ref * ((bn_in.float() * sc + sf) > 0)
    tol = 0.02 * float(ref.abs().max())
    nbad, worst = 0, 0.0
    for rep in range(a.reps):
        r = C.dgrad_gpu(dy, wt, (hw, hw), (1, 1), (1, 1), cfg=C._FPROP_C64_CFG, bn_input=bn_in, bn_scale=sc, bn_shift=sf,
                        bn_sums=True)
        if rep == 0:
            first = r[0]
        else:
            nbad += int((first != r[0]).sum())
        d = (r[0].float() - ref).abs()
        # (elements whose mask argument sits within rounding of zero may legitimately flip: count only real errors)
        near = (bn_in.float() * sc + sf).abs() < 2e-3
        nbad += int(((d > tol) & ~near).sum())
        worst = max(worst, float(d[~near].max()) / tol)
    print(f"weight-stationary 3x3 masked dgrad at {hw}x{hw}: {'ok' if nbad == 0 else 'BAD ' + str(nbad)}({worst:.2f})", flush=True)
    return bad_total + nbad


def dgrad_epilogues(a) -> int:
    """The input-gradient kernels with the fused epilogue (residual gradient added, the producer's ReLU mask recomputed
    from its conv output, the two BatchNorm-backward sums): dx against fp32 and bit for bit across launches."""
    bad_total = 0
    for (hw, ci, co, k) in ((56, 256, 64, 1), (28, 128, 128, 3), (14, 256, 256, 3), (14, 1024, 256, 1)):
        torch.manual_seed(hw + ci)
        pad = k // 2
        n = a.batch
        dy = torch.randn(n, hw, hw, co, device=DEV).half()
        w = (torch.randn(co, k, k, ci, device=DEV) / (k * k * ci) ** 0.5).half()
        wt = C.krsc_to_crsk(w, torch.float16)
        res = torch.randn(n, hw, hw, ci, device=DEV).half()
        bn_in = torch.randn(n, hw, hw, ci, device=DEV).half()
        sc, sf = torch.rand(ci, device=DEV) + 0.5, torch.randn(ci, device=DEV) * 0.3
        ref = torch.nn.grad.conv2d_input((n, ci, hw, hw), w.float().permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2),
                                         stride=1, padding=pad).permute(0, 2, 3, 1)
        arg = bn_in.float() * sc + sf
        ref = (ref + res.float()) * (arg > 0)
        near = arg.abs() < 2e-3
        tol = 0.02 * float(ref.abs().max())
        line = []
        for cfg in (1, 2, 3, 9, 10, 11, 5, 6):
            nbad, worst, first, ran = 0, 0.0, None, True
            for rep in range(a.reps):
                r = C.dgrad_gpu(dy, wt, (hw, hw), (1, 1), (pad, pad), res=res, cfg=cfg, bn_input=bn_in, bn_scale=sc,
                                bn_shift=sf, bn_sums=True)
                if r is None:
                    ran = False
                    break
                dx = r[0]
                d = (dx.float() - ref).abs()
                nbad += int(((d > tol) & ~near).sum())
                worst = max(worst, float(d[~near].max()) / tol)
                if first is None:
                    first = dx
                else:
                    nbad += int((first != dx).sum())
            if ran:
                line.append(f"e{cfg}:{'ok' if nbad == 0 else 'BAD ' + str(nbad)}({worst:.2f})")
                bad_total += nbad
        print(f"dgrad + epilogue {hw:3d} {co:5d}->{ci:5d} k{k}: " + " ".join(line), flush=True)
        del dy, w, res, bn_in, ref, arg, near
        torch.cuda.empty_cache()
    return bad_total


def grouped_convs(a) -> int:
    """ResNeXt's grouped 3x3 layers (32 groups) on the chunk kernels: forward, input and weight gradient through the
    autograd op, against fp32; forward and input gradient bit for bit across launches."""
    from pytorch_distributed_amd import ops
    bad_total = 0
    # (c, groups): ResNeXt's 32 groups, then depthwise layers of MobileNetV2 / ShuffleNetV2 (58 and 116 channels: the
    # 4-byte-per-lane instantiations)
    for (hw, c, s_, g) in ((56, 128, 1, 32), (56, 256, 2, 32), (28, 256, 1, 32), (14, 512, 1, 32), (7, 1024, 1, 32),
                           (112, 32, 1, 32), (112, 96, 2, 96), (56, 144, 1, 144), (28, 58, 1, 58), (14, 116, 2, 116),
                           (14, 576, 1, 576)):
        torch.manual_seed(hw + c)
        n = a.batch
        x = torch.randn(n, hw, hw, c, device=DEV).half()
        w = (torch.randn(c, c // g, 3, 3, device=DEV) / (9 * c // g) ** 0.5)
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        wr = w.half().float().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride=s_, padding=1, groups=g)
        gy = torch.randn(yr.shape[0], yr.shape[2], yr.shape[3], c, device=DEV).half()
        yr.backward(gy.float().permute(0, 3, 1, 2))
        ref_y, ref_dx, ref_dw = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad
        tols = (0.02 * float(ref_y.abs().max()), 0.02 * float(ref_dx.abs().max()), 3e-3 * float(ref_dw.abs().max()))
        nbad, worst, first = 0, 0.0, None
        for rep in range(a.reps):
            xg = x.clone().requires_grad_(True)
            wg = w.clone().requires_grad_(True)
            y = ops.conv2d(xg, wg, None, (s_, s_), (1, 1), groups=g)
            y.backward(gy)
            for got, ref, tol in ((y.detach(), ref_y, tols[0]), (xg.grad, ref_dx, tols[1]), (wg.grad, ref_dw, tols[2])):
                d = (got.float() - ref).abs()
                nbad += int((d > tol).sum())
                worst = max(worst, float(d.max()) / tol)
            if first is None:
                first = (y.detach().clone(), xg.grad.clone())
            else:
                nbad += int((first[0] != y.detach()).sum()) + int((first[1] != xg.grad).sum())
        print(f"{'depthwise' if g == c else 'grouped  '} 3x3 {hw:3d} {c:5d} g{g} s{s_}: {'ok' if nbad == 0 else 'BAD ' + str(nbad)}({worst:.2f})", flush=True)
        bad_total += nbad
        del x, xr, yr, gy, ref_y, ref_dx
        torch.cuda.empty_cache()
    return bad_total


def stem_backward(a) -> int:
    """The stem's one-pass backward (BatchNorm backward apply + 7x7 weight gradient) at 224x224 against the separate
    kernels: dW of the conv and the BatchNorm gradients, several launches (float atomics on dW: equal to rounding)."""
    from pytorch_distributed_amd import nn as pnn, ops
    from pytorch_distributed_amd.ops import fused
    torch.manual_seed(9)
    n = a.batch
    conv = pnn.Conv2d(3, 64, 7, 2, 3, bias=False).to(DEV)
    bn = pnn.BatchNorm2d(64).to(DEV)
    img = torch.randn(n, 3, 224, 224, device=DEV)
    g = None

    def run(fuse_on):
        nonlocal g
        old = fused._FUSE_STEM_BWD
        fused._FUSE_STEM_BWD = fuse_on
        try:
            for p in (conv.weight, bn.weight, bn.bias):
                p.grad = None
            y = ops.stem_conv_bn_relu_pool(img, conv.weight, bn.weight, bn.bias, conv.stride, conv.padding, bn.state(),
                                           torch.float16, pool=(3, 2, 1))
            if g is None:
                g = torch.randn_like(y)
            y.backward(g)
            return [p.grad.float().clone() for p in (conv.weight, bn.weight, bn.bias)]
        finally:
            fused._FUSE_STEM_BWD = old

    ref = run(False)
    nbad, worst = 0, 0.0
    calls = []
    real = C.stem_bwd_fused
    C.stem_bwd_fused = lambda *args, **kw: (calls.append(1), real(*args, **kw))[1]
    for rep in range(a.reps):
        got = run(True)
        for r_, g_ in zip(ref, got):
            tol = 3e-3 * float(r_.abs().max())
            d = (r_ - g_).abs()
            nbad += int((d > tol).sum())
            worst = max(worst, float(d.max()) / tol)
    C.stem_bwd_fused = real
    if len(calls) != a.reps:
        print("stem backward: the one-pass kernel did not run")
        return 1
    print(f"stem backward in one pass at 224x224: {'ok' if nbad == 0 else 'BAD ' + str(nbad)}({worst:.2f})", flush=True)
    return nbad


if __name__ == "__main__":
    sys.exit(main())
