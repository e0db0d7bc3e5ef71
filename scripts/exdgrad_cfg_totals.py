"""Per-candidate time of the input-gradient kernels WITH the fused epilogue (residual + recomputed ReLU mask +
BatchNorm-backward sums) summed over the ResNet-50 layers that use it (batch 400, fp16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd.ops import conv as C

DEV = "cuda"
B = int(os.environ.get("BATCH", "400"))
# (hw of dx, dx channels, dy channels, k, stride, multiplicity): conv1 of every block (1x1, dx = block input) and the 3x3 convs
LAYERS = [(56, 256, 64, 1, 1, 2), (56, 256, 128, 1, 1, 1), (28, 512, 128, 1, 1, 3), (28, 512, 256, 1, 1, 1), (14, 1024, 256, 1, 1, 5),
          (14, 1024, 512, 1, 1, 1), (7, 2048, 512, 1, 1, 2), (56, 128, 128, 3, 2, 1), (28, 128, 128, 3, 1, 3), (28, 256, 256, 3, 2, 1),
          (14, 256, 256, 3, 1, 5), (14, 512, 512, 3, 2, 1), (7, 512, 512, 3, 1, 2)]
CFGS = (0, 1, 2, 3, 9, 10, 11, 5, 6)
tot = {c: 0.0 for c in CFGS}
best_tot = 0.0
for (hw, ci, co, k, s, mult) in LAYERS:
    pad = k // 2
    ho = (hw + 2 * pad - k) // s + 1
    dy = torch.randn(B, ho, ho, co, device=DEV).half()
    w = torch.randn(co, k, k, ci, device=DEV).half()
    wt = C.krsc_to_crsk(w, torch.float16)
    res = torch.randn(B, hw, hw, ci, device=DEV).half() if k == 1 else None
    bn_in = torch.randn(B, hw, hw, ci, device=DEV).half()
    sc, sf = torch.rand(ci, device=DEV) + 0.5, torch.randn(ci, device=DEV) * 0.3
    t = {}
    for c in CFGS:
        fn = lambda: C.dgrad_gpu(dy, wt, (hw, hw), (s, s), (pad, pad), res=res, cfg=c or (1 << 30), bn_input=bn_in, bn_scale=sc,
                                 bn_shift=sf, bn_sums=True)
        if fn() is None:
            t[c] = float("inf")
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t[c] = e0.elapsed_time(e1) / 5 * 1000
    best_tot += min(t.values()) * mult
    for c in CFGS:
        tot[c] += (t[c] if t[c] != float("inf") else t[0]) * mult
    print(f"{hw:3d} dx {ci:5d} <- dy {co:5d} k{k} s{s} x{mult}: " + " ".join(f"{c}:{t[c]:.0f}" for c in CFGS), flush=True)
    del dy, w, wt, res, bn_in
    torch.cuda.empty_cache()
print(f"per-step totals (ms): best per layer {best_tot / 1000:.2f} | " + " ".join(f"{c}:{tot[c] / 1000:.2f}" for c in CFGS))
