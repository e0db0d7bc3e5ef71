"""Per-candidate forward / input-gradient time summed over ResNet-50's conv layers (batch 400, fp16): the best static
choice when the tuner is off, and how far it is from the per-layer best."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd.ops import conv as C

DEV = "cuda"
B = int(os.environ.get("BATCH", "400"))
LAYERS = [(56, 64, 64, 1, 1, 1), (56, 64, 64, 3, 1, 3), (56, 64, 256, 1, 1, 4), (56, 256, 64, 1, 1, 2), (56, 256, 128, 1, 1, 1),
          (56, 128, 128, 3, 2, 1), (28, 128, 512, 1, 1, 4), (56, 256, 512, 1, 2, 1), (28, 512, 128, 1, 1, 3), (28, 128, 128, 3, 1, 3),
          (28, 512, 256, 1, 1, 1), (28, 256, 256, 3, 2, 1), (14, 256, 1024, 1, 1, 6), (28, 512, 1024, 1, 2, 1), (14, 1024, 256, 1, 1, 5),
          (14, 256, 256, 3, 1, 5), (14, 1024, 512, 1, 1, 1), (14, 512, 512, 3, 2, 1), (7, 512, 2048, 1, 1, 3), (14, 1024, 2048, 1, 2, 1),
          (7, 2048, 512, 1, 1, 2), (7, 512, 512, 3, 1, 2)]
CFGS = (0, 1, 2, 9, 12, 5, 6, 7, 13)


def timeit(fn):
    if fn() is None:
        return float("inf")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1000


for kind in ("fprop", "dgrad"):
    tot = {c: 0.0 for c in CFGS}
    best_tot = 0.0
    for (hw, ci, co, k, s, mult) in LAYERS:
        pad = k // 2
        ho = (hw + 2 * pad - k) // s + 1
        x = torch.randn(B, hw, hw, ci, device=DEV).half()
        w = torch.randn(co, k, k, ci, device=DEV).half()
        dy = torch.randn(B, ho, ho, co, device=DEV).half()
        wt = C.krsc_to_crsk(w, torch.float16)
        t = {}
        for c in CFGS:
            cc = c or (1 << 30)
            if kind == "fprop":
                t[c] = timeit(lambda: C.fprop_gpu(x, w, (s, s), (pad, pad), stats=True, cfg=cc))
            else:
                t[c] = timeit(lambda: C.dgrad_gpu(dy, wt, (hw, hw), (s, s), (pad, pad), cfg=cc))
        best_tot += min(t.values()) * mult
        for c in CFGS:
            tot[c] += (t[c] if t[c] != float("inf") else t[0]) * mult
        print(f"{kind} {hw:3d} {ci:5d}->{co:5d} k{k} s{s} x{mult}: " + " ".join(f"{c}:{t[c]:.0f}" for c in CFGS), flush=True)
        del x, w, dy, wt
        torch.cuda.empty_cache()
    print(f"{kind} per-step totals (ms): best per layer {best_tot / 1000:.2f} | " + " ".join(f"{c}:{tot[c] / 1000:.2f}" for c in CFGS), flush=True)
