"""Race detector of last resort: with backends.deterministic every model's step should be bit-reproducible.  Runs each
architecture's forward + backward three times on the same batch (fp16 O2-style compute) and reports what differs."""
import argparse, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_distributed_amd import backends, models, ops
from pytorch_distributed_amd.ops import elementwise as E

DEFAULT = "resnet18,resnet50,wide_resnet50_2,resnext50_32x4d,vgg16_bn,alexnet,squeezenet1_1,densenet121,mobilenet_v2,shufflenet_v2_x1_0,mnasnet1_0,googlenet,inception_v3"


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--archs", default=DEFAULT)
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--det", type=int, default=1)
    a = p.parse_args()
    backends.deterministic = bool(a.det)
    dev = "cuda"
    for arch in a.archs.split(","):
        torch.manual_seed(0)
        m = models.get_model(arch).to(dev)
        if hasattr(m, "set_compute_dtype"):
            m.set_compute_dtype(torch.float16)
        size = 299 if arch == "inception_v3" else 224
        x = torch.randn(a.batch, 3, size, size, device=dev)
        t = torch.randint(0, 1000, (a.batch,), device=dev)
        runs = []
        state = {k: v.clone() for k, v in m.state_dict().items()}
        for rep in range(3):
            m.load_state_dict(state)          # BatchNorm running statistics back to the start
            m.zero_grad(set_to_none=True)
            torch.manual_seed(1234)            # the same dropout masks in every run: Philox key (seed, call counter)
            E._DROPOUT_CALLS = 0
            logits = m(x)
            (ops.cross_entropy(logits, t) * 128.0).backward()
            runs.append((logits.detach().float().clone(), [pp.grad.float().clone() for pp in m.parameters() if pp.grad is not None]))
        ldiff = max(float((runs[i][0] - runs[0][0]).abs().max()) for i in (1, 2))
        gdiff, gcount = 0.0, 0
        for i in (1, 2):
            for g0, g1 in zip(runs[0][1], runs[i][1]):
                d = float((g0 - g1).abs().max()) / (float(g0.abs().max()) + 1e-30)
                gdiff = max(gdiff, d)
                gcount += int(d > 0)
        nl = max(int((runs[i][0] != runs[0][0]).sum()) for i in (1, 2))
        print(f"{arch:20s} logits: {nl} of {runs[0][0].numel()} differ, max diff {ldiff:.3e}   grads: {gcount} of {2 * len(runs[0][1])} tensors differ, worst rel {gdiff:.3e}", flush=True)
        del m, runs, state
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
