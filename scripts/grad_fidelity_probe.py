"""How close do the fp16 O2 gradients of a random-init ResNet-50 (224x224, batch 64) come to an fp32 oracle, for this
package under its fusion switches and for stock autocast?  Cosine per layer, three seeds.  (Set the PDA_* switches in the
environment; the label is printed with the numbers.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pytorch_distributed_amd import amp, backends, models, ops
from pytorch_distributed_amd.optim import FusedSGD

DEV = "cuda"
NAMES = ["fc.weight", "layer4.2.conv3.weight", "layer3.5.conv2.weight", "layer2.0.conv1.weight", "layer1.0.conv1.weight",
         "conv1.weight"]


def cos(a, b):
    a, b = a.detach().float().flatten(), b.detach().float().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def main():
    label = os.environ.get("PROBE_LABEL", "default")
    backends.benchmark = os.environ.get("PROBE_TUNE", "1") == "1"
    backends.deterministic = os.environ.get("PROBE_DET", "0") == "1"
    for seed in (64, 65, 66):
        torch.manual_seed(seed)
        m = models.get_model("resnet50").to(DEV)
        ref = models.torch_resnet("resnet50").to(DEV)
        ref.load_state_dict(m.state_dict())
        opt = FusedSGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        m, opt = amp.initialize(m, opt, opt_level="O2", loss_scale=1024.0, verbosity=0)
        nb = int(os.environ.get("PROBE_BATCH", "64"))
        x = torch.randn(nb, 3, 224, 224, device=DEV)
        t = torch.randint(0, 1000, (nb,), device=DEV)
        for _ in range(2):
            opt.zero_grad(set_to_none=False)
            loss = ops.cross_entropy(m(x), t)
            with amp.scale_loss(loss, opt) as scaled:
                scaled.backward()
        ref.train()
        (F.cross_entropy(ref(x), t) * 1024.0).backward()
        g, gr = dict(m.named_parameters()), dict(ref.named_parameters())
        ours = [cos(g[k].grad, gr[k].grad) for k in NAMES]
        orc = {k: gr[k].grad.detach().clone() for k in NAMES}
        ref.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            la = F.cross_entropy(ref(x), t)
        (la * 1024.0).backward()
        stock = [cos(gr[k].grad, orc[k]) for k in NAMES]
        print(f"{label:28s} seed {seed}  ours  " + " ".join(f"{v:.3f}" for v in ours), flush=True)
        print(f"{label:28s} seed {seed}  stock " + " ".join(f"{v:.3f}" for v in stock), flush=True)
        del m, ref, opt, g, gr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
