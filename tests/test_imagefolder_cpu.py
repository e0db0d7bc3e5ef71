"""Image-directory input pipeline (reference: ImageFolder + RandomResizedCrop / RandomHorizontalFlip and
Resize(256) / CenterCrop(224) + Normalize, /root/reference/6.distributed_slurm_main.py:126-159).
torchvision is not installed here, so there is no side-by-side oracle: the tests pin the documented
semantics (class indexing by sorted directory name, output shapes and value ranges, exact centre crop,
determinism under a seed, uint8 hand-off + device-side normalisation equal to ToTensor + Normalize)."""
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from pytorch_distributed_amd import data as D  # noqa: E402
from pytorch_distributed_amd.data import imagefolder as IF  # noqa: E402


def _make_tree(root, sizes=((64, 48), (300, 400), (500, 333), (224, 224))):
    rng = np.random.RandomState(0)
    for split in ("train", "val"):
        for cls in ("n02_dog", "n01_cat"):
            d = os.path.join(root, split, cls)
            os.makedirs(d)
            for i, (w, h) in enumerate(sizes):
                a = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
                Image.fromarray(a).save(os.path.join(d, f"img_{i}.png"))
        with open(os.path.join(root, split, "n02_dog", "notes.txt"), "w") as f:
            f.write("not an image")
    return root


def test_imagefolder_indexing_and_pipelines(tmp_path):
    root = _make_tree(str(tmp_path))
    assert D.imagenet_available(root)
    tr, va = D.imagenet(root, True), D.imagenet(root, False)
    assert tr.classes == ["n01_cat", "n02_dog"] and tr.class_to_idx == {"n01_cat": 0, "n02_dog": 1}
    assert len(tr) == 8 and len(va) == 8 and tr.targets == [0] * 4 + [1] * 4
    assert all(p.endswith(".png") for p, _ in tr.samples)
    for ds in (tr, va):
        for i in range(len(ds)):
            x, y = ds[i]
            assert x.dtype == torch.uint8 and x.shape == (3, 224, 224) and y in (0, 1)
    # float pipeline = the reference's tensors: normalised, finite, roughly zero-centred for uniform noise
    vf = D.imagenet(root, False, uint8=False)
    x, _ = vf[3]
    assert x.dtype == torch.float32 and x.shape == (3, 224, 224) and torch.isfinite(x).all()
    assert abs(float(x.mean())) < 0.5 and -3.0 < float(x.min()) and float(x.max()) < 3.0
    with pytest.raises(FileNotFoundError):
        D.ImageFolder(str(tmp_path / "train" / "n01_cat"))     # no class sub-directories


def test_val_transform_is_resize_then_exact_centre_crop(tmp_path):
    """A 224x224 image resized to 256 and centre-cropped: equals PIL's own resize + crop of the same box."""
    a = np.random.RandomState(1).randint(0, 256, (300, 400, 3), dtype=np.uint8)
    img = Image.fromarray(a)
    out = D.imagenet_val_transform(224, 256)(img)
    ref = img.resize((int(256 * 400 / 300), 256), Image.BILINEAR)
    left, top = int(round((ref.size[0] - 224) / 2.0)), 16
    ref = torch.from_numpy(np.asarray(ref.crop((left, top, left + 224, top + 224))).transpose(2, 0, 1).copy())
    assert torch.equal(out, ref)
    # an image that already has the target shorter side is not resampled
    b = Image.fromarray(a[:256, :256])
    assert torch.equal(D.Compose([D.Resize(256), D.ToUint8Tensor()])(b),
                       torch.from_numpy(a[:256, :256].transpose(2, 0, 1).copy()))


def test_train_transform_is_deterministic_under_seed_and_random_otherwise(tmp_path):
    root = _make_tree(str(tmp_path))
    tr = D.imagenet(root, True)
    torch.manual_seed(7)
    a = [tr[i][0] for i in range(4)]
    torch.manual_seed(7)
    b = [tr[i][0] for i in range(4)]
    c = [tr[i][0] for i in range(4)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert any(not torch.equal(x, y) for x, y in zip(a, c))
    # crop geometry stays inside the image and respects the scale / ratio bounds
    rrc = D.RandomResizedCrop(224)
    for _ in range(200):
        top, left, h, w = rrc.get_params(500, 333)
        assert 0 <= top and top + h <= 333 and 0 <= left and left + w <= 500 and h > 0 and w > 0
        assert 0.08 * 500 * 333 * 0.9 <= h * w <= 500 * 333 and 0.7 < w / h < 1.4
    flips = sum(float(torch.rand(())) < 0.5 for _ in range(1000))
    assert 400 < flips < 600
    flipped = D.RandomHorizontalFlip(1.0)(Image.fromarray(np.arange(12, dtype=np.uint8).reshape(2, 2, 3)))
    assert np.asarray(flipped)[0, 0, 0] == 3


def test_uint8_batches_normalised_downstream_match_float_pipeline(tmp_path):
    """loader ships bytes; DevicePrefetcher(normalize=..., channels_scale=255) and the ResNet stem's own
    normalisation both reproduce ToTensor + Normalize."""
    root = _make_tree(str(tmp_path))
    vu, vf = D.imagenet(root, False), D.imagenet(root, False, uint8=False)
    lu = D.make_loader(vu, 4, workers=0, pin_memory=False)
    lf = D.make_loader(vf, 4, workers=0, pin_memory=False)
    pre = D.DevicePrefetcher(lu, torch.device("cpu"), normalize=(D.IMAGENET_MEAN, D.IMAGENET_STD), channels_scale=255.0)
    for (xu, yu), (xf, yf) in zip(pre, lf):
        assert torch.equal(yu, yf) and xu.dtype == torch.float32
        assert torch.allclose(xu, xf, atol=1e-5)
    from pytorch_distributed_amd import models
    torch.manual_seed(0)
    m = models.get_model("resnet18", num_classes=4).eval()
    xb, _ = next(iter(lu))
    xfb, _ = next(iter(lf))
    ref = m(xfb)
    m.set_input_normalization(D.IMAGENET_MEAN, D.IMAGENET_STD, 1.0 / 255.0)
    out = m(xb)
    assert torch.allclose(out, ref, atol=1e-3, rtol=1e-3)


def test_engine_trains_from_an_image_directory_cpu(tmp_path):
    from pytorch_distributed_amd import engine
    root = _make_tree(str(tmp_path), sizes=((80, 64), (96, 120)))
    args = engine.build_parser().parse_args(["--dataset", "imagenet", "--data", root, "-a", "resnet18", "--epochs", "1",
                                             "-b", "4", "--num-classes", "2", "--cpu", "-j", "0", "-p", "1"])
    tr, va = engine.make_datasets(args)
    assert isinstance(tr, D.ImageFolder) and args.input_uint8[0] == D.IMAGENET_MEAN
    dev = torch.device("cpu")
    model = engine.make_model(args, dev)
    opt = engine.make_optimizer(args, model)
    model, opt = engine.setup_amp(args, model, opt)
    os.chdir(tmp_path)
    acc = engine.fit(args, model, opt, dev, tr, va, 4, distributed=False)
    assert 0.0 <= acc <= 100.0 and model._in_mean is not None and abs(model.input_scale - 1 / 255) < 1e-9
