"""Image-directory datasets and the ImageNet augmentation pipeline.

What the reference's multi-node script builds from torchvision
(/root/reference/6.distributed_slurm_main.py:126-159): ``datasets.ImageFolder(traindir,
Compose([RandomResizedCrop(224), RandomHorizontalFlip(), ToTensor(), normalize]))`` for training and
``ImageFolder(valdir, Compose([Resize(256), CenterCrop(224), ToTensor(), normalize]))`` for validation.
torchvision is not part of this stack, so the directory scan, the PIL decode and the geometric transforms
live here.  One deliberate difference: by default the pipelines END IN RAW ``uint8`` CHW tensors
(``ToUint8Tensor``) and the float conversion + mean/std normalisation + NCHW->NHWC repack happen on the
GPU inside the model's stem kernel (``ResNet.set_input_normalization``) or in the side-stream
``DevicePrefetcher(normalize=...)`` - a quarter of the host->device bytes and no per-pixel float math on
the loader's CPU workers.  ``uint8=False`` gives the reference's normalised fp32 tensors instead.

Randomness comes from torch's global generator, which ``DataLoader`` seeds per worker from its base seed:
``torch.manual_seed(s)`` before iterating makes an epoch's crops and flips reproducible.
"""
from __future__ import annotations

import math
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from .datasets import IMAGENET_MEAN, IMAGENET_STD

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def pil_loader(path: str):
    from PIL import Image
    with open(path, "rb") as f:
        img = Image.open(f)
        return img.convert("RGB")


def find_classes(root: str) -> Tuple[List[str], dict]:
    classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
    if not classes:
        raise FileNotFoundError(f"no class directories under {root}")
    return classes, {c: i for i, c in enumerate(classes)}


def make_dataset(root: str, class_to_idx: dict, extensions: Sequence[str] = IMG_EXTENSIONS) -> List[Tuple[str, int]]:
    samples = []
    for cls in sorted(class_to_idx):
        d = os.path.join(root, cls)
        for dirpath, _, files in sorted(os.walk(d, followlinks=True)):
            for name in sorted(files):
                if name.lower().endswith(tuple(extensions)):
                    samples.append((os.path.join(dirpath, name), class_to_idx[cls]))
    return samples


class ImageFolder(Dataset):
    """``root/<class>/<image>`` -> ``(transform(image), class index)``; classes are the sorted sub-directory
    names, samples are sorted by path inside each class (torchvision's ordering, so the same directory
    gives the same index <-> sample map)."""

    def __init__(self, root: str, transform: Optional[Callable] = None, target_transform: Optional[Callable] = None,
                 loader: Callable = pil_loader, extensions: Sequence[str] = IMG_EXTENSIONS):
        self.root = root
        self.classes, self.class_to_idx = find_classes(root)
        self.samples = make_dataset(root, self.class_to_idx, extensions)
        if not self.samples:
            raise FileNotFoundError(f"found no image files under {root} (extensions: {', '.join(extensions)})")
        self.targets = [t for _, t in self.samples]
        self.imgs = self.samples
        self.transform, self.target_transform, self.loader = transform, target_transform, loader

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, i: int):
        path, target = self.samples[i]
        img = self.loader(path)
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return img, target


# ------------------------------------------------------------------------------------------------
# transforms (PIL image in, PIL image out, until one of the To*Tensor steps)
# ------------------------------------------------------------------------------------------------
def _rand() -> float:
    return float(torch.rand(()))


class Compose:
    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class RandomResizedCrop:
    """A crop of random area (``scale`` of the image) and aspect ratio (log-uniform in ``ratio``), resized to
    ``size`` x ``size``; after 10 failed draws the central crop with the nearest valid ratio."""

    def __init__(self, size: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.size, self.scale, self.ratio = int(size), scale, ratio

    def get_params(self, width: int, height: int):
        area = width * height
        log_r = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * (self.scale[0] + (self.scale[1] - self.scale[0]) * _rand())
            ar = math.exp(log_r[0] + (log_r[1] - log_r[0]) * _rand())
            w = int(round(math.sqrt(target * ar)))
            h = int(round(math.sqrt(target / ar)))
            if 0 < w <= width and 0 < h <= height:
                top = int(_rand() * (height - h + 1))
                left = int(_rand() * (width - w + 1))
                return min(top, height - h), min(left, width - w), h, w
        in_ratio = width / height
        if in_ratio < self.ratio[0]:
            w, h = width, int(round(width / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            h, w = height, int(round(height * self.ratio[1]))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, img):
        from PIL import Image
        top, left, h, w = self.get_params(*img.size)
        return img.resize((self.size, self.size), Image.BILINEAR, box=(left, top, left + w, top + h))


class RandomHorizontalFlip:
    def __init__(self, p: float = 0.5):
        self.p = p

    def __call__(self, img):
        from PIL import Image
        return img.transpose(Image.FLIP_LEFT_RIGHT) if _rand() < self.p else img


class Resize:
    """Shorter side -> ``size`` (aspect ratio kept), or an exact ``(h, w)``."""

    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        from PIL import Image
        if isinstance(self.size, int):
            w, h = img.size
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return img
            if w < h:
                ow, oh = self.size, int(self.size * h / w)
            else:
                oh, ow = self.size, int(self.size * w / h)
            return img.resize((ow, oh), Image.BILINEAR)
        return img.resize((self.size[1], self.size[0]), Image.BILINEAR)


class CenterCrop:
    def __init__(self, size: int):
        self.size = int(size)

    def __call__(self, img):
        w, h = img.size
        if w < self.size or h < self.size:       # pad small images with black, as torchvision does
            from PIL import Image
            canvas = Image.new(img.mode, (max(w, self.size), max(h, self.size)))
            canvas.paste(img, ((canvas.size[0] - w) // 2, (canvas.size[1] - h) // 2))
            img, (w, h) = canvas, canvas.size
        left, top = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
        return img.crop((left, top, left + self.size, top + self.size))


class ToUint8Tensor:
    """PIL image -> ``uint8`` tensor ``[C, H, W]`` with values 0..255 (normalise on the device)."""

    def __call__(self, img) -> torch.Tensor:
        a = np.asarray(img, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


class ToTensor:
    """PIL image -> ``float32`` tensor ``[C, H, W]`` in [0, 1]."""

    def __call__(self, img) -> torch.Tensor:
        return ToUint8Tensor()(img).float().div_(255.0)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        return (t - self.mean) / self.std


def imagenet_train_transform(size: int = 224, uint8: bool = True) -> Compose:
    """/root/reference/6.distributed_slurm_main.py:130-139."""
    tail = [ToUint8Tensor()] if uint8 else [ToTensor(), Normalize(IMAGENET_MEAN, IMAGENET_STD)]
    return Compose([RandomResizedCrop(size), RandomHorizontalFlip()] + tail)


def imagenet_val_transform(size: int = 224, resize: int = 256, uint8: bool = True) -> Compose:
    """/root/reference/6.distributed_slurm_main.py:148-156."""
    tail = [ToUint8Tensor()] if uint8 else [ToTensor(), Normalize(IMAGENET_MEAN, IMAGENET_STD)]
    return Compose([Resize(resize), CenterCrop(size)] + tail)


def imagenet_available(root: str) -> bool:
    return bool(root) and os.path.isdir(os.path.join(root, "train")) and os.path.isdir(os.path.join(root, "val"))


def imagenet(root: str, train: bool = True, size: int = 224, uint8: bool = True) -> ImageFolder:
    """``root/train`` or ``root/val`` with the matching pipeline (``traindir`` / ``valdir`` of the reference,
    /root/reference/6.distributed_slurm_main.py:126-127)."""
    if train:
        return ImageFolder(os.path.join(root, "train"), imagenet_train_transform(size, uint8))
    return ImageFolder(os.path.join(root, "val"), imagenet_val_transform(size, int(round(size * 256 / 224)), uint8))
