"""Datasets.  There is no network in this environment, so the reference's
``datasets.CIFAR10(download=True)`` / ``datasets.MNIST(download=True)``
(/root/reference/2.distributed.py:137, /root/reference/5.2.horovod_pytorch_mnist.py:135) become:

* readers for the standard on-disk formats if the files are already there: CIFAR-10 from the binary version
  ``cifar-10-batches-bin`` or from the ``cifar-10-batches-py`` directory that the reference's download leaves
  behind (python pickles, read with an unpickler that admits numpy array reconstruction and nothing else, so a
  crafted file cannot run code); MNIST from the idx-ubyte files, and
* synthetic datasets of the same shapes (the benchmark configurations are synthetic).
"""
from __future__ import annotations

import gzip
import os
import pickle
import struct
from typing import Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

CIFAR10_MEAN, CIFAR10_STD = (0.4915, 0.4823, 0.4468), (0.2470, 0.2435, 0.2616)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
MNIST_MEAN, MNIST_STD = (0.1307,), (0.3081,)


class SyntheticImages(Dataset):
    """Deterministic random images + labels; ``uint8=True`` yields raw bytes for the on-device normaliser."""

    def __init__(self, num_samples: int, shape: Tuple[int, int, int], num_classes: int, seed: int = 0,
                 uint8: bool = False, materialize: bool = True):
        self.n, self.shape, self.num_classes, self.seed, self.uint8 = num_samples, tuple(shape), num_classes, seed, uint8
        g = torch.Generator().manual_seed(seed)
        self.labels = torch.randint(0, num_classes, (num_samples,), generator=g)
        pool = min(num_samples, 1024) if materialize else 0
        self.pool = None
        if pool:
            if uint8:
                self.pool = torch.randint(0, 256, (pool, *self.shape), generator=g, dtype=torch.uint8)
            else:
                self.pool = torch.randn((pool, *self.shape), generator=g)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if self.pool is not None:
            return self.pool[i % self.pool.shape[0]], int(self.labels[i])
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        x = torch.randint(0, 256, self.shape, generator=g, dtype=torch.uint8) if self.uint8 \
            else torch.randn(self.shape, generator=g)
        return x, int(self.labels[i])


def synthetic_cifar10(train: bool = True, **kw):
    return SyntheticImages(50000 if train else 10000, (3, 32, 32), 10, seed=1 if train else 2, **kw)


def synthetic_imagenet(train: bool = True, num_samples: Optional[int] = None, **kw):
    return SyntheticImages(num_samples or (1281167 if train else 50000), (3, 224, 224), 1000, seed=3 if train else 4,
                           **kw)


def synthetic_mnist(train: bool = True, **kw):
    return SyntheticImages(60000 if train else 10000, (1, 28, 28), 10, seed=5 if train else 6, **kw)


class CIFAR10Bin(Dataset):
    """CIFAR-10 binary version (data_batch_[1-5].bin / test_batch.bin: 1 label byte + 3072 pixel bytes)."""

    def __init__(self, root: str, train: bool = True, normalize: bool = True):
        d = os.path.join(root, "cifar-10-batches-bin")
        files = [f"data_batch_{i}.bin" for i in range(1, 6)] if train else ["test_batch.bin"]
        recs = [np.fromfile(os.path.join(d, f), dtype=np.uint8).reshape(-1, 3073) for f in files]
        a = np.concatenate(recs)
        self.labels = torch.from_numpy(a[:, 0].astype(np.int64))
        self.images = torch.from_numpy(a[:, 1:].reshape(-1, 3, 32, 32).copy())
        self.normalize = normalize
        self.mean = torch.tensor(CIFAR10_MEAN).view(3, 1, 1)
        self.std = torch.tensor(CIFAR10_STD).view(3, 1, 1)

    @staticmethod
    def available(root: str) -> bool:
        return os.path.exists(os.path.join(root, "cifar-10-batches-bin", "data_batch_1.bin"))

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        x = self.images[i]
        if self.normalize:
            x = (x.float() / 255.0 - self.mean) / self.std
        return x, int(self.labels[i])


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Unpickler for the CIFAR python batches: dicts of bytes / lists / numpy arrays.  Every global a numpy array
    pickle refers to is listed here; anything else in the stream is refused instead of imported."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError(f"cifar-10-batches-py: refusing to load global {module}.{name}")
        return super().find_class(module, name)


class CIFAR10Py(CIFAR10Bin):
    """CIFAR-10 python version (``cifar-10-batches-py/data_batch_[1-5]``, ``test_batch``): what
    ``torchvision.datasets.CIFAR10(download=True)`` (/root/reference/2.distributed.py:137) leaves under ``root``."""

    def __init__(self, root: str, train: bool = True, normalize: bool = True):
        d = os.path.join(root, "cifar-10-batches-py")
        files = [f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]
        images, labels = [], []
        for f in files:
            with open(os.path.join(d, f), "rb") as fh:
                entry = _ArraysOnlyUnpickler(fh, encoding="latin1").load()
            images.append(np.asarray(entry["data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
            labels.extend(entry["labels"] if "labels" in entry else entry["fine_labels"])
        self.images = torch.from_numpy(np.concatenate(images).copy())
        self.labels = torch.tensor([int(v) for v in labels], dtype=torch.int64)
        self.normalize = normalize
        self.mean = torch.tensor(CIFAR10_MEAN).view(3, 1, 1)
        self.std = torch.tensor(CIFAR10_STD).view(3, 1, 1)

    @staticmethod
    def available(root: str) -> bool:
        return os.path.exists(os.path.join(root, "cifar-10-batches-py", "data_batch_1"))


class MNISTIdx(Dataset):
    def __init__(self, root: str, train: bool = True, normalize: bool = True):
        pre = "train" if train else "t10k"
        self.images = torch.from_numpy(self._read(root, f"{pre}-images-idx3-ubyte", 16).reshape(-1, 1, 28, 28).copy())
        self.labels = torch.from_numpy(self._read(root, f"{pre}-labels-idx1-ubyte", 8).astype(np.int64))
        self.normalize = normalize

    @staticmethod
    def _path(root, name):
        for cand in (os.path.join(root, name), os.path.join(root, name + ".gz"),
                     os.path.join(root, "MNIST", "raw", name), os.path.join(root, "MNIST", "raw", name + ".gz")):
            if os.path.exists(cand):
                return cand
        return None

    @classmethod
    def available(cls, root: str) -> bool:
        return cls._path(root, "train-images-idx3-ubyte") is not None

    @classmethod
    def _read(cls, root, name, header):
        p = cls._path(root, name)
        if p is None:
            raise FileNotFoundError(name)
        op = gzip.open if p.endswith(".gz") else open
        with op(p, "rb") as f:
            raw = f.read()
        struct.unpack(">I", raw[:4])
        return np.frombuffer(raw, dtype=np.uint8, offset=header)

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        x = self.images[i]
        if self.normalize:
            x = (x.float() / 255.0 - MNIST_MEAN[0]) / MNIST_STD[0]
        return x, int(self.labels[i])


def cifar10(root: str, train: bool = True, synthetic: Optional[bool] = None):
    """Real CIFAR-10 if its files (binary or python version) are under ``root``, else the synthetic stand-in."""
    if root and CIFAR10Bin.available(root) and synthetic is not True:
        return CIFAR10Bin(root, train)
    if root and CIFAR10Py.available(root) and synthetic is not True:
        return CIFAR10Py(root, train)
    if synthetic is False:
        return CIFAR10Bin(root, train)      # asked for the real data: let the missing file be the error
    return synthetic_cifar10(train)


def mnist(root: str, train: bool = True, synthetic: Optional[bool] = None):
    if synthetic is False or (synthetic is None and root and MNISTIdx.available(root)):
        return MNISTIdx(root, train)
    return synthetic_mnist(train)
