from .sampler import DistributedSampler
from .datasets import (SyntheticImages, synthetic_cifar10, synthetic_imagenet, synthetic_mnist, CIFAR10Bin, CIFAR10Py, MNISTIdx,
                       cifar10, mnist, CIFAR10_MEAN, CIFAR10_STD, IMAGENET_MEAN, IMAGENET_STD, MNIST_MEAN, MNIST_STD)
from .loader import make_loader, DevicePrefetcher, SyntheticDeviceLoader
from .imagefolder import (ImageFolder, Compose, RandomResizedCrop, RandomHorizontalFlip, Resize, CenterCrop,
                          ToTensor, ToUint8Tensor, Normalize, imagenet, imagenet_available,
                          imagenet_train_transform, imagenet_val_transform)

data_prefetcher = DevicePrefetcher  # the reference's class name

__all__ = ["DistributedSampler", "SyntheticImages", "synthetic_cifar10", "synthetic_imagenet", "synthetic_mnist",
           "CIFAR10Bin", "CIFAR10Py", "MNISTIdx", "cifar10", "mnist", "make_loader", "DevicePrefetcher", "data_prefetcher",
           "SyntheticDeviceLoader", "CIFAR10_MEAN", "CIFAR10_STD", "IMAGENET_MEAN", "IMAGENET_STD", "MNIST_MEAN",
           "MNIST_STD", "ImageFolder", "Compose", "RandomResizedCrop", "RandomHorizontalFlip", "Resize", "CenterCrop",
           "ToTensor", "ToUint8Tensor", "Normalize", "imagenet", "imagenet_available", "imagenet_train_transform",
           "imagenet_val_transform"]
