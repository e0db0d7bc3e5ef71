"""The 2-conv MNIST CNN of the Horovod example
(/root/reference/5.2.horovod_pytorch_mnist.py:36-52): conv(1->10,k5) -> maxpool2 ->
relu -> conv(10->20,k5) -> Dropout2d -> maxpool2 -> relu -> flatten(320) -> fc(50) ->
relu -> dropout -> fc(10) -> log_softmax.  21 840 parameters; state_dict keys
``conv1/conv2/fc1/fc2 .weight/.bias`` as upstream.

Activations run NHWC; the flatten re-orders to the upstream (C,H,W) feature order
so ``fc1.weight`` means the same thing as in the reference model.
"""
from __future__ import annotations

import torch
import torch.nn as tnn

from .. import nn as pnn
from .. import ops


class Net(tnn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = pnn.Conv2d(1, 10, kernel_size=5)
        self.conv2 = pnn.Conv2d(10, 20, kernel_size=5)
        self.conv2_drop = pnn.Dropout2d()
        self.fc1 = pnn.Linear(320, 50)
        self.fc2 = pnn.Linear(50, 10)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.permute(0, 2, 3, 1).contiguous()            # NCHW loader batch -> NHWC
        x = ops.relu(ops.max_pool2d(self.conv1(x), 2))
        x = ops.relu(ops.max_pool2d(self.conv2_drop(self.conv2(x)), 2))
        x = x.permute(0, 3, 1, 2).reshape(-1, 320)        # upstream's view(-1, 320) order
        x = ops.relu(self.fc1(x))
        x = ops.dropout(x, 0.5, self.training)
        x = self.fc2(x)
        return ops.log_softmax(x)


def mnist_net(pretrained: bool = False) -> Net:
    return Net()
