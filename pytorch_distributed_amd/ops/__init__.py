"""Device ops: hand-written gfx950 HIP kernels behind autograd functions, with a
stock-PyTorch path for CPU tensors (used by the CPU test-suite and the gloo
multi-process tests)."""
from . import _lib
from .conv import conv2d, WeightCache, bump_param_generation, make_weight, krsc
from .norm import batch_norm_act, BNState
from .fused import conv_bn_act, stem_conv_bn_relu_pool, input_to_nhwc
from .pool import max_pool2d, global_avg_pool, avg_pool2d, adaptive_avg_pool2d
from .loss import cross_entropy, log_softmax, nll_loss, new_metrics
from .elementwise import relu, relu6, dropout, dropout2d
from . import multi_tensor

kernels_available = _lib.available

__all__ = [
    "conv2d", "WeightCache", "bump_param_generation", "make_weight", "krsc", "batch_norm_act", "BNState",
    "conv_bn_act", "stem_conv_bn_relu_pool", "input_to_nhwc", "max_pool2d", "global_avg_pool", "cross_entropy",
    "log_softmax", "nll_loss", "new_metrics", "relu", "relu6", "avg_pool2d", "adaptive_avg_pool2d", "dropout", "dropout2d", "multi_tensor", "kernels_available",
]
