from .fused_sgd import FusedSGD

__all__ = ["FusedSGD"]
