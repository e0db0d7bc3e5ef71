"""Dynamic loss scaler whose state lives on the device.

apex semantics (reached in the reference through ``amp.initialize`` /
``amp.scale_loss``, /root/reference/4.apex_distributed2.py:177,289-291): initial scale
2**16, halve on overflow and skip that optimizer step, double after 2000
consecutive clean steps, cap at 2**24.  Here {scale, growth_tracker, skipped_steps,
last_overflow} is a 4-float device tensor and the overflow flag an int32 device
word: the gradient check writes the flag, the fused optimizer reads it, and
``after_step`` launches the one-thread update kernel - the training loop never
synchronises with the device to take the skip decision.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from ..ops import multi_tensor as mt


class LossScaler:
    def __init__(self, loss_scale: Union[str, float] = "dynamic", init_scale: float = 2.0 ** 16,
                 growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 min_loss_scale: Optional[float] = None, max_loss_scale: float = 2.0 ** 24, device=None,
                 enabled: bool = True):
        self.dynamic = loss_scale == "dynamic"
        self.enabled = enabled
        init = init_scale if self.dynamic else float(loss_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.min_loss_scale = min_loss_scale if min_loss_scale is not None else 1.0
        self.max_loss_scale = max_loss_scale
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.state = torch.tensor([init, 0.0, 0.0, 0.0], dtype=torch.float32, device=self.device)
        self.overflow_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.grads_scaled = False   # True between scale_loss exit and the optimizer step when unscale is deferred
        self._warned = False

    def to(self, device) -> "LossScaler":
        self.device = torch.device(device)
        self.state = self.state.to(self.device)
        self.overflow_flag = self.overflow_flag.to(self.device)
        return self

    @property
    def scale_tensor(self) -> torch.Tensor:
        return self.state[0:1]

    def loss_scale(self) -> float:
        """Current scale (synchronises with the device: for logging/tests, not for the hot loop)."""
        return float(self.state[0].item())

    def skipped_steps(self) -> int:
        return int(self.state[2].item())

    def last_step_overflowed(self) -> bool:
        return bool(self.state[3].item() != 0)

    def after_step(self) -> None:
        """Consume the overflow flag: update the scale and clear the flag."""
        if self.dynamic:
            mt.scaler_update(self.state, self.overflow_flag, self.growth_factor, self.backoff_factor,
                             self.growth_interval, self.min_loss_scale, self.max_loss_scale)
        else:
            mt.scaler_update(self.state, self.overflow_flag, 1.0, 1.0, 1 << 30, 0.0, float("inf"))
        self.grads_scaled = False

    def state_dict(self) -> dict:
        s = self.state.tolist()
        return {"loss_scale": s[0], "unskipped": int(s[1]), "skipped_steps": int(s[2]), "dynamic": self.dynamic}

    def load_state_dict(self, sd: dict) -> None:
        self.state.copy_(torch.tensor([float(sd["loss_scale"]), float(sd.get("unskipped", 0)),
                                       float(sd.get("skipped_steps", 0)), 0.0]))
        self.dynamic = bool(sd.get("dynamic", self.dynamic))
