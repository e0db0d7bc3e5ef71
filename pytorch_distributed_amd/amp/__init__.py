"""Mixed precision with dynamic loss scaling (apex.amp-compatible surface)."""
from .frontend import initialize, scale_loss, state_dict, load_state_dict, master_params
from .scaler import LossScaler

__all__ = ["initialize", "scale_loss", "state_dict", "load_state_dict", "master_params", "LossScaler"]
