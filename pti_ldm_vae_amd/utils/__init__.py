from .config import load_vae_config, parse_config, read_config, resolve_ar_settings, resolve_bool
from .distributed import setup_ddp
from .losses import ensure_three_channels
from .metrics import compute_regression_metrics

__all__ = ["compute_regression_metrics", "ensure_three_channels", "load_vae_config", "parse_config", "read_config", "resolve_ar_settings",
           "resolve_bool", "setup_ddp"]
