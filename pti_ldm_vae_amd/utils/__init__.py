from .config import load_vae_config, parse_config, read_config, resolve_ar_settings, resolve_bool
from .distributed import setup_ddp
from .losses import ensure_three_channels

__all__ = ["ensure_three_channels", "load_vae_config", "parse_config", "read_config", "resolve_ar_settings",
           "resolve_bool", "setup_ddp"]
