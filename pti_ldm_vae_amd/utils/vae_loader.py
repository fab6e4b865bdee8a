"""Config + checkpoint -> eval-mode ``VAEModel`` (reference ``src/pti_ldm_vae/utils/vae_loader.py:27-43``)."""
from __future__ import annotations

from typing import Any

import torch

from ..models.autoencoder import VAEModel
from .config import load_vae_config  # noqa: F401  (re-exported like the reference module)


def load_vae_model(config: Any, checkpoint_path: str, device: torch.device) -> VAEModel:
    """Accepts a bare state-dict file or a training checkpoint holding ``autoencoder_state_dict``
    (vae_loader.py:39-41).  Files are read with ``weights_only=True``."""
    autoencoder = VAEModel.from_config(config.autoencoder_def).to(device)
    checkpoint = torch.load(checkpoint_path, map_location=device, weights_only=True)
    state_dict = checkpoint.get("autoencoder_state_dict", checkpoint)
    autoencoder.load_state_dict(state_dict)
    autoencoder.eval()
    return autoencoder
