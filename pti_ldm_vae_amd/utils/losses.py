"""Reference ``src/pti_ldm_vae/utils/losses.py:8-28``: 1 -> 3 channel repeat in front of a perceptual net."""
from __future__ import annotations

import torch


def ensure_three_channels(tensor: torch.Tensor) -> torch.Tensor:
    if tensor.ndim != 4:
        raise ValueError(f"Expected 4D tensor (B, C, H, W), got shape {tensor.shape}")
    c = tensor.shape[1]
    if c == 3:
        return tensor
    if c == 1:
        return tensor.repeat(1, 3, 1, 1)
    raise ValueError(f"Perceptual loss expects 1 or 3 channels, got {c}")
