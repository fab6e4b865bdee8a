"""Regression head on frozen-encoder latents (reference ``src/pti_ldm_vae/models/regression_head.py``:
``LatentRegressor`` :30-78, ``VAELatentRegressor`` :81-169).  The encoder runs on the HIP engine under
``no_grad`` (encoder-only inference, BASELINE config 5); the MLP is three tiny ``nn.Linear`` layers."""
from __future__ import annotations

import warnings
from collections.abc import Iterable, Sequence

import torch
from torch import nn

from .autoencoder import VAEModel

_ACT = {"relu": nn.ReLU, "gelu": nn.GELU, "leaky_relu": nn.LeakyReLU, "elu": nn.ELU}


class LatentRegressor(nn.Module):
    def __init__(self, in_features: int, hidden_dims: Sequence[int], output_dim: int, dropout: float = 0.0,
                 activation: str = "relu") -> None:
        super().__init__()
        if in_features <= 0:
            raise ValueError("in_features must be positive.")
        if output_dim <= 0:
            raise ValueError("output_dim must be positive.")
        if activation not in _ACT:
            raise ValueError(f"Unsupported activation: {activation}. Choose from {', '.join(_ACT)}.")
        dims = [in_features, *hidden_dims, output_dim]
        layers: list[nn.Module] = []
        for i in range(len(dims) - 2):
            layers += [nn.Linear(dims[i], dims[i + 1]), _ACT[activation]()]
            if dropout > 0:
                layers.append(nn.Dropout(p=dropout))
        layers.append(nn.Linear(dims[-2], dims[-1]))
        self.mlp = nn.Sequential(*layers)

    def forward(self, latent_flat: torch.Tensor) -> torch.Tensor:
        return self.mlp(latent_flat)


class VAELatentRegressor(nn.Module):
    def __init__(self, vae: VAEModel, regressor: LatentRegressor, *, latent_dim: int,
                 flatten_warning_threshold: int = 131072) -> None:
        super().__init__()
        self.vae, self.regressor, self.latent_dim = vae, regressor, latent_dim
        first = next((m for m in regressor.mlp if isinstance(m, nn.Linear)), None)
        if first is None or first.in_features != latent_dim:
            raise ValueError(f"Regression head expects in_features={latent_dim}, "
                             f"got {first.in_features if first else 'unknown'}.")
        for p in self.vae.parameters():
            p.requires_grad = False
        self.vae.eval()
        self.flatten_warning_threshold = flatten_warning_threshold

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            latent = self.vae.encode_deterministic(images)
        flat = torch.flatten(latent, start_dim=1)
        if flat.shape[1] > self.flatten_warning_threshold:
            warnings.warn(f"Flattened latent dimension {flat.shape[1]} is large; consider reducing patch size or "
                          "latent channels.", stacklevel=2)
        return self.regressor(flat)

    @staticmethod
    def compute_flat_dim(latent: torch.Tensor) -> int:
        return int(torch.flatten(latent, start_dim=1).shape[1])

    @staticmethod
    def infer_flat_dim_from_patch(vae: VAEModel, patch_size: Iterable[int], device: torch.device, *,
                                  channels: int | None = None) -> int:
        h, w = patch_size
        c = channels if channels is not None else getattr(vae.autoencoder, "in_channels", 1)
        with torch.no_grad():
            latent = vae.encode_deterministic(torch.zeros(1, c, h, w, device=device))
        return VAELatentRegressor.compute_flat_dim(latent)
