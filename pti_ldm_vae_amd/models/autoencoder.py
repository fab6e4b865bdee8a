"""``VAEModel`` — the drop-in boundary of this framework.

Mirrors ``pti_ldm_vae.models.VAEModel`` of the reference (``src/pti_ldm_vae/models/autoencoder.py:6-171``):
same constructor, ``from_config`` defaults, methods and un-prefixed ``state_dict`` keys, so every
reference caller (``vae_scripts/train_vae.py:264,385,555``, ``regression_head.py:129,165``,
``utils/vae_loader.py:38-42``) can use it unchanged.  The inner network is this package's HIP
``AutoencoderKL`` instead of MONAI's.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .autoencoderkl import AutoencoderKL


class VAEModel(nn.Module):
    """Variational autoencoder wrapper (reference autoencoder.py:48-79).

    ``forward(x) -> (reconstruction, z_mu, z_third)`` where ``z_third`` is what MONAI's
    ``AutoencoderKL.forward`` returns there: sigma.  Pass ``third_output="logvar"``
    to get the clamped log-variance instead (what the reference's docstrings assume).
    """

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, latent_channels: int,
                 channels: list[int], num_res_blocks: int = 2, norm_num_groups: int = 32, norm_eps: float = 1e-6,
                 attention_levels: list[bool] | None = None, with_encoder_nonlocal_attn: bool = True,
                 with_decoder_nonlocal_attn: bool = True, third_output: str = "sigma") -> None:
        super().__init__()
        if attention_levels is None:
            attention_levels = [False] * len(channels)
        self.autoencoder = AutoencoderKL(
            spatial_dims=spatial_dims, in_channels=in_channels, out_channels=out_channels,
            latent_channels=latent_channels, channels=channels, num_res_blocks=num_res_blocks,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, attention_levels=attention_levels,
            with_encoder_nonlocal_attn=with_encoder_nonlocal_attn,
            with_decoder_nonlocal_attn=with_decoder_nonlocal_attn, third_output=third_output)

    @classmethod
    def from_config(cls, config: dict) -> "VAEModel":
        """Reference autoencoder.py:81-103 (same keys, same ``.get`` defaults)."""
        return cls(
            spatial_dims=config["spatial_dims"], in_channels=config["in_channels"],
            out_channels=config["out_channels"], latent_channels=config["latent_channels"],
            channels=config["channels"], num_res_blocks=config.get("num_res_blocks", 2),
            norm_num_groups=config.get("norm_num_groups", 32), norm_eps=config.get("norm_eps", 1e-6),
            attention_levels=config.get("attention_levels"),
            with_encoder_nonlocal_attn=config.get("with_encoder_nonlocal_attn", True),
            with_decoder_nonlocal_attn=config.get("with_decoder_nonlocal_attn", True),
            third_output=config.get("third_output", "sigma"))

    def forward(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return self.autoencoder(x)

    def encode_stage_2_inputs(self, x: torch.Tensor) -> torch.Tensor:
        return self.autoencoder.encode_stage_2_inputs(x)

    def encode_deterministic(self, x: torch.Tensor) -> torch.Tensor:
        z_mu, _ = self.autoencoder.encode(x)
        return z_mu

    def decode_stage_2_outputs(self, z: torch.Tensor) -> torch.Tensor:
        return self.autoencoder.decode_stage_2_outputs(z)

    def reconstruct_deterministic(self, x: torch.Tensor) -> torch.Tensor:
        return self.decode_stage_2_outputs(self.encode_deterministic(x))

    def mark_weights_dirty(self) -> None:
        """Not in the reference: tell the HIP engine to re-pack its 16-bit weight operands after a write the parameters'
        version counters cannot see (``p.data.copy_()``, direct writes into the flat arena); see
        ``AutoencoderKL.mark_weights_dirty``."""
        self.autoencoder.mark_weights_dirty()

    # reference autoencoder.py:165-171 delegates to the inner net => un-prefixed keys.  The extra
    # (ignored-by-the-reference) arguments are accepted so parents' recursive state_dict() works.
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        return self.autoencoder.load_state_dict(state_dict, strict=strict)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if destination is not None or prefix:
            return self.autoencoder.state_dict(destination=destination, prefix=prefix + "autoencoder.",
                                               keep_vars=keep_vars)
        return self.autoencoder.state_dict(keep_vars=keep_vars)
