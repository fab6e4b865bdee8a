"""TEST INFRASTRUCTURE ONLY — plain ``torch.nn`` CPU fp32 restatement of
``monai.networks.nets.PatchDiscriminator`` and ``monai.losses.PatchAdversarialLoss`` (MONAI 1.5.1, pinned in the
reference's ``uv.lock``) as the reference builds and calls them:

* ``vae_scripts/train_vae.py:266-279``: ``PatchDiscriminator(spatial_dims=2, num_layers_d=3, channels=32, in_channels=1,
  out_channels=1, norm="INSTANCE")``;
* ``:298``: ``PatchAdversarialLoss(criterion="least_squares")``;
* ``:399-401`` generator term, ``:447-458`` discriminator step.

MONAI is not vendored in /root/reference and is not installed here, so this file restates its published structure:
``PatchDiscriminator`` is an ``nn.Sequential`` of ``Convolution`` blocks (kernel 4, padding 1) —
``initial_conv`` (stride 2, bias, LeakyReLU(0.2), no norm), ``"0" .. str(num_layers_d - 1)`` (channels doubling,
stride 2 except the last, no bias, ADN ordering "NDA": norm -> dropout(0) -> LeakyReLU(0.2)), ``final_conv`` (stride 1,
bias, ``conv_only``) — whose ``forward`` returns the list of every block's output; Conv weights are initialised
``normal(0, 0.02)``.  In each block the Conv2d is registered as ``conv`` and the norm/activation as ``adn``, so the
``state_dict`` keys are ``initial_conv.conv.{weight,bias}``, ``{0,1,2}.conv.weight``, ``final_conv.conv.{weight,bias}``
(InstanceNorm2d has no affine parameters / buffers by default).  ``PatchAdversarialLoss("least_squares")`` applies
``LeakyReLU(0.05)`` to the logits (``no_activation_leastsq=False``) and takes the MSE against a constant 1 (real) / 0
(fake) target; for the generator the target is always "real".

Parity: UNPINNED for this file (the reference holds no discriminator output and MONAI cannot be run here).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ADN(nn.Module):
    """MONAI ``ADN(ordering="NDA")`` with dropout 0: InstanceNorm2d (``N``), then LeakyReLU (``A``)."""

    def __init__(self, channels: int | None, slope: float):
        super().__init__()
        if channels is not None:
            self.N = nn.InstanceNorm2d(channels)          # affine=False, eps=1e-5, no running statistics
        self.A = nn.LeakyReLU(negative_slope=slope)

    def forward(self, x):
        if hasattr(self, "N"):
            x = self.N(x)
        return self.A(x)


class Convolution(nn.Module):
    def __init__(self, cin, cout, stride, bias, norm, conv_only=False, slope=0.2):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 4, stride=stride, padding=1, bias=bias)
        if not conv_only:
            self.adn = _ADN(cout if norm else None, slope)

    def forward(self, x):
        x = self.conv(x)
        return self.adn(x) if hasattr(self, "adn") else x


class PatchDiscriminator(nn.Sequential):
    def __init__(self, spatial_dims: int = 2, num_layers_d: int = 3, channels: int = 32, in_channels: int = 1,
                 out_channels: int = 1, norm: str = "INSTANCE"):
        super().__init__()
        if spatial_dims != 2 or norm != "INSTANCE":
            raise ValueError("oracle PatchDiscriminator: 2-D, norm='INSTANCE' only (what the reference builds)")
        self.num_layers_d, self.num_channels = num_layers_d, channels
        self.add_module("initial_conv", Convolution(in_channels, channels, 2, True, False))
        cin, cout = channels, channels * 2
        for l_ in range(num_layers_d):
            self.add_module(str(l_), Convolution(cin, cout, 1 if l_ == num_layers_d - 1 else 2, False, True))
            cin, cout = cout, cout * 2
        self.add_module("final_conv", Convolution(cin, out_channels, 1, True, False, conv_only=True))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight.data, 0.0, 0.02)

    def forward(self, x):
        out = [x]
        for block in self.children():
            out.append(block(out[-1]))
        return out[1:]


def patch_adversarial_loss(logits: torch.Tensor, target_is_real: bool, for_discriminator: bool, slope: float = 0.05):
    """``PatchAdversarialLoss(criterion="least_squares")(logits, target_is_real, for_discriminator)`` for one
    discriminator output: mean((LeakyReLU_0.05(logits) - target)^2), target 1 for real / 0 for fake; the generator
    (``for_discriminator=False``) always aims at "real"."""
    if not for_discriminator:
        target_is_real = True
    a = F.leaky_relu(logits, slope)
    return F.mse_loss(a, torch.full_like(a, 1.0 if target_is_real else 0.0))
