"""TEST INFRASTRUCTURE ONLY — plain ``torch.nn`` CPU fp32 restatement of
``monai.networks.nets.AutoencoderKL`` (MONAI 1.5.1, pinned in the reference's
``uv.lock:859-860``) as constructed by ``src/pti_ldm_vae/models/autoencoder.py:67-79``.

MONAI is not vendored in /root/reference and is not installed here, so this
file restates its published structure.  Module
names are chosen so that ``state_dict()`` keys equal the MONAI key map
(Appendix A.3): a real reference checkpoint loads with ``strict=True``.

Parity: UNPINNED for this file (the reference has no tests and MONAI cannot be
run here).  Self-checks that do exist: parameter totals 4,562,593 (config A) and
12,324,885 (config AR) — see tests/test_oracle.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class Convolution(nn.Module):
    """``monai.networks.blocks.Convolution(conv_only=True)``: the Conv2d is
    registered under the name ``conv`` (Appendix A.1) => ``...conv.weight`` keys."""

    def __init__(self, cin: int, cout: int, k: int, stride: int = 1, padding: int = 0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)

    def forward(self, x):
        return self.conv(x)


class AEKLResBlock(nn.Module):
    """norm1 -> silu -> conv1 -> norm2 -> silu -> conv2, + nin_shortcut(x)."""

    def __init__(self, cin: int, cout: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
        self.conv1 = Convolution(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
        self.conv2 = Convolution(cout, cout, 3, 1, 1)
        self.nin_shortcut = Convolution(cin, cout, 1, 1, 0) if cin != cout else nn.Identity()

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return self.nin_shortcut(x) + h


class AEKLDownsample(nn.Module):
    """F.pad(x,(0,1,0,1)) (right/bottom only) then 3x3 stride-2 pad-0 conv."""

    def __init__(self, c: int):
        super().__init__()
        self.conv = Convolution(c, c, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0))


class Upsample(nn.Module):
    """MONAI ``Upsample(mode="nontrainable", interp_mode="nearest", scale_factor=2,
    post_conv=Convolution 3x3)``: submodules ``upsample_non_trainable`` / ``postconv``."""

    def __init__(self, c: int):
        super().__init__()
        self.upsample_non_trainable = nn.Upsample(scale_factor=2.0, mode="nearest")
        self.postconv = Convolution(c, c, 3, 1, 1)

    def forward(self, x):
        return self.postconv(self.upsample_non_trainable(x))


class SABlock(nn.Module):
    """Single-head self-attention, separate to_q/to_k/to_v (bias) and out_proj."""

    def __init__(self, c: int):
        super().__init__()
        self.to_q = nn.Linear(c, c, bias=True)
        self.to_k = nn.Linear(c, c, bias=True)
        self.to_v = nn.Linear(c, c, bias=True)
        self.out_proj = nn.Linear(c, c)
        self.scale = float(c) ** -0.5

    def forward(self, x):  # x: [B, L, C]
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        att = torch.softmax(torch.einsum("blc,bmc->blm", q, k) * self.scale, dim=-1)
        return self.out_proj(torch.einsum("blm,bmc->blc", att, v))


class SpatialAttentionBlock(nn.Module):
    def __init__(self, c: int, groups: int, eps: float):
        super().__init__()
        self.norm = nn.GroupNorm(groups, c, eps=eps, affine=True)
        self.attn = SABlock(c)

    def forward(self, x):
        b, c, h, w = x.shape
        r = x
        y = self.norm(x).reshape(b, c, h * w).transpose(1, 2)
        y = self.attn(y)
        return y.transpose(1, 2).reshape(b, c, h, w) + r


def _expand(num_res_blocks, n):
    return [num_res_blocks] * n if isinstance(num_res_blocks, int) else list(num_res_blocks)


class Encoder(nn.Module):
    def __init__(self, in_channels, channels, latent_channels, num_res_blocks, groups, eps,
                 attention_levels, with_nonlocal_attn):
        super().__init__()
        blocks: list[nn.Module] = [Convolution(in_channels, channels[0], 3, 1, 1)]
        cout = channels[0]
        for i, c in enumerate(channels):
            cin, cout = cout, c
            last = i == len(channels) - 1
            for _ in range(num_res_blocks[i]):
                blocks.append(AEKLResBlock(cin, cout, groups, eps))
                cin = cout
                if attention_levels[i]:
                    blocks.append(SpatialAttentionBlock(cin, groups, eps))
            if not last:
                blocks.append(AEKLDownsample(cin))
        if with_nonlocal_attn:
            blocks.append(AEKLResBlock(channels[-1], channels[-1], groups, eps))
            blocks.append(SpatialAttentionBlock(channels[-1], groups, eps))
            blocks.append(AEKLResBlock(channels[-1], channels[-1], groups, eps))
        blocks.append(nn.GroupNorm(groups, channels[-1], eps=eps, affine=True))  # NO SiLU after it
        blocks.append(Convolution(channels[-1], latent_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class Decoder(nn.Module):
    def __init__(self, channels, latent_channels, out_channels, num_res_blocks, groups, eps,
                 attention_levels, with_nonlocal_attn):
        super().__init__()
        rc = list(reversed(channels))
        blocks: list[nn.Module] = [Convolution(latent_channels, rc[0], 3, 1, 1)]
        if with_nonlocal_attn:
            blocks.append(AEKLResBlock(rc[0], rc[0], groups, eps))
            blocks.append(SpatialAttentionBlock(rc[0], groups, eps))
            blocks.append(AEKLResBlock(rc[0], rc[0], groups, eps))
        r_att = list(reversed(attention_levels))
        r_nrb = list(reversed(num_res_blocks))
        cout = rc[0]
        for i, c in enumerate(rc):
            cin, cout = cout, c
            last = i == len(rc) - 1
            for _ in range(r_nrb[i]):
                blocks.append(AEKLResBlock(cin, cout, groups, eps))
                cin = cout
                if r_att[i]:
                    blocks.append(SpatialAttentionBlock(cin, groups, eps))
            if not last:
                blocks.append(Upsample(cin))
        blocks.append(nn.GroupNorm(groups, cin, eps=eps, affine=True))  # NO SiLU after it
        blocks.append(Convolution(cin, out_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class AutoencoderKLOracle(nn.Module):
    """fp32 CPU oracle of MONAI AutoencoderKL (2-D only; Appendix A.1)."""

    def __init__(self, spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4,
                 channels=(32, 64, 128, 128), num_res_blocks=2, norm_num_groups=32, norm_eps=1e-6,
                 attention_levels=None, with_encoder_nonlocal_attn=True,
                 with_decoder_nonlocal_attn=True):
        super().__init__()
        if spatial_dims != 2:
            raise ValueError("oracle restates the 2-D network only")
        channels = list(channels)
        if attention_levels is None:
            attention_levels = [False] * len(channels)
        attention_levels = list(attention_levels)
        if any(c % norm_num_groups != 0 for c in channels):
            raise ValueError("AutoencoderKL expects all channels being multiple of norm_num_groups")
        if len(channels) != len(attention_levels):
            raise ValueError("AutoencoderKL expects channels being same size of attention_levels")
        nrb = _expand(num_res_blocks, len(channels))
        if len(nrb) != len(channels):
            raise ValueError("num_res_blocks must have the same length as channels")
        self.in_channels = in_channels
        self.latent_channels = latent_channels
        self.encoder = Encoder(in_channels, channels, latent_channels, nrb, norm_num_groups,
                               norm_eps, attention_levels, with_encoder_nonlocal_attn)
        self.decoder = Decoder(channels, latent_channels, out_channels, nrb, norm_num_groups,
                               norm_eps, attention_levels, with_decoder_nonlocal_attn)
        self.quant_conv_mu = Convolution(latent_channels, latent_channels, 1)
        self.quant_conv_log_sigma = Convolution(latent_channels, latent_channels, 1)
        self.post_quant_conv = Convolution(latent_channels, latent_channels, 1)

    # --- MONAI API (Appendix A.1) ------------------------------------------------
    def encode(self, x):
        h = self.encoder(x)
        z_mu = self.quant_conv_mu(h)
        z_log_var = torch.clamp(self.quant_conv_log_sigma(h), -30.0, 20.0)
        z_sigma = torch.exp(z_log_var / 2)
        return z_mu, z_sigma

    def encode_with_logvar(self, x):
        h = self.encoder(x)
        z_mu = self.quant_conv_mu(h)
        z_log_var = torch.clamp(self.quant_conv_log_sigma(h), -30.0, 20.0)
        return z_mu, torch.exp(z_log_var / 2), z_log_var

    def sampling(self, z_mu, z_sigma, eps=None):
        if eps is None:
            eps = torch.randn_like(z_sigma)
        return z_mu + eps * z_sigma

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def reconstruct(self, x):
        return self.decode(self.encode(x)[0])

    def forward(self, x, eps=None):
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma, eps)
        return self.decode(z), z_mu, z_sigma

    def encode_stage_2_inputs(self, x, eps=None):
        return self.sampling(*self.encode(x), eps)

    def decode_stage_2_outputs(self, z):
        return self.decode(z)


CONFIG_A = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4,
                channels=[32, 64, 128, 128], num_res_blocks=2, norm_num_groups=16, norm_eps=1e-6,
                attention_levels=[False, False, False, False],
                with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True)
CONFIG_AR = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=10,
                 channels=[64, 128, 256], num_res_blocks=2, norm_num_groups=32, norm_eps=1e-6,
                 attention_levels=[False, False, False],
                 with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True)


def build_oracle(cfg: dict, seed: int = 42) -> AutoencoderKLOracle:
    """Seeded default-init oracle (``set_determinism(seed)`` analogue, train_vae.py:808)."""
    torch.manual_seed(seed)
    return AutoencoderKLOracle(**cfg).float()


def synthetic_images(batch: int, channels: int, size: int, seed: int = 42) -> torch.Tensor:
    """Synthetic inputs of SURVEY.md §8(d): z-scored elliptical foreground (~40% of the
    pixels), exact-zero background — mimics ``LocalNormalizeByMask`` output."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, channels, size, size, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    mask = ((xx / 0.80) ** 2 + (yy / 0.64) ** 2 <= 1.0).float()
    return x * mask
