"""MI355X-native stand-in for ``monai.networks.nets.AutoencoderKL`` (MONAI 1.5.1) as the reference
builds it in ``src/pti_ldm_vae/models/autoencoder.py:67-79``.

The ``nn.Module`` tree below only HOLDS parameters, under exactly the names MONAI registers them
(SURVEY.md Appendix A.3), so ``state_dict()`` / ``load_state_dict()`` / ``parameters()`` /
``DistributedDataParallel`` / ``torch.optim.Adam`` see what they would see with MONAI.  All
arithmetic runs in the HIP engine (``pti_ldm_vae_amd/engine.py``) through the C-ABI; there is no
PyTorch or CPU fallback — calling the model on a CPU tensor raises.

Memory layout: every parameter is a view into ONE flat fp32 arena -- region 1 = encoder blocks, quant_conv_mu,
quant_conv_log_sigma; region 2 = post_quant_conv, then the decoder blocks (registration order inside each, q|k|v
projections adjacent) -- and gradients are written by the kernels into a second arena of the same layout.  That is what lets the data-parallel loop all-reduce gradients as a few large
contiguous buckets and the optimiser run as one kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("parameter holder: the HIP engine computes this block")


class Convolution(_Holder):
    """MONAI ``Convolution(conv_only=True)``: the conv is registered as ``.conv``."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)


class AEKLResBlock(_Holder):
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
        self.conv1 = Convolution(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
        self.conv2 = Convolution(cout, cout, 3, 1, 1)
        self.nin_shortcut = Convolution(cin, cout, 1, 1, 0) if cin != cout else nn.Identity()


class AEKLDownsample(_Holder):
    def __init__(self, c):
        super().__init__()
        self.conv = Convolution(c, c, 3, 2, 0)


class Upsample(_Holder):
    def __init__(self, c):
        super().__init__()
        self.upsample_non_trainable = nn.Upsample(scale_factor=2.0, mode="nearest")
        self.postconv = Convolution(c, c, 3, 1, 1)


class SABlock(_Holder):
    def __init__(self, c):
        super().__init__()
        self.to_q = nn.Linear(c, c, bias=True)
        self.to_k = nn.Linear(c, c, bias=True)
        self.to_v = nn.Linear(c, c, bias=True)
        self.out_proj = nn.Linear(c, c)


class SpatialAttentionBlock(_Holder):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.norm = nn.GroupNorm(groups, c, eps=eps, affine=True)
        self.attn = SABlock(c)


class Encoder(_Holder):
    def __init__(self, in_channels, channels, latent_channels, nrb, groups, eps, attention_levels, nonlocal_attn):
        super().__init__()
        blocks = [Convolution(in_channels, channels[0], 3, 1, 1)]
        cout = channels[0]
        for i, c in enumerate(channels):
            cin, cout = cout, c
            for _ in range(nrb[i]):
                blocks.append(AEKLResBlock(cin, cout, groups, eps))
                cin = cout
                if attention_levels[i]:
                    blocks.append(SpatialAttentionBlock(cin, groups, eps))
            if i != len(channels) - 1:
                blocks.append(AEKLDownsample(cin))
        if nonlocal_attn:
            blocks += [AEKLResBlock(channels[-1], channels[-1], groups, eps),
                       SpatialAttentionBlock(channels[-1], groups, eps),
                       AEKLResBlock(channels[-1], channels[-1], groups, eps)]
        blocks.append(nn.GroupNorm(groups, channels[-1], eps=eps, affine=True))
        blocks.append(Convolution(channels[-1], latent_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)


class Decoder(_Holder):
    def __init__(self, channels, latent_channels, out_channels, nrb, groups, eps, attention_levels, nonlocal_attn):
        super().__init__()
        rc = list(reversed(channels))
        blocks = [Convolution(latent_channels, rc[0], 3, 1, 1)]
        if nonlocal_attn:
            blocks += [AEKLResBlock(rc[0], rc[0], groups, eps), SpatialAttentionBlock(rc[0], groups, eps),
                       AEKLResBlock(rc[0], rc[0], groups, eps)]
        r_att, r_nrb = list(reversed(attention_levels)), list(reversed(nrb))
        cout = rc[0]
        for i, c in enumerate(rc):
            cin, cout = cout, c
            for _ in range(r_nrb[i]):
                blocks.append(AEKLResBlock(cin, cout, groups, eps))
                cin = cout
                if r_att[i]:
                    blocks.append(SpatialAttentionBlock(cin, groups, eps))
            if i != len(rc) - 1:
                blocks.append(Upsample(cin))
        blocks.append(nn.GroupNorm(groups, cin, eps=eps, affine=True))
        blocks.append(Convolution(cin, out_channels, 3, 1, 1))
        self.blocks = nn.ModuleList(blocks)


def _attn_first(named):
    """Arena order: registration order, except that inside an attention block the three projection
    weights are adjacent and the three projection biases are adjacent (the fused q|k|v 1x1 conv
    and its gradient then address them as one [3C,C] / [3C] tensor without any copy)."""
    out, held = [], {}
    for name, p in named:
        if ".attn.to_" in name:
            held[name] = p
            if name.endswith("attn.to_v.bias"):
                pre = name[: -len("to_v.bias")]
                for leaf in ("to_q.weight", "to_k.weight", "to_v.weight", "to_q.bias", "to_k.bias", "to_v.bias"):
                    out.append((pre + leaf, held.pop(pre + leaf)))
        else:
            out.append((name, p))
    assert not held
    return out


class AutoencoderKL(nn.Module):
    """Drop-in for MONAI's AutoencoderKL on MI355X (2-D).  ``third_output`` selects what the third
    element of ``forward`` is: ``"sigma"`` (MONAI behaviour, default) or
    ``"logvar"`` (the clamped log-variance the reference's docstrings assume)."""

    def __init__(self, spatial_dims=2, in_channels=1, out_channels=1, latent_channels=3, channels=(32, 64, 64),
                 num_res_blocks=(1, 1, 2), norm_num_groups=32, norm_eps=1e-6, attention_levels=(False, False, True),
                 with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True, third_output="sigma"):
        super().__init__()
        if spatial_dims != 2:
            raise ValueError("pti_ldm_vae_amd AutoencoderKL: only spatial_dims=2 has a HIP path")
        channels = list(channels)
        attention_levels = list(attention_levels)
        if any(c % norm_num_groups != 0 for c in channels):
            raise ValueError("AutoencoderKL expects all channels being multiple of norm_num_groups")
        if len(channels) != len(attention_levels):
            raise ValueError("AutoencoderKL expects channels being same size of attention_levels")
        nrb = [num_res_blocks] * len(channels) if isinstance(num_res_blocks, int) else list(num_res_blocks)
        if len(nrb) != len(channels):
            raise ValueError("`num_res_blocks` should be a single integer or a tuple of integers with the same "
                             "length as `channels`.")
        if third_output not in ("sigma", "logvar"):
            raise ValueError("third_output must be 'sigma' or 'logvar'")
        if latent_channels > 16:
            raise ValueError("latent_channels > 16 is not supported by the HIP latent-head kernel")
        self.in_channels, self.out_channels, self.latent_channels = in_channels, out_channels, latent_channels
        self.channels, self.norm_num_groups, self.norm_eps = channels, norm_num_groups, norm_eps
        self.third_output = third_output
        self.encoder = Encoder(in_channels, channels, latent_channels, nrb, norm_num_groups, norm_eps,
                               attention_levels, with_encoder_nonlocal_attn)
        self.decoder = Decoder(channels, latent_channels, out_channels, nrb, norm_num_groups, norm_eps,
                               attention_levels, with_decoder_nonlocal_attn)
        self.quant_conv_mu = Convolution(latent_channels, latent_channels, 1)
        self.quant_conv_log_sigma = Convolution(latent_channels, latent_channels, 1)
        self.post_quant_conv = Convolution(latent_channels, latent_channels, 1)
        self._engine = None
        self._build_arena()

    # ---- flat arenas ---------------------------------------------------------------------------
    def _build_arena(self):
        named = list(self.named_parameters())
        enc = [(n, p) for n, p in named if n.startswith(("encoder.", "quant_conv_"))]
        # post_quant_conv sits FIRST in the decoder region (right behind quant_conv_*): backward finishes the decoder
        # region with it, so its gradients merge into the bucket of decoder.blocks.0 and, across the region boundary,
        # with the latent heads' that encode_backward reports first
        dec = ([(n, p) for n, p in named if n.startswith("post_quant_conv.")] +
               [(n, p) for n, p in named if n.startswith("decoder.")])
        assert len(enc) + len(dec) == len(named)
        order = _attn_first(enc) + _attn_first(dec)
        slots, off = {}, 0
        for name, p in order:
            slots[name] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4          # 16-byte aligned slots (vector bias loads)
        self._enc_end = max(o + (n + 3) // 4 * 4 for nme, (o, n, _) in slots.items()
                            if nme.startswith(("encoder.", "quant_conv_")))
        arena = torch.zeros(off, dtype=torch.float32)
        for name, p in order:
            o, n, shp = slots[name]
            arena[o:o + n].copy_(p.data.reshape(-1))
        self._slots, self._arena, self._grad_arena = slots, arena, None
        self._param_by_name = dict(named)
        self._repoint()

    def _repoint(self):
        for name, p in self._param_by_name.items():
            o, n, shp = self._slots[name]
            p.data = self._arena[o:o + n].view(shp)
            p.grad = None
        self._engine = None

    def _apply(self, fn, recurse=True):
        new = fn(self._arena)
        if new.dtype != torch.float32:
            raise TypeError("pti_ldm_vae_amd AutoencoderKL keeps fp32 master weights; bf16 copies are derived")
        self._arena = new
        self._grad_arena = None
        self._repoint()
        return self

    @property
    def param_arena(self) -> torch.Tensor:
        return self._arena

    @property
    def grad_arena(self) -> torch.Tensor:
        if self._grad_arena is None or self._grad_arena.device != self._arena.device:
            self._grad_arena = torch.zeros_like(self._arena)
        return self._grad_arena

    def slot_view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        """The view of ``arena`` (parameter / gradient arena, an Adam moment buffer) with the parameter's shape."""
        o, n, shp = self._slots[name]
        return arena[o:o + n].view(shp)

    def grad_view(self, name):
        return self.slot_view(self.grad_arena, name)

    def arena_regions(self):
        """(encoder+quant region, post_quant+decoder region) as (start, end) element offsets."""
        return (0, self._enc_end), (self._enc_end, self._arena.numel())

    def attach_grads(self):
        """Make every ``p.grad`` a view of the gradient arena (used by the native training loop)."""
        for name, p in self._param_by_name.items():
            p.grad = self.grad_view(name)

    def mark_weights_dirty(self):
        """Call after updating weights in a way the parameters' version counters do not see.  The engine re-derives its
        packed 16-bit operands when ``sum(p._version)`` changes: optimiser steps, ``load_state_dict``, ``p.add_()`` under
        ``no_grad`` all bump it; writes through ``p.data`` (``p.data.copy_()``, EMA loops on ``.data``) or directly into
        ``param_arena`` do NOT -- call this (or ``VAEModel.mark_weights_dirty``) after them."""
        if self._engine is not None:
            self._engine.packed_version = -1

    # ---- engine --------------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            if not self._arena.is_cuda:
                raise RuntimeError("pti_ldm_vae_amd AutoencoderKL runs on MI355X only: move the model to a cuda "
                                   "(HIP) device first. There is no CPU fallback for the VAE hot path.")
            from ..engine import Engine
            self._engine = Engine(self)
        return self._engine

    # ---- MONAI API -----------------------------------------------------------------------------
    def encode(self, x: torch.Tensor):
        """-> (z_mu, z_sigma), z_sigma = exp(clamp(log_var, -30, 20) / 2)."""
        return self.engine().encode(x)

    def sampling(self, z_mu, z_sigma):
        eps = torch.randn_like(z_sigma)
        return z_mu + eps * z_sigma

    def decode(self, z: torch.Tensor):
        return self.engine().decode(z)

    def reconstruct(self, x):
        z_mu, _ = self.encode(x)
        return self.decode(z_mu)

    def forward(self, x):
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma)
        third = z_sigma if self.third_output == "sigma" else 2.0 * torch.log(z_sigma)
        return self.decode(z), z_mu, third

    def encode_stage_2_inputs(self, x):
        return self.sampling(*self.encode(x))

    def decode_stage_2_outputs(self, z):
        return self.decode(z)
