"""Perceptual (LPIPS) term of the training loss — SURVEY 8f N3.

Reference: ``vae_scripts/train_vae.py:299`` builds ``monai.losses.PerceptualLoss(spatial_dims=2, network_type="squeeze")``
(= the ``lpips`` package's ``LPIPS(net="squeeze")`` on torchvision's SqueezeNet-1.1 features, result averaged over the
batch) and ``:395-397`` applies it to ``ensure_three_channels(reconstruction)`` vs ``ensure_three_channels(images)`` with
weight 1.0 in every shipped config.

Status here: the pretrained weights (torchvision ``squeezenet1_1`` + lpips ``squeeze.pth``) cannot be fetched — there is
no network — and neither ``lpips`` nor ``torchvision`` is installed.  This module is the PARAMETER HOLDER (``torch.nn``
modules only so that ``state_dict()`` carries the two packages' key names; ``PerceptualLoss(weights=(backbone_file,
lin_file))`` loads files the user supplies locally with ``torch.load(..., weights_only=True)``) plus the autograd glue of
the HIP path.  Like every other module of this package it is **HIP-only**: the arithmetic is ``perceptual_engine.py`` +
``csrc/squeeze.hip`` / ``csrc/lpips.hip`` behind the C-ABI -- one-channel images (what the VAE produces) run entirely on
the library (first layer folded to one input channel, Fire modules on the MFMA convs, pooling, comparison tail);
three-channel inputs take ``torch``'s convolution for the first layer only (a device op, plumbing next to the hot path)
and enter the trunk through the layout kernels.  CPU tensors are REFUSED; there is no torch formulation of the term in
the product -- the checker lives in ``oracle/perceptual.py`` (test infrastructure).
The gradient w.r.t. the reconstruction is ADDED to the native step's ``d_recon`` (``VAETrainer(perceptual=...)``).
Parity: UNPINNED (restated from the published structure of both packages; no weights, no reference output available).
Without supplied weights the class refuses to build unless ``allow_random_init=True`` (tests, throughput runs).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.losses import ensure_three_channels

LPIPS_CHANNELS = (64, 128, 256, 384, 384, 512, 512)


class Fire(nn.Module):
    """torchvision ``squeezenet.Fire`` (key names ``squeeze``, ``expand1x1``, ``expand3x3``)."""

    def __init__(self, cin: int, squeeze: int, e1: int, e3: int):
        super().__init__()
        self.squeeze = nn.Conv2d(cin, squeeze, 1)
        self.expand1x1 = nn.Conv2d(squeeze, e1, 1)
        self.expand3x3 = nn.Conv2d(squeeze, e3, 3, padding=1)

    def forward(self, x):
        raise NotImplementedError("parameter holder: a Fire module runs as perceptual_engine._Fire (two MFMA convolutions); "
                                  "the torch restatement used as the tests' checker is oracle/perceptual.py")


class _Pool(nn.Module):
    def forward(self, x):
        raise NotImplementedError("parameter-free marker: the pooling runs as ops.maxpool3s2_fwd / _bwd")


class _Relu(nn.Module):
    def forward(self, x):
        return F.relu(x)


def squeezenet1_1_features() -> nn.Sequential:
    """``torchvision.models.squeezenet1_1().features`` (indices 0..12 => keys ``features.N...`` of its state_dict)."""
    return nn.Sequential(
        nn.Conv2d(3, 64, 3, stride=2), _Relu(), _Pool(),
        Fire(64, 16, 64, 64), Fire(128, 16, 64, 64), _Pool(),
        Fire(128, 32, 128, 128), Fire(256, 32, 128, 128), _Pool(),
        Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256), Fire(512, 64, 256, 256))


class _Lin(nn.Module):
    """lpips ``NetLinLayer``: Dropout (index 0, inactive in eval) + 1x1 conv without bias (index 1) => ``linK.model.1.weight``."""

    def __init__(self, cin: int):
        super().__init__()
        self.model = nn.Sequential(nn.Identity(), nn.Conv2d(cin, 1, 1, bias=False))

    def forward(self, x):
        return self.model(x)


class _TrunkCompareFn(torch.autograd.Function):
    """LPIPS value [N] from tap 0 of the reconstruction (NCHW fp32, output of the torch first layer) and the target's
    seven taps (``SqueezeLPIPS.native_target_taps``: tap 0 NCHW fp32, taps 1..6 NHWC fp16), with the rest of the feature
    network on the HIP trunk (``perceptual_engine.SqueezeTrunk``) and the comparison on the tail kernels in each map's own
    layout; gradient to tap 0 only."""

    @staticmethod
    def forward(ctx, t0, trunk, target, lin_ws):
        from .. import ops
        t0 = t0.contiguous()
        taps, saved = trunk.forward(ops.nchw_f32_to_nhwc_f16(t0), save=True)
        total, sv0 = ops.lpips_tap_fwd(t0, target[0], lin_ws[0])
        tails = [sv0]
        for a, b, w in zip(taps, target[1:], lin_ws[1:]):
            val, sv = ops.lpips_tap_nhwc_fwd(a, b, w)
            tails.append(sv)
            total = total + val
        ctx.trunk, ctx.saved, ctx.feats, ctx.target, ctx.lin_ws, ctx.tails = trunk, saved, [t0] + taps, target, lin_ws, tails
        return total

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        g = g.contiguous().float()
        f, t, w, sv = ctx.feats, ctx.target, ctx.lin_ws, ctx.tails
        g_t0 = ops.lpips_tap_bwd(f[0], t[0], w[0], sv[0], g)
        tap_grads = [ops.lpips_tap_nhwc_bwd(f[k], t[k], w[k], sv[k], g) for k in range(1, len(f))]
        g0 = ctx.trunk.backward(ctx.saved, tap_grads)
        ctx.saved = ctx.feats = ctx.tails = None
        return ops.nhwc_bf16_add_to_nchw_f32_(g0.contiguous(), g_t0), None, None, None


class _OneChannelCompareFn(torch.autograd.Function):
    """LPIPS value [N] of a ONE-channel reconstruction [N,1,H,W] against the target's seven taps (all NHWC fp16, from
    ``SqueezeLPIPS.native_target_taps`` on a one-channel target): the whole feature network on the HIP library -- first
    layer folded to one input channel (``perceptual_engine.fold_first_layer``), trunk, comparison -- and the gradient
    w.r.t. the one-channel image (the sum over the three repeated channels is part of the fold)."""

    @staticmethod
    def forward(ctx, x1, trunk, target, lin_ws):
        from .. import ops
        x1 = x1.contiguous()
        t0 = ops.squeeze_conv1_fwd(x1, trunk.w10)
        taps, saved = trunk.forward(t0, save=True)
        feats, total, tails = [t0] + taps, None, []
        for a, b, w in zip(feats, target, lin_ws):
            val, sv = ops.lpips_tap_nhwc_fwd(a, b, w)
            tails.append(sv)
            total = val if total is None else total + val
        ctx.trunk, ctx.saved, ctx.feats, ctx.target, ctx.lin_ws, ctx.tails = trunk, saved, feats, target, lin_ws, tails
        ctx.hw = (x1.shape[-2], x1.shape[-1])
        return total

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        g = g.contiguous().float()
        f, t, w, sv = ctx.feats, ctx.target, ctx.lin_ws, ctx.tails
        tap_grads = [ops.lpips_tap_nhwc_bwd(f[k], t[k], w[k], sv[k], g) for k in range(len(f))]
        g0 = ops.relu_bwd_add_(tap_grads[0], ctx.trunk.backward(ctx.saved, tap_grads[1:]).contiguous(), f[0])
        dx = ops.squeeze_conv1_bwd(g0, None, ctx.trunk.w10, *ctx.hw)       # g0 carries the ReLU mask already
        ctx.saved = ctx.feats = ctx.tails = None
        return dx, None, None, None


class SqueezeLPIPS(nn.Module):
    """``lpips.LPIPS(net="squeeze", lpips=True, spatial=False)`` in eval mode: input scaling, seven SqueezeNet-1.1
    feature taps (``features[0:2], [2:5], [5:8], [8:10], [10:11], [11:12], [12:13]``), channel-unit-normalised squared
    differences weighted by the ``lin`` layers, spatially averaged and summed.  Returns [N,1,1,1]."""

    SLICES = ((0, 2), (2, 5), (5, 8), (8, 10), (10, 11), (11, 12), (12, 13))
    _trunk = None

    def __init__(self):
        super().__init__()
        self.features = squeezenet1_1_features()
        for k, c in enumerate(LPIPS_CHANNELS):
            setattr(self, f"lin{k}", _Lin(c))
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1), persistent=False)
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()

    # ---- the only path: the HIP library (CPU tensors are refused) ------------------------------------------------------
    def require_device(self, x, what="input"):
        if not (x.is_cuda and self.features[0].weight.is_cuda):
            raise RuntimeError(f"pti_ldm_vae_amd PerceptualLoss runs on MI355X only ({what} / module on "
                               f"{x.device} / {self.features[0].weight.device}): there is no CPU or torch fallback for the "
                               "term -- the CPU restatement used as the tests' checker is oracle/perceptual.py")
        if x.dtype != torch.float32 or x.dim() != 4:
            raise TypeError(f"PerceptualLoss: expected fp32 [N,C,H,W], got {x.dtype} {tuple(x.shape)}")

    def trunk(self):
        dev = self.features[0].weight.device
        if self._trunk is None or self._trunk_dev != dev:
            from ..perceptual_engine import SqueezeTrunk
            self._trunk, self._trunk_dev = SqueezeTrunk(self), dev
        return self._trunk

    def tap0(self, x):
        """First layer of a THREE-channel batch: torch's convolution on the device (one-channel images never come here)."""
        x = (x - self.shift) / self.scale
        return self.features[1](self.features[0](x))

    @torch.no_grad()
    def native_target_taps(self, x):
        """The seven taps of a three-channel TARGET as ``native_compare`` consumes them (tap 0 NCHW fp32 as the first
        layer leaves it, taps 1..6 NHWC fp16 from the HIP trunk), no autograd graph."""
        from .. import ops
        self.require_device(x, "target")
        t0 = self.tap0(x).contiguous()
        taps, _ = self.trunk().forward(ops.nchw_f32_to_nhwc_f16(t0), save=False)
        return [t0] + taps

    def native_compare(self, in0, target_taps):
        self.require_device(in0)
        lin_ws = [getattr(self, f"lin{k}").model[1].weight.view(-1) for k in range(len(self.SLICES))]
        return _TrunkCompareFn.apply(self.tap0(in0), self.trunk(), target_taps, lin_ws).view(-1, 1, 1, 1)

    # one-channel images (what the VAE produces): the first layer too runs on the HIP library, folded to one channel
    @torch.no_grad()
    def native_target_taps_1ch(self, x1):
        from .. import ops
        self.require_device(x1, "target")
        tr = self.trunk()
        t0 = ops.squeeze_conv1_fwd(x1.contiguous(), tr.w10)
        taps, _ = tr.forward(t0, save=False)
        return [t0] + taps

    def native_compare_1ch(self, x1, target_taps):
        self.require_device(x1)
        lin_ws = [getattr(self, f"lin{k}").model[1].weight.view(-1) for k in range(len(self.SLICES))]
        return _OneChannelCompareFn.apply(x1, self.trunk(), target_taps, lin_ws).view(-1, 1, 1, 1)

    def forward(self, in0, in1):
        """LPIPS of two three-channel batches on the device -> [N,1,1,1]."""
        return self.native_compare(in0, self.native_target_taps(in1))

    def load_state_dict(self, *args, **kwargs):
        self._trunk = None          # the HIP trunk packs the weights once: rebuild it from the loaded ones
        return super().load_state_dict(*args, **kwargs)

    # ---- local weight files ------------------------------------------------------------------------------------------
    def load_local_weights(self, backbone_file: str, lin_file: str) -> None:
        """``backbone_file``: torchvision's ``squeezenet1_1-*.pth`` state_dict (keys ``features.N.*``; the classifier's
        keys are ignored); ``lin_file``: lpips' ``weights/v0.1/squeeze.pth`` (keys ``linK.model.1.weight``).  Both are
        read with ``weights_only=True``.  Raises if a key the network needs is missing or has the wrong shape."""
        bb = torch.load(backbone_file, map_location="cpu", weights_only=True)
        lin = torch.load(lin_file, map_location="cpu", weights_only=True)
        own = self.state_dict()
        picked = {}
        for k in own:
            src = bb if k.startswith("features.") else lin
            if k not in src:
                raise KeyError(f"perceptual weights: '{k}' not found in {'backbone' if src is bb else 'lin'} file")
            if tuple(src[k].shape) != tuple(own[k].shape):
                raise ValueError(f"perceptual weights: shape of '{k}' is {tuple(src[k].shape)}, expected {tuple(own[k].shape)}")
            picked[k] = src[k]
        self.load_state_dict(picked, strict=True)
        self._trunk = None          # the HIP trunk packs the weights once: rebuild it from the loaded ones


class PerceptualLoss(nn.Module):
    """Counterpart of ``monai.losses.PerceptualLoss(spatial_dims=2, network_type="squeeze")`` as the reference uses it:
    ``forward(input, target) -> scalar`` = batch mean of LPIPS(input, target); 1-channel inputs are repeated to three
    (``ensure_three_channels``, which the reference applies itself before the call — doing it again is a no-op)."""

    def __init__(self, spatial_dims: int = 2, network_type: str = "squeeze", weights: tuple[str, str] | None = None,
                 allow_random_init: bool = False):
        super().__init__()
        if spatial_dims != 2 or network_type != "squeeze":
            raise ValueError("pti_ldm_vae_amd PerceptualLoss: spatial_dims=2, network_type='squeeze' (the reference's call)")
        self.net = SqueezeLPIPS()
        if weights is not None:
            self.net.load_local_weights(*weights)
        elif not allow_random_init:
            raise RuntimeError("PerceptualLoss: the pretrained SqueezeNet-1.1 / LPIPS weights are not available offline. "
                               "Pass weights=(squeezenet1_1_state_dict.pth, lpips_squeeze.pth) from local files "
                               "(or allow_random_init=True for tests / throughput runs).")
        self.pretrained = weights is not None

    @staticmethod
    def _one_channel(t: torch.Tensor) -> bool:
        return t.dim() == 4 and t.shape[1] == 1 and min(t.shape[2:]) >= 3

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if not (self._one_channel(input) and self._one_channel(target)):      # mixed channel counts: both as three
            input, target = ensure_three_channels(input.float()), ensure_three_channels(target.float())
        return self.from_taps(input, self.target_taps(target))

    @torch.no_grad()
    def target_taps(self, target: torch.Tensor):
        """The target's feature taps alone (no autograd graph): they do not depend on the reconstruction, so the trainer
        computes them on its side stream while the VAE forward runs, and hands them to ``from_taps``."""
        target = target.float()
        self.net.require_device(target, "target")
        if self._one_channel(target):
            return self.net.native_target_taps_1ch(target)
        return self.net.native_target_taps(ensure_three_channels(target))

    def from_taps(self, input: torch.Tensor, target_taps) -> torch.Tensor:
        """``forward(input, target)`` with the target's taps precomputed by ``target_taps(target)``."""
        input = input.float()
        self.net.require_device(input)
        if self._one_channel(input) and target_taps[0].dtype == torch.float16:
            return self.net.native_compare_1ch(input, target_taps).mean()
        return self.net.native_compare(ensure_three_channels(input), target_taps).mean()
