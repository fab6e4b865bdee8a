"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the perceptual (LPIPS, SqueezeNet-1.1) term, the checker of SURVEY 8f N3.

What it restates: ``vae_scripts/train_vae.py:299`` builds ``monai.losses.PerceptualLoss(spatial_dims=2,
network_type="squeeze")`` and ``:395-397`` calls it on ``ensure_three_channels(reconstruction)`` /
``ensure_three_channels(images)`` (``src/pti_ldm_vae/utils/losses.py:8-28``: a one-channel batch is repeated three times).
MONAI's 2-D "squeeze" network is the ``lpips`` package's ``LPIPS(net="squeeze", lpips=True)`` in eval mode on
torchvision's ``squeezenet1_1().features``; MONAI returns the batch mean.  Neither package (monai 1.5.1, lpips 0.1.4,
torchvision) is in ``/root/reference`` nor importable here and the pretrained weight files cannot be fetched, so this file
restates the PUBLISHED structure of the two packages as plain functions over a ``state_dict`` with their key names
(``features.N.*`` of torchvision, ``linK.model.1.weight`` of lpips):

  * input scaling  (x - shift) / scale  with lpips' ScalingLayer constants;
  * features: conv 3->64 k3 s2 (no padding) + ReLU | MaxPool(3,2,ceil) | Fire x2 | pool | Fire x2 | pool | Fire x4, a Fire
    being  s = relu(conv1x1(x));  cat(relu(conv1x1(s)), relu(conv3x3 pad 1 (s)));
  * seven taps = outputs of features[0:2], [2:5], [5:8], [8:10], [10:11], [11:12], [12:13];
  * per tap: unit-normalise every pixel's channel vector (x / (||x||_2 + 1e-10)), squared difference, 1x1 ``lin`` layer
    (no bias; the Dropout in front of it is inactive in eval mode), spatial mean; summed over the taps -> [N,1,1,1].

**Parity unpinned**: no weights, no reference output and no fixture exist for this term (the reference holds no tests);
it is anchored on the reference's call sites above only.  The product (``pti_ldm_vae_amd/models/perceptual.py``) is
HIP-only and imports nothing from here; the GPU tests load the product's ``state_dict()`` into these functions.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# lpips.ScalingLayer
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)
# torchvision squeezenet1_1().features: index -> kind
POOLS = (2, 5, 8)
FIRES = (3, 4, 6, 7, 9, 10, 11, 12)
TAP_AFTER = (1, 4, 7, 9, 10, 11, 12)          # last feature index of each lpips slice
TAP_CHANNELS = (64, 128, 256, 384, 384, 512, 512)


def three_channels(x: torch.Tensor) -> torch.Tensor:
    """``ensure_three_channels`` (utils/losses.py:8-28): [N,1,H,W] -> repeated to [N,3,H,W]; 3-channel input unchanged."""
    if x.dim() != 4 or x.shape[1] not in (1, 3):
        raise ValueError(f"expected [N,1|3,H,W], got {tuple(x.shape)}")
    return x.repeat(1, 3, 1, 1) if x.shape[1] == 1 else x


def scale_input(x: torch.Tensor) -> torch.Tensor:
    shift = torch.tensor(SHIFT, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - shift) / scale


def fire(sd: dict, idx: int, x: torch.Tensor) -> torch.Tensor:
    p = f"features.{idx}."
    s = F.relu(F.conv2d(x, sd[p + "squeeze.weight"], sd[p + "squeeze.bias"]))
    e1 = F.relu(F.conv2d(s, sd[p + "expand1x1.weight"], sd[p + "expand1x1.bias"]))
    e3 = F.relu(F.conv2d(s, sd[p + "expand3x3.weight"], sd[p + "expand3x3.bias"], padding=1))
    return torch.cat([e1, e3], dim=1)


def feature_layer(sd: dict, idx: int, x: torch.Tensor) -> torch.Tensor:
    """Layer ``idx`` of ``squeezenet1_1().features`` alone."""
    if idx == 0:
        return F.conv2d(x, sd["features.0.weight"], sd["features.0.bias"], stride=2)
    if idx == 1:
        return F.relu(x)
    if idx in POOLS:
        return F.max_pool2d(x, 3, 2, ceil_mode=True)
    return fire(sd, idx, x)


def taps(sd: dict, x3: torch.Tensor) -> list:
    """The seven feature taps of a 3-channel batch (input scaling included)."""
    x, out = scale_input(x3), []
    for idx in range(13):
        x = feature_layer(sd, idx, x)
        if idx in TAP_AFTER:
            out.append(x)
    return out


def tap_distance(a: torch.Tensor, b: torch.Tensor, lin_weight: torch.Tensor) -> torch.Tensor:
    """One tap of LPIPS: channel-unit-normalised squared difference through the 1x1 ``lin`` layer, spatial mean -> [N]."""
    an = a / (a.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
    bn = b / (b.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
    return F.conv2d((an - bn) ** 2, lin_weight.reshape(1, -1, 1, 1)).mean(dim=(2, 3)).reshape(-1)


def lpips(sd: dict, x3: torch.Tensor, y3: torch.Tensor) -> torch.Tensor:
    """``lpips.LPIPS(net="squeeze")(x, y)`` on 3-channel batches -> [N,1,1,1]."""
    total = 0.0
    for k, (a, b) in enumerate(zip(taps(sd, x3), taps(sd, y3))):
        total = total + tap_distance(a, b, sd[f"lin{k}.model.1.weight"])
    return total.reshape(-1, 1, 1, 1)


def perceptual_loss(sd: dict, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """``monai.losses.PerceptualLoss("squeeze")(input, target)`` as the reference calls it: batch mean of LPIPS on the
    (repeated) three-channel images -> scalar."""
    return lpips(sd, three_channels(input.float() if input.dtype not in (torch.float32, torch.float64) else input),
                 three_channels(target.float() if target.dtype not in (torch.float32, torch.float64) else target)).mean()


def cpu_state(module_or_sd, dtype=torch.float32) -> dict:
    """A product module's (or a plain) ``state_dict`` as CPU tensors of ``dtype`` for the functions above."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: v.detach().to("cpu", dtype) for k, v in sd.items()}


def random_state(seed: int = 0, nonneg_lin: bool = True) -> dict:
    """A seeded random ``state_dict`` with both packages' key names and shapes (PyTorch default conv init)."""
    g = torch.Generator().manual_seed(seed)

    def conv(cout, cin, k):
        bound = 1.0 / (cin * k * k) ** 0.5
        return ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound, (torch.rand(cout, generator=g) * 2 - 1) * bound)
    sd = {}
    sd["features.0.weight"], sd["features.0.bias"] = conv(64, 3, 3)
    cin = 64
    for idx, (sq, ex) in zip(FIRES, ((16, 64), (16, 64), (32, 128), (32, 128), (48, 192), (48, 192), (64, 256), (64, 256))):
        p = f"features.{idx}."
        sd[p + "squeeze.weight"], sd[p + "squeeze.bias"] = conv(sq, cin, 1)
        sd[p + "expand1x1.weight"], sd[p + "expand1x1.bias"] = conv(ex, sq, 1)
        sd[p + "expand3x3.weight"], sd[p + "expand3x3.bias"] = conv(ex, sq, 3)
        cin = 2 * ex
    for k, c in enumerate(TAP_CHANNELS):
        w = (torch.rand(1, c, 1, 1, generator=g) * 2 - 1) / c ** 0.5
        sd[f"lin{k}.model.1.weight"] = w.abs() if nonneg_lin else w
    return sd
