"""Native trunk of the perceptual (LPIPS / SqueezeNet-1.1) feature network -- SURVEY 8f N3 ("HIP convs reused from K1").

Reference: ``vae_scripts/train_vae.py:299,395-397`` -> ``monai.losses.PerceptualLoss(network_type="squeeze")`` ->
``lpips.LPIPS(net="squeeze")`` -> ``torchvision.models.squeezenet1_1().features``.  ``models/perceptual.py`` holds the
parameters (both packages' key names) and the torch formulation; this module runs everything AFTER the first
convolution on the HIP library:

  * a Fire module = squeeze 1x1 + ReLU, then expand1x1 and expand3x3 + ReLU, concatenated.  Here: ONE 1x1
    ``pti_conv2d_mfma`` (squeeze channels zero-padded to a multiple of 32) and ONE 3x3 ``pti_conv2d_mfma`` whose first
    half of output channels carries the 1x1 expand weights at the centre tap -- the concatenation is the output layout;
  * ReLU fused into the convolutions' stores (``relu_out``); ReLU-backward / MaxPool(3, 2, ceil) forward + backward and
    the first layer folded to one input channel: ``csrc/squeeze.hip``;
  * activations NHWC fp16 (forward MFMA operands fp16, as the VAE's forward), gradients NHWC bf16 through the same
    data-gradient kernels as the VAE's backward (transposed / flipped weight packs).  The network is frozen: no weight
    gradients, weights packed once.

The first layer (3 -> 64, stride 2, no padding; 3 input channels are outside the MFMA kernels' shapes) stays a torch
convolution; its ReLU output is tap 0 and the trunk's input.  Parity vs the reference: UNPINNED like the rest of the
term (no weights available); pinned to the torch formulation of the same network by tests/test_gpu_perceptual.py."""
from __future__ import annotations

import torch

from . import ops

F16, BF16 = torch.float16, torch.bfloat16

# squeezenet1_1.features indices after the first conv + ReLU (0, 1): pools at 2, 5, 8; Fire modules elsewhere.
# taps (lpips slices): after 1 (tap 0 = trunk input), 4, 7, 9, 10, 11, 12.
_PLAN = (("pool",), ("fire", 3, False), ("fire", 4, True), ("pool",), ("fire", 6, False), ("fire", 7, True), ("pool",),
         ("fire", 9, True), ("fire", 10, True), ("fire", 11, True), ("fire", 12, True))


def fold_first_layer(net):
    """The first convolution of the network for a ONE-channel image (the reference repeats it three times,
    ``utils/losses.py:8-28``, and lpips' ScalingLayer scales every copy): fp32 [10, 64] = {W'[tap][co] =
    sum_c W[co][c][tap] / scale_c, b'[co] = b[co] - sum_c shift_c / scale_c * sum_tap W[co][c][tap]} -- exact, the layer
    has no padding.  Operand of ``ops.squeeze_conv1_fwd`` / ``_bwd``."""
    conv = net.features[0]
    if conv.in_channels != 3 or conv.out_channels != 64 or tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (2, 2) \
            or tuple(conv.padding) != (0, 0):
        raise ValueError("fold_first_layer: expected squeezenet1_1's first convolution (3 -> 64, 3x3, stride 2, no padding)")
    w = conv.weight.detach().double()                                    # [64, 3, 3, 3]
    scale, shift = net.scale.detach().double().view(3), net.shift.detach().double().view(3)
    wf = (w / scale.view(1, 3, 1, 1)).sum(1).reshape(64, 9)               # [co][tap]
    bf = conv.bias.detach().double() - (w.sum((2, 3)) * (shift / scale).view(1, 3)).sum(1)
    return torch.cat([wf.t().contiguous(), bf.view(1, 64)], 0).float().contiguous()


class _Fire:
    def __init__(self, f):
        dev = f.squeeze.weight.device
        cin, s = f.squeeze.in_channels, f.squeeze.out_channels
        e1, e3 = f.expand1x1.out_channels, f.expand3x3.out_channels
        sp = (s + 31) // 32 * 32
        self.cin, self.sp, self.cout = cin, sp, e1 + e3
        wsq = torch.zeros(sp, cin, 1, 1, device=dev)
        wsq[:s] = f.squeeze.weight.detach().float()
        self.bsq = torch.zeros(sp, device=dev)
        self.bsq[:s] = f.squeeze.bias.detach().float()
        wex = torch.zeros(e1 + e3, sp, 3, 3, device=dev)
        wex[:e1, :s, 1, 1] = f.expand1x1.weight.detach().float()[:, :, 0, 0]
        wex[e1:, :s] = f.expand3x3.weight.detach().float()
        self.bex = torch.cat([f.expand1x1.bias.detach().float(), f.expand3x3.bias.detach().float()]).contiguous()
        self.wsq = ops.pack_conv_weight(wsq.contiguous(), 1, f16=True)
        self.wex = ops.pack_conv_weight(wex.contiguous(), 3, f16=True)
        self.wsq_t = ops.pack_conv_weight(wsq.contiguous(), 1, flip=True)
        self.wex_t = ops.pack_conv_weight(wex.contiguous(), 3, flip=True)

    def fwd(self, x):
        n, h, w, _ = x.shape
        s = torch.empty(n, h, w, self.sp, dtype=F16, device=x.device)
        ops.conv_mfma(x, self.wsq, self.bsq, s, cout=self.sp, ksize=1, relu=True)      # ReLU fused into the store
        e = torch.empty(n, h, w, self.cout, dtype=F16, device=x.device)
        ops.conv_mfma(s, self.wex, self.bex, e, cout=self.cout, ksize=3, relu=True)
        return s, e

    def bwd(self, ge, s, e, g_next=None):
        """ge: gradient w.r.t. the module's (post-ReLU) output, consumed in place (``g_next``: a second gradient of the
        same output to be added first -- a tap's own gradient + the one from the layers after it) -> gradient w.r.t.
        its input."""
        n, h, w, _ = e.shape
        if g_next is not None:
            ops.relu_bwd_add_(ge, g_next, e)
        else:
            ops.relu_bwd_(ge, e)
        gs = torch.empty(n, h, w, self.sp, dtype=BF16, device=e.device)
        ops.conv_mfma(ge, self.wex_t, None, gs, cout=self.sp, ksize=3)
        ops.relu_bwd_(gs, s)
        gx = torch.empty(n, h, w, self.cin, dtype=BF16, device=e.device)
        ops.conv_mfma(gs, self.wsq_t, None, gx, cout=self.cin, ksize=1)
        return gx


class SqueezeTrunk:
    """Built from a ``models.perceptual.SqueezeLPIPS`` on the HIP device (weights are read once: the network is frozen)."""

    def __init__(self, net):
        self.fires = {i: _Fire(net.features[i]) for st in _PLAN if st[0] == "fire" for i in (st[1],)}
        self.c0 = net.features[0].out_channels
        self.w10 = fold_first_layer(net)

    def forward(self, x0, save: bool):
        """x0: NHWC fp16 [N, H, W, 64] (ReLU output of the first convolution) -> (taps 1..6 as NHWC fp16, saved)."""
        if x0.dtype != F16 or x0.dim() != 4 or x0.shape[3] != self.c0 or not x0.is_contiguous():
            raise ValueError("SqueezeTrunk.forward: expected a contiguous NHWC fp16 tensor with %d channels" % self.c0)
        taps, saved, x = [], [], x0
        for st in _PLAN:
            if st[0] == "pool":
                y = ops.maxpool3s2_fwd(x)
                if save:
                    saved.append(("pool", x, y))
                x = y
            else:
                s, e = self.fires[st[1]].fwd(x)
                if save:
                    saved.append(("fire", st[1], st[2], s, e))
                if st[2]:
                    taps.append(e)
                x = e
        return taps, saved

    def backward(self, saved, tap_grads):
        """tap_grads: NHWC bf16 gradients w.r.t. taps 1..6 (consumed) -> NHWC bf16 gradient w.r.t. x0."""
        g, k = None, len(tap_grads) - 1
        for st in reversed(saved):
            if st[0] == "pool":
                g = ops.maxpool3s2_bwd(g, st[1], st[2])
            else:
                _, idx, is_tap, s, e = st
                if is_tap:
                    ge, g_next = tap_grads[k], g
                    k -= 1
                else:
                    ge, g_next = g, None
                g = self.fires[idx].bwd(ge, s, e, g_next)
        return g
