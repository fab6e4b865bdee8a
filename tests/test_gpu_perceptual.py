"""GPU tests of the LPIPS comparison tail kernels (csrc/lpips.hip; reference: train_vae.py:299,:395-397 ->
monai PerceptualLoss("squeeze") -> lpips normalize_tensor / squared difference / lin layer / spatial mean).

The checker is ``oracle/perceptual.py`` (test infrastructure: a functional CPU restatement of the lpips / torchvision
structure over a state_dict; the product module is HIP-only and holds no torch formulation), in float64 for the tap
kernels and fp32 for the whole network, fed with the PRODUCT's state_dict.  Tolerances: the kernels compute in fp32 with a different summation order than torch -> value 1e-5
relative, gradient rel-L2 1e-5.  Parity of the whole term vs the reference stays UNPINNED (no weights, see
oracle/perceptual.py); these tests pin the kernels to the published formula."""
import pytest
import torch

from oracle import perceptual as OP


def _tap_hip(a, b, wt, gout, dev):
    """value [N] and gradient w.r.t. ``a`` of one tap through the NCHW tail kernels (what _TrunkCompareFn uses for tap 0)."""
    from pti_ldm_vae_amd import ops
    ad, bd, wd = a.to(dev).contiguous(), b.to(dev).contiguous(), wt.to(dev).contiguous()
    v, saved = ops.lpips_tap_fwd(ad, bd, wd)
    return v, ops.lpips_tap_bwd(ad, bd, wd, saved, gout.float().to(dev).contiguous())

pytestmark = pytest.mark.gpu

# (n, c, h, w): every tap shape class of SqueezeNet-1.1 at 256^2 input (odd map sizes, 1/2/4/8 channel slices) + ragged
SHAPES = [(2, 64, 127, 127), (3, 128, 63, 63), (2, 256, 31, 31), (4, 384, 15, 15), (2, 512, 15, 15), (1, 48, 5, 7),
          (2, 100, 9, 9), (1, 8, 1, 1)]


def _inputs(shape, seed, relu=True):
    g = torch.Generator().manual_seed(seed)
    n, c, h, w = shape
    a = torch.randn(n, c, h, w, generator=g)
    b = a + 0.3 * torch.randn(n, c, h, w, generator=g)
    if relu:
        a, b = a.relu(), b.relu()
    wt = torch.rand(c, generator=g) * 0.2
    return a, b, wt


@pytest.mark.parametrize("shape", SHAPES)
def test_tap_forward_and_gradient_vs_torch_float64(dev, shape):
    a, b, wt = _inputs(shape, seed=sum(shape))
    a64 = a.double().requires_grad_(True)
    v64 = OP.tap_distance(a64, b.double(), wt.double())
    gout = torch.linspace(0.5, 1.5, shape[0], dtype=torch.float64)
    g64, = torch.autograd.grad((v64 * gout).sum(), a64)
    v, gd = _tap_hip(a, b, wt, gout, dev)
    torch.cuda.synchronize()
    relv = ((v.cpu().double() - v64).abs() / v64.abs().clamp_min(1e-12)).max().item()
    relg = ((gd.cpu().double() - g64).norm() / g64.norm()).item()
    print(f"[lpips tap {shape}] value rel {relv:.2e}, grad relL2 {relg:.2e}")
    assert relv <= 1e-5 and relg <= 1e-5
    assert torch.isfinite(gd).all()


def test_tap_is_zero_on_identical_maps_and_bitwise_reproducible(dev):
    from pti_ldm_vae_amd import ops
    a, b, wt = _inputs((3, 128, 63, 63), seed=5)
    a, b, wt = a.to(dev), b.to(dev), wt.to(dev)
    v0, _ = ops.lpips_tap_fwd(a, a.clone(), wt)
    assert float(v0.abs().max()) == 0.0
    v1, s1 = ops.lpips_tap_fwd(a, b, wt)
    v2, s2 = ops.lpips_tap_fwd(a, b, wt)
    g = torch.ones(3, device=dev)
    assert torch.equal(v1, v2) and torch.equal(s1, s2)
    assert torch.equal(ops.lpips_tap_bwd(a, b, wt, s1, g), ops.lpips_tap_bwd(a, b, wt, s2, g))


def test_all_zero_pixels_give_finite_gradients(dev):
    """A pixel whose features are all zero after ReLU: torch's autograd formula yields 0/0 = NaN there; the kernel keeps
    the finite first term (documented in csrc/lpips.hip).  Everywhere else the two agree."""
    a, b, wt = _inputs((2, 64, 9, 9), seed=9)
    a[:, :, 4, 4] = 0.0
    a64 = a.double().requires_grad_(True)
    g64, = torch.autograd.grad(OP.tap_distance(a64, b.double(), wt.double()).sum(), a64)
    assert torch.isnan(g64[:, :, 4, 4]).all()          # the behaviour being documented
    _, gd = _tap_hip(a, b, wt, torch.ones(2), dev)
    assert torch.isfinite(gd).all()
    mask = torch.ones_like(a, dtype=torch.bool)
    mask[:, :, 4, 4] = False
    rel = ((gd.cpu().double()[mask] - g64[mask]).norm() / g64[mask].norm()).item()
    assert rel <= 1e-5


def test_bad_arguments_are_refused(dev):
    from pti_ldm_vae_amd import ops
    a = torch.zeros(2, 16, 4, 4, device=dev)
    with pytest.raises((ValueError, TypeError)):
        ops.lpips_tap_fwd(a, torch.zeros(2, 16, 4, 5, device=dev), torch.zeros(16, device=dev))
    with pytest.raises((ValueError, TypeError)):
        ops.lpips_tap_fwd(a, a, torch.zeros(8, device=dev))
    with pytest.raises((ValueError, TypeError)):
        ops.lpips_tap_fwd(a.half(), a.half(), torch.zeros(16, device=dev))
    with pytest.raises((ValueError, TypeError)):
        ops.lpips_tap_fwd(a.cpu(), a.cpu(), torch.zeros(16))


# ---- the trunk of the feature network on the HIP library (perceptual_engine.SqueezeTrunk) --------------------------------
def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("shape", [(2, 127, 127, 64), (1, 63, 63, 128), (3, 31, 31, 256), (2, 8, 10, 16), (1, 3, 3, 8), (1, 2, 5, 8)])
def test_maxpool_ceil_forward_backward_vs_torch(dev, shape):
    """MaxPool2d(3, 2, ceil_mode=True) on NHWC fp16 (odd sizes: every window complete; even sizes: clipped windows):
    forward bit-equal to torch, backward (gather) equal to torch's where the maximum is unique -- the inputs are
    continuous random values, so ties have probability ~0 -- and accumulation into an existing gradient."""
    import torch.nn.functional as F
    from pti_ldm_vae_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    # fp16-exact values that are pairwise distinct inside every 3x3 window (random fp16 values tie in ~1 % of windows,
    # where torch picks the first maximum and the gather shares the gradient)
    n_, h_, w_, c_ = shape
    ii = (torch.arange(h_).view(1, -1, 1, 1) * 37 + torch.arange(w_).view(1, 1, -1, 1) * 101
          + torch.arange(c_).view(1, 1, 1, -1) * 7 + torch.arange(n_).view(-1, 1, 1, 1) * 13) % 1024
    x = ((ii.float() - 512.0) / 64.0).half().to(dev)
    y = ops.maxpool3s2_fwd(x)
    xt = x.permute(0, 3, 1, 2).float().requires_grad_(True)
    yt = F.max_pool2d(xt, 3, 2, ceil_mode=True)
    assert tuple(y.shape) == (shape[0], yt.shape[2], yt.shape[3], shape[3])
    assert torch.equal(y.permute(0, 3, 1, 2).float(), yt.detach())
    gy = torch.randn(*y.shape, generator=g).to(dev).bfloat16()
    gxt, = torch.autograd.grad(yt, xt, gy.permute(0, 3, 1, 2).float())
    gx = ops.maxpool3s2_bwd(gy, x, y)
    # up to 4 bf16 gradients are summed in fp32 and rounded once; torch sums fp32 values exactly
    assert torch.allclose(gx.permute(0, 3, 1, 2).float(), gxt, rtol=1e-2, atol=1e-3)
    base = torch.randn(*shape, generator=g).to(dev).bfloat16()
    acc = ops.maxpool3s2_bwd(gy, x, y, gx=base.clone())
    assert torch.allclose(acc.float(), base.float() + gxt.permute(0, 2, 3, 1), rtol=2e-2, atol=2e-2)


def test_relu_passes(dev):
    from pti_ldm_vae_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 7, 16, generator=g).half().to(dev)
    y = ops.relu_f16_(x.clone())
    assert torch.equal(y, x.clamp_min(0))
    gr = torch.randn(3, 5, 7, 16, generator=g).to(dev).bfloat16()
    out = ops.relu_bwd_(gr.clone(), y)
    assert torch.equal(out, torch.where(y > 0, gr, torch.zeros_like(gr)))
    g2 = torch.randn(3, 5, 7, 16, generator=g).to(dev).bfloat16()
    out2 = ops.relu_bwd_add_(gr.clone(), g2, y)
    assert torch.equal(out2, torch.where(y > 0, (gr.float() + g2.float()).bfloat16(), torch.zeros_like(gr)))
    with pytest.raises((ValueError, TypeError, RuntimeError)):
        ops.relu_f16_(torch.zeros(7, device=dev).half())          # not a multiple of 8


@pytest.mark.parametrize("idx,hw", [(3, 63), (7, 31), (10, 15), (12, 15)])
def test_fire_module_forward_and_input_gradient_vs_torch(dev, idx, hw):
    """One Fire module on the HIP library (two MFMA convolutions, merged expands) vs the oracle's Fire in fp32 on the CPU: fp16 forward
    operands -> rel-L2 3e-3 forward (measured 4e-4); bf16 gradients (the incoming one is rounded to bf16 too) and ReLU
    masks taken from the fp16 squeeze output (a value within rounding of 0 flips its mask) -> gradient rel-L2 3e-2
    (measured 1.1-1.8e-2), cosine >= 0.9995."""
    from pti_ldm_vae_amd.models.perceptual import SqueezeLPIPS
    from pti_ldm_vae_amd.perceptual_engine import _Fire
    torch.manual_seed(idx)
    net = SqueezeLPIPS().to(dev)
    sd = OP.cpu_state(net)
    f = net.features[idx]
    cin = f.squeeze.in_channels
    x = torch.randn(2, cin, hw, hw).relu()
    xt = x.clone().requires_grad_(True)
    yt = OP.fire(sd, idx, xt)                                   # the checker: oracle Fire on the CPU, fp32
    gy = torch.randn_like(yt) * (yt > 0)
    gxt, = torch.autograd.grad(yt, xt, gy)
    x, gy, yt, gxt = x.to(dev), gy.to(dev), yt.detach().to(dev), gxt.to(dev)
    fire = _Fire(f)
    xn = x.half().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
    s, e = fire.fwd(xn)
    r_f = _rel(e.permute(0, 3, 1, 2).float(), yt.detach())
    gx = fire.bwd(gy.bfloat16().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).clone(), s, e)
    r_g = _rel(gx.permute(0, 3, 1, 2).float(), gxt)
    print(f"[fire {idx} @{hw}] forward relL2 {r_f:.2e}, input-gradient relL2 {r_g:.2e}")
    cos = torch.nn.functional.cosine_similarity(gx.permute(0, 3, 1, 2).flatten().double(), gxt.flatten().double(), dim=0).item()
    assert r_f <= 3e-3 and r_g <= 3e-2 and cos >= 0.9995


def test_whole_term_three_channel_path_vs_oracle(dev):
    """The whole perceptual term on three-channel inputs (torch first layer on the device, HIP trunk + tail) vs the oracle
    network on the CPU at 256x256: value within 5e-3 relative, gradient w.r.t. the reconstruction cosine >= 0.999 (16-bit
    feature maps vs fp32)."""
    from pti_ldm_vae_amd.models import PerceptualLoss
    torch.manual_seed(3)
    pl = PerceptualLoss(allow_random_init=True).to(dev)
    sd = OP.cpu_state(pl.net)
    g = torch.Generator().manual_seed(4)
    y = torch.rand(3, 1, 256, 256, generator=g)
    x = y + 0.1 * torch.randn(3, 1, 256, 256, generator=g)
    y, x = y.repeat(1, 3, 1, 1), x.repeat(1, 3, 1, 1)
    xo = x.clone().requires_grad_(True)
    l2 = OP.perceptual_loss(sd, xo, y)
    g2, = torch.autograd.grad(l2, xo)
    xd, yd = x.to(dev).requires_grad_(True), y.to(dev)
    l1 = pl(xd, yd)
    g1, = torch.autograd.grad(l1, xd)
    g1c = g1.cpu()
    cos = torch.nn.functional.cosine_similarity(g1c.flatten().double(), g2.flatten().double(), dim=0).item()
    print(f"[lpips term, three-channel path] {l1.item():.6e} vs oracle {l2.item():.6e}; grad cosine {cos:.6f}, relL2 {_rel(g1c, g2):.2e}")
    assert l1.item() == pytest.approx(l2.item(), rel=5e-3)
    assert cos >= 0.999 and torch.isfinite(g1).all()
    # split API used by the trainer: target taps first, then the comparison
    l3 = pl.from_taps(xd, pl.target_taps(yd))
    assert l3.item() == l1.item()
    # identical images: exactly zero
    assert pl(yd, yd).item() == 0.0


@pytest.mark.parametrize("shape", [(3, 128, 63, 63), (2, 256, 31, 31), (4, 384, 15, 15), (2, 512, 15, 15), (1, 64, 5, 7), (2, 1024, 3, 3)])
def test_tap_nhwc_kernels_vs_torch_float64(dev, shape):
    """The comparison kernels on the trunk's layout (NHWC fp16 maps, bf16 gradient) vs the torch formula in float64 on the
    SAME fp16-rounded values: value 1e-5 relative; gradient rel-L2 3e-3 (one bf16 rounding of the result)."""
    from pti_ldm_vae_amd import ops
    a, b, wt = _inputs(shape, seed=sum(shape) + 1)
    a, b = a.half(), b.half()
    a64 = a.double().requires_grad_(True)
    v64 = OP.tap_distance(a64, b.double(), wt.double())
    gout = torch.linspace(0.5, 1.5, shape[0], dtype=torch.float64)
    g64, = torch.autograd.grad((v64 * gout).sum(), a64)
    an = a.permute(0, 2, 3, 1).contiguous().to(dev)
    bn = b.permute(0, 2, 3, 1).contiguous().to(dev)
    v, sv = ops.lpips_tap_nhwc_fwd(an, bn, wt.to(dev))
    ga = ops.lpips_tap_nhwc_bwd(an, bn, wt.to(dev), sv, gout.float().to(dev))
    relv = ((v.cpu().double() - v64).abs() / v64.abs().clamp_min(1e-12)).max().item()
    relg = _rel(ga.cpu().permute(0, 3, 1, 2).float(), g64)
    print(f"[lpips tap nhwc {shape}] value rel {relv:.2e}, grad relL2 {relg:.2e}")
    assert relv <= 1e-5 and relg <= 3e-3
    v0, _ = ops.lpips_tap_nhwc_fwd(an, an.clone(), wt.to(dev))
    assert float(v0.abs().max()) == 0.0
    assert not ops.lpips_tap_nhwc_supported(40) and ops.lpips_tap_nhwc_supported(384)


def test_layout_kernels_at_the_trunk_boundary(dev):
    from pti_ldm_vae_amd import ops
    g = torch.Generator().manual_seed(2)
    for n, c, h, w in ((2, 64, 127, 127), (1, 128, 5, 9), (3, 64, 1, 1)):
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        y = ops.nchw_f32_to_nhwc_f16(x)
        assert torch.equal(y, x.permute(0, 2, 3, 1).half())
        gr = torch.randn(n, h, w, c, generator=g).to(dev).bfloat16()
        base = torch.randn(n, c, h, w, generator=g).to(dev)
        out = ops.nhwc_bf16_add_to_nchw_f32_(gr, base.clone())
        assert torch.equal(out, base + gr.permute(0, 3, 1, 2).float())
    with pytest.raises((ValueError, TypeError, RuntimeError)):
        ops.nchw_f32_to_nhwc_f16(torch.zeros(1, 48, 4, 4, device=dev))


def test_folded_first_layer_vs_torch(dev):
    """The first layer for a one-channel image (three scaled copies folded into a 1 -> 64 convolution, ReLU fused) vs the
    oracle's first layer + ReLU on the repeated, scaled image (CPU): forward rel-L2 1e-3 (fp16 output), gradient w.r.t. the one-channel
    image rel-L2 5e-3 (bf16 incoming gradient)."""
    from pti_ldm_vae_amd import ops
    from pti_ldm_vae_amd.models.perceptual import SqueezeLPIPS
    from pti_ldm_vae_amd.perceptual_engine import fold_first_layer
    torch.manual_seed(6)
    net = SqueezeLPIPS().to(dev)
    sd = OP.cpu_state(net)
    for n, h, w in ((2, 256, 256), (1, 17, 30), (3, 3, 3)):
        xc = torch.rand(n, 1, h, w)
        xt = xc.clone().requires_grad_(True)
        t0 = OP.feature_layer(sd, 1, OP.feature_layer(sd, 0, OP.scale_input(OP.three_channels(xt))))   # [n, 64, ho, wo] fp32, CPU
        x = xc.to(dev)
        w10 = fold_first_layer(net)
        y = ops.squeeze_conv1_fwd(x, w10)
        assert tuple(y.shape) == (n, t0.shape[2], t0.shape[3], 64)
        r_f = _rel(y.permute(0, 3, 1, 2).float().cpu(), t0.detach())
        # the incoming gradient is zero where the fp32 and the fp16 outputs may disagree about the sign (|t0| tiny)
        gy = torch.randn_like(t0) * (t0.detach() > 1e-2)
        gxt, = torch.autograd.grad(t0, xt, gy)
        t0, gy, gxt = t0.detach().to(dev), gy.to(dev), gxt.to(dev)
        dx = ops.squeeze_conv1_bwd(gy.permute(0, 2, 3, 1).contiguous().bfloat16(), y, w10, h, w)
        r_g = _rel(dx, gxt)
        # a gradient that already carries the mask (what the term's backward hands over): same result without t0
        gm = ops.relu_bwd_(gy.permute(0, 2, 3, 1).contiguous().bfloat16(), y)
        # (equal up to fp32 summation order: the two branches of the kernel add the eight products differently)
        d_nomask, d_mask = ops.squeeze_conv1_bwd(gm, None, w10, h, w), ops.squeeze_conv1_bwd(gm, y, w10, h, w)
        assert (d_nomask - d_mask).abs().max().item() <= 1e-5 * d_mask.abs().max().item()     # (absolute: sums with cancellation)
        print(f"[first layer {n}x{h}x{w}] forward relL2 {r_f:.2e}, gradient relL2 {r_g:.2e}")
        assert r_f <= 1e-3 and r_g <= 5e-3


def test_whole_term_one_channel_path_vs_oracle(dev):
    """One-channel images through PerceptualLoss: everything on the HIP library (folded first layer, trunk, comparison)
    vs the oracle network on the repeated image (CPU, fp32)."""
    from pti_ldm_vae_amd.models import PerceptualLoss
    torch.manual_seed(3)
    pl = PerceptualLoss(allow_random_init=True).to(dev)
    sd = OP.cpu_state(pl.net)
    g = torch.Generator().manual_seed(4)
    yc = torch.rand(3, 1, 256, 256, generator=g)
    xc = yc + 0.1 * torch.randn(3, 1, 256, 256, generator=g)
    xo = xc.clone().requires_grad_(True)
    l2 = OP.perceptual_loss(sd, xo, yc)
    g2, = torch.autograd.grad(l2, xo)
    y, x = yc.to(dev), xc.to(dev).requires_grad_(True)
    l1 = pl(x, y)
    g1, = torch.autograd.grad(l1, x)
    g1c = g1.cpu()
    cos = torch.nn.functional.cosine_similarity(g1c.flatten().double(), g2.flatten().double(), dim=0).item()
    print(f"[lpips term, one-channel path] {l1.item():.6e} vs oracle {l2.item():.6e}; grad cosine {cos:.6f}, relL2 {_rel(g1c, g2):.2e}")
    assert l1.item() == pytest.approx(l2.item(), rel=5e-3)
    assert cos >= 0.999 and tuple(g1.shape) == (3, 1, 256, 256) and torch.isfinite(g1).all()
    assert pl.from_taps(x, pl.target_taps(y)).item() == l1.item()
    assert pl(y, y).item() == 0.0
    # three-channel inputs keep the torch first layer + HIP trunk
    x3 = x.detach().repeat(1, 3, 1, 1)
    assert pl(x3, y.repeat(1, 3, 1, 1)).item() == pytest.approx(l2.item(), rel=5e-3)
    # mixed channel counts: both sides are taken as three-channel images
    assert pl(x3, y).item() == pytest.approx(l2.item(), rel=5e-3)


def test_fused_relu_store_of_the_plain_forward_conv(dev):
    """``relu_out``: y = max(conv + bias, 0) in the store of the plain fp16 forward convolution -- bit-equal to the same
    launch followed by the separate ReLU pass; refused on launches whose kernel does not carry it."""
    from pti_ldm_vae_amd import ops
    g = torch.Generator().manual_seed(8)
    for cin, cout, k, hw in ((64, 32, 1, 63), (32, 128, 3, 31), (64, 512, 3, 15)):
        w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        x = torch.randn(2, hw, hw, cin, generator=g).half().to(dev)
        wp = ops.pack_conv_weight(w, k, f16=True)
        y0 = torch.empty(2, hw, hw, cout, dtype=torch.float16, device=dev)
        y1 = torch.empty_like(y0)
        ops.conv_mfma(x, wp, b, y0, cout=cout, ksize=k)
        ops.conv_mfma(x, wp, b, y1, cout=cout, ksize=k, relu=True)
        assert torch.equal(y1, ops.relu_f16_(y0.clone())) and (y0 < 0).any()
    wpb = ops.pack_conv_weight(w, k)                                   # bf16 pack: a data-gradient style launch
    with pytest.raises(RuntimeError):
        ops.conv_mfma(x.bfloat16(), wpb, None, torch.empty(2, hw, hw, cout, dtype=torch.bfloat16, device=dev), cout=cout, ksize=k,
                      relu=True)


def test_train_script_with_local_perceptual_weight_files(dev, tmp_path):
    """``train_vae --perceptual-weights A B``: the weight files (written here with both packages' key names; random
    values -- the real ones cannot be fetched) are loaded into the module, the HIP trunk is packed from THEM (not from
    the constructor's random init), the term is part of the step (``train/perceptual_loss`` logged, > 0) and of the
    validation pass.  Also: the step with the term is bitwise reproducible."""
    import json
    import os
    from pti_ldm_vae_amd import train_vae
    from pti_ldm_vae_amd.models import PerceptualLoss, VAEModel
    from pti_ldm_vae_amd.models.perceptual import SqueezeLPIPS
    from pti_ldm_vae_amd.trainer import VAETrainer
    torch.manual_seed(11)
    src = SqueezeLPIPS()
    sd = src.state_dict()
    bb = {k: v for k, v in sd.items() if k.startswith("features.")}
    bb["classifier.1.weight"] = torch.zeros(1000, 512, 1, 1)            # torchvision's file also holds the classifier
    lin = {k: v.abs() for k, v in sd.items() if k.startswith("lin")}
    torch.save(bb, tmp_path / "squeezenet1_1.pth")
    torch.save(lin, tmp_path / "lpips_squeeze.pth")
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "config", "vae_dente_no_adv.json")))
    cfg["run_dir"] = str(tmp_path / "run")
    cfg["autoencoder_def"]["channels"] = [32, 64]
    cfg["autoencoder_def"]["attention_levels"] = [False, False]
    cfg["autoencoder_def"]["num_res_blocks"] = 1
    cfg["autoencoder_train"].update(batch_size=2, patch_size=[64, 64], max_epochs=2, perceptual_weight=1.0)
    cf = tmp_path / "cfg.json"
    cf.write_text(json.dumps(cfg))
    train_vae.main(["-c", str(cf), "--synthetic", "8", "--log-every", "1", "--perceptual-weights",
                    str(tmp_path / "squeezenet1_1.pth"), str(tmp_path / "lpips_squeeze.pth")])
    lines = [json.loads(l) for l in open(tmp_path / "run" / "metrics.jsonl")]
    p_train = [l["train/perceptual_loss"] for l in lines if "train/perceptual_loss" in l]
    assert p_train and all(p > 0 for p in p_train) and any("val/perceptual_loss" in l for l in lines)
    # the packed trunk follows the loaded weights
    pl = PerceptualLoss(weights=(str(tmp_path / "squeezenet1_1.pth"), str(tmp_path / "lpips_squeeze.pth"))).to(dev)
    x = torch.rand(2, 1, 64, 64, device=dev)
    y = torch.rand(2, 1, 64, 64, device=dev)
    v_native = pl(x, y).item()
    assert v_native == pytest.approx(OP.perceptual_loss(OP.cpu_state(pl.net), x.cpu(), y.cpu()).item(), rel=5e-3)
    # bitwise reproducible step with the term
    outs = []
    for _ in range(2):
        torch.manual_seed(5)
        m = VAEModel.from_config(dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4, channels=[32, 64],
                                      num_res_blocks=1, norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False],
                                      with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)).to(dev)
        p2 = PerceptualLoss(weights=(str(tmp_path / "squeezenet1_1.pth"), str(tmp_path / "lpips_squeeze.pth"))).to(dev)
        tr = VAETrainer(m, lr=1e-4, perceptual=p2, perceptual_weight=1.0)
        e = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(6)).to(dev)
        for _ in range(2):
            out = tr.step(x, e)
        torch.cuda.synchronize()
        outs.append((out["perceptual"].item(), m.autoencoder.param_arena.detach().clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


def test_trunk_entry_points_refuse_bad_arguments(dev):
    """Shapes the kernels do not cover are refused on the host, before any launch (the library never faults on them)."""
    from pti_ldm_vae_amd import ops
    from pti_ldm_vae_amd.models.perceptual import SqueezeLPIPS
    x12 = torch.zeros(1, 4, 4, 12, device=dev).half()                     # channels not a multiple of 8
    with pytest.raises(RuntimeError):
        ops.maxpool3s2_fwd(x12)
    with pytest.raises((ValueError, TypeError)):
        ops.maxpool3s2_fwd(torch.zeros(1, 4, 4, 16, device=dev))          # fp32 instead of fp16
    x = torch.zeros(1, 5, 5, 16, device=dev).half()
    y = ops.maxpool3s2_fwd(x)
    with pytest.raises((ValueError, TypeError)):
        ops.maxpool3s2_bwd(torch.zeros(1, 3, 3, 16, device=dev).bfloat16(), x, y)      # gy of the wrong size
    with pytest.raises((ValueError, TypeError)):
        ops.relu_bwd_(torch.zeros(8, device=dev).bfloat16(), torch.zeros(16, device=dev).half())
    with pytest.raises((ValueError, TypeError)):
        ops.squeeze_conv1_fwd(torch.zeros(1, 3, 8, 8, device=dev), torch.zeros(10, 64, device=dev))   # three channels
    with pytest.raises(RuntimeError):
        ops.squeeze_conv1_fwd(torch.zeros(1, 1, 2, 8, device=dev), torch.zeros(10, 64, device=dev))   # smaller than the kernel
    with pytest.raises((ValueError, TypeError)):
        ops.lpips_tap_nhwc_fwd(torch.zeros(1, 2, 2, 40, device=dev).half(), torch.zeros(1, 2, 2, 40, device=dev).half(),
                               torch.zeros(40, device=dev))               # 40 channels: no lane layout
    # the trunk itself refuses anything but a contiguous NHWC fp16 map with the first layer's width
    tr = SqueezeLPIPS().to(dev).trunk()
    with pytest.raises(ValueError):
        tr.forward(torch.zeros(1, 9, 9, 32, device=dev).half(), save=False)
    with pytest.raises(ValueError):
        tr.forward(torch.zeros(1, 64, 9, 9, device=dev).half().permute(0, 2, 3, 1), save=False)
