"""GPU parity of the PatchDiscriminator path (reference ``vae_scripts/train_vae.py:266-279,298,399-401,
447-458``) against ``oracle/patch_discriminator.py`` (restatement of MONAI's PatchDiscriminator(norm="INSTANCE") and
PatchAdversarialLoss("least_squares") -- parity UNPINNED w.r.t. MONAI itself, see that file's header).

  * every discriminator-specific kernel against plain torch (unfold / fold autograd, instance_norm, leaky_relu, mse);
  * the engine: logits, the generator term's gradient w.r.t. the image, and every parameter gradient of the
    discriminator loss against oracle autograd, at 64x64 and at the real 256x256 (logit map 30x30);
  * the drop-in ``nn.Module`` under autograd (``discriminator(x)[-1]`` + ``loss.backward()`` + ``torch.optim.Adam``);
  * bitwise reproducibility of forward + backward.
Tolerances (bf16 MFMA operands and bf16 storage of conv outputs / gradients, fp32 accumulation): logits max-abs <= 2 % of
their range; gradients cosine >= 0.999 and rel-L2 <= 3e-2; losses 2e-3 relative."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))


def _pair(dev, seed=0):
    from oracle.patch_discriminator import PatchDiscriminator as Oracle
    from pti_ldm_vae_amd.models import PatchDiscriminator
    torch.manual_seed(seed)
    ref = Oracle()
    with torch.no_grad():          # a trained-looking state: weights 5x the init scale so InstanceNorm inputs vary
        for p in ref.parameters():
            p.mul_(5.0)
    net = PatchDiscriminator()
    net.load_state_dict(ref.state_dict())
    return ref, net.to(dev)


# ---- kernels -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c,h,w,stride", [(32, 16, 16, 2), (64, 12, 20, 2), (128, 9, 9, 1), (256, 7, 6, 1)])
def test_im2col_and_col2im_vs_unfold(dev, c, h, w, stride):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(c + h)
    n = 2
    y = torch.randn(n, h, w, c).to(BF16)
    t = torch.stack([torch.randn(n, c) * 0.3, torch.rand(n, c) + 0.5], -1).contiguous()          # {mean, rstd}
    ho, wo = ops.pd_out_hw(h, w, stride)
    P = torch.empty(n, ho, wo, 16 * c, dtype=BF16, device=dev)
    ops.pd_im2col(y.to(dev), t.to(dev), P, stride=stride, act=True, slope=0.2)
    yf = y.float().requires_grad_(True)
    a = F.leaky_relu((yf - t[:, None, None, :, 0]) * t[:, None, None, :, 1], 0.2)                 # [n,h,w,c]
    cols = F.unfold(a.permute(0, 3, 1, 2), 4, padding=1, stride=stride)                           # [n, c*16, L] (c, ky, kx)
    want = cols.view(n, c, 16, ho, wo).permute(0, 3, 4, 2, 1).reshape(n, ho, wo, 16 * c)          # (ky*4+kx)*c + ch
    assert torch.equal(P.cpu(), want.detach().to(BF16)), "patch gather is a copy of bf16-rounded activations"
    # col2im: gradient of sum(P * dP) w.r.t. y, with the LeakyReLU' factor, and the InstanceNorm-backward sums
    dP = torch.randn(n, ho, wo, 16 * c).to(BF16)
    (want.float() * dP.float()).sum().backward()
    xhat = (y.float() - t[:, None, None, :, 0]) * t[:, None, None, :, 1]
    g_want = yf.grad / t[:, None, None, :, 1]          # d/d xhat (autograd went through the rstd scale as well)
    g = torch.empty(n, h, w, c, dtype=BF16, device=dev)
    g, sums = ops.pd_col2im(dP.to(dev), y.to(dev), t.to(dev), g, stride=stride, slope=0.2)
    torch.cuda.synchronize()
    assert _rel(g.cpu().float(), g_want) < 4e-3
    s_want = torch.stack([g_want.sum((1, 2)), (g_want * xhat).sum((1, 2))], -1)
    assert _rel(sums.cpu(), s_want) < 2e-3
    # no normalisation in front of the activation (first block): xhat = y, no sums
    g2 = torch.empty(n, h, w, c, dtype=BF16, device=dev)
    g2, none = ops.pd_col2im(dP.to(dev), y.to(dev), None, g2, stride=stride, slope=0.2)
    yf2 = y.float().requires_grad_(True)
    cols2 = F.unfold(F.leaky_relu(yf2, 0.2).permute(0, 3, 1, 2), 4, padding=1, stride=stride)
    w2 = cols2.view(n, c, 16, ho, wo).permute(0, 3, 4, 2, 1).reshape(n, ho, wo, 16 * c)
    (w2 * dP.float()).sum().backward()
    assert none is None and _rel(g2.cpu().float(), yf2.grad) < 4e-3


def test_image_patches_and_image_gradient(dev):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(3)
    n, h, w = 2, 20, 12
    x = torch.randn(n, 1, h, w)
    P = torch.empty(n, h // 2, w // 2, 32, dtype=BF16, device=dev)
    ops.pd_im2col_image(x.to(dev), P)
    xf = x.clone().requires_grad_(True)
    cols = F.unfold(xf, 4, padding=1, stride=2).view(n, 16, h // 2, w // 2).permute(0, 2, 3, 1)
    assert torch.equal(P.cpu()[..., :16], cols.detach().to(BF16)) and float(P[..., 16:].abs().max()) == 0.0
    dP = torch.randn(n, h // 2, w // 2, 32).to(BF16)
    (cols * dP[..., :16].float()).sum().backward()
    d_img = torch.full((n, 1, h, w), 2.0, device=dev)
    ops.pd_col2im_image(dP.to(dev), d_img, scale=0.5, accumulate=True)
    assert torch.allclose(d_img.cpu(), 2.0 + 0.5 * xf.grad, atol=1e-5, rtol=1e-5)
    ops.pd_col2im_image(dP.to(dev), d_img, scale=1.0, accumulate=False)
    assert torch.allclose(d_img.cpu(), xf.grad, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("c,h,w", [(64, 8, 8), (256, 31, 31), (32, 5, 3)])
def test_instance_norm_stats_and_backward(dev, c, h, w):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(c)
    n = 3
    y = (torch.randn(n, h, w, c) * 2 + 0.7).to(BF16)
    t = ops.pd_in_stats(y.to(dev), 1e-5).cpu()
    yf = y.float().requires_grad_(True)
    mean, var = yf.mean((1, 2)), yf.var((1, 2), unbiased=False)
    assert torch.allclose(t[..., 0], mean.detach(), atol=2e-5, rtol=1e-5)
    assert torch.allclose(t[..., 1], (var.detach() + 1e-5).rsqrt(), rtol=2e-4)
    out = F.instance_norm(yf.permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1)
    g = torch.randn(n, h, w, c).to(BF16)
    (out * g.float()).sum().backward()
    xhat = out.detach()
    sums = torch.stack([g.float().sum((1, 2)), (g.float() * xhat).sum((1, 2))], -1).contiguous()
    dy = ops.pd_in_bwd_apply(g.to(dev).clone(), y.to(dev), t.to(dev), sums.to(dev))
    torch.cuda.synchronize()
    assert _rel(dy.cpu().float(), yf.grad) < 5e-3


@pytest.mark.parametrize("real,fmt", [(True, "bf16"), (False, "bf16"), (True, "f16"), (True, "f32"), (False, "f32")])
def test_lsgan_vs_oracle(dev, real, fmt):
    """16-bit padded rows (logit in column 0 of 32) and the fp32 [M,1] form the direct final block produces."""
    from oracle.patch_discriminator import patch_adversarial_loss
    from pti_ldm_vae_amd import ops
    torch.manual_seed(5)
    m = 2 * 30 * 30
    lg = torch.randn(m) * 1.5 + 0.3
    if fmt == "f32":
        rows = lg.view(m, 1).clone()
        d = torch.full((m, 1), 9.0, device=dev)
    else:
        rows = torch.full((m, 32), 7.0)                  # the padding columns must not matter
        rows[:, 0] = lg
        rows = rows.to(torch.float16 if fmt == "f16" else BF16)
        d = torch.full((m, 32), 9.0, dtype=BF16, device=dev)
    loss = ops.pd_lsgan(rows.to(dev), target=1.0 if real else 0.0, slope=0.05, grad_scale=0.3 * 2.0 / m, d_logits=d)
    lf = rows[:, 0].float().requires_grad_(True)
    want = patch_adversarial_loss(lf.view(2, 1, 30, 30), target_is_real=real, for_discriminator=True)
    (0.3 * want).backward()
    assert abs(float(loss) - float(want.detach())) < 1e-5 * max(1.0, float(want.detach()))
    if fmt != "f32":
        assert float(d[:, 1:].abs().max()) == 0.0
    assert _rel(d[:, 0].cpu().float(), lf.grad) < (1e-6 if fmt == "f32" else 4e-3)   # bf16 rounding of the stored gradient


@pytest.mark.parametrize("c,h,w", [(256, 31, 31), (64, 7, 5), (32, 9, 12)])
def test_final_block_direct_kernels(dev, c, h, w):
    """The one-output-channel final block without a patch matrix: forward, data gradient (+ LeakyReLU' + InstanceNorm
    partial sums) and weight / bias gradient against torch autograd of conv2d(leaky_relu(instance_norm(y)))."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(c + h)
    n = 3
    y = (torch.randn(n, h, w, c) * 1.5 + 0.2).to(BF16)
    t = ops.pd_in_stats(y.to(dev), 1e-5)
    tc = t.cpu()
    wt = (torch.randn(1, c, 4, 4) * 0.05).requires_grad_(True)
    bias = torch.tensor([0.3], requires_grad=True)
    xh = ((y.float() - tc[:, None, None, :, 0]) * tc[:, None, None, :, 1]).requires_grad_(True)    # same table on both sides
    out = F.conv2d(F.leaky_relu(xh, 0.2).permute(0, 3, 1, 2), wt, bias, padding=1)
    dl = torch.randn_like(out)
    out.backward(dl)
    w16c = wt.detach().permute(0, 2, 3, 1).reshape(-1).contiguous().to(dev)        # [ky][kx][c]
    bd = torch.zeros(32, device=dev)
    bd[0] = 0.3
    logits = torch.empty(n, h - 1, w - 1, device=dev)
    ops.pd_final_fwd(y.to(dev), t, w16c, bd, logits, slope=0.2)
    assert _rel(logits.cpu(), out.detach()[:, 0]) < 1e-5
    g = torch.empty(n, h, w, c, dtype=BF16, device=dev)
    g, sums = ops.pd_final_dgrad(dl[:, 0].contiguous().to(dev), y.to(dev), t, w16c, g, slope=0.2)
    assert _rel(g.float().cpu(), xh.grad) < 4e-3
    s_want = torch.stack([xh.grad.sum((1, 2)), (xh.grad * xh.detach()).sum((1, 2))], -1)
    assert _rel(sums.cpu(), s_want) < 1e-4
    gw = ops.pd_final_wgrad(dl[:, 0].contiguous().to(dev), y.to(dev), t, slope=0.2).cpu()
    assert _rel(gw[:16 * c], wt.grad.permute(0, 2, 3, 1).reshape(-1)) < 1e-4
    assert abs(float(gw[16 * c]) - float(bias.grad)) < 1e-4 * max(1.0, abs(float(bias.grad)))
    assert float(gw[16 * c + 1:].abs().max()) == 0.0


# ---- engine ------------------------------------------------------------------------------------------------------------
def _pin_forward_state(ref, ctxs):
    """Forward hooks that make the oracle's conv outputs take the values the HIP pass stored (``ctx.y``), gradients
    flowing straight through: the oracle's autograd then differentiates at the SAME forward point.  Needed because the
    path has kinks: LeakyReLU(0.2)' is 1 or 0.2 (and LeakyReLU(0.05)' on the logits 1 or 0.05) depending on a sign, and
    with conv outputs rounded to bf16 ~0.5 % of the pre-activations sit on the other side of zero from the fp32
    oracle's -- sqrt(0.005) * 0.8 = 6 % gradient deviation per layer that says nothing about the backward kernels.
    ``ctxs``: the contexts of the successive oracle forward calls, in order."""
    blocks = list(ref.children())
    state = {"call": 0}
    handles = []

    def make(i):
        def hook(m, inp, out):
            ctx = ctxs[state["call"] // len(blocks)]
            state["call"] += 1
            yv = ctx.y[i]      # last block: fp32 logits [B,Ho,Wo]; the others: bf16 NHWC conv outputs
            y = yv.unsqueeze(1).cpu() if yv.dim() == 3 else yv[..., :out.shape[1]].float().permute(0, 3, 1, 2).cpu()
            return out + (y - out).detach()
        return hook
    for i, blk in enumerate(blocks):
        handles.append(blk.conv.register_forward_hook(make(i)))
    return handles


@pytest.mark.parametrize("size,batch", [(64, 2), (256, 2), (96, 3)])
def test_engine_vs_oracle(dev, size, batch):
    """(1) forward: logits and losses against the plain fp32 oracle; (2) backward: gradient w.r.t. the image (generator
    term) and every parameter gradient of the discriminator loss against oracle autograd evaluated at the HIP pass's
    forward state (see _pin_forward_state); (3) the same gradients against the un-pinned fp32 oracle, at the looser
    bound the kinks allow (measured cosine 0.994-0.996 at these sizes)."""
    from oracle.patch_discriminator import patch_adversarial_loss as pal
    ref, net = _pair(dev, seed=size)
    torch.manual_seed(size + 1)
    x = torch.randn(batch, 1, size, size) * 0.8
    real = torch.randn(batch, 1, size, size) * 0.8 + 0.2
    aw = 0.1
    eng = net.engine()
    xd, rd = x.to(dev), real.to(dev)
    ctx = eng.forward(xd, save=True)
    rctx = eng.forward(rd, save=True)
    torch.cuda.synchronize()

    def oracle_grads():
        xg = x.clone().requires_grad_(True)
        logits_o = ref(xg)[-1]
        gl_o = pal(logits_o, True, False)
        dx_o, = torch.autograd.grad(aw * gl_o, xg)
        ref.zero_grad(set_to_none=True)
        loss_f, loss_r = pal(ref(x)[-1], False, True), pal(ref(real)[-1], True, True)
        (0.5 * aw * (loss_f + loss_r)).backward()
        return logits_o.detach(), gl_o.detach(), loss_f.detach(), loss_r.detach(), dx_o, {n: p.grad.clone() for n, p in ref.named_parameters()}

    logits_o, gl_o, lf_o, lr_o, dx_free, grads_free = oracle_grads()
    # (1) forward
    logits = eng.logits(ctx).cpu()
    assert logits.shape == logits_o.shape
    span = float(logits_o.max() - logits_o.min())
    assert float((logits - logits_o).abs().max()) <= 2e-2 * span and _rel(logits, logits_o) <= 2e-2
    gl, d_gen = eng.lsgan(ctx, target_is_real=True, weight=aw)
    lf, d_fake = eng.lsgan(ctx, target_is_real=False, weight=0.5 * aw)
    lr, d_real = eng.lsgan(rctx, target_is_real=True, weight=0.5 * aw)
    assert float(gl) == pytest.approx(float(gl_o), rel=1e-2) and float(lf) == pytest.approx(float(lf_o), rel=1e-2)
    assert float(lr) == pytest.approx(float(lr_o), rel=1e-2)
    # HIP backward: generator term (image gradient only), then the discriminator step (fake pass reused + real pass)
    d_img = torch.zeros_like(xd)
    eng.backward(ctx, d_gen, want_wgrad=False, d_img=d_img)
    assert float(net.grad_arena.abs().max()) == 0.0, "want_wgrad=False must not touch the gradient arena"
    eng.backward(ctx, d_fake, want_wgrad=True)
    eng.backward(rctx, d_real, want_wgrad=True)
    torch.cuda.synchronize()
    d_img = d_img.cpu()
    # (2) oracle autograd at the HIP forward state: three oracle forward calls = (fake, fake, real)
    handles = _pin_forward_state(ref, [ctx, ctx, rctx])
    _, _, _, _, dx_pin, grads_pin = oracle_grads()
    for h in handles:
        h.remove()
    assert _cos(d_img, dx_pin) >= 0.9995 and _rel(d_img, dx_pin) <= 3e-2, (_cos(d_img, dx_pin), _rel(d_img, dx_pin))
    for name, go in grads_pin.items():
        gm = net.grad_view(name).cpu()
        assert gm.shape == go.shape
        assert _cos(gm, go) >= 0.9995 and _rel(gm, go) <= 3e-2, (name, _cos(gm, go), _rel(gm, go))
    # (3) against the free-running fp32 oracle
    worst = min([_cos(d_img, dx_free)] + [_cos(net.grad_view(n).cpu(), g) for n, g in grads_free.items()])
    print(f"[disc {size}x{size} b{batch}] vs un-pinned oracle: image-gradient cosine {_cos(d_img, dx_free):.4f}, worst {worst:.4f}")
    assert worst >= 0.985
    # the padding of the arena (first layer's 16 spare columns, last layer's 31 spare rows) received exactly zero
    real_elems = sum(net.grad_view(n).abs().sum() for n in grads_pin)
    assert abs(float(net.grad_arena.abs().sum()) - float(real_elems)) <= 1e-6 * float(real_elems)


def test_dropin_module_autograd_and_adam(dev):
    """``discriminator(x)[-1]`` under autograd, as train_vae.py:451-458 uses the MONAI module; one torch Adam step."""
    from oracle.patch_discriminator import patch_adversarial_loss as pal
    ref, net = _pair(dev, seed=11)
    torch.manual_seed(12)
    x, real = torch.randn(2, 1, 64, 64), torch.randn(2, 1, 64, 64) + 0.3
    opt_o, opt = torch.optim.Adam(ref.parameters(), lr=1e-3), torch.optim.Adam(net.parameters(), lr=1e-3)
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    for model, optim, put in ((ref, opt_o, lambda t: t), (net, opt, lambda t: t.to(dev))):
        optim.zero_grad(set_to_none=True)
        loss = 0.5 * (pal(model(put(x))[-1], False, True) + pal(model(put(real))[-1], True, True))
        loss.backward()
        optim.step()
    # Adam's first update is lr * sign(g): its cosine counts sign agreement, and the kinks (see _pin_forward_state) plus
    # gradients at the bf16 noise floor flip a few per cent of the signs -- gated at 0.85 per tensor; the gradients
    # themselves (p.grad after backward) at 0.985
    for (n, po), (_, p) in zip(ref.named_parameters(), net.named_parameters()):
        assert _cos(p.grad.cpu(), po.grad) >= 0.985, (n, _cos(p.grad.cpu(), po.grad))
        upd_o, upd = po.detach() - before[n], p.detach().cpu() - before[n]
        assert _cos(upd, upd_o) >= 0.85, (n, _cos(upd, upd_o))
    # the generator side: gradient flows to the input, parameters untouched when they do not require grad
    xg = x.to(dev).requires_grad_(True)
    out = net(xg)
    assert isinstance(out, list) and out[-1].shape == (2, 1, 6, 6)
    pal(out[-1], True, False).backward()
    assert xg.grad is not None and float(xg.grad.abs().sum()) > 0
    net.return_intermediates = True
    with torch.no_grad():
        outs, outs_o = net(x.to(dev)), ref(x)
    assert [tuple(o.shape) for o in outs] == [tuple(o.shape) for o in outs_o]


def test_forward_backward_bitwise_reproducible(dev):
    _, net = _pair(dev, seed=21)
    torch.manual_seed(22)
    x = torch.randn(4, 1, 128, 128, device=dev)
    eng = net.engine()
    res = []
    for _ in range(2):
        net.grad_arena.zero_()
        ctx = eng.forward(x, save=True)
        loss, d = eng.lsgan(ctx, target_is_real=False, weight=0.5)
        d_img = torch.zeros_like(x)
        eng.backward(ctx, d, want_wgrad=True, d_img=d_img)
        torch.cuda.synchronize()
        res.append((eng.logits(ctx).clone(), d_img.clone(), net.grad_arena.clone(), loss.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ---- the native training step with the adversarial branch on --------------------------------------------------------
def test_native_adversarial_step_vs_oracle(dev):
    """``VAETrainer.step(adversarial=True)`` (train_vae.py:385-458 with ``adv_enabled`` and ``epoch > 5``) at config A,
    64x64, batch 2, injected eps, against the oracle VAE + oracle discriminator: generator loss = L1 + kl_weight*KL +
    adv_weight * LSGAN(D(recon), real) -> backward -> Adam(G); then 0.5*(LSGAN(D(recon.detach()), fake) +
    LSGAN(D(images), real)) * adv_weight -> backward -> Adam(D).  adv_weight is chosen so that the adversarial term carries
    about 0.3 of the generator's gradient norm (a wrong sign or scale of it fails the cosine; the discriminator's
    own kink-limited accuracy, see _pin_forward_state, does not dominate it), and PatchAdversarialLoss runs with ``no_activation_leastsq=True`` on both sides: with MONAI's default
    LeakyReLU(0.05) on the logits, which of the 72 logits of this map sit on the other side of zero after 16-bit
    rounding decides the comparison (the default activation is covered by test_lsgan_vs_oracle)."""
    from oracle.autoencoderkl import CONFIG_A, build_oracle, synthetic_images
    from oracle.losses import train_step_losses
    from oracle.patch_discriminator import patch_adversarial_loss as pal
    from pti_ldm_vae_amd.models import VAEModel
    from pti_ldm_vae_amd.trainer import VAETrainer
    lr = 1e-4
    oracle = build_oracle(CONFIG_A, 42)
    model = VAEModel.from_config(CONFIG_A)
    model.load_state_dict(oracle.state_dict())
    model = model.to(dev)
    dref, dnet = _pair(dev, seed=31)
    x = synthetic_images(2, 1, 64, seed=7)
    lat = 64 // 2 ** (len(CONFIG_A["channels"]) - 1)
    eps = torch.randn(2, CONFIG_A["latent_channels"], lat, lat, generator=torch.Generator().manual_seed(8))

    opt_g, opt_d = torch.optim.Adam(oracle.parameters(), lr=lr), torch.optim.Adam(dref.parameters(), lr=lr)
    d0 = {n: p.detach().clone() for n, p in dref.named_parameters()}
    loss_o, rec_o, kl_o, (recon_o, _, _) = train_step_losses(oracle, x, eps)
    gen_o = pal(dref(recon_o)[-1], True, False, slope=1.0)
    params = list(oracle.parameters())
    g_plain = torch.cat([g.flatten() for g in torch.autograd.grad(loss_o, params, retain_graph=True)])
    g_adv = torch.cat([g.flatten() for g in torch.autograd.grad(gen_o, params, retain_graph=True)])
    aw = float(0.3 * g_plain.norm() / g_adv.norm())      # the adversarial term carries ~0.3 of the generator's gradient
    opt_g.zero_grad(set_to_none=True)
    (loss_o + aw * gen_o).backward()
    g_o = torch.cat([p.grad.flatten() for p in params])
    opt_g.step()
    opt_d.zero_grad(set_to_none=True)
    disc_o = 0.5 * (pal(dref(recon_o.detach())[-1], False, True, slope=1.0) + pal(dref(x)[-1], True, True, slope=1.0))
    (aw * disc_o).backward()
    gd_o = {n: p.grad.clone() for n, p in dref.named_parameters()}
    opt_d.step()

    tr = VAETrainer(model, lr=lr, discriminator=dnet, adv_weight=aw, adv_no_activation_leastsq=True)
    out = tr.step(x.to(dev), eps.to(dev), adversarial=True)
    torch.cuda.synchronize()
    ae = model.autoencoder
    g_h = torch.cat([ae.grad_view(n).detach().cpu().flatten() for n, _ in ae.named_parameters()])
    share = float((g_o - g_plain).norm() / g_o.norm())
    print(f"[adv step] gen {float(out['adv_gen']):.5f} vs {float(gen_o):.5f}  disc {float(out['adv_disc']):.5f} vs "
          f"{float(disc_o):.5f}  G-grad cosine {_cos(g_h, g_o):.5f} (without the adversarial term: {_cos(g_h, g_plain):.5f}; "
          f"its share of the gradient norm {share:.3f})")
    assert 0.15 < share < 0.6, "the test must be sensitive to the adversarial gradient without being dominated by it"
    # the discriminator here sees the HIP reconstruction, the oracle's sees the oracle's: with its weights at 5x the
    # initialisation scale it amplifies the 1e-3 reconstruction difference -- measured 2e-3 .. 6e-3 on the two losses
    assert float(out["adv_gen"]) == pytest.approx(float(gen_o), rel=1e-2)
    assert float(out["adv_disc"]) == pytest.approx(float(disc_o), rel=1e-2)
    assert float(out["loss"]) == pytest.approx(float(loss_o + aw * gen_o), rel=2e-3)
    # measured 0.9984 at a share of 0.37 (the discriminator part alone is kink-limited to ~0.99, see above); without
    # the adversarial gradient the cosine would be 0.93-0.95
    assert _cos(g_h, g_o) >= 0.998 and _cos(g_h, g_plain) < 0.97
    # discriminator gradients / Adam update against the free-running oracle: LeakyReLU kinks bound these (see
    # _pin_forward_state; the backward kernels themselves are held to 0.9995 in test_engine_vs_oracle)
    for n, go in gd_o.items():
        assert _cos(dnet.grad_view(n).cpu(), go) >= 0.985, (n, _cos(dnet.grad_view(n).cpu(), go))
    upd_o = torch.cat([(p.detach() - d0[n]).flatten() for n, p in dref.named_parameters()])
    upd_h = torch.cat([(p.detach().cpu() - d0[n]).flatten() for n, p in dnet.named_parameters()])
    assert _cos(upd_h, upd_o) >= 0.9
    # validation path: same terms under no_grad, nothing is updated
    before = dnet.param_arena.clone()
    res, _ = tr.eval_losses(x.to(dev), adversarial=True)
    assert torch.equal(before, dnet.param_arena) and float(res["adv_gen"]) > 0 and float(res["adv_disc"]) > 0


def test_adversarial_steps_are_bitwise_reproducible(dev):
    """Three optimiser steps with the adversarial branch on: generator AND discriminator parameters bit-identical run to
    run (no float atomics in the discriminator passes either)."""
    from pti_ldm_vae_amd.models import PatchDiscriminator, VAEModel
    from pti_ldm_vae_amd.trainer import VAETrainer
    small = dict(spatial_dims=2, in_channels=1, out_channels=1, latent_channels=4, channels=[32, 64], num_res_blocks=1,
                 norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False], with_encoder_nonlocal_attn=True,
                 with_decoder_nonlocal_attn=True)
    torch.manual_seed(3)
    x = torch.randn(4, 1, 128, 128, device=dev)
    eps = torch.randn(3, 4, 4, 64, 64, device=dev)
    torch.manual_seed(4)
    g0, d0 = VAEModel.from_config(small).state_dict(), PatchDiscriminator().state_dict()
    runs = []
    for _ in range(2):
        m, d = VAEModel.from_config(small).to(dev), PatchDiscriminator().to(dev)
        m.load_state_dict(g0)
        d.load_state_dict(d0)
        tr = VAETrainer(m, lr=1e-3, discriminator=d, adv_weight=0.5)
        outs = [tr.step(x, eps[i], adversarial=True) for i in range(3)]
        torch.cuda.synchronize()
        runs.append((m.autoencoder.param_arena.clone(), d.param_arena.clone(), outs[-1]["adv_disc"].clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][1]).all()


def test_train_script_with_adversarial_branch(dev, tmp_path):
    """train_vae.py with ``adv_enabled`` (the branch switched on from epoch 1 instead of 6 to keep the test short): the
    reference's discriminator files and checkpoint entries appear (train_vae.py:696-698,741-758), the adversarial
    metrics are logged, and a resumed run restores the discriminator and its optimiser."""
    import json
    import os
    from pti_ldm_vae_amd import train_vae
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "config", "vae_dente_no_adv.json")))
    cfg["run_dir"] = str(tmp_path / "run")
    cfg["autoencoder_def"].update(channels=[32, 64], attention_levels=[False, False], num_res_blocks=1)
    cfg["autoencoder_train"].update(batch_size=2, patch_size=[64, 64], max_epochs=3, perceptual_weight=0.0, adv_enabled=True,
                                    adv_weight=0.1)
    cf = tmp_path / "cfg.json"
    cf.write_text(json.dumps(cfg))
    train_vae.main(["-c", str(cf), "--synthetic", "8", "--log-every", "1", "--adv-start-epoch", "1"])
    wdir = tmp_path / "run" / "trained_weights"
    files = sorted(os.listdir(wdir))
    assert "discriminator_last.pt" in files and "autoencoder_last.pt" in files
    best = [f for f in files if f.startswith("checkpoint_epoch")][0]
    ep = best[16:-4]
    assert f"discriminator_epoch{ep}.pth" in files
    ck = torch.load(wdir / best, weights_only=True)
    dsd = ck["discriminator_state_dict"]
    assert list(dsd) == ["initial_conv.conv.weight", "initial_conv.conv.bias", "0.conv.weight", "1.conv.weight",
                         "2.conv.weight", "final_conv.conv.weight", "final_conv.conv.bias"]
    assert tuple(dsd["2.conv.weight"].shape) == (256, 128, 4, 4) and ck["optimizer_d_state_dict"] is not None
    from oracle.patch_discriminator import PatchDiscriminator as Oracle
    Oracle().load_state_dict(torch.load(wdir / "discriminator_last.pt", weights_only=True), strict=True)
    lines = [json.loads(l) for l in open(tmp_path / "run" / "metrics.jsonl")]
    adv = [l for l in lines if l.get("train/adv_disc_loss", 0.0) != 0.0]
    assert adv and all(l["train/adv_gen_loss"] > 0 for l in adv)
    cfg["resume_ckpt"], cfg["checkpoint_dir"] = True, str(wdir / best)
    cfg["autoencoder_train"]["max_epochs"] = 4
    cf.write_text(json.dumps(cfg))
    train_vae.main(["-c", str(cf), "--synthetic", "8", "--adv-start-epoch", "1"])


# ---- the perceptual term in the native step ----------------------------------------------------------------
def test_native_step_with_perceptual_term_vs_oracle(dev):
    """``VAETrainer(perceptual=PerceptualLoss(...), perceptual_weight=w).step`` against the oracle VAE + the oracle
    perceptual network (oracle/perceptual.py, same weights) on the CPU (random-initialised network with non-negative lin weights: the pretrained files cannot be fetched,
    so this checks the plumbing -- value, scaling, sign and that the gradient really reaches the HIP backward -- not
    LPIPS itself).  The weight is chosen so that the term carries about half of the generator gradient."""
    from oracle.autoencoderkl import CONFIG_A, build_oracle, synthetic_images
    from oracle.losses import train_step_losses
    from pti_ldm_vae_amd.models import PerceptualLoss, VAEModel
    from pti_ldm_vae_amd.trainer import VAETrainer
    torch.manual_seed(5)
    from oracle import perceptual as OP
    ploss = PerceptualLoss(allow_random_init=True)
    with torch.no_grad():
        for k in range(7):
            getattr(ploss.net, f"lin{k}").model[1].weight.abs_()
    psd = OP.cpu_state(ploss.net)                        # the CPU checker: oracle/perceptual.py on the same weights
    oracle = build_oracle(CONFIG_A, 42)
    model = VAEModel.from_config(CONFIG_A)
    model.load_state_dict(oracle.state_dict())
    model = model.to(dev)
    x = synthetic_images(2, 1, 64, seed=9)
    lat = 64 // 2 ** (len(CONFIG_A["channels"]) - 1)
    eps = torch.randn(2, CONFIG_A["latent_channels"], lat, lat, generator=torch.Generator().manual_seed(10))
    loss_o, _, _, (recon_o, _, _) = train_step_losses(oracle, x, eps)
    p_o = OP.perceptual_loss(psd, recon_o, x)
    params = list(oracle.parameters())
    g_plain = torch.cat([g.flatten() for g in torch.autograd.grad(loss_o, params, retain_graph=True)])
    g_p = torch.cat([g.flatten() for g in torch.autograd.grad(p_o, params, retain_graph=True)])
    w = float(g_plain.norm() / g_p.norm())
    g_o = g_plain + w * g_p
    import copy
    tr = VAETrainer(model, lr=1e-4, perceptual=copy.deepcopy(ploss).to(dev), perceptual_weight=w)
    out = tr.step(x.to(dev), eps.to(dev))
    torch.cuda.synchronize()
    ae = model.autoencoder
    g_h = torch.cat([ae.grad_view(n).detach().cpu().flatten() for n, _ in ae.named_parameters()])
    print(f"[perceptual step] p {float(out['perceptual']):.6f} vs {float(p_o):.6f}; grad cosine {_cos(g_h, g_o):.5f} "
          f"(without the term {_cos(g_h, g_plain):.5f})")
    assert float(out["perceptual"]) == pytest.approx(float(p_o), rel=2e-3)
    assert float(out["loss"]) == pytest.approx(float(loss_o + w * p_o), rel=2e-3)
    assert _cos(g_h, g_o) >= 0.999 and _cos(g_h, g_plain) < 0.95
    res, _ = tr.eval_losses(x.to(dev))
    assert float(res["perceptual"]) > 0
    with pytest.raises(ValueError):
        VAETrainer(model, lr=1e-4, perceptual_weight=1.0)


def test_engine_vs_frozen_golden_vectors(dev):
    """HIP discriminator against tests/golden/disc_golden.npz (oracle outputs frozen by oracle/make_golden.py on a seeded
    state and input): logits, the three least-squares losses, parameter-gradient norms of the discriminator loss."""
    import os
    import numpy as np
    from test_discriminator_cpu import _golden_state
    from pti_ldm_vae_amd.models import PatchDiscriminator
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "disc_golden.npz"))
    ref, x, real = _golden_state()
    net = PatchDiscriminator()
    net.load_state_dict(ref.state_dict())
    net = net.to(dev)
    eng = net.engine()
    ctx, rctx = eng.forward(x.to(dev), save=True), eng.forward(real.to(dev), save=True)
    logits = eng.logits(ctx).cpu().numpy()
    span = float(gold["logits"].max() - gold["logits"].min())
    assert float(np.abs(logits - gold["logits"]).max()) <= 2e-2 * span
    gen, _ = eng.lsgan(ctx, target_is_real=True, want_grad=False)
    fake, d_f = eng.lsgan(ctx, target_is_real=False, weight=0.5)
    realv, d_r = eng.lsgan(rctx, target_is_real=True, weight=0.5)
    assert float(gen) == pytest.approx(float(gold["gen"]), rel=1e-2)
    assert float(fake) == pytest.approx(float(gold["fake"]), rel=1e-2)
    assert float(realv) == pytest.approx(float(gold["real"]), rel=1e-2)
    eng.backward(ctx, d_f, want_wgrad=True)
    eng.backward(rctx, d_r, want_wgrad=True)
    torch.cuda.synchronize()
    for n, _ in net.named_parameters():
        want = float(gold["gnorm_" + n.replace(".", "_")])
        assert float(net.grad_view(n).norm()) == pytest.approx(want, rel=6e-2), n   # kink-limited, see _pin_forward_state


def test_discriminator_step_on_its_own_stream_is_bit_identical(dev):
    """The discriminator's step runs on its own stream beside the VAE backward (default) -- generator and discriminator
    parameters after three adversarial steps must equal, bit for bit, those of the serial schedule
    (``_adv_stream = None`` = PTI_ADV_STREAM=0), and the returned loss must be readable right after ``step``."""
    import dp_gpu_worker as W
    from pti_ldm_vae_amd.trainer import VAETrainer
    x, eps = W.fixed_inputs()
    x, eps = x.to(dev), eps.to(dev)
    res = []
    for own_stream in (True, False):
        model, disc = W.build_model(dev), W.build_disc(dev)
        tr = VAETrainer(model, lr=W.LR, discriminator=disc, adv_weight=0.1)
        assert tr._adv_stream is not None
        if not own_stream:
            tr._adv_stream = None
        losses = [float(tr.step(x, eps, adversarial=True)["adv_disc"]) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, model.autoencoder.param_arena.detach().clone(), disc.param_arena.detach().clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_full_objective_step_concurrent_terms_vs_serial_schedule(dev):
    """Perceptual + adversarial in one step: the generator's adversarial term (and then the discriminator's step) run on
    the discriminator's stream while the perceptual term runs on the main stream.  Against the serial schedule
    (``_adv_stream = None``): the adversarial gradient joins d_recon through one extra fp32 rounding (buffer + add
    instead of an accumulating store), so the parameters agree to ~1e-6 relative after two steps, not bit for bit;
    run to run the concurrent schedule IS bit-identical."""
    import dp_gpu_worker as W
    from pti_ldm_vae_amd.models import PerceptualLoss
    from pti_ldm_vae_amd.trainer import VAETrainer
    x, eps = W.fixed_inputs()
    x, eps = x.to(dev), eps.to(dev)
    torch.manual_seed(21)
    ploss = PerceptualLoss(allow_random_init=True).to(dev)
    res = []
    for mode in ("concurrent", "concurrent", "serial"):
        model, disc = W.build_model(dev), W.build_disc(dev)
        tr = VAETrainer(model, lr=W.LR, discriminator=disc, adv_weight=0.1, perceptual=ploss, perceptual_weight=1.0)
        if mode == "serial":
            tr._adv_stream = None
        for _ in range(2):
            out = tr.step(x, eps, adversarial=True)
        vals = [float(out[k]) for k in ("loss", "perceptual", "adv_gen", "adv_disc")]
        torch.cuda.synchronize()
        res.append((vals, model.autoencoder.param_arena.detach().clone(), disc.param_arena.detach().clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][0], res[2][0]):
        assert a == pytest.approx(b, rel=1e-4)
    # Adam's first steps move every parameter by ~lr whatever the gradient's size: compare against that scale
    assert (res[0][1] - res[2][1]).abs().max().item() <= 0.05 * W.LR
    assert (res[0][2] - res[2][2]).abs().max().item() <= 0.05 * W.LR
