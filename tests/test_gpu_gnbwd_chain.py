"""The GroupNorm backward between a ResBlock's two convs applied inside the first conv's data-gradient launch
(csrc/conv_mfma.hip, prologue PRO_GNB; C-ABI pti_conv2d_mfma_gnbwd_chain): against the unchained composition it
replaces -- pti_conv2d_mfma_gnbwd -> pti_gn_bwd_apply -> pti_conv2d_mfma_gnbwd, i.e. autograd of
GroupNorm -> SiLU -> Conv2d inside MONAI's AEKLResBlock (reference src/pti_ldm_vae/models/autoencoder.py:67-79) -- on the
same inputs, and end to end through the training step with the knob on and off."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("n,h,w,c", [(2, 16, 16, 128), (3, 13, 21, 128), (1, 8, 16, 256)])
def test_chained_launch_equals_apply_then_conv(dev, n, h, w, c):
    from pti_ldm_vae_amd import ops
    G = 16
    g = torch.Generator(device=dev).manual_seed(100 + h)
    dy = (torch.randn(n, h, w, c, device=dev, generator=g) * 0.3).bfloat16()
    h1 = (torch.randn(n, h, w, c, device=dev, generator=g) * 1.2 + 0.1).half()
    x = (torch.randn(n, h, w, c, device=dev, generator=g) * 0.9).half()
    w1 = torch.randn(c, c, 3, 3, device=dev, generator=g) * 0.03
    w2 = torch.randn(c, c, 3, 3, device=dev, generator=g) * 0.03
    gm1, bt1 = 1 + 0.2 * torch.randn(c, device=dev, generator=g), 0.1 * torch.randn(c, device=dev, generator=g)
    gm2, bt2 = 1 + 0.2 * torch.randn(c, device=dev, generator=g), 0.1 * torch.randn(c, device=dev, generator=g)
    wpt1 = ops.pack_conv_weight(w1, 3, ops.PTI_CONV_S1, flip=True)
    wpt2 = ops.pack_conv_weight(w2, 3, ops.PTI_CONV_S1, flip=True)
    st1, st2 = ops.gn_stats(x, G), ops.gn_stats(h1, G)
    assert ops.gnbwd_chain_supported(c, c, 3, x.dtype)
    # the layer above: d act2 -> g2 = d act2 * act'(GN2(h1)), sums2
    g2 = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=dev)
    sums2 = torch.zeros(n * c * 2, device=dev)
    ops.conv_mfma_gnbwd(dy, wpt2, h1, st2, gm2, bt2, g2, sums2, cout=c, groups=G, silu=True)
    # unchained: apply, then conv1's data gradient + GN1 backward reduction
    dh1 = torch.empty_like(g2)
    dg2, db2 = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.gn_bwd_apply(h1, g2, dh1, st2, gm2, bt2, sums2, dg2, db2, groups=G)
    g1 = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=dev)
    sums1 = torch.zeros(n * c * 2, device=dev)
    ops.conv_mfma_gnbwd(dh1, wpt1, x, st1, gm1, bt1, g1, sums1, cout=c, groups=G, silu=True)
    # chained
    dh1c, g1c = torch.empty_like(g2), torch.empty_like(g1)
    sums1c = torch.zeros(n * c * 2, device=dev)
    ops.conv_mfma_gnbwd_chain(g2, h1, st2, gm2, sums2, dh1c, wpt1, x, st1, gm1, bt1, g1c, sums1c, cout=c, groups=G, silu=True)
    dg2c, db2c = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.gn_affine_grads(sums2, dg2c, db2c, n, c)
    torch.cuda.synchronize()
    # the applied gradient: same formula, different association (a*g + b*h + d) -> bf16 outputs may differ by one rounding
    assert _rel(dh1c, dh1) <= 4e-3
    diff = (dh1c.float() - dh1.float()).abs()
    assert (diff > 2.0 ** -7 * dh1.float().abs().clamp_min(1e-3)).float().mean().item() <= 1e-3
    # conv1's data gradient on it, and its GroupNorm-backward sums
    assert _rel(g1c, g1) <= 6e-3
    assert _rel(sums1c, sums1) <= 6e-3
    # the affine gradients of the chained GroupNorm are bit-identical to pti_gn_bwd_apply's (same sums, same order)
    assert torch.equal(dg2c, dg2) and torch.equal(db2c, db2)
    # run to run
    dh1d, g1d = torch.empty_like(g2), torch.empty_like(g1)
    sums1d = torch.zeros(n * c * 2, device=dev)
    dg2d, db2d = torch.zeros(c, device=dev), torch.zeros(c, device=dev)     # ... with the affine gradients riding the finalize launch
    ops.conv_mfma_gnbwd_chain(g2, h1, st2, gm2, sums2, dh1d, wpt1, x, st1, gm1, bt1, g1d, sums1d, cout=c, groups=G, silu=True,
                              in_dgamma=dg2d, in_dbeta=db2d)
    torch.cuda.synchronize()
    assert torch.equal(dh1c, dh1d) and torch.equal(g1c, g1d) and torch.equal(sums1c, sums1d)
    assert torch.equal(dg2d, dg2) and torch.equal(db2d, db2)


def test_unsupported_shapes_are_refused(dev):
    from pti_ldm_vae_amd import ops
    assert not ops.gnbwd_chain_supported(64, 64, 3, torch.float16)
    assert not ops.gnbwd_chain_supported(128, 64, 3, torch.float16)
    assert not ops.gnbwd_chain_supported(128, 128, 3, torch.bfloat16)
    assert not ops.gnbwd_chain_supported(128, 128, 1, torch.float16)


def test_training_step_with_and_without_the_chain(dev, monkeypatch):
    """Config A at 64x64 (its 128-channel ResBlocks take the chained launch): loss identical (the forward is untouched),
    every gradient within 16-bit rounding of the unchained schedule's."""
    from oracle.autoencoderkl import CONFIG_A, build_oracle, synthetic_images
    from pti_ldm_vae_amd.models import VAEModel, compute_kl_loss
    x = synthetic_images(2, 1, 64, seed=42).to(dev)
    eps = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(43)).to(dev)
    sd = build_oracle(CONFIG_A, 42).state_dict()
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PTI_GNBWD_CHAIN", flag)
        model = VAEModel.from_config(CONFIG_A)
        model.load_state_dict(sd)
        model = model.to(dev)
        mu, sig = model.autoencoder.encode(x)
        rec = model.autoencoder.decode(mu + eps * sig)
        loss = torch.nn.functional.l1_loss(rec, x) + 1e-3 * compute_kl_loss(mu, sig)
        loss.backward()
        torch.cuda.synchronize()
        out[flag] = (loss.item(), {k: p.grad.detach().clone() for k, p in model.autoencoder.named_parameters()})
    assert out["1"][0] == out["0"][0]
    f1 = torch.cat([g.flatten() for g in out["1"][1].values()])
    f0 = torch.cat([out["0"][1][k].flatten() for k in out["1"][1]])
    cos = (f1.double() @ f0.double() / (f1.double().norm() * f0.double().norm())).item()
    assert cos >= 0.9999, cos
    assert _rel(f1, f0) <= 1e-2
