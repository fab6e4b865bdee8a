"""GPU parity tests (through the C-ABI) of the backward / bottleneck / optimiser kernels against
PyTorch fp32 autograd on CPU.  Tolerances: bf16 outputs max-abs <= 1% of the reference scale and
rel-L2 <= 3e-3; fp32 reductions rel-L2 <= 2e-3 when their inputs went through a bf16 prologue
(the oracle applies the same bf16 rounding of the activation), 1e-4 otherwise.
"""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import _gn_ref, _nhwc, _r, _report

pytestmark = pytest.mark.gpu

WG_CASES = [
    # n, cin, cout, h, w, ks, mode, prologue
    (2, 32, 32, 16, 16, 3, "s1", 2),
    (3, 32, 32, 24, 40, 3, "s1", 0),     # many tiles, several splits, ragged
    # plain bf16 inputs of stride-1 3x3 convs (the training step's saved activated inputs): the LDS-DMA ring kernel (v4)
    (2, 64, 64, 16, 16, 3, "s1", 0),
    (2, 128, 128, 16, 32, 3, "s1", 0),
    (1, 128, 64, 13, 19, 3, "s1", 0),    # ragged edges on both axes: zero-page pieces
    (1, 256, 256, 8, 16, 3, "s1", 0),    # 64 (co, ci) blocks, one pixel split
    (5, 32, 64, 40, 24, 3, "s1", 0),     # more tiles per workgroup than ring slots, tiles past the end
    (1, 32, 32, 8, 16, 3, "s1", 0),      # a single tile: falls back to one split
    (2, 64, 64, 16, 16, 3, "s1", 2),
    (2, 32, 64, 13, 19, 3, "s1", 2),
    (2, 128, 128, 16, 16, 3, "s1", 2),
    (1, 128, 64, 16, 16, 3, "s1", 2),
    (1, 256, 256, 8, 16, 3, "s1", 2),
    (2, 64, 64, 16, 32, 3, "s2", 0),
    (1, 128, 128, 16, 16, 3, "s2", 0),
    (2, 128, 128, 8, 8, 3, "up", 0),
    (2, 32, 64, 16, 16, 1, "s1", 0),
    (2, 128, 384, 8, 8, 1, "s1", 1),
]


@pytest.mark.parametrize("n,cin,cout,h,w,ks,mode,pro", WG_CASES)
def test_conv_wgrad_mfma(dev, n, cin, cout, h, w, ks, mode, pro):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(4)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.3 + 0.2)
    gamma = 1 + 0.2 * torch.randn(cin)
    beta = 0.1 * torch.randn(cin)
    a = _r(_gn_ref(x, groups, gamma, beta, eps, pro == 2)) if pro else x
    wt = torch.zeros(cout, cin, ks, ks, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    if mode == "s1":
        y = F.conv2d(a, wt, b, padding=ks // 2); m = ops.PTI_CONV_S1
    elif mode == "s2":
        y = F.conv2d(F.pad(a, (0, 1, 0, 1)), wt, b, stride=2); m = ops.PTI_CONV_S2PAD
    else:
        y = F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), wt, b, padding=1); m = ops.PTI_CONV_UP2
    dy = _r(torch.randn_like(y))
    y.backward(dy)
    xd = _nhwc(x).to(dev, torch.bfloat16)
    dyd = _nhwc(dy).to(dev, torch.bfloat16)
    dw = torch.full((cout, cin, ks, ks), float("nan"), device=dev)
    db = torch.full((cout,), float("nan"), device=dev)
    st = ops.gn_stats(xd, groups) if pro else None
    ops.conv_wgrad_mfma(xd, dyd, dw, db, ksize=ks, mode=m, prologue=pro, in_stats=st,
                        gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups, eps=eps)
    torch.cuda.synchronize()
    _report(f"wgrad_mfma[{mode},k{ks},{cin}->{cout},pro{pro}] dW", dw, wt.grad, max_frac=2e-3, l2=2e-4)
    _report("wgrad_mfma db", db, b.grad, max_frac=1e-4, l2=2e-5)
    # accumulate=True adds on top
    ops.conv_wgrad_mfma(xd, dyd, dw, db, ksize=ks, mode=m, prologue=pro, in_stats=st,
                        gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups, eps=eps,
                        accumulate=True)
    torch.cuda.synchronize()
    _report("wgrad_mfma accumulate", dw, 2 * wt.grad, max_frac=2e-3, l2=2e-4)


@pytest.mark.parametrize("n,c,h,w,g,silu,res", [(2, 32, 16, 16, 16, True, True), (2, 64, 9, 7, 16, True, False),
                                                (2, 128, 8, 8, 16, False, False), (1, 256, 8, 8, 32, True, True),
                                                (2, 32, 64, 64, 16, True, True)])
def test_gn_bwd(dev, n, c, h, w, g, silu, res):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(5)
    eps = 1e-6
    x = _r(torch.randn(n, c, h, w) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(c)).requires_grad_(True)
    beta = (0.1 * torch.randn(c)).requires_grad_(True)
    da = _r(torch.randn(n, c, h, w))
    dres = _r(torch.randn(n, c, h, w)) if res else None
    _gn_ref(x, g, gamma, beta, eps, silu).backward(da)
    ref_dx = x.grad + (dres if res else 0)
    xd = _nhwc(x.detach()).to(dev, torch.bfloat16)
    dad = _nhwc(da).to(dev, torch.bfloat16)
    st = ops.gn_stats(xd, g)
    dx = torch.full_like(xd, float("nan"))
    sums = torch.zeros(n, c, 2, device=dev)
    dg = torch.zeros(c, device=dev)
    dbt = torch.zeros(c, device=dev)
    ops.gn_bwd(xd, dad, dx, st, gamma.detach().to(dev), beta.detach().to(dev), sums, dg, dbt, groups=g, eps=eps, silu=silu,
               dres=_nhwc(dres).to(dev, torch.bfloat16) if res else None)
    torch.cuda.synchronize()
    _report("gn_bwd dx", dx.float().cpu().permute(0, 3, 1, 2), ref_dx, max_frac=1e-2, l2=3e-3)
    _report("gn_bwd dgamma", dg, gamma.grad, max_frac=1e-4, l2=2e-5)
    _report("gn_bwd dbeta", dbt, beta.grad, max_frac=1e-4, l2=2e-5)


def test_pool2x2_sum(dev):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(6)
    x = _r(torch.randn(2, 64, 12, 20))
    y = torch.empty(2, 6, 10, 64, dtype=torch.bfloat16, device=dev)
    ops.pool2x2_sum(_nhwc(x).to(dev, torch.bfloat16), y)
    torch.cuda.synchronize()
    _report("pool2x2", y.float().cpu().permute(0, 3, 1, 2), F.avg_pool2d(x, 2) * 4)


@pytest.mark.parametrize("b,l,hs", [(2, 4, 8), (1, 10, 8), (3, 4, 5), (2, 16, 16), (1, 3, 13)])
def test_latent_head(dev, b, l, hs):
    """encode tail + sampling + post_quant_conv, forward and backward, vs torch autograd."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(7)
    hw = hs * hs
    h = (torch.randn(b, l, hs, hs) * 2).requires_grad_(True)
    h.data[0, 0, 0, 0] = 100.0   # drive one log-variance through the clamp
    eps = torch.randn(b, l, hs, hs)
    ws = [(torch.randn(l, l, 1, 1) * 0.5).requires_grad_(True) for _ in range(3)]
    bs = [(torch.randn(l) * 0.1).requires_grad_(True) for _ in range(3)]
    ws[1].data[0, 0, 0, 0] = 1.0
    mu = F.conv2d(h, ws[0], bs[0])
    lv = torch.clamp(F.conv2d(h, ws[1], bs[1]), -30.0, 20.0)
    sg = torch.exp(lv / 2)
    zq = F.conv2d(mu + eps * sg, ws[2], bs[2])
    dzq, dmu, dsg = torch.randn_like(zq), torch.randn_like(mu), torch.randn_like(sg)
    (zq * dzq).sum().add((mu * dmu).sum()).add((sg * dsg).sum()).backward()
    hd = _nhwc(h.detach()).reshape(b, hw, l).contiguous().to(dev)
    wd = [w.detach().reshape(l, l).contiguous().to(dev) for w in ws]
    bd = [x.detach().to(dev) for x in bs]
    mu_d, sg_d, lv_d = (torch.empty(b, l, hs, hs, device=dev) for _ in range(3))
    zq_d = torch.empty(b, hw, l, device=dev)
    epsd = eps.to(dev)
    ops.latent_head_fwd(hd, epsd, wd[0], bd[0], wd[1], bd[1], wd[2], bd[2], mu_d, sg_d, lv_d, zq_d)
    torch.cuda.synchronize()
    _report("latent mu", mu_d, mu.detach(), max_frac=1e-5, l2=1e-6)
    _report("latent sigma", sg_d, sg.detach(), max_frac=1e-5, l2=1e-5)
    _report("latent logvar", lv_d, lv.detach(), max_frac=1e-5, l2=1e-6)
    _report("latent zq", zq_d.reshape(b, hs, hs, l).permute(0, 3, 1, 2), zq.detach(), max_frac=1e-5, l2=1e-5)
    dh = torch.empty(b, hw, l, device=dev)
    gw = [torch.zeros(l, l, device=dev) for _ in range(3)]
    gb = [torch.zeros(l, device=dev) for _ in range(3)]
    ops.latent_head_bwd(hd, epsd, wd[0], bd[0], wd[1], bd[1], wd[2], bd[2],
                        _nhwc(dzq).reshape(b, hw, l).contiguous().to(dev), dmu.to(dev), dsg.to(dev), dh,
                        gw[0], gb[0], gw[1], gb[1], gw[2], gb[2])
    torch.cuda.synchronize()
    _report("latent dh", dh.reshape(b, hs, hs, l).permute(0, 3, 1, 2), h.grad, max_frac=1e-4, l2=1e-5)
    for i, nm in enumerate(("mu", "log_sigma", "post")):
        _report(f"latent dW {nm}", gw[i], ws[i].grad.reshape(l, l), max_frac=1e-4, l2=1e-5)
        _report(f"latent db {nm}", gb[i], bs[i].grad, max_frac=1e-4, l2=1e-5)
    # deterministic path (eps = None) and decode-only entry
    ops.latent_head_fwd(hd, None, wd[0], bd[0], wd[1], bd[1], wd[2], bd[2], mu_d, sg_d, None, zq_d)
    zq2 = torch.empty(b, hw, l, device=dev)
    ops.post_quant(mu_d, wd[2], bd[2], zq2)
    torch.cuda.synchronize()
    _report("latent det zq", zq_d, F.conv2d(mu, ws[2], bs[2]).detach().permute(0, 2, 3, 1).reshape(b, hw, l), 1e-5, 1e-5)
    _report("post_quant", zq2, zq_d, 1e-6, 1e-6)


@pytest.mark.parametrize("l2,mode", [(False, 0), (True, 0), (False, 1)])
def test_vae_loss(dev, l2, mode):
    """recon + kl_weight*KL and its gradient seeds vs the oracle restatement of losses.py."""
    from oracle.losses import kl_loss
    from pti_ldm_vae_amd import ops
    torch.manual_seed(8)
    rec = torch.randn(3, 1, 32, 32, requires_grad=True)
    img = torch.randn(3, 1, 32, 32)
    mu = torch.randn(3, 4, 4, 4, requires_grad=True)
    third = (torch.rand(3, 4, 4, 4) + 0.3).requires_grad_(True)
    klw = 1e-3
    r = F.mse_loss(rec, img) if l2 else F.l1_loss(rec, img)
    k = kl_loss(mu, third, input_is_logvar=(mode == 0))
    (r + klw * k).backward()
    out = torch.zeros(2, device=dev)
    d_rec, d_mu, d_th = torch.empty_like(rec, device=dev), torch.empty_like(mu, device=dev), torch.empty_like(third, device=dev)
    ops.vae_loss(rec.detach().to(dev), img.to(dev), mu.detach().to(dev), third.detach().to(dev), out, d_rec, d_mu, d_th,
                 l2=l2, third_mode=mode, kl_weight=klw)
    torch.cuda.synchronize()
    _report("loss scalars", out, torch.stack([r.detach(), k.detach()]), 1e-5, 1e-5)
    _report("d_recon", d_rec, rec.grad, 1e-5, 1e-5)
    _report("d_mu", d_mu, mu.grad, 1e-5, 1e-5)
    _report("d_third", d_th, third.grad, 1e-4, 1e-5)


def test_adam_matches_torch(dev):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(9)
    p0 = torch.randn(10007)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=2.5e-5)
    p = p0.clone().to(dev)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.randn(10007) * 0.01
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(dev), m, v, lr=2.5e-5, step=step)
    torch.cuda.synchronize()
    _report("adam", p, p_ref.detach(), 1e-6, 1e-6)
    # the update itself (~lr) against torch, to one fp32 ulp of the O(1) parameters
    assert ((p.cpu() - p0) - (p_ref.detach() - p0)).abs().max().item() <= 2.4e-7


def test_casts(dev):
    from pti_ldm_vae_amd import ops
    x = _r(torch.randn(2, 3, 5, 7))
    y = torch.empty(2, 5, 7, 3, dtype=torch.bfloat16, device=dev)
    ops.cast_nchw_f32_to_nhwc_bf16(x.to(dev), y)
    z = torch.empty(2, 3, 5, 7, device=dev)
    ops.cast_nhwc_bf16_to_nchw_f32(y, z)
    torch.cuda.synchronize()
    assert torch.equal(z.cpu(), x)


def test_conv_wgrad_mfma_batched(dev):
    """pti_conv_wgrad_mfma_batched: heterogeneous jobs (different channel counts, map sizes, ragged edges, batch sizes) in one
    partial launch + one reduction launch; every job against autograd of F.conv2d; accumulate adds; a repeat is bitwise
    identical (fixed summation order)."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(41)
    shapes = [(2, 32, 32, 16, 16), (1, 64, 128, 13, 19), (3, 128, 64, 24, 16), (2, 128, 128, 8, 8), (1, 32, 64, 40, 24),
              (2, 256, 256, 8, 16), (1, 64, 64, 8, 16)]
    jobs, refs = [], []
    for n, cin, cout, h, w in shapes:
        x = _r(torch.randn(n, cin, h, w) * 1.2 + 0.1)
        wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        b = torch.zeros(cout, requires_grad=True)
        y = F.conv2d(x, wt, b, padding=1)
        dy = _r(torch.randn_like(y))
        y.backward(dy)
        dw = torch.full((cout * cin * 9,), float("nan"), device=dev)
        db = torch.full((cout,), float("nan"), device=dev)
        jobs.append((_nhwc(x).to(dev, torch.bfloat16), _nhwc(dy).to(dev, torch.bfloat16), dw, db))
        refs.append((wt.grad, b.grad))
    ops.conv_wgrad_mfma_batched(jobs, accumulate=False)
    torch.cuda.synchronize()
    first = [(j[2].clone(), j[3].clone()) for j in jobs]
    for (n, cin, cout, h, w), (_, _, dw, db), (rw, rb) in zip(shapes, jobs, refs):
        _report(f"batched wgrad [{cin}->{cout} {h}x{w} b{n}] dW", dw.view(cout, cin, 3, 3), rw, max_frac=2e-3, l2=2e-4)
        _report("batched wgrad db", db, rb, max_frac=1e-4, l2=2e-5)
    ops.conv_wgrad_mfma_batched(jobs, accumulate=True)
    torch.cuda.synchronize()
    for (_, _, dw, db), (rw, rb) in zip(jobs, refs):
        _report("batched wgrad accumulate", dw.view(rw.shape), 2 * rw, max_frac=2e-3, l2=2e-4)
    for j in jobs:
        j[2].fill_(float("nan")); j[3].fill_(float("nan"))
    ops.conv_wgrad_mfma_batched(jobs, accumulate=False)
    torch.cuda.synchronize()
    assert all(torch.equal(a, j[2]) and torch.equal(b, j[3]) for (a, b), j in zip(first, jobs))
    with pytest.raises(ValueError):
        ops.conv_wgrad_mfma_batched([])
    with pytest.raises(ValueError):
        ops.conv_wgrad_mfma_batched(jobs * 3)


def test_image_pad_and_slice_kernels(dev):
    """csrc/narrow_pad.hip: [N,C,H,W] fp32 -> zero-padded 32-channel NHWC 16-bit copies (one pass, two formats) and the
    slice back; both against plain torch indexing (bit-exact: the only arithmetic is the 16-bit rounding torch also does)."""
    from pti_ldm_vae_amd import ops
    for n, c, h, w in [(2, 3, 8, 16), (3, 1, 5, 7), (1, 8, 4, 4), (2, 2, 33, 17)]:
        x = torch.randn(n, c, h, w, device=dev)
        ya, yb = ops.pad_nchw_to_nhwc32(x, torch.float16, torch.bfloat16)
        ref = torch.zeros(n, h, w, 32, device=dev)
        ref[..., :c] = x.permute(0, 2, 3, 1)
        assert torch.equal(ya, ref.half()) and torch.equal(yb, ref.bfloat16())
        y1, none = ops.pad_nchw_to_nhwc32(x, torch.bfloat16)
        assert none is None and torch.equal(y1, ref.bfloat16())
        t = torch.randn(n, h, w, 32, device=dev).half()
        assert torch.equal(ops.slice_nhwc32_to_nchw(t, c), t[..., :c].permute(0, 3, 1, 2).float().contiguous())
        tb = t.bfloat16()
        assert torch.equal(ops.slice_nhwc32_to_nchw(tb, c), tb[..., :c].permute(0, 3, 1, 2).float().contiguous())
    with pytest.raises(Exception):
        ops.pad_nchw_to_nhwc32(torch.randn(1, 9, 4, 4, device=dev), torch.float16)     # more than 8 image channels


def test_direct_repack_one_launch(dev):
    """pti_direct_repack: the degenerate-channel convs' operands from the fp32 master weight -- [tap][ci][co], the
    tap-reversed [tap][co][ci] of the data gradient, and the zero-padded master copies the MFMA packer reads -- against
    torch permutes, several entries of different shapes in one launch."""
    from pti_ldm_vae_amd import ops
    ents, refs = [], []
    for cout, cin, pad in [(32, 1, None), (4, 128, "rows"), (128, 4, "cols"), (3, 32, "rows"), (32, 3, "cols")]:
        w = torch.randn(cout, cin, 3, 3, device=dev)
        b = torch.randn(cout, device=dev)
        e = dict(w=w, b=b, w_tck=torch.empty(9, cin, cout, device=dev), w_tck_t=torch.empty(9, cout, cin, device=dev))
        if pad == "rows":
            e["wpad"], e["bpad"] = torch.zeros(32, cin, 3, 3, device=dev), torch.zeros(32, device=dev)
        elif pad == "cols":
            e["wpad"] = torch.zeros(cout, 32, 3, 3, device=dev)
        ents.append(e)
        refs.append((w.permute(2, 3, 1, 0).reshape(9, cin, cout), w.flip(2, 3).permute(2, 3, 0, 1).reshape(9, cout, cin)))
    rp = ops.DirectRepack(ents)
    rp.run()
    torch.cuda.synchronize()
    for e, (r1, r2) in zip(ents, refs):
        assert torch.equal(e["w_tck"], r1) and torch.equal(e["w_tck_t"], r2)
        w = e["w"]
        if "wpad" in e:
            ref = torch.zeros_like(e["wpad"])
            ref[:w.shape[0], :w.shape[1]] = w
            assert torch.equal(e["wpad"], ref)
        if "bpad" in e:
            assert torch.equal(e["bpad"][:w.shape[0]], e["b"]) and float(e["bpad"][w.shape[0]:].abs().sum()) == 0.0
    with pytest.raises(ValueError):
        ops.DirectRepack([dict(w=torch.randn(4, 4, 3, 3, device=dev), wpad=torch.zeros(2, 4, 3, 3, device=dev))])
