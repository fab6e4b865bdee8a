"""The weight-stationary persistent conv kernel for the 128 -> 128 3x3 layers (csrc/conv_ws.hip) against (1) the v2 MFMA
kernel it stands in for -- BIT FOR BIT where no GroupNorm parameters are involved (both accumulate the 72 k-blocks in the
same order into the same fragments, so any difference is a bug in the persistent loop: halo double buffer, tile walk,
epilogue tables), within one ulp on a few elements where they are (see ``_same``) -- and (2) plain fp32 torch ops
on the CPU (the tolerances of tests/test_gpu_ops.py).  ``PTI_CONV_WS_MAX_WGS`` caps the grid so that every workgroup walks
several tiles on these small tensors (the path the full-size launches take: 1024-4096 tiles over 256 workgroups);
``PTI_CONV_WS=0`` selects the v2 kernel.  Both variables are read per launch.

Replaces nn.Conv2d(128,128,3) inside MONAI's AEKLResBlock / Upsample (reference src/pti_ldm_vae/models/autoencoder.py:67-79).
"""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import _gn_ref, _nhwc, _r, _report, _stats_ref

pytestmark = pytest.mark.gpu


def _both(monkeypatch, fn, cap, names_ok=True):
    """Run ``fn()`` (which launches and returns a tuple of tensors) with the v2 kernel, then with the weight-stationary
    kernel at the grid caps ``cap`` (0 = no cap); returns (v2 outputs, [ws outputs per cap])."""
    from pti_ldm_vae_amd import ops
    monkeypatch.setenv("PTI_CONV_WS", "0")
    ref = fn()
    name0 = ops.last_kernel_name()
    outs, names = [], []
    for c in cap:
        monkeypatch.setenv("PTI_CONV_WS", "1")
        if c:
            monkeypatch.setenv("PTI_CONV_WS_MAX_WGS", str(c))
        else:
            monkeypatch.delenv("PTI_CONV_WS_MAX_WGS", raising=False)
        outs.append(fn())
        names.append(ops.last_kernel_name())
    monkeypatch.delenv("PTI_CONV_WS_MAX_WGS", raising=False)
    monkeypatch.delenv("PTI_CONV_WS", raising=False)
    if names_ok:     # (the fused-backward wrapper ends with the finalize launch: its last kernel is not the conv)
        assert name0.startswith("conv_mfma2_kernel"), name0
        assert all(n.startswith("conv_ws128_kernel") for n in names), names
    return ref, outs


def _same(ref, outs, what, prologue=False):
    """No prologue: bit for bit.  With the GroupNorm prologue the two kernels derive scale / shift in separately compiled code
    (under -ffast-math the FMA contraction of var = E[x^2] - mean^2 and of the affine may differ): rstd / scale can differ by
    one ulp, which flips the 16-bit rounding of a few activations and through them of a few hundred outputs -- each by one
    ulp of its format.  Allowed: <= 2e-3 of the elements, each within 2^-8 of the tensor's scale."""
    for o in outs:
        for k, (a, b) in enumerate(zip(ref, o)):
            if a is None:
                continue
            if not prologue:
                assert torch.equal(a, b), f"{what}: output {k} differs from the v2 kernel, max |diff| {(a.float() - b.float()).abs().max().item():.3e}"
                continue
            d = (a.double() - b.double()).abs()
            nbad, scale = int((d > 0).sum()), a.double().abs().max().item()
            if a.dtype in (torch.int64, torch.float32):      # statistics / GroupNorm-backward sums: long sums of slightly different terms
                assert d.max().item() <= 2e-4 * scale, f"{what}: output {k}: max |diff| {d.max().item():.3e} (scale {scale:.3e})"
                continue
            assert nbad <= max(8, 2e-3 * d.numel()), f"{what}: output {k}: {nbad} of {d.numel()} elements differ"
            assert d.max().item() <= 2.0 ** -8 * scale, f"{what}: output {k}: max |diff| {d.max().item():.3e} (scale {scale:.3e})"


FWD_CASES = [
    # n, h, w, mode, prologue, residual, out_stats, fp16 storage (fp16 MFMA operands), groups
    (3, 24, 40, "s1", 2, True, True, True, 16),       # 3 x 3 x 3 = 27 tiles, side output
    (2, 13, 19, "s1", 2, False, True, True, 16),      # ragged on both axes
    (2, 16, 32, "s1", 2, True, False, True, 32),      # 4 channels per group: two groups per 8-channel piece
    (2, 16, 16, "s1", 1, False, True, True, 16),      # GroupNorm without SiLU
    (2, 16, 32, "s1", 0, True, True, True, 16),       # no prologue
    (2, 8, 24, "up", 0, False, True, True, 16),       # nearest-2x gather in the loader -> 16 x 48 output
    (2, 16, 32, "s1", 2, True, True, False, 16),      # bf16 storage + prologue: run-time-format launch, the WS kernel declines it
]


@pytest.mark.parametrize("n,h,w,mode,pro,res,ostats,f16,groups", FWD_CASES)
def test_forward_launches(dev, monkeypatch, n, h, w, mode, pro, res, ostats, f16, groups):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(h * 100 + w)
    eps, c = 1e-6, 128
    sdt = torch.float16 if f16 else torch.bfloat16
    rnd = (lambda t: t.to(sdt).float())
    x = rnd(torch.randn(n, c, h, w) * 1.3 + 0.2)
    wt = rnd(torch.randn(c, c, 3, 3) / (c * 9) ** 0.5)
    bias = torch.randn(c) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(c), 0.1 * torch.randn(c)
    a = x
    if pro:
        a = rnd(_gn_ref(x, groups, gamma, beta, eps, pro == 2))
    if mode == "s1":
        ref, m = F.conv2d(a, wt, bias, padding=1), ops.PTI_CONV_S1
    else:
        ref, m = F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), wt, bias, padding=1), ops.PTI_CONV_UP2
    rs = rnd(torch.randn_like(ref)) if res else None
    if res:
        ref = ref + rs
    xd = _nhwc(x).to(dev, sdt)
    wp = ops.pack_conv_weight(wt.to(dev), 3, m, f16=f16)
    ho, wo = ops.conv_out_hw(h, w, m)
    st = ops.gn_stats(xd, groups) if pro else None
    rsd = _nhwc(rs).to(dev, sdt) if res else None
    gd, bd, biasd = gamma.to(dev), beta.to(dev), bias.to(dev)
    save = bool(pro == 2 and mode == "s1")

    def run():
        y = torch.full((n, ho, wo, c), float("nan"), dtype=sdt, device=dev)
        ost = torch.zeros(n, 16, 2, dtype=torch.int64, device=dev) if ostats else None
        act = torch.full((n, h, w, c), float("nan"), dtype=torch.bfloat16, device=dev) if save else None
        ops.conv_mfma(xd, wp, biasd, y, cout=c, ksize=3, mode=m, prologue=pro, in_stats=st, gamma=gd if pro else None,
                      beta=bd if pro else None, groups=groups, eps=eps, residual=rsd, out_stats=ost, out_groups=16, act_out=act)
        torch.cuda.synchronize()
        return y, ost, act

    if not f16:
        monkeypatch.setenv("PTI_CONV_WS", "1")
        run()
        assert ops.last_kernel_name().startswith("conv_mfma2_kernel")
        return
    ref_out, outs = _both(monkeypatch, run, cap=(0, 2, 5))
    _same(ref_out, outs, f"fwd[{mode},pro{pro},res{res},g{groups}]", prologue=pro != 0)
    y, ost, act = outs[1]
    _report(f"conv_ws fwd[{mode},pro{pro}]", y.float().cpu().permute(0, 3, 1, 2), ref)
    if act is not None:
        _report("conv_ws act_out", act.float().cpu().permute(0, 3, 1, 2), a, max_frac=1e-2, l2=3e-3)
    if ostats:
        _report("conv_ws fused stats", ops.stats_to_float(ost), _stats_ref(y.float().cpu().permute(0, 3, 1, 2), 16), max_frac=1e-3, l2=1e-4)


@pytest.mark.parametrize("n,h,w,mode,res,pool", [(3, 24, 40, "s1", True, False), (2, 13, 19, "s1", False, False),
                                                 (2, 16, 32, "s1", False, True), (2, 8, 24, "zins", False, False)])
def test_plain_data_gradient_launches(dev, monkeypatch, n, h, w, mode, res, pool):
    """bf16 in / out, no prologue: the data gradient of a 3x3 conv (flipped pack), optionally + residual gradient, with the
    2x2-sum-pooled store (gradient of nearest-2x up-sampling), and the zero-insert gather (gradient of the stride-2 conv)."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(h + w)
    c = 128
    wf = _r(torch.randn(c, c, 3, 3) / (c * 9) ** 0.5)          # forward weight [cout, cin, 3, 3]
    dy = _r(torch.randn(n, c, h, w))
    if mode == "s1":
        xin = torch.zeros(n, c, h, w, requires_grad=True)
        F.conv2d(xin, wf, None, padding=1).backward(dy)
        ref, m = xin.grad, ops.PTI_CONV_S1
    else:
        xin = torch.zeros(n, c, 2 * h, 2 * w, requires_grad=True)
        F.conv2d(F.pad(xin, (0, 1, 0, 1)), wf, None, stride=2).backward(dy)
        ref, m = xin.grad, ops.PTI_CONV_ZINS
    rs = _r(torch.randn_like(ref)) if res else None
    if res:
        ref = ref + rs
    if pool:
        ref = F.avg_pool2d(ref, 2) * 4
    wp = ops.pack_conv_weight(wf.to(dev), 3, m, flip=True)
    dyd = _nhwc(dy).to(dev, torch.bfloat16)
    rsd = _nhwc(rs).to(dev, torch.bfloat16) if res else None
    ho, wo = ops.conv_out_hw(h, w, m)

    def run():
        shape = (n, ho // 2, wo // 2, c) if pool else (n, ho, wo, c)
        y = torch.full(shape, float("nan"), dtype=torch.bfloat16, device=dev)
        ops.conv_mfma(dyd, wp, None, y, cout=c, ksize=3, mode=m, residual=rsd, pool2=pool)
        torch.cuda.synchronize()
        return (y,)

    ref_out, outs = _both(monkeypatch, run, cap=(0, 3))
    _same(ref_out, outs, f"dgrad[{mode},res{res},pool{pool}]")
    _report(f"conv_ws dgrad[{mode}]", outs[1][0].float().cpu().permute(0, 3, 1, 2), ref, max_frac=1.5e-2 if pool else 1e-2,
            l2=4e-3 if pool else 2e-3)


@pytest.mark.parametrize("n,h,w,silu,groups", [(3, 24, 40, True, 16), (2, 13, 19, True, 16), (2, 16, 16, False, 32)])
def test_data_gradient_fused_with_groupnorm_backward(dev, monkeypatch, n, h, w, silu, groups):
    """``pti_conv2d_mfma_gnbwd``: dy_out = conv^T(dy_in) * act'(GN(gx)) and the per-tile partial sums {sum dy, sum dy*xhat};
    WS vs v2 bit for bit (outputs AND the finalised sums), and vs autograd of act(GN(gx)) on the CPU."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(3 * h + w)
    c, eps = 128, 1e-6
    wf = _r(torch.randn(c, c, 3, 3) / (c * 9) ** 0.5)
    dy = _r(torch.randn(n, c, h, w))
    gx = (torch.randn(n, c, h, w) * 1.3 + 0.2).half().float()
    gamma, beta = 1 + 0.2 * torch.randn(c), 0.1 * torch.randn(c)
    # reference: da = conv^T(dy);  dz = da * act'(GN(gx))  (the GroupNorm-backward APPLY is a separate launch)
    xin = torch.zeros(n, c, h, w, requires_grad=True)
    F.conv2d(xin, wf, None, padding=1).backward(dy)
    da = xin.grad
    if silu:
        zz = F.group_norm(gx, groups, gamma, beta, eps).clone().requires_grad_(True)
        F.silu(zz).backward(da)
        dz = zz.grad
    else:
        dz = da
    xg = gx.reshape(n, groups, -1)
    xhat = ((xg - xg.mean(-1, keepdim=True)) / (xg.var(-1, unbiased=False, keepdim=True) + eps).sqrt()).reshape(n, c, h, w)
    dzr = _r(dz)
    sums_ref = torch.stack([dzr.sum((2, 3)), (dzr * xhat).sum((2, 3))], -1)       # [n, c, 2]
    wp = ops.pack_conv_weight(wf.to(dev), 3, ops.PTI_CONV_S1, flip=True)
    dyd = _nhwc(dy).to(dev, torch.bfloat16)
    gxd = _nhwc(gx).to(dev, torch.float16)
    st = ops.gn_stats(gxd, groups)
    gd, bd = gamma.to(dev), beta.to(dev)

    def run():
        out = torch.full((n, h, w, c), float("nan"), dtype=torch.bfloat16, device=dev)
        sums = torch.full((n, c, 2), float("nan"), dtype=torch.float32, device=dev)
        ops.conv_mfma_gnbwd(dyd, wp, gxd, st, gd, bd, out, sums, cout=c, ksize=3, groups=groups, eps=eps, silu=silu)
        torch.cuda.synchronize()
        return out, sums

    ref_out, outs = _both(monkeypatch, run, cap=(0, 2, 7), names_ok=False)
    _same(ref_out, outs, f"dgrad+gnbwd[silu{silu},g{groups}]", prologue=True)    # (the epilogue's GroupNorm parameters: same caveat)
    out, sums = outs[1]
    _report("conv_ws gnbwd dy_out", out.float().cpu().permute(0, 3, 1, 2), dz, max_frac=1.5e-2, l2=4e-3)
    _report("conv_ws gnbwd sums", sums.cpu(), sums_ref, max_frac=2e-2, l2=1e-2)


def test_full_size_tile_walk_matches_v2_kernel(dev, monkeypatch):
    """Config A's most frequent 128-channel launches at full size (batch 32: 64^2 = 1024 tiles, 32^2 = 256 tiles): forward with
    prologue + side output + residual + statistics; every workgroup walks 4 tiles (or 1).  WS == v2 bit for bit."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(0)
    c, groups, eps = 128, 16, 1e-6
    for hw in (64, 32):
        n = 32
        xd = (torch.randn(n, hw, hw, c, device=dev) * 1.3 + 0.2).half()
        rsd = torch.randn(n, hw, hw, c, device=dev).half()
        wp = ops.pack_conv_weight((torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5), 3, ops.PTI_CONV_S1, f16=True)
        bias, gamma, beta = torch.randn(c, device=dev) * 0.1, 1 + 0.2 * torch.randn(c, device=dev), 0.1 * torch.randn(c, device=dev)
        st = ops.gn_stats(xd, groups)

        def run():
            y = torch.full((n, hw, hw, c), float("nan"), dtype=torch.float16, device=dev)
            ost = torch.zeros(n, 16, 2, dtype=torch.int64, device=dev)
            act = torch.full((n, hw, hw, c), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.conv_mfma(xd, wp, bias, y, cout=c, ksize=3, prologue=2, in_stats=st, gamma=gamma, beta=beta, groups=groups, eps=eps,
                          residual=rsd, out_stats=ost, out_groups=16, act_out=act)
            torch.cuda.synchronize()
            return y, ost, act

        ref_out, outs = _both(monkeypatch, run, cap=(0,))
        # Not bit for bit at this size: the two kernels evaluate var = E[x^2] - mean^2 of the GroupNorm prologue in separately
        # compiled code, and under -ffast-math the FMA contraction of that expression may differ -- rstd then differs by one
        # ulp for a few (sample, group) pairs, which flips the fp16 rounding of a handful of activations (measured: a few dozen of
        # 16.8 M side-output elements, ~7 k of 16.8 M outputs, each by one fp16 ulp).  Everything else is identical.
        (y0, st0, a0), (y1, st1, a1) = ref_out, outs[0]
        assert torch.isfinite(y1.float()).all()
        for name, t0, t1, frac in (("side output", a0, a1, 1e-4), ("output", y0, y1, 2e-3)):
            d = (t0.float() - t1.float()).abs()
            nbad = int((d > 0).sum())
            print(f"[ws full size {hw}^2] {name}: {nbad} of {d.numel()} elements differ, max |diff| {d.max().item():.2e}")
            assert nbad <= frac * d.numel() and d.max().item() <= 2.0 ** -9 * t0.float().abs().max().item()
        assert (st0 - st1).abs().max().item() <= 2e-4 * st0.abs().max().item()
