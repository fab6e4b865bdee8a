"""GPU parity of the fp16 STORAGE format of forward activations (pti_conv_desc.in_f16 / res_f16 / out_f16 and the
x_f16 / wide_f16 arguments): every kernel that reads or writes a forward activation must give the same result as on
bf16 storage, only with fp16's finer rounding.  MFMA operands, gradients and saved activated inputs stay bf16.

Inputs are pre-rounded to bf16 (exactly representable in fp16 at these magnitudes), references are fp32 CPU ops.
Tolerances: outputs stored as fp16 carry a 2^-11 rounding step instead of bf16's 2^-8, but the MFMA operands are
still bf16-rounded after the prologue, so conv outputs keep the 1e-2 / 3e-3 bounds of tests/test_gpu_ops.py;
pure format conversions (no prologue) are exact on these inputs and are held to 2e-3 / 5e-4."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import _gn_ref, _nhwc, _r, _report, _stats_ref

pytestmark = pytest.mark.gpu
H16 = torch.float16
B16 = torch.bfloat16


def _r16(t):
    return t.to(H16).float()


@pytest.mark.parametrize("n,cin,cout,h,w,ks,mode,pro,res,ostats", [
    (2, 32, 32, 16, 16, 3, "s1", 2, True, True),
    (2, 64, 64, 16, 32, 3, "s1", 2, True, True),
    (2, 128, 128, 16, 16, 3, "s1", 2, True, True),
    (1, 64, 32, 13, 19, 3, "s1", 2, False, True),
    (1, 256, 256, 8, 16, 3, "s1", 2, True, False),
    (2, 128, 128, 8, 8, 3, "up", 0, False, True),      # no prologue: fp16 -> bf16 operand conversion in the loader
    (1, 64, 64, 8, 16, 3, "up", 0, False, False),
    (2, 32, 64, 16, 16, 1, "s1", 0, False, False),      # nin_shortcut
    (2, 128, 384, 8, 8, 1, "s1", 1, False, False),      # q,k,v projection: fp16 in, bf16 out
    (1, 64, 64, 16, 32, 3, "s2", 0, False, True),       # v1 kernel (stride 2)
    (2, 32, 32, 16, 16, 3, "s2", 0, False, False),
])
def test_conv_mfma_fp16_storage(dev, n, cin, cout, h, w, ks, mode, pro, res, ostats):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(21)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.3 + 0.2)
    wt = _r(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    bias = torch.randn(cout) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(cin), 0.1 * torch.randn(cin)
    a = _r(_gn_ref(x, groups, gamma, beta, eps, pro == 2)) if pro else x
    if mode == "s1":
        ref, m = F.conv2d(a, wt, bias, padding=ks // 2), ops.PTI_CONV_S1
    elif mode == "s2":
        ref, m = F.conv2d(F.pad(a, (0, 1, 0, 1)), wt, bias, stride=2), ops.PTI_CONV_S2PAD
    else:
        ref, m = F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), wt, bias, padding=1), ops.PTI_CONV_UP2
    rs = _r(torch.randn_like(ref)) if res else None
    if res:
        ref = ref + rs
    out_dt = B16 if cout == 384 else H16
    xd = _nhwc(x).to(dev, H16)
    wp = ops.pack_conv_weight(wt.to(dev), ks, m)
    ho, wo = ops.conv_out_hw(h, w, m)
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=out_dt, device=dev)
    st = ops.gn_stats(xd, groups) if pro else None
    ost = torch.zeros(n, 16, 2, dtype=torch.int64, device=dev) if ostats else None
    act = torch.full((n, h, w, cin), float("nan"), dtype=B16, device=dev) if (pro and mode == "s1" and ks == 3) else None
    ops.conv_mfma(xd, wp, bias.to(dev), y, cout=cout, ksize=ks, mode=m, prologue=pro, in_stats=st,
                  gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups, eps=eps,
                  residual=_nhwc(rs).to(dev, H16) if res else None, out_stats=ost, out_groups=16, act_out=act)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    tight = not pro
    _report(f"conv_mfma fp16[{mode},k{ks},{cin}->{cout},pro{pro}]", got, ref, max_frac=2e-3 if tight else 1e-2,
            l2=5e-4 if tight else 3e-3)
    if ostats:   # statistics are those of the values as stored
        _report("fused stats (fp16 out)", ops.stats_to_float(ost), _stats_ref(got, 16), max_frac=1e-3, l2=1e-4)
    if act is not None:   # the saved activated input stays bf16 (it is the weight gradient's MFMA operand)
        _report("act_out (bf16)", act.float().cpu().permute(0, 3, 1, 2), a, max_frac=1e-2, l2=3e-3)


@pytest.mark.parametrize("n,cin,cout,h,w,ks,mode,pro,res,ostats", [
    (2, 32, 32, 16, 16, 3, "s1", 2, True, True),
    (2, 64, 64, 16, 32, 3, "s1", 2, True, True),
    (2, 128, 128, 16, 16, 3, "s1", 2, True, True),
    (1, 64, 32, 13, 19, 3, "s1", 2, False, True),
    (1, 256, 256, 8, 16, 3, "s1", 2, True, False),
    (2, 128, 128, 8, 8, 3, "up", 0, False, True),       # no prologue: the fp16 activation IS the operand
    (2, 32, 64, 16, 16, 1, "s1", 0, False, False),      # nin_shortcut
    (1, 64, 64, 16, 32, 3, "s2", 0, False, True),       # v1 kernel (stride 2)
    (2, 32, 32, 16, 16, 3, "s2", 0, False, False),
    (1, 256, 32, 8, 16, 3, "s1", 1, False, False),      # GroupNorm without SiLU: encoder conv_out on the padded latent tile
    (2, 128, 128, 8, 8, 3, "s1", 1, False, True),
    (2, 64, 64, 8, 8, 1, "s1", 1, False, False),
])
def test_conv_mfma_fp16_operands(dev, n, cin, cout, h, w, ks, mode, pro, res, ostats):
    """Forward convs of the default engine: fp16 storage AND fp16 MFMA operands (pti_conv_desc.w_f16, weights packed
    as fp16, v_mfma_f32_32x32x16_f16).  Inputs are bf16-representable (exact in fp16), weights are rounded to fp16 for
    the reference, so the only error sources are the fp16 rounding of the activated operand (2^-11) and of the output:
    max-abs <= 2e-3 of scale, rel-L2 <= 5e-4 -- 5-6x tighter than the bf16-operand bounds of the test above.  The saved
    activated input stays bf16 (weight-gradient operand)."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(31)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.3 + 0.2)
    wt = _r16(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    bias = torch.randn(cout) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(cin), 0.1 * torch.randn(cin)
    a = _gn_ref(x, groups, gamma, beta, eps, pro == 2) if pro else x
    if mode == "s1":
        ref, m = F.conv2d(a, wt, bias, padding=ks // 2), ops.PTI_CONV_S1
    elif mode == "s2":
        ref, m = F.conv2d(F.pad(a, (0, 1, 0, 1)), wt, bias, stride=2), ops.PTI_CONV_S2PAD
    else:
        ref, m = F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), wt, bias, padding=1), ops.PTI_CONV_UP2
    rs = _r(torch.randn_like(ref)) if res else None
    if res:
        ref = ref + rs
    xd = _nhwc(x).to(dev, H16)
    wp = ops.pack_conv_weight(wt.to(dev), ks, m, f16=True)
    assert wp.dtype == H16
    ho, wo = ops.conv_out_hw(h, w, m)
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=H16, device=dev)
    st = ops.gn_stats(xd, groups) if pro else None
    ost = torch.zeros(n, 16, 2, dtype=torch.int64, device=dev) if ostats else None
    act = torch.full((n, h, w, cin), float("nan"), dtype=B16, device=dev) if (pro == 2 and mode == "s1" and ks == 3) else None
    ops.conv_mfma(xd, wp, bias.to(dev), y, cout=cout, ksize=ks, mode=m, prologue=pro, in_stats=st,
                  gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups, eps=eps,
                  residual=_nhwc(rs).to(dev, H16) if res else None, out_stats=ost, out_groups=16, act_out=act)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    _report(f"conv_mfma f16 operands[{mode},k{ks},{cin}->{cout},pro{pro}]", got, ref, max_frac=2e-3, l2=5e-4)
    if ostats:
        _report("fused stats (fp16 out)", ops.stats_to_float(ost), _stats_ref(got, 16), max_frac=1e-3, l2=1e-4)
    if act is not None:
        _report("act_out (bf16)", act.float().cpu().permute(0, 3, 1, 2), a, max_frac=1e-2, l2=3e-3)


def test_fp16_operands_need_fp16_tensors(dev):
    """fp16-packed weights with a bf16 activation are refused before any launch (PTI_EUNSUPPORTED)."""
    from pti_ldm_vae_amd import ops
    from pti_ldm_vae_amd._lib import PtiError
    wp = ops.pack_conv_weight(torch.zeros(32, 32, 3, 3, device=dev), 3, f16=True)
    x = torch.zeros(1, 8, 8, 32, dtype=B16, device=dev)
    with pytest.raises(PtiError):
        ops.conv_mfma(x, wp, None, torch.empty(1, 8, 8, 32, dtype=H16, device=dev), cout=32)


def test_fp16_storage_saturates_instead_of_overflowing(dev):
    """the un-normalised residual stream is stored as fp16.  A residual branch scaled up to ~1e5 must not
    turn into inf (and then NaN through the next GroupNorm's statistics): the kernels that store fp16 run with
    MODE.FP16_OVFL, so the conversion saturates at +-65504 and everything downstream stays finite."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(33)
    n, c, h, w = 1, 32, 16, 16
    x = _r(torch.randn(n, c, h, w))
    wt = _r16(torch.randn(c, c, 3, 3) / (c * 9) ** 0.5)
    big = _r(torch.randn(n, c, h, w) * 6.0e4)                     # |residual| up to ~2e5 > fp16 max once added
    xd = _nhwc(x).to(dev, H16)
    rs = _nhwc(big).to(dev, B16)                                   # bf16 residual: representable input, fp16 output
    y = torch.full((n, h, w, c), float("nan"), dtype=H16, device=dev)
    ost = torch.zeros(n, 16, 2, dtype=torch.int64, device=dev)
    ops.conv_mfma(xd, ops.pack_conv_weight(wt.to(dev), 3), None, y, cout=c, residual=rs, out_stats=ost, out_groups=16)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all(), "fp16 store overflowed to inf"
    assert y.float().abs().max().item() == 65504.0                # saturated, not wrapped
    # and a consumer of that tensor (GroupNorm+SiLU prologue of the next conv) stays finite too
    y2 = torch.full((n, h, w, c), float("nan"), dtype=H16, device=dev)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    ops.conv_mfma(y, ops.pack_conv_weight(wt.to(dev), 3, f16=True), None, y2, cout=c, prologue=ops.PTI_PRO_GN_SILU,
                  in_stats=ops.gn_stats(y, 16), gamma=g, beta=b, groups=16, eps=1e-6)
    torch.cuda.synchronize()
    assert torch.isfinite(y2.float()).all()


def test_gn_stats_fp16(dev):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(22)
    x = _r(torch.randn(3, 64, 9, 7) * 2 + 0.5)
    st = ops.gn_stats(_nhwc(x).to(dev, H16), 16)
    torch.cuda.synchronize()
    _report("gn_stats fp16", ops.stats_to_float(st), _stats_ref(x, 16), max_frac=1e-4, l2=1e-5)


@pytest.mark.parametrize("n,c,h,w,silu,res", [(2, 32, 16, 16, True, True), (2, 128, 8, 8, False, False)])
def test_gn_bwd_fp16_input(dev, n, c, h, w, silu, res):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(23)
    g, eps = 16, 1e-6
    x = _r(torch.randn(n, c, h, w) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(c)).requires_grad_(True)
    beta = (0.1 * torch.randn(c)).requires_grad_(True)
    da = _r(torch.randn(n, c, h, w))
    dres = _r(torch.randn(n, c, h, w)) if res else None
    _gn_ref(x, g, gamma, beta, eps, silu).backward(da)
    xd = _nhwc(x.detach()).to(dev, H16)
    st = ops.gn_stats(xd, g)
    dx = torch.full(xd.shape, float("nan"), dtype=B16, device=dev)
    sums, dg, dbt = torch.zeros(n, c, 2, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.gn_bwd(xd, _nhwc(da).to(dev, B16), dx, st, gamma.detach().to(dev), beta.detach().to(dev), sums, dg, dbt,
               groups=g, eps=eps, silu=silu, dres=_nhwc(dres).to(dev, B16) if res else None)
    torch.cuda.synchronize()
    _report("gn_bwd dx (x fp16)", dx.float().cpu().permute(0, 3, 1, 2), x.grad + (dres if res else 0), max_frac=1e-2, l2=3e-3)
    _report("gn_bwd dgamma", dg, gamma.grad, max_frac=1e-4, l2=2e-5)
    _report("gn_bwd dbeta", dbt, beta.grad, max_frac=1e-4, l2=2e-5)


@pytest.mark.parametrize("n,cin,cout,h,w,ks,silu", [(2, 32, 32, 16, 16, 3, True), (1, 128, 128, 13, 19, 3, True),
                                                    (2, 128, 384, 8, 8, 1, False)])
def test_fused_gnbwd_fp16_gn_input(dev, n, cin, cout, h, w, ks, silu):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(24)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.4 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(cin)).requires_grad_(True)
    beta = (0.1 * torch.randn(cin)).requires_grad_(True)
    wt = _r(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    a = F.group_norm(x, groups, gamma, beta, eps)
    y = F.conv2d(F.silu(a) if silu else a, wt, None, padding=ks // 2)
    dy = _r(torch.randn_like(y))
    y.backward(dy)
    xd = _nhwc(x.detach()).to(dev, H16)
    st = ops.gn_stats(xd, groups)
    wpt = ops.pack_conv_weight(wt.to(dev), ks, ops.PTI_CONV_S1, flip=True)
    dyt = torch.full((n, h, w, cin), float("nan"), dtype=B16, device=dev)
    sums = torch.zeros(n, cin, 2, device=dev)
    g, b = gamma.detach().to(dev), beta.detach().to(dev)
    ops.conv_mfma_gnbwd(_nhwc(dy).to(dev, B16), wpt, xd, st, g, b, dyt, sums, cout=cin, ksize=ks, groups=groups, eps=eps,
                        silu=silu)
    dx = torch.full(xd.shape, float("nan"), dtype=B16, device=dev)
    dg, db = torch.zeros(cin, device=dev), torch.zeros(cin, device=dev)
    ops.gn_bwd_apply(xd, dyt, dx, st, g, b, sums, dg, db, groups=groups, eps=eps)
    torch.cuda.synchronize()
    _report("fused gnbwd dx (gx fp16)", dx.float().cpu().permute(0, 3, 1, 2), x.grad, max_frac=1.5e-2, l2=6e-3)
    _report("fused gnbwd dgamma", dg, gamma.grad, max_frac=1e-2, l2=5e-3)
    _report("fused gnbwd dbeta", db, beta.grad, max_frac=1e-2, l2=5e-3)


@pytest.mark.parametrize("n,cin,cout,h,w,ks,mode,pro", [
    (2, 32, 32, 16, 16, 3, "s1", 0), (2, 64, 64, 16, 16, 3, "s1", 2), (2, 128, 128, 8, 8, 3, "up", 0),
    (2, 64, 64, 16, 32, 3, "s2", 0), (2, 32, 64, 16, 16, 1, "s1", 0), (2, 128, 384, 8, 8, 1, "s1", 1)])
def test_conv_wgrad_fp16_input(dev, n, cin, cout, h, w, ks, mode, pro):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(25)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.3 + 0.2)
    gamma, beta = 1 + 0.2 * torch.randn(cin), 0.1 * torch.randn(cin)
    a = _r(_gn_ref(x, groups, gamma, beta, eps, pro == 2)) if pro else x
    wt = torch.zeros(cout, cin, ks, ks, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    if mode == "s1":
        y, m = F.conv2d(a, wt, b, padding=ks // 2), ops.PTI_CONV_S1
    elif mode == "s2":
        y, m = F.conv2d(F.pad(a, (0, 1, 0, 1)), wt, b, stride=2), ops.PTI_CONV_S2PAD
    else:
        y, m = F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), wt, b, padding=1), ops.PTI_CONV_UP2
    dy = _r(torch.randn_like(y))
    y.backward(dy)
    xd = _nhwc(x).to(dev, H16)
    dw = torch.full((cout, cin, ks, ks), float("nan"), device=dev)
    db = torch.full((cout,), float("nan"), device=dev)
    st = ops.gn_stats(xd, groups) if pro else None
    ops.conv_wgrad_mfma(xd, _nhwc(dy).to(dev, B16), dw, db, ksize=ks, mode=m, prologue=pro, in_stats=st,
                        gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups, eps=eps)
    torch.cuda.synchronize()
    _report(f"wgrad fp16 x [{mode},k{ks},{cin}->{cout},pro{pro}]", dw, wt.grad, max_frac=2e-3, l2=2e-4)
    _report("wgrad db", db, b.grad, max_frac=1e-4, l2=2e-5)


def test_conv_direct_fp16_wide_side(dev):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(26)
    n, h, w = 2, 16, 16
    # few-cin: fp32 NCHW image -> fp16 NHWC wide output
    x = torch.randn(n, 1, h, w)
    wt, b = torch.randn(32, 1, 3, 3) * 0.3, torch.randn(32) * 0.1
    y = torch.full((n, h, w, 32), float("nan"), dtype=H16, device=dev)
    ops.conv_direct(x.to(dev), wt.permute(2, 3, 1, 0).reshape(9, 1, 32).contiguous().to(dev), b.to(dev), y, n=n, h=h, w=w,
                    cin=1, cout=32, x_layout="nchw")
    torch.cuda.synchronize()
    _report("direct few-cin fp16 out", y.float().cpu().permute(0, 3, 1, 2), F.conv2d(x, wt, b, padding=1), max_frac=2e-3, l2=5e-4)
    # few-cout with GroupNorm prologue: fp16 NHWC wide input -> fp32 NCHW output
    xs = _r(torch.randn(n, 32, h, w) * 1.2 + 0.1)
    gamma, beta = 1 + 0.2 * torch.randn(32), 0.1 * torch.randn(32)
    w2, b2 = torch.randn(1, 32, 3, 3) * 0.1, torch.randn(1) * 0.1
    xd = _nhwc(xs).to(dev, H16)
    st = ops.gn_stats(xd, 16)
    out = torch.full((n, 1, h, w), float("nan"), device=dev)
    ops.conv_direct(xd, w2.permute(2, 3, 1, 0).reshape(9, 32, 1).contiguous().to(dev), b2.to(dev), out, n=n, h=h, w=w, cin=32,
                    cout=1, y_layout="nchw", prologue=ops.PTI_PRO_GN, in_stats=st, gamma=gamma.to(dev), beta=beta.to(dev),
                    groups=16, eps=1e-6)
    torch.cuda.synchronize()
    _report("direct few-cout fp16 in", out, F.conv2d(F.group_norm(xs, 16, gamma, beta, 1e-6), w2, b2, padding=1),
            max_frac=1e-4, l2=1e-5)
