"""GPU parity tests of the individual HIP kernels, called through the C-ABI (ctypes), against
plain PyTorch fp32 CPU ops on the same (bf16-representable) inputs.

Tolerances (stated here, floating point): inputs/weights are pre-rounded to bf16 so the only
differences are fp32 accumulation order and the final bf16 rounding of the output
(rel 2^-9 = 0.2%): atol = 1e-2 * max|ref|, checked on max-abs error, plus a 2e-3 relative
L2 bound.  Reductions (fp32 outputs) use rel L2 <= 1e-4.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _r(t):
    return t.to(torch.bfloat16).float()


def _report(name, got, ref, max_frac=1e-2, l2=2e-3):
    got = got.float().cpu()
    ref = ref.float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-12
    rel_l2 = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    idx = err.argmax().item()
    msg = (f"{name}: max|err|={err.max().item():.4e} (ref scale {scale:.3e}) relL2={rel_l2:.3e} "
           f"at flat idx {idx}: got {got.flatten()[idx].item():.5f} ref {ref.flatten()[idx].item():.5f} "
           f"nan={torch.isnan(got).sum().item()}")
    print(msg)
    assert not torch.isnan(got).any(), msg
    assert err.max().item() <= max_frac * scale, msg
    assert rel_l2 <= l2, msg


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _gn_ref(x, groups, gamma, beta, eps, silu):
    y = F.group_norm(x, groups, gamma, beta, eps)
    return F.silu(y) if silu else y


def _stats_ref(x, groups):
    n, c, h, w = x.shape
    xg = x.reshape(n, groups, -1).double()
    return torch.stack([xg.sum(-1), (xg * xg).sum(-1)], -1).float()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,h,w,g", [(2, 32, 16, 16, 16), (3, 64, 9, 7, 16), (2, 128, 8, 8, 16), (1, 256, 8, 8, 32),
                                       (2, 32, 64, 64, 16)])
def test_gn_stats(dev, n, c, h, w, g):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(0)
    x = _r(torch.randn(n, c, h, w) * 1.5 + 0.3)
    st = ops.gn_stats(_nhwc(x).to(dev, torch.bfloat16), g)
    torch.cuda.synchronize()
    _report("gn_stats", ops.stats_to_float(st), _stats_ref(x, g), max_frac=1e-4, l2=1e-5)


CONV_CASES = [
    # n, cin, cout, h, w, ksize, mode, prologue, residual, out_stats
    (2, 32, 32, 16, 16, 3, "s1", 0, False, False),
    (2, 32, 32, 16, 16, 3, "s1", 2, True, True),
    (1, 32, 64, 13, 19, 3, "s1", 2, False, True),      # ragged tile edges
    (2, 64, 64, 16, 32, 3, "s1", 2, True, False),
    (2, 64, 128, 8, 8, 3, "s1", 2, False, True),
    (2, 128, 128, 16, 16, 3, "s1", 2, True, True),
    (1, 128, 64, 16, 16, 3, "s1", 2, False, False),
    (1, 64, 32, 16, 16, 3, "s1", 2, False, False),
    (1, 256, 256, 8, 16, 3, "s1", 2, True, True),       # two cin chunks, two cout tiles (AR config)
    (2, 32, 32, 16, 16, 3, "s2", 0, False, False),
    (1, 64, 64, 16, 32, 3, "s2", 0, False, True),
    (1, 128, 128, 16, 16, 3, "s2", 0, False, False),
    (2, 128, 128, 8, 8, 3, "up", 0, False, True),
    (1, 64, 64, 8, 16, 3, "up", 0, False, False),
    (1, 32, 32, 8, 8, 3, "zins", 0, False, False),
    (1, 128, 128, 8, 8, 3, "zins", 0, False, False),
    (2, 32, 64, 16, 16, 1, "s1", 0, False, False),
    (2, 128, 64, 8, 16, 1, "s1", 0, True, False),
    (2, 128, 384, 8, 8, 1, "s1", 1, False, False),      # fused q,k,v projection with GN prologue
    (1, 128, 128, 8, 8, 1, "s1", 0, True, False),
]


@pytest.mark.parametrize("n,cin,cout,h,w,ks,mode,pro,res,ostats", CONV_CASES)
def test_conv_mfma(dev, n, cin, cout, h, w, ks, mode, pro, res, ostats):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(1)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.3 + 0.2)
    wt = _r(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    bias = torch.randn(cout) * 0.1
    gamma = 1 + 0.2 * torch.randn(cin)
    beta = 0.1 * torch.randn(cin)
    a = x
    if pro:
        a = _r(_gn_ref(x, groups, gamma, beta, eps, pro == 2))
    if mode == "s1":
        ref = F.conv2d(a, wt, bias, padding=ks // 2)
        m = ops.PTI_CONV_S1
    elif mode == "s2":
        ref = F.conv2d(F.pad(a, (0, 1, 0, 1)), wt, bias, stride=2)
        m = ops.PTI_CONV_S2PAD
    elif mode == "up":
        ref = F.conv2d(F.interpolate(a, scale_factor=2.0, mode="nearest"), wt, bias, padding=1)
        m = ops.PTI_CONV_UP2
    else:  # zins: data gradient of the stride-2 asymmetric-pad conv whose weight is wt^T (cin<->cout)
        # forward conv: in [n,cout,2h,2w] -> out [n,cin,h,w] with weight wf[cin, cout,3,3]
        wf = _r(torch.randn(cin, cout, 3, 3) / (cout * 9) ** 0.5)
        xin = torch.zeros(n, cout, 2 * h, 2 * w, requires_grad=True)
        out = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wf, None, stride=2)
        out.backward(a)
        ref = xin.grad + bias.view(1, -1, 1, 1)
        m = ops.PTI_CONV_ZINS
    rs = None
    if res:
        rs = _r(torch.randn_like(ref))
        ref = ref + rs
    xd = _nhwc(x).to(dev, torch.bfloat16)
    if mode == "zins":
        wp = ops.pack_conv_weight(wf.to(dev), 3, m, flip=True)
    else:
        wp = ops.pack_conv_weight(wt.to(dev), ks, m)
    ho, wo = ops.conv_out_hw(h, w, m)
    y = torch.full((n, ho, wo, cout), float("nan"), dtype=torch.bfloat16, device=dev)
    st = ops.gn_stats(xd, groups) if pro else None
    og = 16
    ost = torch.zeros(n, og, 2, dtype=torch.int64, device=dev) if ostats else None   # Q47.16 fixed-point sums
    # 3x3 S1 + prologue launches also exercise the side output act_out = prologue(x) (pti_conv2d_mfma_saveact)
    act = torch.full_like(xd, float("nan")) if (pro and mode == "s1" and ks == 3) else None
    ops.conv_mfma(xd, wp, bias.to(dev), y, cout=cout, ksize=ks, mode=m, prologue=pro, in_stats=st,
                  gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups, eps=eps,
                  residual=_nhwc(rs).to(dev, torch.bfloat16) if res else None, out_stats=ost, out_groups=og, act_out=act)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    _report(f"conv_mfma[{mode},k{ks},{cin}->{cout},pro{pro}]", got, ref)
    if act is not None:   # bf16 of an fp32 GN+SiLU: one rounding step of slack against the CPU reference
        _report("conv_mfma act_out", act.float().cpu().permute(0, 3, 1, 2), a, max_frac=1e-2, l2=3e-3)
    if ostats:
        _report("conv_mfma fused stats", ops.stats_to_float(ost), _stats_ref(_r(got), og), max_frac=1e-3, l2=1e-4)


def test_conv_mfma_rejects_bad_shapes(dev):
    from pti_ldm_vae_amd import ops
    from pti_ldm_vae_amd._lib import PtiError
    x = torch.zeros(1, 8, 8, 48, dtype=torch.bfloat16, device=dev)
    with pytest.raises((PtiError, ValueError)):
        ops.pack_conv_weight(torch.zeros(32, 48, 3, 3, device=dev), 3)
    y = torch.zeros(1, 8, 8, 32, dtype=torch.bfloat16, device=dev)
    wp = torch.zeros(32 * 32 * 9, dtype=torch.bfloat16, device=dev)
    with pytest.raises((PtiError, ValueError)):
        ops.conv_mfma(x, wp, None, y, cout=32)


DIRECT_CASES = [
    # n, cin, cout, h, w, prologue, in layout/dtype, out layout/dtype
    (2, 1, 32, 16, 16, 0, "nchw_f32", "nhwc_bf16"),    # encoder conv_in
    (2, 3, 64, 9, 11, 0, "nchw_f32", "nhwc_bf16"),     # 3-channel variant
    (2, 4, 128, 8, 8, 0, "nhwc_f32", "nhwc_bf16"),     # decoder conv_in
    (1, 10, 256, 8, 8, 0, "nhwc_f32", "nhwc_bf16"),    # AR decoder conv_in
    (2, 32, 1, 16, 16, 1, "nhwc_bf16", "nchw_f32"),    # decoder norm+conv_out
    (2, 64, 3, 12, 10, 1, "nhwc_bf16", "nchw_f32"),
    (2, 128, 4, 8, 8, 1, "nhwc_bf16", "nhwc_f32"),     # encoder norm+conv_out
    (1, 256, 10, 8, 8, 1, "nhwc_bf16", "nhwc_f32"),
    (2, 128, 4, 8, 8, 0, "nhwc_bf16", "nhwc_f32"),     # data gradient of decoder conv_in
]


@pytest.mark.parametrize("n,cin,cout,h,w,pro,il,ol", DIRECT_CASES)
def test_conv_direct(dev, n, cin, cout, h, w, pro, il, ol):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(2)
    groups, eps = (16 if cin % 16 == 0 else 1), 1e-6
    x = torch.randn(n, cin, h, w) * 1.2 + 0.1
    if il.endswith("bf16"):
        x = _r(x)
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    bias = torch.randn(cout) * 0.1
    gamma = 1 + 0.2 * torch.randn(cin)
    beta = 0.1 * torch.randn(cin)
    a = _gn_ref(x, groups, gamma, beta, eps, pro == 2) if pro else x
    ref = F.conv2d(a, wt, bias, padding=1)
    w_tck = wt.permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous().to(dev)
    if il == "nchw_f32":
        xd, xl = x.to(dev), "nchw"
    elif il == "nhwc_f32":
        xd, xl = _nhwc(x).to(dev), "nhwc"
    else:
        xd, xl = _nhwc(x).to(dev, torch.bfloat16), "nhwc"
    if ol == "nhwc_bf16":
        y, yl = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device=dev), "nhwc"
    elif ol == "nchw_f32":
        y, yl = torch.full((n, cout, h, w), float("nan"), device=dev), "nchw"
    else:
        y, yl = torch.full((n, h, w, cout), float("nan"), device=dev), "nhwc"
    st = ops.gn_stats(xd, groups) if pro else None
    ops.conv_direct(xd, w_tck, bias.to(dev), y, n=n, h=h, w=w, cin=cin, cout=cout, x_layout=xl, y_layout=yl,
                    prologue=pro, in_stats=st, gamma=gamma.to(dev) if pro else None,
                    beta=beta.to(dev) if pro else None, groups=groups, eps=eps)
    torch.cuda.synchronize()
    got = y.float().cpu()
    if yl == "nhwc":
        got = got.permute(0, 3, 1, 2)
    tol = dict(max_frac=1e-2, l2=3e-3) if ol.endswith("bf16") else dict(max_frac=1e-4, l2=2e-5)
    _report(f"conv_direct[{cin}->{cout},pro{pro}]", got, ref, **tol)


@pytest.mark.parametrize("kind,n,cw,cn,h,w,pro", [("fewcout", 2, 32, 1, 16, 16, 1), ("fewcout", 2, 128, 4, 8, 8, 1),
                                                   ("fewcin", 2, 32, 1, 16, 16, 0), ("fewcin", 2, 128, 4, 8, 8, 0),
                                                   ("fewcin", 1, 64, 3, 10, 12, 0)])
def test_wgrad_direct(dev, kind, n, cw, cn, h, w, pro):
    """Weight/bias gradients of the degenerate-channel convs vs torch autograd."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(3)
    groups, eps = 16, 1e-6
    gamma = 1 + 0.2 * torch.randn(cw)
    beta = 0.1 * torch.randn(cw)
    if kind == "fewcout":   # y[cn] = conv(P(x[cw])), wide = x, narrow = dY
        x = _r(torch.randn(n, cw, h, w) + 0.2)
        wt = torch.randn(cn, cw, 3, 3, requires_grad=True)
        b = torch.zeros(cn, requires_grad=True)
        a = _gn_ref(x, groups, gamma, beta, eps, pro == 2) if pro else x
        dy = torch.randn(n, cn, h, w)
        F.conv2d(a, wt, b, padding=1).backward(dy)
        wide = _nhwc(x).to(dev, torch.bfloat16)
        narrow = dy.to(dev)  # NCHW fp32
        dw = torch.zeros(cn, cw, 3, 3, device=dev)
        db = torch.zeros(cn, device=dev)
        st = ops.gn_stats(wide, groups) if pro else None
        ops.wgrad_direct(wide, narrow, dw, n=n, h=h, w=w, cw=cw, cn=cn, ksize=3, sgn=1, narrow_layout="nchw",
                         dw_strides=(1, 9, cw * 9), dbias_narrow=db, prologue=pro, in_stats=st,
                         gamma=gamma.to(dev) if pro else None, beta=beta.to(dev) if pro else None, groups=groups,
                         eps=eps)
    else:                   # y[cw] = conv(x[cn]), wide = dY (bf16), narrow = x
        x = torch.randn(n, cn, h, w)
        wt = torch.randn(cw, cn, 3, 3, requires_grad=True)
        b = torch.zeros(cw, requires_grad=True)
        dy = _r(torch.randn(n, cw, h, w))
        F.conv2d(x, wt, b, padding=1).backward(dy)
        wide = _nhwc(dy).to(dev, torch.bfloat16)
        narrow = x.to(dev)
        dw = torch.zeros(cw, cn, 3, 3, device=dev)
        db = torch.zeros(cw, device=dev)
        ops.wgrad_direct(wide, narrow, dw, n=n, h=h, w=w, cw=cw, cn=cn, ksize=3, sgn=-1, narrow_layout="nchw",
                         dw_strides=(1, cn * 9, 9), dbias_wide=db)
    torch.cuda.synchronize()
    _report(f"wgrad_direct[{kind}] dW", dw, wt.grad, max_frac=2e-3 if pro else 1e-4, l2=1e-3 if pro else 2e-5)
    _report(f"wgrad_direct[{kind}] db", db, b.grad, max_frac=1e-4, l2=2e-5)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 32, 32, 16, 16), (1, 64, 64, 12, 20), (2, 128, 128, 8, 16), (1, 128, 64, 10, 6)])
def test_conv_mfma_pooled_output(dev, n, cin, cout, h, w):
    """pool2x2_out: y = 2x2 sum pool of the stride-1 conv output (the nearest-2x up-sampling data gradient)."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(3)
    x = _r(torch.randn(n, cin, h, w))
    wt = _r(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5)
    full = _r(F.conv2d(x, wt, None, padding=1))        # the unfused path rounds the full-resolution map to bf16 first
    ref = 4.0 * F.avg_pool2d(full, 2)
    y = torch.full((n, h // 2, w // 2, cout), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.conv_mfma(_nhwc(x).to(dev, torch.bfloat16), ops.pack_conv_weight(wt.to(dev), 3, ops.PTI_CONV_S1), None, y, cout=cout,
                  pool2=True)
    torch.cuda.synchronize()
    _report(f"conv_mfma pooled[{cin}->{cout} {h}x{w}]", y.float().cpu().permute(0, 3, 1, 2), ref)
