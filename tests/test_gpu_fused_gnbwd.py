"""GPU parity of the fused data-gradient + GroupNorm(+SiLU)-backward path (pti_conv2d_mfma_gnbwd followed by
pti_gn_bwd_apply) against torch autograd of  conv(act(GroupNorm(x)))  on CPU fp32.
Tolerance: dx is bf16 and the intermediate dy is rounded to bf16 once -> max-abs <= 1.5 % of scale,
rel-L2 <= 6e-3; dgamma/dbeta are fp32 reductions of bf16 values -> rel-L2 <= 5e-3."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import _nhwc, _r, _report

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,cin,cout,h,w,ks,silu,res", [(2, 32, 32, 16, 16, 3, True, True), (2, 64, 128, 16, 16, 3, True, False),
                                                        (1, 128, 128, 13, 19, 3, True, True), (2, 128, 384, 8, 8, 1, False, True),
                                                        (1, 256, 256, 8, 16, 3, True, False)])
def test_dgrad_with_fused_gn_backward(dev, n, cin, cout, h, w, ks, silu, res):
    from pti_ldm_vae_amd import ops
    torch.manual_seed(11)
    groups, eps = 16, 1e-6
    x = _r(torch.randn(n, cin, h, w) * 1.4 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(cin)).requires_grad_(True)
    beta = (0.1 * torch.randn(cin)).requires_grad_(True)
    wt = _r(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    a = F.group_norm(x, groups, gamma, beta, eps)
    a = F.silu(a) if silu else a
    y = F.conv2d(a, wt, None, padding=ks // 2)
    dy = _r(torch.randn_like(y))
    y.backward(dy)
    dres = _r(torch.randn(n, cin, h, w)) if res else None
    ref_dx = x.grad + (dres if res else 0)
    xd = _nhwc(x.detach()).to(dev, torch.bfloat16)
    st = ops.gn_stats(xd, groups)
    wpt = ops.pack_conv_weight(wt.to(dev), ks, ops.PTI_CONV_S1, flip=True)
    dyt = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(n, cin, 2, device=dev)
    g, b = gamma.detach().to(dev), beta.detach().to(dev)
    ops.conv_mfma_gnbwd(_nhwc(dy).to(dev, torch.bfloat16), wpt, xd, st, g, b, dyt, sums, cout=cin, ksize=ks, groups=groups,
                        eps=eps, silu=silu)
    dx = torch.full_like(xd, float("nan"))
    dg, db = torch.zeros(cin, device=dev), torch.zeros(cin, device=dev)
    ops.gn_bwd_apply(xd, dyt, dx, st, g, b, sums, dg, db, groups=groups, eps=eps,
                     dres=_nhwc(dres).to(dev, torch.bfloat16) if res else None)
    torch.cuda.synchronize()
    _report("fused gnbwd dx", dx.float().cpu().permute(0, 3, 1, 2), ref_dx, max_frac=1.5e-2, l2=6e-3)
    _report("fused gnbwd dgamma", dg, gamma.grad, max_frac=1e-2, l2=5e-3)
    _report("fused gnbwd dbeta", db, beta.grad, max_frac=1e-2, l2=5e-3)


@pytest.mark.parametrize("n,tiles,row", [(3, 1, 64), (2, 37, 128), (4, 512, 64), (1, 100, 24)])
def test_gn_sums_finalize_adds_tile_rows_in_a_fixed_order(dev, n, tiles, row):
    """pti_gn_sums_finalize: sums[n][i] = sum over tiles of partials[n][t][i]; equal to an fp64 sum within fp32
    rounding, identical bits on every call, rows not a multiple of 64 handled."""
    import ctypes as C
    from pti_ldm_vae_amd import _lib as L
    torch.manual_seed(5)
    part = torch.randn(n, tiles, row, device=dev)
    outs = []
    for _ in range(3):
        sums = torch.full((n, row), float("nan"), device=dev)
        L.check(L.lib().pti_gn_sums_finalize(C.c_void_p(part.data_ptr()), C.c_void_p(sums.data_ptr()), n, tiles, row,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pti_gn_sums_finalize")
        outs.append(sums)
    torch.cuda.synchronize()
    ref = part.double().sum(dim=1)
    err = (outs[0].double() - ref).abs().max().item()
    assert err <= 1e-6 * tiles ** 0.5 * 8 + 1e-6, err
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
