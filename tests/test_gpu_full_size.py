"""Per-kernel checks at the sizes the training step really launches (the other per-kernel tests
use maps of at most 64x64): batch 32 x 256 x 256 tensors -- 8192+ workgroups per launch, byte offsets up to 2^28..2^30,
the 32-bit buffer offsets of the conv kernel's loads and stores near their far end, the batched weight-gradient planner
with thousands of tiles per job.  References: fp32 CPU ops on bf16-representable inputs, computed on crops (forward / data
gradient: six windows per launch, placed at the first and last samples and rows) or in full (weight gradient: torch's own
conv2d_weight over the whole batch).  Tolerances as in test_gpu_ops.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16, F16 = torch.bfloat16, torch.float16


def _r(t):
    return t.to(BF16).float()


def _windows(n, h, w, size=24):
    return [(0, 0, 0), (0, h - size, w - size), (n // 2, h // 2 - 7, 3), (n - 1, 0, w - size), (n - 1, h - size, 0),
            (n - 1, h - size, w - size)]


def _crop_with_halo(x_nchw, b, y0, x0, size):
    """x[b, :, y0-1 : y0+size+1, x0-1 : x0+size+1] with zeros outside the image (the conv's own padding)."""
    _, c, h, w = x_nchw.shape
    out = torch.zeros(1, c, size + 2, size + 2)
    ys, ye, xs, xe = max(y0 - 1, 0), min(y0 + size + 1, h), max(x0 - 1, 0), min(x0 + size + 1, w)
    out[0, :, ys - (y0 - 1):ye - (y0 - 1), xs - (x0 - 1):xe - (x0 - 1)] = x_nchw[b, :, ys:ye, xs:xe]
    return out


@pytest.mark.parametrize("n,c,h,w", [(32, 32, 256, 256), (32, 64, 128, 128), (32, 128, 64, 64)])
def test_conv_forward_and_dgrad_at_step_sizes(dev, n, c, h, w):
    """The forward ResBlock conv of config A at batch 32 (fp16 storage + operands, GroupNorm+SiLU prologue, residual,
    fused output statistics, saved activated input) and its plain data gradient (bf16), checked on six windows."""
    from pti_ldm_vae_amd import ops
    torch.manual_seed(n + c)
    groups, eps, size = 16, 1e-6, 24
    g = torch.Generator().manual_seed(c)
    x = _r(torch.randn(n, c, h, w, generator=g) * 1.2 + 0.1)
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    wt16, wtb = wt.to(F16).float(), _r(wt)
    bias = torch.randn(c, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    res = _r(torch.randn(n, c, h, w, generator=g))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, F16)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev, F16)
    y = torch.full((n, h, w, c), float("nan"), dtype=F16, device=dev)
    act = torch.full((n, h, w, c), float("nan"), dtype=BF16, device=dev)
    st = ops.gn_stats(xd, groups)
    ost = torch.zeros(n, groups, 2, dtype=torch.int64, device=dev)
    ops.conv_mfma(xd, ops.pack_conv_weight(wt.to(dev), 3, f16=True), bias.to(dev), y, cout=c, ksize=3,
                  prologue=ops.PTI_PRO_GN_SILU, in_stats=st, gamma=gamma.to(dev), beta=beta.to(dev), groups=groups, eps=eps,
                  residual=rd, out_stats=ost, out_groups=groups, act_out=act)
    dy = _r(torch.randn(n, c, h, w, generator=g))
    dx = torch.full((n, h, w, c), float("nan"), dtype=BF16, device=dev)
    ops.conv_mfma(dy.permute(0, 2, 3, 1).contiguous().to(dev, BF16), ops.pack_conv_weight(wtb.to(dev), 3, flip=True), None, dx,
                  cout=c, ksize=3)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all() and torch.isfinite(act.float()).all()
    yc, dxc, actc = y.float().cpu(), dx.float().cpu(), act.float().cpu()
    wt_t = wtb.flip(2, 3).permute(1, 0, 2, 3).contiguous()          # data-gradient operand
    for b, y0, x0 in _windows(n, h, w, size):
        # GroupNorm statistics are per sample over the whole map
        xg = x[b].reshape(groups, -1)
        mean, var = xg.mean(1), xg.var(1, unbiased=False)
        cpg = c // groups
        sc = (gamma / (var + eps).sqrt().repeat_interleave(cpg))
        sh = beta - mean.repeat_interleave(cpg) * sc
        crop = _crop_with_halo(x, b, y0, x0, size)
        a = F.silu(crop * sc[None, :, None, None] + sh[None, :, None, None])
        inside = torch.zeros(1, 1, size + 2, size + 2)
        ys, ye, xs, xe = max(y0 - 1, 0), min(y0 + size + 1, h), max(x0 - 1, 0), min(x0 + size + 1, w)
        inside[0, 0, ys - (y0 - 1):ye - (y0 - 1), xs - (x0 - 1):xe - (x0 - 1)] = 1.0
        a = a * inside                                               # the padding is a zero of the ACTIVATED tensor
        ref = F.conv2d(a.to(F16).float(), wt16, bias) + res[b:b + 1, :, y0:y0 + size, x0:x0 + size]
        got = yc[b:b + 1, y0:y0 + size, x0:x0 + size].permute(0, 3, 1, 2)
        err = (got - ref).abs().max().item()
        assert err <= 4e-3 * ref.abs().max().item(), (b, y0, x0, err)
        a_in = a[:, :, 1:-1, 1:-1]
        got_a = actc[b:b + 1, y0:y0 + size, x0:x0 + size].permute(0, 3, 1, 2)
        assert (got_a - a_in).abs().max().item() <= 1e-2 * a_in.abs().max().item()
        ref_dx = F.conv2d(_crop_with_halo(dy, b, y0, x0, size), wt_t)
        got_dx = dxc[b:b + 1, y0:y0 + size, x0:x0 + size].permute(0, 3, 1, 2)
        assert (got_dx - ref_dx).abs().max().item() <= 1e-2 * ref_dx.abs().max().item(), (b, y0, x0)
    # fused statistics of the stored output: sums over whole samples (first and last)
    for b in (0, n - 1):
        ref_st = torch.stack([yc[b].reshape(-1, groups, c // groups).double().sum((0, 2)),
                              (yc[b].reshape(-1, groups, c // groups).double() ** 2).sum((0, 2))], -1)
        got_st = ops.stats_to_float(ost[b]).cpu()
        assert ((got_st - ref_st).norm() / ref_st.norm()).item() <= 1e-4


@pytest.mark.parametrize("n,c,h,w", [(32, 32, 256, 256), (32, 128, 64, 64)])
def test_weight_gradient_at_step_sizes(dev, n, c, h, w):
    """Batched weight gradient (two jobs in one launch) over full batch-32 tensors against torch's conv2d_weight."""
    from pti_ldm_vae_amd import ops
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(7 * c)
    jobs, refs = [], []
    for j in range(2):
        x = _r(torch.randn(n, c, h, w, generator=g))
        dy = _r(torch.randn(n, c, h, w, generator=g) * 0.05)
        dw = torch.zeros(c, c, 3, 3, device=dev)
        db = torch.zeros(c, device=dev)
        jobs.append((x.permute(0, 2, 3, 1).contiguous().to(dev, BF16), dy.permute(0, 2, 3, 1).contiguous().to(dev, BF16), dw, db))
        refs.append((torch.nn.grad.conv2d_weight(x, (c, c, 3, 3), dy, padding=1), dy.sum((0, 2, 3))))
    if all(ops.wgrad_batch_eligible(x, dy, 3, ops.PTI_CONV_S1, ops.PTI_PRO_NONE) for x, dy, _, _ in jobs):
        ops.conv_wgrad_mfma_batched(jobs, accumulate=False)
    else:
        for x, dy, dw, db in jobs:
            ops.conv_wgrad_mfma(x, dy, dw, db, ksize=3)
    torch.cuda.synchronize()
    for (x, dy, dw, db), (rw, rb) in zip(jobs, refs):
        assert ((dw.cpu() - rw).norm() / rw.norm()).item() <= 2e-3
        assert ((db.cpu() - rb).norm() / rb.norm()).item() <= 1e-4
